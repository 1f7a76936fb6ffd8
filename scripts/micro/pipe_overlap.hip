// Do the matrix pipe and the vector ALU of one gfx950 SIMD run concurrently -- within a wave, and across two waves? (profiles/r06_pipe_overlap.txt)
//   hipcc --offload-arch=gfx950 -O3 -o pipe_overlap scripts/micro/pipe_overlap.hip && ./pipe_overlap
// One workgroup per CU (256 of them), 4 or 8 waves (= 1 or 2 per SIMD); every wave runs ITERS iterations of a body chosen by `mode` and its role; the
// result is shader clocks per iteration of wave 0 (s_memtime) and the kernel's wall time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

#define VFMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(c1), "v"(c2))
#define VEXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define VPK(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(p1), "v"(p2))

// body kinds: 0 = 4 MFMAs (independent accumulators), 1 = NV plain fma on 8 independent registers, 2 = both interleaved (NV/4 fma behind each MFMA),
//             3 = NV v_exp_f32, 4 = MFMAs + v_exp interleaved, 5 = NV/2 v_pk_fma_f32 (two floats each)
template <int KIND, int NV>
__device__ __forceinline__ void body(f32x16_t (&acc)[4], float (&x)[8], bf16x8_t a, bf16x8_t b, float c1, float c2) {
  typedef __attribute__((__vector_size__(2 * sizeof(float)))) float f32x2_t;
  f32x2_t p1 = {c1, c1}, p2 = {c2, c2};
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    if (KIND == 0 || KIND == 2 || KIND == 4) acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m], 0, 0, 0);
#pragma unroll
    for (int v = 0; v < NV / 4; ++v) {
      if (KIND == 1 || KIND == 2) VFMA(x[(m * (NV / 4) + v) & 7]);
      if (KIND == 3 || KIND == 4) VEXP(x[(m * (NV / 4) + v) & 7]);
      if (KIND == 5 && (v & 1) == 0) { f32x2_t t = {x[(m * 2) & 7], x[(m * 2 + 1) & 7]}; VPK(t); x[(m * 2) & 7] = t[0]; x[(m * 2 + 1) & 7] = t[1]; }
    }
  }
}

template <int KA, int KB, int NV>
__global__ __launch_bounds__(512) void probe(float* out, long long* clk, int iters, float c1, float c2) {
  f32x16_t acc[4];
  float x[8];
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  for (int i = 0; i < 8; ++i) x[i] = 0.001f * (threadIdx.x + i);
  bf16x8_t a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (threadIdx.x & 7)); b[i] = (__bf16)(0.02f * i); }
  const int role = __builtin_amdgcn_readfirstlane(threadIdx.x >> 8);   // waves 0-3: role 0, waves 4-7: role 1
  __syncthreads();
  const long long t0 = __builtin_readcyclecounter();
  if (role == 0) {
    for (int it = 0; it < iters; ++it) body<KA, NV>(acc, x, a, b, c1, c2);
  } else {
    for (int it = 0; it < iters; ++it) body<KB, NV>(acc, x, a, b, c1, c2);
  }
  const long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int m = 0; m < 4; ++m)
    for (int r = 0; r < 16; ++r) s += acc[m][r];
  for (int i = 0; i < 8; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 255) == 0) clk[blockIdx.x * 2 + role] = t1 - t0;
}

template <int KA, int KB, int NV>
static void run(const char* name, int waves, int iters) {
  float* out; long long* clk;
  hipMalloc(&out, 256 * 512 * sizeof(float));
  hipMalloc(&clk, 256 * 2 * sizeof(long long));
  hipMemset(clk, 0, 256 * 2 * sizeof(long long));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<KA, KB, NV>), dim3(256), dim3(waves * 64), 0, 0, out, clk, iters, 1.0001f, 0.0001f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  std::vector<long long> h(512);
  hipMemcpy(h.data(), clk, 512 * sizeof(long long), hipMemcpyDeviceToHost);
  double a = 0, b = 0;
  for (int i = 0; i < 256; ++i) { a += h[2 * i]; b += h[2 * i + 1]; }
  printf("%-58s waves/SIMD %d: role0 %7.1f clk/iter  role1 %7.1f clk/iter   kernel %.3f ms\n", name, waves / 4, a / 256 / iters, b / 256 / iters, ms);
  hipFree(out); hipFree(clk);
}

int main() {
  const int it = 4000;
  printf("per iteration: 4 MFMA 32x32x16 bf16 (128 clocks of matrix pipe) and / or NV vector instructions\n");
  run<0, 0, 16>("MFMA only", 4, it);
  run<1, 1, 16>("16 v_fma only", 4, it);
  run<1, 1, 32>("32 v_fma only", 4, it);
  run<3, 3, 16>("16 v_exp only", 4, it);
  run<5, 5, 16>("8 v_pk_fma (16 floats) only", 4, it);
  run<2, 2, 16>("one wave: 4 MFMA + 16 v_fma interleaved", 4, it);
  run<2, 2, 32>("one wave: 4 MFMA + 32 v_fma interleaved", 4, it);
  run<4, 4, 16>("one wave: 4 MFMA + 16 v_exp interleaved", 4, it);
  run<0, 0, 16>("two waves: MFMA | MFMA", 8, it);
  run<1, 1, 32>("two waves: 32 v_fma | 32 v_fma", 8, it);
  run<3, 3, 16>("two waves: 16 v_exp | 16 v_exp", 8, it);
  run<0, 1, 16>("two waves: 4 MFMA | 16 v_fma", 8, it);
  run<0, 1, 32>("two waves: 4 MFMA | 32 v_fma", 8, it);
  run<0, 3, 16>("two waves: 4 MFMA | 16 v_exp", 8, it);
  run<0, 3, 32>("two waves: 4 MFMA | 32 v_exp", 8, it);
  run<2, 2, 16>("two waves: (4 MFMA + 16 v_fma) x 2", 8, it);
  return 0;
}
