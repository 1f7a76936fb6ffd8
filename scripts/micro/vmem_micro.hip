// Load-path microbenchmark for gfx950 (experiment, not part of the product): how many bytes per clock per CU can a
// 512-thread workgroup per CU pull from L2 in the GEMM's access pattern (8 rows x 128 B per wave-instruction),
//   mode 0: global_load_lds_dwordx4 (LDS-DMA)      mode 1: global_load_dwordx4 -> VGPR
//   mode 2: global_load_dwordx4 -> VGPR -> ds_write_b128
// build: hipcc --offload-arch=gfx950 -O3 -o vmem_micro vmem_micro.hip ; run: ./vmem_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

template <int MODE, int PIECES, bool PIPE>
__global__ __launch_bounds__(512, 2) void k(const char* base, long long region, int row_stride, int iters, float* sink,
                                             long long* clk, int blocked) {
  // blocked = 1: the operand is stored slice-major ([panel][k-slice][256 rows][128 B]): one K-slice of a panel is ONE
  // contiguous 32 KiB block (row pitch 128 B, the next slice follows directly) instead of 256 rows `row_stride` apart
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
  // every workgroup streams 512 rows x 128 B per iteration: rows [0,256) private ("A panel"), [256,512) shared by the XCD ("W")
  const char* priv = base + ((long long)(xcd * 32 + idx) * 256 * row_stride) % region;
  const char* shrd = base + ((long long)(xcd) * 256 * row_stride + region / 2) % region;
  const int pitch = blocked ? 128 : row_stride;
  const int kstep = blocked ? 32768 : 128;
  const int kwrap = blocked ? (256 * row_stride) : row_stride;     // bytes of one panel = what a tile streams through
  const char* src[PIECES];
#pragma unroll
  for (int i = 0; i < PIECES; ++i) {
    const int s = (i * 8 + w) * 64 + lane;          // 16-B chunk index within the 64 KB slice
    const int r = s >> 3, c = (s & 7) ^ ((r >> 1) & 7);
    src[i] = (r < 256 ? priv + (long long)r * pitch : shrd + (long long)(r - 256) * pitch) + c * 16;
  }
  const unsigned lbase = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  float4 accv = make_float4(0, 0, 0, 0);
  long long t0 = __builtin_readcyclecounter();
  if (MODE == 0 && PIPE) {   // slice 0 up front: the loop then issues slice it+1 and waits for slice it
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
      unsigned keep, la = lbase + (i * 8 + w) * 1024;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(src[i]), "s"(la) : "memory");
    }
  }
  for (int it = 0; it < iters; ++it) {
    const int koff = (int)(((long long)(it + (MODE == 0 && PIPE ? 1 : 0)) * kstep) % kwrap);
    const int stage = ((it + (MODE == 0 && PIPE ? 1 : 0)) & 1) * 65536;
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        unsigned keep, la = lbase + stage + (i * 8 + w) * 1024;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(src[i] + koff), "s"(la) : "memory");
      }
      if (PIPE && it + 1 < iters) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      float4 v[PIECES];
#pragma unroll
      for (int i = 0; i < PIECES; ++i) v[i] = *reinterpret_cast<const float4*>(src[i] + koff);
      if (MODE == 2) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) *reinterpret_cast<float4*>(smem + stage + (i * 8 + w) * 1024 + lane * 16) = v[i];
      } else {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) { accv.x += v[i].x; accv.y += v[i].y; accv.z += v[i].z; accv.w += v[i].w; }
      }
    }
    __syncthreads();
  }
  long long t1 = __builtin_readcyclecounter();
  if (MODE != 1) accv = *reinterpret_cast<float4*>(smem + tid * 16);
  if (accv.x == 12345.f) sink[0] = accv.x + accv.y + accv.z + accv.w;
  if (tid == 0) clk[blockIdx.x] = t1 - t0;
}

template <int MODE, bool PIPE>
void run(const char* buf, long long region, int row_stride, int iters, float* sink, long long* clk, const char* name, int grid = 256, int blocked = 0) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k<MODE, 8, PIPE>), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k<MODE, 8, PIPE>), dim3(grid), dim3(512), 131072, 0, buf, region, row_stride, iters, sink, clk, blocked);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
  }
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  long long h[256]; CK(hipMemcpy(h, clk, sizeof(h), hipMemcpyDeviceToHost));
  double c = 0; for (int i = 0; i < grid; ++i) c += (double)h[i]; c /= grid;
  const double bytes = 65536.0 * iters;
  printf("%-28s %s grid %3d region %6lld MB stride %5d: %.3f ms  %.1f B/clk/CU (shader clocks)  %.2f TB/s aggregate  (%.2f GHz)\n", name, blocked ? "BLOCKED" : "strided", grid,
         region >> 20, row_stride, ms, bytes / c, bytes * grid / (ms * 1e-3) / 1e12, c / (ms * 1e6));
}

int main() {
  const long long cap = 1LL << 30;
  char* buf; float* sink; long long* clk;
  CK(hipMalloc(&buf, cap + (1 << 20))); CK(hipMemset(buf, 1, cap + (1 << 20)));
  CK(hipMalloc(&sink, 16)); CK(hipMalloc(&clk, 256 * 8));
  const int iters = 2000;
  for (long long region : {16LL << 20, 512LL << 20}) {
    for (int stride : {1536, 6144}) {
      for (int grid : {256, 64}) {
        for (int blocked : {0, 1}) {
          run<0, true>(buf, region, stride, iters, sink, clk, "lds-dma pipelined", grid, blocked);
          run<1, false>(buf, region, stride, iters, sink, clk, "global_load", grid, blocked);
        }
      }
    }
  }
  return 0;
}
