// probe of ds_read_b64_tr_b16 semantics on gfx950 (experiment): prints, for natural per-lane addresses (lane i -> 4
// contiguous bf16 at element 4 i), which LDS element each (lane, j) receives
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short s4;
__global__ void k(short* out, int stride_elems) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int i = threadIdx.x & 15, g = threadIdx.x >> 4;
  // group g: 4 rows (row stride `stride_elems`), lane i -> row i/4, quad i%4; groups 1024 elements apart
  const int a = g * 1024 + (i / 4) * stride_elems + (i % 4) * 4;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + a));
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  for (int stride : {16, 64}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, stride);
    short h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row stride %d elements:\n", stride);
    for (int l = 0; l < 64; ++l) { printf("lane %2d:", l); for (int j = 0; j < 4; ++j) printf(" %5d", h[l * 4 + j]); printf("%s", (l % 4 == 3) ? "\n" : "   |"); }
  }
  return 0;
}
