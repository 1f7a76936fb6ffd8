// Common device/host helpers for the VIMA MI355X (gfx950 / CDNA4) hot path.
// Wavefront = 64 lanes everywhere in this code base; there is no other target.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace vima {

typedef unsigned short bf16_t;  // raw bfloat16 bits (activations / weights operand type in bf16 mode)

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

constexpr int kWave = 64;

// hipFuncSetAttribute (the dynamic-LDS limit of a kernel) applies to the CURRENT device only: every launcher that raises the limit
// keeps one of these per kernel instantiation and asks it before launching, so that a process driving several GPUs (or a handle
// created on a device other than the first one used) gets the attribute on each of them.
struct PerDeviceOnce {
  unsigned long long done = 0;   // bit d: attribute set on device d (devices >= 64 are simply set every time)
  // returns hipSuccess when the attribute is (now) set on the current device
  template <typename F> hipError_t ensure(F&& set_attribute) {
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64 && ((__atomic_load_n(&done, __ATOMIC_ACQUIRE) >> dev) & 1ull)) return hipSuccess;
    e = set_attribute();
    if (e == hipSuccess && dev >= 0 && dev < 64) __atomic_fetch_or(&done, 1ull << dev, __ATOMIC_RELEASE);
    return e;
  }
};

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even; lowers to the gfx950 hardware conversion (v_cvt_pk_bf16_f32)
__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return __builtin_bit_cast(bf16_t, (__bf16)f); }
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack2_bf16(float a, float b) {   // low half = a
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

template <typename T> struct Elem;
template <> struct Elem<float> {
  static __device__ __forceinline__ float load(const float* p) { return *p; }
  static __device__ __forceinline__ void store(float* p, float v) { *p = v; }
  static __device__ __forceinline__ float from_f32(float v) { return v; }
  static __device__ __forceinline__ float to_f32(float v) { return v; }
};
template <> struct Elem<bf16_t> {
  static __device__ __forceinline__ float load(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void store(bf16_t* p, float v) { *p = f32_to_bf16(v); }
  static __device__ __forceinline__ bf16_t from_f32(float v) { return f32_to_bf16(v); }
  static __device__ __forceinline__ float to_f32(bf16_t v) { return bf16_to_f32(v); }
};

// 4 consecutive elements <-> float4
__device__ __forceinline__ float4 load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 load4(const bf16_t* p) {
  uint2 u = *reinterpret_cast<const uint2*>(p);
  float4 r;
  r.x = __uint_as_float(u.x << 16);
  r.y = __uint_as_float(u.x & 0xffff0000u);
  r.z = __uint_as_float(u.y << 16);
  r.w = __uint_as_float(u.y & 0xffff0000u);
  return r;
}
__device__ __forceinline__ void store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ void store4(bf16_t* p, float4 v) {
  uint2 u;
  u.x = pack2_bf16(v.x, v.y);
  u.y = pack2_bf16(v.z, v.w);
  *reinterpret_cast<uint2*>(p) = u;
}

// 4 floats -> 4 OCP e4m3 bytes of e4m3(v * inv), saturating at 448, round to nearest even (byte 0 = v.x)
__device__ __forceinline__ uint32_t pack4_fp8(float4 v, float inv) {
  int w = 0;
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.x * inv, -448.0f, 448.0f), __builtin_amdgcn_fmed3f(v.y * inv, -448.0f, 448.0f), w, false);
  w = __builtin_amdgcn_cvt_pk_fp8_f32(__builtin_amdgcn_fmed3f(v.z * inv, -448.0f, 448.0f), __builtin_amdgcn_fmed3f(v.w * inv, -448.0f, 448.0f), w, true);
  return (uint32_t)w;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// activations (ids shared with the host side)
enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_QUICKGELU = 3 };

__device__ __forceinline__ float apply_act(float v, int act) {
  switch (act) {
    case ACT_RELU: return fmaxf(v, 0.0f);
    case ACT_GELU: return 0.5f * v * (1.0f + erff(v * 0.70710678118654752440f));  // exact erf GELU (nn.GELU())
    case ACT_QUICKGELU: return v * __builtin_amdgcn_rcpf(1.0f + __expf(-1.702f * v));          // x * sigmoid(1.702 x), v_exp + v_rcp
    default: return v;
  }
}

// GELU (erf form, nn.GELU()) for the bf16-operand kernels, whose output is rounded to bf16 (relative step 2^-8) anyway:
// 1 + erf(v / sqrt 2) from the complementary error function in the Abramowitz-Stegun 7.1.26 form, erfc(x) = (a1 t + .. + a5 t^5) e^(-x^2),
// t = 1 / (1 + p x), x = |v| / sqrt 2 (|error of erf| <= 1.5e-7), taken as erfc for v < 0 (no cancellation: the small values keep
// their relative accuracy) and as 2 - erfc otherwise. One v_rcp, one v_exp and 11 plain VALU operations instead of libm's erff
// (~50 instructions: measured 236 clocks per value, the larger part of a GELU GEMM's epilogue -- scripts/small_m_stamps.py).
// The fp32-operand parity kernels keep erff.
__device__ __forceinline__ float gelu_erf_bf16(float v) {
  const float x = fabsf(v) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(__builtin_fmaf(0.3275911f, x, 1.0f));
  float q = __builtin_fmaf(1.061405429f, t, -1.453152027f);
  q = __builtin_fmaf(q, t, 1.421413741f);
  q = __builtin_fmaf(q, t, -0.284496736f);
  q = __builtin_fmaf(q, t, 0.254829592f);
  const float e = (q * t) * __builtin_amdgcn_exp2f((x * x) * -1.44269504088896340736f);   // erfc(x), x >= 0
  return (0.5f * v) * (v < 0.0f ? e : 2.0f - e);
}
// activation of a kernel with operand type T
template <typename T> __device__ __forceinline__ float apply_act_t(float v, int act) {
  if (sizeof(T) == 2 && act == ACT_GELU) return gelu_erf_bf16(v);
  return apply_act(v, act);
}

}  // namespace vima
