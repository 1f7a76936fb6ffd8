// Attention kernels of the VIMA policy hot path (gfx950, wave64).
//
//  * vit_attn      : nn.MultiheadAttention(768, 24) over 5-token crop sequences (vit.py:203,217-231). One thread per
//                    (crop, query token, head); everything in registers; HBM-bound.
//  * attn_generic  : exact fp32-softmax attention for any operand type (parity mode + fallback), one wave per query.
//  * attn_mfma / attn_mfma4 / attn_split : bf16 flash attention on the matrix cores (online softmax in the log2 domain,
//                    fp32 statistics): one wave per 32 queries (small shapes), 4 waves sharing LDS-staged K/V tiles
//                    (>= 64 queries; V read with ds_read_b64_tr_b16), or 4 waves splitting the keys of <= 32 queries
//                    (one env step of cross attention); head dims 32 / 64 (16 / 128 run on attn_generic). Flavours:
//       ATTN_T5     T5Attention.forward (prompt_encoder.py:769-816): NO 1/sqrt(d); + (rel-bias[bucket(j-i)] + mask)
//       ATTN_CROSS  XAttention.forward (components.py:184-214): /sqrt(d); + (1-mask)*finfo.min key mask
//       ATTN_CAUSAL Attention._attn (components.py:51-80): /sqrt(d); w*b + -1e4*(1-b) causal fill; + key mask
//     Masked keys carry the score finfo(fp32).min exactly like the reference (so an all-masked row degenerates to the
//     same uniform distribution); the -1e4 causal fill is kept literally (no causal tile skipping).
#include "kernels.h"
#include <float.h>
#include <type_traits>
#include <stdlib.h>

namespace vima {
namespace {

// =============================================================================================== ViT (S <= 8, D = 32)
// One (crop, query token, head) of nn.MultiheadAttention over a 5-token crop sequence, everything in registers. ONE body for the
// three kernels below (operands from global memory, from LDS, cls query only): the last ViT block may be evaluated for the cls row
// alone (vit_prune_last) and must give the bits the full block gives, so the arithmetic is pinned -- explicit fmaf for the dot
// products and the weighted sum, no contraction of anything else (the same source compiled into different kernels was otherwise
// contracted differently: 4.6e-3 on the object tokens between the LDS-staged kernel and the per-thread one).
__device__ __forceinline__ void unpack8(const uint4& u, float* f) {
  f[0] = __uint_as_float(u.x << 16); f[1] = __uint_as_float(u.x & 0xffff0000u);
  f[2] = __uint_as_float(u.y << 16); f[3] = __uint_as_float(u.y & 0xffff0000u);
  f[4] = __uint_as_float(u.z << 16); f[5] = __uint_as_float(u.z & 0xffff0000u);
  f[6] = __uint_as_float(u.w << 16); f[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void ld8(const bf16_t* p, float* f) { unpack8(*reinterpret_cast<const uint4*>(p), f); }
__device__ __forceinline__ void ld8(const float* p, float* f) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// q: the query's 32 values; krow(j) / vrow(j): pointer to the 32 values of key / value row j of this head; o: the 32 outputs
// co[c / 8]: element offset of the head's 8-value chunk c / 8 inside its 32-value row segment -- {0, 8, 16, 24} everywhere except in
// vit_attn_lds_kernel, whose LDS image stores the four 16-byte chunks of a head permuted (bank conflicts); the VALUES and the order of
// the arithmetic do not depend on it.
struct ChunkId { __device__ __forceinline__ int operator[](int i) const { return i * 8; } };
// SM = the largest sequence the instantiation handles (8: object crops, 5 tokens; 16: the baseline policies' whole frames, cls + 8 patches = 9 tokens). Rows
// j >= S contribute exp = 0.f to the sum and nothing else, so the result does not depend on SM.
template <typename T, typename KR, typename VR, typename CO = ChunkId, int SM = 8>
__device__ __forceinline__ void vit_head_attention(const T* qp, int S, KR krow, VR vrow, float (&o)[32], CO co = CO()) {
#pragma clang fp contract(off)
  constexpr int D = 32;
  float q[D];
#pragma unroll
  for (int c = 0; c < D; c += 8) ld8(qp + co[c >> 3], q + c);
  const float scale = 0.17677669529663687f;  // 1/sqrt(32)
  float s[SM];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < SM; ++j) {
    s[j] = -INFINITY;
    if (j < S) {
      const T* kp = krow(j);
      float d = 0.f;
#pragma unroll
      for (int c = 0; c < D; c += 8) {
        float v[8];
        ld8(kp + co[c >> 3], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) d = __builtin_fmaf(q[c + e], v[e], d);
      }
      s[j] = d * scale;
      mx = fmaxf(mx, s[j]);
    }
  }
  float l = 0.f;
#pragma unroll
  for (int j = 0; j < SM; ++j) {
    s[j] = j < S ? expf(s[j] - mx) : 0.f;
    l += s[j];
  }
  const float inv = 1.0f / l;
#pragma unroll
  for (int c = 0; c < D; ++c) o[c] = 0.f;
#pragma unroll
  for (int j = 0; j < SM; ++j) {
    if (j < S) {
      const T* vp = vrow(j);
      const float pj = s[j] * inv;
#pragma unroll
      for (int c = 0; c < D; c += 8) {
        float v[8];
        ld8(vp + co[c >> 3], v);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[c + e] = __builtin_fmaf(pj, v[e], o[c + e]);
      }
    }
  }
}

template <typename T>
__device__ __forceinline__ void vit_store_head(const float (&o)[32], T* op, uint8_t* o8p, float inv8) {
  constexpr int D = 32;
  if (o8p) {   // precision "fp8": the consumer GEMM (out_proj) takes e4m3 activations
    uint32_t w[8];
#pragma unroll
    for (int c = 0; c < D; c += 4) w[c >> 2] = pack4_fp8(make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]), inv8);
    uint4* o8 = reinterpret_cast<uint4*>(o8p);
    o8[0] = make_uint4(w[0], w[1], w[2], w[3]);
    o8[1] = make_uint4(w[4], w[5], w[6], w[7]);
    return;
  }
#pragma unroll
  for (int c = 0; c < D; c += 4) store4(op + c, make_float4(o[c], o[c + 1], o[c + 2], o[c + 3]));
}

template <typename T, int SM = 8>
__global__ __launch_bounds__(256) void vit_attn_kernel(const T* __restrict__ qkv, T* __restrict__ out, long long total,
                                                        int S, int W, int heads, uint8_t* out8 = nullptr, float inv8 = 1.0f) {
  constexpr int D = 32;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int h = (int)(idx % heads);
  const long long mi = idx / heads;       // m * S + i
  const long long m = mi / S;
  const int ld = 3 * W;
  float o[D];
  auto kr = [&](int j) { return qkv + (m * S + j) * ld + W + h * D; };
  auto vr = [&](int j) { return qkv + (m * S + j) * ld + 2 * W + h * D; };
  vit_head_attention<T, decltype(kr), decltype(vr), ChunkId, SM>(qkv + mi * ld + h * D, S, kr, vr, o);
  vit_store_head<T>(o, out + mi * W + h * D, out8 ? out8 + mi * W + h * D : nullptr, inv8);
}

// The same attention for the hot shape (bf16, 5 tokens, width 768, 24 heads) with the operands staged in LDS. In the kernel above every
// (query, head) thread re-reads its crop's K and V rows itself: 84 KB per crop go through the vector L1 in 8-byte requests for 23 KB
// of HBM data (4.2 TB/s). Here a workgroup copies the contiguous q|k|v rows of TWO crops (10 x 4 608 B) into LDS once with 16-byte
// LDS-DMA requests and its 240 (crop, query, head) threads read them from there with ds_read_b128; three workgroups per CU keep
// 135 KB of loads in flight (5.4 TB/s). Same per-head body as the other two kernels: bit-identical results.
// Round 4: consecutive lanes are consecutive HEADS, 64 bytes apart, so the 16 lanes of a ds_read_b128 group hit every bank four times (SQ
// counters: LDS conflict ratio 0.67). The image is therefore stored with the four 16-byte chunks of head h permuted, position x holds
// chunk x ^ ((h >> 2) & 3) (applied on the SOURCE address of the LDS-DMA, which is per lane; the destination stays lane-linear), and read
// back through the same XOR: the 16 lanes of a group (heads a + 4 b: every 64-byte bank quarter a four times, with four different b) then
// touch 16 distinct 16-byte slots of the bank row (K / V reads conflict-free, the query reads of lanes on different rows 2-way: ratio 0.30).
// The kernel time did not move (92 us): it reads 415 MB and WRITES 138 MB per launch = 6.0 TB/s, the copy rate of this chip (6.3 TB/s).
constexpr int VA_S = 5, VA_W = 768, VA_H = 24, VA_ROWB = 3 * VA_W * 2;   // bytes per qkv row
__global__ __launch_bounds__(256) void vit_attn_lds_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out, int M,
                                                            uint8_t* out8, float inv8) {
  extern __shared__ __attribute__((aligned(16))) char va_smem[];
  constexpr int D = 32;
  const int tid = threadIdx.x, w = tid >> 6;
  const long long m0 = (long long)blockIdx.x * 2;
  const int nc = (M - m0) >= 2 ? 2 : 1;
  const int chunks = nc * VA_S * VA_ROWB / 16;                // 16-byte chunks of this workgroup's rows (contiguous in memory)
  const char* src = reinterpret_cast<const char*>(qkv) + m0 * VA_S * VA_ROWB;
#pragma unroll
  for (int it = 0; it < (2 * VA_S * VA_ROWB / 16 + 255) / 256; ++it) {
    const int c = it * 256 + tid;                     // LDS chunk position (lane-linear destination)
    const int pr = c % (VA_ROWB / 16);                // position inside its qkv row: [q | k | v] x 24 heads x 4 chunks
    const int g = c ^ (((pr % (VA_W / 8)) >> 4) & 3);   // global chunk: x ^ ((h >> 2) & 3), h = (pr % 96) / 4
    if (c < chunks)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (long long)g * 16),
                                       (__attribute__((address_space(3))) void*)(va_smem + (it * 256 + w * 64) * 16), 16, 0, 0);
  }
  __syncthreads();
  if (tid >= nc * VA_S * VA_H) return;
  const int cl = tid / (VA_S * VA_H), rem = tid % (VA_S * VA_H);
  const int i = rem / VA_H, h = rem % VA_H;
  const bf16_t* base = reinterpret_cast<const bf16_t*>(va_smem) + cl * VA_S * (3 * VA_W);
  float o[D];
  struct ChunkSw { int b; __device__ __forceinline__ int operator[](int i) const { return (i ^ b) * 8; } };
  vit_head_attention<bf16_t>(base + i * (3 * VA_W) + h * D, VA_S, [&](int j) { return base + j * (3 * VA_W) + VA_W + h * D; },
                             [&](int j) { return base + j * (3 * VA_W) + 2 * VA_W + h * D; }, o, ChunkSw{(h >> 2) & 3});
  const long long mi = (m0 + cl) * VA_S + i;
  vit_store_head<bf16_t>(o, out + mi * VA_W + h * D, out8 ? out8 + mi * VA_W + h * D : nullptr, inv8);
}

// Last ViT block: only the cls token (token 0) is read by ln_post (vit.py:186), so only its query is needed.
// q: T [M, W] (cls rows), kv: T [M*S, 2W] (K | V of every token) -> out T [M, W]. One thread per (crop, head).
template <typename T>
__global__ __launch_bounds__(256) void vit_attn_cls_kernel(const T* __restrict__ qb, const T* __restrict__ kv,
                                                            T* __restrict__ out, long long total, int S, int W, int heads,
                                                            uint8_t* out8 = nullptr, float inv8 = 1.0f) {
  constexpr int D = 32;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int h = (int)(idx % heads);
  const long long m = idx / heads;
  const int ld = 2 * W;
  float o[D];
  vit_head_attention<T>(qb + m * W + h * D, S, [&](int j) { return kv + (m * S + j) * ld + h * D; },
                        [&](int j) { return kv + (m * S + j) * ld + W + h * D; }, o);
  vit_store_head<T>(o, out + m * W + h * D, out8 ? out8 + m * W + h * D : nullptr, inv8);
}

// =============================================================================================== generic exact kernel

struct AttnDev {
  const void* q; int ldq;
  const void* k; int ldk;
  const void* v; int ldv;
  void* out; int ldo;
  const uint8_t* kmask;
  const float* relbias;
  int bias_far;   // see AttnArgs::bias_far (0 = no promise)
  int B, H, Lq, Lk;
  float scale;
  int mode;
  int Lkr;      // K / V / mask rows per sample in memory (>= Lk)
  long long k_bs, v_bs; int k_hs, v_hs;   // K / V batch and head strides in elements (AttnArgs; defaults Lkr * ld and D)
  int q_off;    // causal: query i sits at global position i + q_off
  long long* dbg;   // optional (4-wave kernel): per workgroup 8 accumulated shader-clock phase totals of wave 0
};

__device__ __forceinline__ float score_fixup(float dot, int mode, float scale, int i, int j, int Lk, bool masked,
                                             const float* relbias_h) {
  // literal restatement of the reference's score arithmetic (order of operations preserved)
  const float madd = masked ? -FLT_MAX : 0.0f;
  float s;
  if (mode == ATTN_T5) {
    s = dot + (relbias_h[(j - i) + Lk - 1] + madd);
  } else if (mode == ATTN_CROSS) {
    s = dot * scale + madd;
  } else {
    s = dot * scale;
    if (j > i) s = -1e4f;
    s = s + madd;
  }
  return s;
}

template <typename T, int D>
__global__ __launch_bounds__(64) void attn_generic_kernel(const AttnDev p) {
  extern __shared__ float sc[];
  const int lane = threadIdx.x;
  const int i = blockIdx.x, h = blockIdx.y, b = blockIdx.z;
  const T* qp = reinterpret_cast<const T*>(p.q) + ((long long)b * p.Lq + i) * p.ldq + h * D;
  float q[D];
#pragma unroll
  for (int c = 0; c < D; c += 4) {
    const float4 v = load4(qp + c);
    q[c] = v.x; q[c + 1] = v.y; q[c + 2] = v.z; q[c + 3] = v.w;
  }
  const float* rb = p.relbias ? p.relbias + (long long)h * (2 * p.Lk - 1) : nullptr;
  float mx = -INFINITY;
  for (int j = lane; j < p.Lk; j += 64) {
    const T* kp = reinterpret_cast<const T*>(p.k) + b * p.k_bs + (long long)j * p.ldk + h * p.k_hs;
    float d = 0.f;
#pragma unroll
    for (int c = 0; c < D; c += 4) {
      const float4 v = load4(kp + c);
      d = fmaf(q[c], v.x, d); d = fmaf(q[c + 1], v.y, d); d = fmaf(q[c + 2], v.z, d); d = fmaf(q[c + 3], v.w, d);
    }
    const bool masked = p.kmask && !p.kmask[(long long)b * p.Lkr + j];
    const float s = score_fixup(d, p.mode, p.scale, i + p.q_off, j, p.Lk, masked, rb);
    sc[j] = s;
    mx = fmaxf(mx, s);
  }
  mx = wave_max(mx);
  float l = 0.f;
  for (int j = lane; j < p.Lk; j += 64) {
    const float e = expf(sc[j] - mx);
    sc[j] = e;
    l += e;
  }
  l = wave_sum(l);
  __syncthreads();
  for (int dl = lane; dl < D; dl += 64) {   // D = 128: two output columns per lane
    float acc = 0.f;
    const T* vp = reinterpret_cast<const T*>(p.v) + b * p.v_bs + h * p.v_hs + dl;
    for (int j = 0; j < p.Lk; ++j) acc = fmaf(sc[j], Elem<T>::load(vp + (long long)j * p.ldv), acc);
    T* op = reinterpret_cast<T*>(p.out) + ((long long)b * p.Lq + i) * p.ldo + h * D + dl;
    Elem<T>::store(op, acc / l);
  }
}

// =============================================================================================== MFMA flash attention
// One wave: 32 queries x all keys of one (batch, head). S^T = K.Q^T is computed with the KEY tile as the MFMA A-operand
// so that a lane owns one query column: row max / sum are in-lane reductions plus one exchange with lane^32, and the
// exponentiated scores are ALREADY laid out as the B-operand (P^T) of O^T += V^T.P^T -- no cross-lane shuffle of P.
// V^T comes from a transposed LDS image of the V tile (rows padded to 72 B: conflict-free ds_read_b64).
constexpr int VT_STRIDE = 36;  // bf16 elements per Vt row (32 keys + 4 pad)

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) { return pack2_bf16(a, b); }

template <int D, int MODE>
__global__ __launch_bounds__(64) void attn_mfma_kernel(const AttnDev p) {
  __shared__ __attribute__((aligned(16))) bf16_t vt[D * VT_STRIDE];
  constexpr int KD = D / 16;   // MFMA k-steps over the head dim
  constexpr int OT = D / 32;   // 32-row tiles of O^T
  const int lane = threadIdx.x;
  const int hi = lane >> 5, l31 = lane & 31;
  const int q0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q);
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k);
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v);

  const int qi = q0 + l31;
  const int qrow = qi < p.Lq ? qi : p.Lq - 1;
  bf16x8_t qf[KD];
#pragma unroll
  for (int dd = 0; dd < KD; ++dd) {
    const uint4 u = *reinterpret_cast<const uint4*>(Q + ((long long)b * p.Lq + qrow) * p.ldq + h * D + dd * 16 + hi * 8);
    qf[dd] = __builtin_bit_cast(bf16x8_t, u);
  }
  f32x16_t ot[OT];
#pragma unroll
  for (int it = 0; it < OT; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[it][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;
  const float* rb = (MODE == ATTN_T5) ? p.relbias + (long long)h * (2 * p.Lk - 1) + (p.Lk - 1) - qi : nullptr;
  const uint8_t* km = p.kmask ? p.kmask + (long long)b * p.Lkr : nullptr;

  for (int k0 = 0; k0 < p.Lk; k0 += 32) {
    // ---- S^T tile: rows = 32 keys, cols = 32 queries
    f32x16_t s;
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = 0.f;
    {
      const int krow = (k0 + l31) < p.Lk ? (k0 + l31) : p.Lk - 1;
      const bf16_t* kp = K + b * p.k_bs + (long long)krow * p.ldk + h * p.k_hs + hi * 8;
#pragma unroll
      for (int dd = 0; dd < KD; ++dd) {
        const uint4 u = *reinterpret_cast<const uint4*>(kp + dd * 16);
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, u), qf[dd], s, 0, 0, 0);
      }
    }
    // ---- stage V^T of this key tile (previous tile's readers are done: single wave + barrier)
    __syncthreads();
#pragma unroll
    for (int c0 = 0; c0 < (32 * D / 8) / 64; ++c0) {
      const int c = lane + c0 * 64;
      const int key = c / (D / 8), dc = c % (D / 8);
      const int vrow = (k0 + key) < p.Lk ? (k0 + key) : p.Lk - 1;
      const uint4 u = *reinterpret_cast<const uint4*>(V + b * p.v_bs + (long long)vrow * p.ldv + h * p.v_hs + dc * 8);
      const uint32_t wds[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        vt[(dc * 8 + 2 * e) * VT_STRIDE + key] = (bf16_t)(wds[e] & 0xffffu);
        vt[(dc * 8 + 2 * e + 1) * VT_STRIDE + key] = (bf16_t)(wds[e] >> 16);
      }
    }
    // ---- scores -> probabilities (fp32), online softmax
    float x[16];
    float mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int key = k0 + (r & 3) + 8 * (r >> 2) + 4 * hi;
      float v;
      if (key < p.Lk) {
        const float madd = (km && !km[key]) ? -FLT_MAX : 0.0f;
        if (MODE == ATTN_T5) {
          v = s[r] + ((qi < p.Lq ? rb[key] : 0.f) + madd);
        } else if (MODE == ATTN_CROSS) {
          v = s[r] * p.scale + madd;
        } else {
          v = s[r] * p.scale;
          if (key > qi + p.q_off) v = -1e4f;
          v = v + madd;
        }
      } else {
        v = -INFINITY;
      }
      x[r] = v;
      mt = fmaxf(mt, v);
    }
    mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
    const float m_new = fmaxf(m_run, mt);
    const float alpha = __expf(m_run - m_new);
    uint32_t pk[8];
    float rs = 0.f;
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const float p0 = __expf(x[r] - m_new), p1 = __expf(x[r + 1] - m_new);
      const uint32_t u = pack_bf16(p0, p1);
      pk[r >> 1] = u;
      rs += __uint_as_float(u << 16) + __uint_as_float(u & 0xffff0000u);   // sum the ROUNDED weights actually used
    }
    rs += __shfl_xor(rs, 32, 64);
    l_run = l_run * alpha + rs;
    m_run = m_new;
#pragma unroll
    for (int it = 0; it < OT; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[it][r] *= alpha;
    __syncthreads();
    // ---- O^T += V^T . P^T   (two K=16 MFMAs per 32-row tile of O^T)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint4 pu;
      pu.x = pk[half * 4 + 0]; pu.y = pk[half * 4 + 1]; pu.z = pk[half * 4 + 2]; pu.w = pk[half * 4 + 3];
      const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
#pragma unroll
      for (int it = 0; it < OT; ++it) {
        const bf16_t* vr = vt + (it * 32 + l31) * VT_STRIDE + 16 * half + 4 * hi;
        const uint2 a0 = *reinterpret_cast<const uint2*>(vr);
        const uint2 a1 = *reinterpret_cast<const uint2*>(vr + 8);
        uint4 vu;
        vu.x = a0.x; vu.y = a0.y; vu.z = a1.x; vu.w = a1.y;
        ot[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vu), pf, ot[it], 0, 0, 0);
      }
    }
  }
  if (qi < p.Lq) {
    const float inv = 1.0f / l_run;
    bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + ((long long)b * p.Lq + qi) * p.ldo + h * D;
#pragma unroll
    for (int it = 0; it < OT; ++it)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int d = it * 32 + 8 * g + 4 * hi;
        store4(op + d, make_float4(ot[it][4 * g] * inv, ot[it][4 * g + 1] * inv, ot[it][4 * g + 2] * inv,
                                   ot[it][4 * g + 3] * inv));
      }
  }
}

// =============================================================================================== MFMA flash attention, 4 waves
// Workgroup = 4 waves = 128 queries of one (batch, head); key tiles of 64 are staged ONCE per workgroup in LDS and shared
// by the 4 waves (K row-major with the GEMM's XOR chunk swizzle -> conflict-free ds_read_b128 A-fragments; V transposed
// with 136-B rows -> conflict-free ds_read_b64). The next tile's global loads are issued before the current tile is
// multiplied and written to the other LDS buffer afterwards (register-staged double buffer, one barrier per tile).
// The T5 relative-position bias (a function of key - query only) and the additive key mask live in LDS for the whole
// workgroup, so the softmax stage makes no global-memory access. Same S^T = K.Q^T / O^T += V^T.P^T formulation and the
// same literal mask semantics as attn_mfma_kernel above.
constexpr float kLog2e = 1.4426950408889634f;
constexpr int VT4_STRIDE = 68;   // sizing only: the V image needs < D * 68 * 2 bytes per buffer
// V is kept ROW-major in LDS as D/16 sub-tiles of [64 keys][16 d] (32-byte rows) and the A-fragments of O^T += V^T.P^T
// are fetched with gfx950's transposing read (ds_read_b64_tr_b16: within a 16-lane group, lane i supplies the address
// of (row i/4, 4 columns 4(i%4)..) and lane l receives column l of those 4 rows). One 16-byte ds_write_b128 per
// loaded chunk replaces the eight 2-byte scatter stores of a software transpose (which were 60 % of this kernel's LDS
// cycles, 4-way bank-conflicted). Sub-tiles 2s and 2s+1 (read together by lanes 0-15 / 16-31) are 128 B apart modulo
// the 256-B bank row -> conflict-free reads; the writes are 2-way conflicted at most.
template <int KEYS = 64>
__device__ __forceinline__ constexpr int vsub_off(int s) {
  return (s >> 1) * (2 * (KEYS * 32 + 128) + 64) + (s & 1) * (KEYS * 32 + 128);
}
typedef __attribute__((__vector_size__(4 * sizeof(short)))) short short4_t;
__device__ __forceinline__ uint2 lds_read_tr16(const char* lds_ptr) {
  const short4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4_t*)(lds_ptr));
  return __builtin_bit_cast(uint2, v);
}

// LDS-DMA (16 bytes per lane straight from global memory into LDS at M0 + lane * 16), SADDR form: wave-uniform 64-bit base in SGPRs +
// per-lane unsigned 32-bit byte offset. Inline asm: hipcc's wait-count pass does not see it, the kernel counts vmcnt itself.
__device__ __forceinline__ void attn_glds16(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

template <int D>
__device__ __forceinline__ int kswz(int r, int c) {
  if (D == 64) return c ^ ((r >> 1) & 7);   // 128-B rows, 8 chunks
  return c ^ ((r >> 2) & 3);                // 64-B rows, 4 chunks
}

// (phase ablations of this kernel -- timing only, wrong results -- live in scripts/ablate/attn_ablate.patch and attn_persistent_r04.patch)
#ifndef VIMA_ATTN_NSTG1
#define VIMA_ATTN_NSTG1 2   // ring stages with 32 queries per wave (3 measured 15 % slower: profiles/r04_attention_ablation.txt)
#endif
#ifndef VIMA_ATTN_NSTG2
#define VIMA_ATTN_NSTG2 3   // ... with 64 queries per wave (option attn_qg = 2)
#endif
// QG = query groups of 32 per wave. QG = 2 (option attn_qg = 2): a wave owns 64 queries, i.e. a workgroup 256; every K / V
// fragment read from LDS feeds TWO MFMAs (one per group), and the staging of a key tile (global loads, LDS stores, barrier) is
// amortised over twice the queries; the softmax VALU work per query is unchanged. Costs registers (two waves per SIMD).
template <int D, int MODE, int QG = 1>
__global__ __launch_bounds__(256, QG > 1 ? 2 : (D == 32 ? 4 : 3)) void attn_mfma4_kernel(const AttnDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem4[];
  constexpr int KD = D / 16, OT = D / 32, CPR = D / 8;       // k-steps, O^T tiles, 16-B chunks per K/V row
  constexpr int ROWB = D * 2;
  constexpr int KS_BYTES = 64 * ROWB;
  constexpr int VT_BYTES = D * VT4_STRIDE * 2;
  // K / V tiles are staged by LDS-DMA into a ring of NSTG stages; NSTG - 1 tiles are in flight while one is multiplied. (Round 3:
  // the register-staged double buffer spent 18 % of a key tile issuing the next tile's loads -- address arithmetic and four 16-byte
  // loads per lane -- and 8 % writing them to LDS; per-phase shader-clock stamps, scripts/attn_micro.py STAMPS=1.)
  constexpr int NSTG = QG > 1 ? VIMA_ATTN_NSTG2 : VIMA_ATTN_NSTG1;   // 64 queries per wave run two workgroups per CU: room for three stages
  constexpr int NI = D / 32;                                   // 1-KiB DMA instructions per wave and tile for K, and as many for V
  char* ks_base = smem4;                                       // [NSTG][64][ROWB]
  char* vt_base = smem4 + NSTG * KS_BYTES;                            // [NSTG] x V image (sub-tiled, see vsub_off)
  float* madd = reinterpret_cast<float*>(smem4 + NSTG * KS_BYTES + NSTG * VT_BYTES);   // [nt*64]
  const int nt = (p.Lk + 63) / 64;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform (LDS-DMA destinations live in SGPRs / M0)
  const int hi = lane >> 5, l31 = lane & 31;
  // XCD-aware order (workgroup id % 8 = XCD): the query blocks of one (batch, head) run back to back on ONE XCD so its
  // K/V (re-read by every query block) stay in that XCD's L2: id = ((bh / 8) * nq + qblk) * 8 + bh % 8
  constexpr int QB = 128 * QG;                                 // queries per workgroup
  const int nq = (p.Lq + QB - 1) / QB;
  const int bid = blockIdx.x;
  const int bh = (bid / (8 * nq)) * 8 + (bid & 7);
  const int qblk = (bid >> 3) % nq;
  if (bh >= p.B * p.H) return;
  const int h = bh % p.H, b = bh / p.H;
  int qi[QG];
#pragma unroll
  for (int g = 0; g < QG; ++g) qi[g] = qblk * QB + w * (32 * QG) + g * 32 + l31;
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q);
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k);
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v);

  // ---- workgroup-wide tables (filled further down, behind the first tile's LDS-DMA and the query loads)
  // madd[j]: additive key mask (0 / finfo.min; -inf beyond Lk so padded keys never contribute)
  // tflag[t]: 1 when tile t needs madd (a masked or out-of-range key), else the adds are skipped for the whole tile
  // btab (T5): relative-position bias indexed by (key - query) + boff for the queries of THIS workgroup, zero-padded so that every index a lane
  //            can form (including padded queries / keys) is in range -> no clamps in the inner loop
  int* tflag = reinterpret_cast<int*>(madd + nt * 64);
  float* btab = reinterpret_cast<float*>(tflag + ((nt + 3) & ~3));
  const int kpad = nt * 64;
  const int boff = qblk * QB + QB - 1;                         // index = key - qi + boff in [0, kpad + QB - 2]
  // T5 / cross mode: the mask enters as the S^T accumulators' initial value (see the key loop); its "minus infinity" for a masked key is then multiplied by
  // log2(e) (or the score scale) with the rest of the score, so it is -1e38 instead of finfo.min: it absorbs any score all the same (masked keys weigh exactly 0,
  // a row without a single valid key still comes out uniform) and stays finite
  constexpr bool CINIT = MODE != ATTN_CAUSAL;
  constexpr float kMaskNeg = -1.0e38f;
  const int far = MODE == ATTN_T5 ? p.bias_far : 0;
  const int qw0 = qblk * QB + w * (32 * QG), qw1 = qw0 + 32 * QG - 1;   // this wave's queries (wave-uniform)

  bf16x8_t qf[QG][KD];
  f32x16_t ot[QG][OT];
  float m_run[QG], l_run[QG];
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    const int qrow = qi[g] < p.Lq ? qi[g] : p.Lq - 1;
#pragma unroll
    for (int dd = 0; dd < KD; ++dd) {
      const uint4 u = *reinterpret_cast<const uint4*>(Q + ((long long)b * p.Lq + qrow) * p.ldq + h * D + dd * 16 + hi * 8);
      qf[g][dd] = __builtin_bit_cast(bf16x8_t, u);
    }
#pragma unroll
    for (int it = 0; it < OT; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) ot[g][it][r] = 0.f;
    m_run[g] = -INFINITY;
    l_run[g] = 0.f;
  }

  // per-lane byte offset of the transposing V reads: 16-lane group g = lane>>4 reads sub-tile (g&1) [d 16(g&1)..+15],
  // keys 4 hi + i/4 (i = lane&15; hi = g>>1), 8-byte column quad i%4. The k-slots of the P^T operand are the S^T
  // accumulator registers: lane-half hi holds keys 4hi+{0..3} and 8+4hi+{0..3} of each 16-key step.
  const int vlane = vsub_off((lane >> 4) & 1) + (4 * (lane >> 5) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
  // DMA geometry. K: instruction j = w * NI + i fills LDS chunks p = j * 64 + lane of the [64 keys][CPR chunks] image, i.e. key
  // p / CPR at swizzled position p % CPR, which holds the row's chunk (p % CPR) ^ swizzle(key): the XOR moves to the GLOBAL side.
  // V: instruction j fills half (j & 1) of sub-tile j / 2 ([64 keys][16 d], 32-byte rows): lane -> key 32 (j & 1) + lane / 2,
  // 16-byte part lane & 1 of the row's 32-byte segment 2 (j / 2) + part. Per-lane byte offsets inside a tile are fixed; the tile's
  // first row is a wave-uniform pointer (SGPRs), so a tile costs no vector address arithmetic.
  const bf16_t* Kb = K + b * p.k_bs + h * p.k_hs;
  const bf16_t* Vb = V + b * p.v_bs + h * p.v_hs;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem4;
  auto k_key = [&](int i, int ln) { return ((w * NI + i) * 64 + ln) / CPR; };
  auto k_chunk = [&](int i, int ln) { const int pp = ((w * NI + i) * 64 + ln) % CPR; return kswz<D>(k_key(i, ln), pp); };
  auto v_key = [&](int i, int ln) { return ((w * NI + i) & 1) * 32 + (ln >> 1); };
  auto v_chunk = [&](int i, int ln) { return 2 * ((w * NI + i) >> 1) + (ln & 1); };
  const unsigned kvo0 = (unsigned)(k_key(0, lane) * p.ldk * 2 + k_chunk(0, lane) * 16), vvo0 = (unsigned)(v_key(0, lane) * p.ldv * 2 + v_chunk(0, lane) * 16);
  const unsigned kvo1 = NI > 1 ? (unsigned)(k_key(NI - 1, lane) * p.ldk * 2 + k_chunk(NI - 1, lane) * 16) : 0u;
  const unsigned vvo1 = NI > 1 ? (unsigned)(v_key(NI - 1, lane) * p.ldv * 2 + v_chunk(NI - 1, lane) * 16) : 0u;
  auto dma_tile = [&](int t, int stg) {
    const char* kt = reinterpret_cast<const char*>(Kb + (long long)t * 64 * p.ldk);
    const char* vtp = reinterpret_cast<const char*>(Vb + (long long)t * 64 * p.ldv);
    const unsigned kl = lds0 + stg * KS_BYTES + (w * NI) * 1024;
    const unsigned vl = lds0 + NSTG * KS_BYTES + stg * VT_BYTES;
    unsigned ko0 = kvo0, ko1 = kvo1, vo0 = vvo0, vo1 = vvo1;
    if ((t + 1) * 64 > p.Lk) {   // workgroup-uniform: ragged last tile -> rows beyond Lk re-read the last key (masked by madd = -inf)
      const int last = p.Lk - 1 - t * 64;
      auto cl = [&](int key) { return key < last ? key : last; };
      int ln = lane;
      asm volatile("" : "+v"(ln));   // opaque: the geometry of this (at most one) tile is recomputed here instead of living in eight registers through the loop
      ko0 = (unsigned)(cl(k_key(0, ln)) * p.ldk * 2 + k_chunk(0, ln) * 16); vo0 = (unsigned)(cl(v_key(0, ln)) * p.ldv * 2 + v_chunk(0, ln) * 16);
      if (NI > 1) {
        ko1 = (unsigned)(cl(k_key(NI - 1, ln)) * p.ldk * 2 + k_chunk(NI - 1, ln) * 16); vo1 = (unsigned)(cl(v_key(NI - 1, ln)) * p.ldv * 2 + v_chunk(NI - 1, ln) * 16);
      }
    }
    attn_glds16(kt, ko0, kl);
    attn_glds16(vtp, vo0, vl + vsub_off((w * NI) >> 1) + ((w * NI) & 1) * 1024);
    if (NI > 1) {
      attn_glds16(kt, ko1, kl + 1024);
      attn_glds16(vtp, vo1, vl + vsub_off((w * NI + 1) >> 1) + ((w * NI + 1) & 1) * 1024);
    }
  };
  // `younger` tiles of this wave's DMA may stay in flight (VMEM retires in order; 2 NI instructions per tile); then the workgroup
  // barrier: every wave's share of the awaited tile has landed and nobody still reads the stage that is refilled next
  auto wait_tile_and_barrier = [&](int younger) {
    if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NI) : "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's LDS reads / writes are done
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  dma_tile(0, 0);   // (the ring is disjoint from the tables)
  if (NSTG > 2 && nt > 1) dma_tile(1, 1);
  // ---- the tables, while the first tile and the query fragments are in flight. Every thread's global reads of a table are issued before the first
  // is used (one memory latency per table instead of one per 256 entries: the prologue is a third of this kernel's time at 8 key tiles per workgroup)
  {
    // key j = u * 256 + tid: wave w of pass u holds exactly tile 4 u + w, so the tile flag is a ballot
    const int npass = (kpad + 255) >> 8;
    const unsigned char* km = p.kmask ? reinterpret_cast<const unsigned char*>(p.kmask) + (long long)b * p.Lkr : nullptr;
    for (int u0 = 0; u0 < npass; u0 += 4) {
      unsigned char mk[4];
      // (unconditional loads at clamped addresses, consumed by an empty asm: otherwise hipcc sinks each load into the branch that uses it and waits there)
      unsigned mw[4] = {1u, 1u, 1u, 1u};
      if (km) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int j = (u0 + u) * 256 + tid;
          mw[u] = km[j < p.Lk ? j : p.Lk - 1];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(mw[u]));
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) mk[u] = (unsigned char)mw[u];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int j = (u0 + u) * 256 + tid;
        const float v = j < p.Lk ? (mk[u] ? 0.0f : (CINIT ? kMaskNeg : -FLT_MAX)) : -INFINITY;
        if (j < kpad) madd[j] = v;
        const bool nz = j < kpad && v != 0.0f;
        const bool anyz = __any(nz);
        const int tile = (u0 + u) * 4 + w;
        if (lane == 0 && tile < nt) tflag[tile] = anyz ? 1 : 0;
      }
    }
  }
  if (MODE == ATTN_T5) {
    const float* rb = p.relbias + (long long)h * (2 * p.Lk - 1);
    const int nb = kpad + QB - 1;
    for (int j0 = tid; j0 < nb; j0 += 1024) {
      float bv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int delta = j0 + u * 256 - boff;                 // key - query
        const int ix = delta + p.Lk - 1;
        bv[u] = rb[ix < 0 ? 0 : (ix > 2 * p.Lk - 2 ? 2 * p.Lk - 2 : ix)];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(bv[u]));
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int delta = j0 + u * 256 - boff;
        if (!(delta > -p.Lk && delta < p.Lk)) bv[u] = 0.0f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (j0 + u * 256 < nb) btab[j0 + u * 256] = bv[u] * kLog2e;   // log2 domain, see below
    }
  }
  wait_tile_and_barrier(NSTG > 2 && nt > 1 ? 1 : 0);   // tile 0 has landed, the tables are visible
  // T5 with a table that is constant beyond +-bias_far (AttnArgs::bias_far): a key tile ALL of whose (key - query) lie beyond it for every query of
  // this WAVE takes the constant as one scalar operand instead of 32 per-score LDS reads (16 ds_read2_b32 + their waits: 7 % of the tile's issue
  // slots, profiles/r04_attention_ablation.txt: "nobias"); 43 % of the (wave, tile) pairs at L = 512, 69 % at L = 1024. Same values, same arithmetic.
  // (An index outside this workgroup's table belongs to a distance none of its tiles has.)
  const float cfar_pos = (far > 0 && boff + far < kpad + QB - 1) ? btab[boff + far] : 0.0f;
  const float cfar_neg = (far > 0 && boff - far >= 0) ? btab[boff - far] : 0.0f;
  // The query fragments were requested above with plain global loads. hipcc's wait-count pass carries "these loads may
  // still be pending" into the key-tile loop (it cannot know they landed during the first iteration) and therefore put
  // `s_waitcnt vmcnt(3..0)` in front of the first S^T MFMAs of EVERY iteration -- right behind the prefetch loads of the next
  // tile issued at the top of the iteration, i.e. the prefetch was waited for immediately and its latency was exposed on
  // every tile (T5 shape: 0.418 -> 0.390 ms with 32, 0.405 -> 0.371 ms with 64 queries per wave). Consuming the fragments
  // here (empty asm with the registers as in/out operands) makes the compiler wait for them ONCE, before the loop.
#pragma unroll
  for (int g = 0; g < QG; ++g)
#pragma unroll
    for (int dd = 0; dd < KD; ++dd) asm volatile("" : "+v"(qf[g][dd]));
  // per-phase shader-clock stamps of wave 0 (scripts/attn_micro.py STAMPS=1): compiled in only with -DVIMA_ATTN_STAMPS=1,
  // a run-time switch would split the loop body into basic blocks and pin the instruction schedule at every mark
#ifndef VIMA_ATTN_STAMPS
#define VIMA_ATTN_STAMPS 0
#endif
  long long ph[6] = {0, 0, 0, 0, 0, 0};
  constexpr bool kStamps = VIMA_ATTN_STAMPS != 0;
  const bool dbg = kStamps && p.dbg != nullptr;
  long long t_prev = dbg ? (long long)__builtin_readcyclecounter() : 0;
  auto mark = [&](int i) {
    if constexpr (kStamps) {
      if (dbg) { const long long t = (long long)__builtin_readcyclecounter(); ph[i] += t - t_prev; t_prev = t; }
    }
  };

  int stg = 0, stg_in = NSTG - 1;                             // stage of tile t / of the tile requested in iteration t
  for (int t = 0; t < nt; ++t) {
    if (t + NSTG - 1 < nt) dma_tile(t + NSTG - 1, stg_in);     // into the stage read in iteration t - 1 (all waves are past its barrier)
    mark(0);   // DMA of tile t + NSTG - 1 issued
    const char* ks = ks_base + stg * KS_BYTES;
    const char* vt = vt_base + stg * VT_BYTES + vlane;
    const int k0 = t * 64;
    // ---- S^T for the two 32-key sub-tiles (alternating the two accumulators per k-step measured 7 % slower); every K
    // fragment is read once and multiplied with the queries of all QG groups
    // T5 / cross mode: a tile with masked keys starts its accumulators from the additive key mask (0 / kMaskNeg / -inf per key row) instead of adding it to
    // the 32 scores afterwards: the huge negative absorbs the dot product exactly as it absorbed the finished score (the benchmark masks 10 % of the prompt
    // objects, i.e. nearly every tile takes this path). The causal mode keeps the explicit add: there the -1e4 fill REPLACES the score before the mask goes on.
    f32x16_t s[QG][2];
    const bool masked_tile = tflag[t] != 0;                    // workgroup-uniform
    auto st_mfmas = [&](auto init_c) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int g = 0; g < QG; ++g) {
          if constexpr (decltype(init_c)::value) {
#pragma unroll
            for (int gg = 0; gg < 4; ++gg) {
              const float4 ma = *reinterpret_cast<const float4*>(madd + k0 + sub * 32 + 8 * gg + 4 * hi);
              s[g][sub][4 * gg + 0] = ma.x; s[g][sub][4 * gg + 1] = ma.y; s[g][sub][4 * gg + 2] = ma.z; s[g][sub][4 * gg + 3] = ma.w;
            }
          } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[g][sub][r] = 0.f;
          }
        }
        const int row = sub * 32 + l31;
#pragma unroll
        for (int dd = 0; dd < KD; ++dd) {
          const uint4 u = *reinterpret_cast<const uint4*>(ks + row * ROWB + (kswz<D>(row, dd * 2 + hi) << 4));
#pragma unroll
          for (int g = 0; g < QG; ++g)
            s[g][sub] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, u), qf[g][dd], s[g][sub], 0, 0, 0);
        }
      }
    };
    if (CINIT && masked_tile) st_mfmas(std::true_type{});
    else st_mfmas(std::false_type{});
    mark(1);   // S^T MFMAs issued
    // ---- scores -> probabilities (fp32), in the LOG2 domain: x' = log2(e) * (scaled score + bias) is ONE fma per element
    // (the bias table is pre-multiplied), p = exp2(x' - m'). The additive key mask (0 / -finfo.max / -inf) is added
    // unscaled: it absorbs x' exactly as it absorbs x in the reference, so fully masked rows still come out uniform.
    // The running maximum is only raised when some row's tile maximum exceeds it by more than 2^8 (softmax is
    // invariant to the reference point; p <= 256 keeps bf16 relative precision and fp32 sums exact enough), which
    // removes the O^T rescale from almost every tile.
    uint32_t pk[QG][2][8];
    const float csc = (MODE == ATTN_T5 ? 1.0f : p.scale) * kLog2e;
    // wave-uniform: every (key - query) of this tile and wave at or beyond +far / -far
    const bool far_pos = MODE == ATTN_T5 && far > 0 && k0 - qw1 >= far, far_neg = MODE == ATTN_T5 && far > 0 && k0 + 63 - qw0 <= -far;
#pragma unroll
    for (int g = 0; g < QG; ++g) {
      float x[2][16];
      float mt = -INFINITY;
      const float* bq = btab + (boff - qi[g] + k0 + 4 * hi);    // T5: bias of key (k0 + 4hi + j) is bq[j]
      if (MODE == ATTN_T5 && (far_pos || far_neg)) {
        const float cb = far_pos ? cfar_pos : cfar_neg;
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int r = 0; r < 16; ++r) x[sub][r] = __builtin_fmaf(s[g][sub][r], csc, cb);
      } else {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub) {
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const int kl = sub * 32 + 8 * gg;                     // key = k0 + kl + 4*hi + e
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float sv = s[g][sub][4 * gg + e];
            float v;
            if (MODE == ATTN_T5) v = __builtin_fmaf(sv, csc, bq[kl + e]);
            else if (MODE == ATTN_CROSS) v = sv * csc;
            else v = (k0 + kl + 4 * hi + e > qi[g] + p.q_off) ? -1e4f * kLog2e : sv * csc;
            x[sub][4 * gg + e] = v;
          }
        }
      }
      }
      if (masked_tile && !CINIT) {
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
          for (int gg = 0; gg < 4; ++gg) {
            const float4 ma = *reinterpret_cast<const float4*>(madd + k0 + sub * 32 + 8 * gg + 4 * hi);
            x[sub][4 * gg + 0] += ma.x; x[sub][4 * gg + 1] += ma.y; x[sub][4 * gg + 2] += ma.z; x[sub][4 * gg + 3] += ma.w;
          }
      }
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; ++r) mt = fmaxf(mt, x[sub][r]);
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      if (__any(mt > m_run[g] + 8.0f)) {                        // wave-uniform, rare after the first tiles
        const float m_new = fmaxf(m_run[g], mt);
        const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);   // exp2(-inf) = 0 on the first tile
        l_run[g] *= alpha;
#pragma unroll
        for (int it = 0; it < OT; ++it)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[g][it][r] *= alpha;
        m_run[g] = m_new;
      }
      // the subtraction of the running maximum and the row sum as PACKED fp32 operations (v_pk_add_f32: two elements per
      // issue slot; exp2 itself is a quarter-rate transcendental and stays scalar)
      const f32x2_t mm = {m_run[g], m_run[g]};
      f32x2_t rs2 = {0.f, 0.f};
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const f32x2_t dlt = f32x2_t{x[sub][r], x[sub][r + 1]} - mm;
          const f32x2_t pp = {__builtin_amdgcn_exp2f(dlt[0]), __builtin_amdgcn_exp2f(dlt[1])};
          pk[g][sub][r >> 1] = pack2_bf16(pp[0], pp[1]);
          rs2 += pp;
        }
      float rs = rs2[0] + rs2[1];
      rs += __shfl_xor(rs, 32, 64);
      l_run[g] += rs;
    }
    mark(2);   // softmax
    // ---- O^T += V^T . P^T: every V fragment is read once and multiplied with the probabilities of all QG groups
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        bf16x8_t pf[QG];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
          uint4 pu;
          pu.x = pk[g][sub][half * 4 + 0]; pu.y = pk[g][sub][half * 4 + 1]; pu.z = pk[g][sub][half * 4 + 2]; pu.w = pk[g][sub][half * 4 + 3];
          pf[g] = __builtin_bit_cast(bf16x8_t, pu);
        }
#pragma unroll
        for (int it = 0; it < OT; ++it) {
          // keys sub*32 + half*16 + 4 hi + {0..3 | 8..11}, d = it*32 + l31
          const char* vr = vt + vsub_off(2 * it) + (sub * 32 + half * 16) * 32;
          const uint2 a0 = lds_read_tr16(vr);
          const uint2 a1 = lds_read_tr16(vr + 8 * 32);
          uint4 vu;
          vu.x = a0.x; vu.y = a0.y; vu.z = a1.x; vu.w = a1.y;
#pragma unroll
          for (int g = 0; g < QG; ++g)
            ot[g][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vu), pf[g], ot[g][it], 0, 0, 0);
        }
      }
    mark(3);   // PV MFMAs issued
    {
      int last = t + NSTG - 1;                                 // tiles requested so far: up to min(nt - 1, t + NSTG - 1)
      last = last < nt - 1 ? last : nt - 1;
      wait_tile_and_barrier(last - (t + 1));
    }
    mark(5);   // tile t + 1 landed, barrier
    stg = stg + 1 == NSTG ? 0 : stg + 1;
    stg_in = stg_in + 1 == NSTG ? 0 : stg_in + 1;
  }
  if (dbg && tid == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) p.dbg[(long long)blockIdx.x * 8 + i] = ph[i];
  }
#pragma unroll
  for (int g = 0; g < QG; ++g) {
    if (qi[g] < p.Lq) {
      const float inv = 1.0f / l_run[g];
      bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + ((long long)b * p.Lq + qi[g]) * p.ldo + h * D;
#pragma unroll
      for (int it = 0; it < OT; ++it)
#pragma unroll
        for (int gg = 0; gg < 4; ++gg) {
          const int d = it * 32 + 8 * gg + 4 * hi;
          store4(op + d, make_float4(ot[g][it][4 * gg] * inv, ot[g][it][4 * gg + 1] * inv, ot[g][it][4 * gg + 2] * inv,
                                     ot[g][it][4 * gg + 3] * inv));
        }
    }
  }
}

// =============================================================================================== MFMA flash attention, split keys
// Few queries (Lq <= 32: the cross-attention of one env step has 8, XAttention components.py:184-214), many keys: one
// workgroup = 4 waves = ONE (batch, head); 128-key tiles are staged cooperatively in LDS (coalesced 16-B loads, next tile
// prefetched in registers) and each wave takes one 32-key quarter of every tile with its own online-softmax state; the
// four partial (max, sum, O^T) states are merged through LDS at the end ("flash decoding" inside a workgroup).
// 4x the parallelism of the one-wave kernel per (batch, head) and no per-lane global loads in the loop.
constexpr int SPLIT_TK = 128;

template <int D, int MODE>
__global__ __launch_bounds__(256, D == 32 ? 4 : 2) void attn_split_kernel(const AttnDev p) {
  extern __shared__ __attribute__((aligned(16))) char smems[];
  constexpr int KD = D / 16, OT = D / 32, CPR = D / 8;
  constexpr int ROWB = D * 2;
  constexpr int KS_BYTES = SPLIT_TK * ROWB;
  constexpr int VTS = SPLIT_TK + 4;                            // bf16 elements per V^T row (264 B: conflict-free b64 reads)
  constexpr int VT_BYTES = D * VTS * 2;
  constexpr int CH = SPLIT_TK * CPR / 256;                     // 16-B chunks per thread per tile
  char* ks_base = smems;
  char* vt_base = smems + 2 * KS_BYTES;                      // [2] x V image: D/16 sub-tiles of [128 keys][16 d]
  float* madd = reinterpret_cast<float*>(smems + 2 * KS_BYTES + 2 * VT_BYTES);
  const int nt = (p.Lk + SPLIT_TK - 1) / SPLIT_TK;

  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int hi = lane >> 5, l31 = lane & 31;
  // Adjacent heads share 128-byte lines of the K/V rows (a head is 64 B wide at D = 32): workgroups id and id + 8 run
  // back to back on the same XCD (id % 8), so they are given heads 2k and 2k + 1 and the second one hits in that L2.
  const int bid = blockIdx.x;
  const int bh = (((bid >> 4) << 3) + (bid & 7)) * 2 + ((bid >> 3) & 1);
  if (bh >= p.B * p.H) return;
  const int h = bh % p.H, b = bh / p.H;
  const int qi = l31;
  const bf16_t* Q = reinterpret_cast<const bf16_t*>(p.q);
  const bf16_t* K = reinterpret_cast<const bf16_t*>(p.k);
  const bf16_t* V = reinterpret_cast<const bf16_t*>(p.v);

  // The additive key mask table. Its reads are ISSUED here, ahead of the query and K / V loads, and consumed behind them (unconditional clamped loads + an empty asm:
  // as a loop of conditional loads hipcc waited for each one where it was issued, i.e. a workgroup that lives for four K / V round trips spent two more on its mask
  // before it requested its first tile)
  const uint8_t* km = p.kmask ? p.kmask + (long long)b * p.Lkr : nullptr;
  const int nk = nt * SPLIT_TK;
  unsigned mw0[4] = {1u, 1u, 1u, 1u};
  if (km) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = tid + u * 256;
      mw0[u] = km[j < p.Lk ? j : p.Lk - 1];
    }
  }
  const int qrow = qi < p.Lq ? qi : p.Lq - 1;
  bf16x8_t qf[KD];
#pragma unroll
  for (int dd = 0; dd < KD; ++dd) {
    const uint4 u = *reinterpret_cast<const uint4*>(Q + ((long long)b * p.Lq + qrow) * p.ldq + h * D + dd * 16 + hi * 8);
    qf[dd] = __builtin_bit_cast(bf16x8_t, u);
  }
  f32x16_t ot[OT];
#pragma unroll
  for (int it = 0; it < OT; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[it][r] = 0.f;
  float m_run = -INFINITY, l_run = 0.f;

  // staging registers (named scalars: arrays captured by the lambdas stay in scratch memory, see attn_mfma4_kernel). TWO sets: the
  // loads of tiles t+1 AND t+2 are in flight while tile t is multiplied (tile tau uses set tau & 1). The kernel reads every K / V
  // byte once and a workgroup lives for Lk / 128 dependent HBM round trips; with one tile in flight per workgroup (4 workgroups per
  // CU = 64 KB) the cross attention of a batch-256 step ran at 3.6 TB/s.
  static_assert(CH == 2 || CH == 4, "staging registers");
  uint4 kreg0, kreg1, kreg2, kreg3, vreg0, vreg1, vreg2, vreg3;
  uint4 kregb0, kregb1, kregb2, kregb3, vregb0, vregb1, vregb2, vregb3;
  const bf16_t* Kbh = K + b * p.k_bs + h * p.k_hs;     // this (batch, head)'s first key / value row
  const bf16_t* Vbh = V + b * p.v_bs + h * p.v_hs;
  auto gaddr = [&](int t, int i, const bf16_t* base, int ld) {
    const int id = tid + i * 256;
    const int key = id / CPR, c = id % CPR;
    int row = t * SPLIT_TK + key;
    row = row < p.Lk ? row : p.Lk - 1;
    return reinterpret_cast<const uint4*>(base + (long long)row * ld + c * 8);
  };
  auto gload = [&](int t, auto set) {
    if constexpr (decltype(set)::value == 0) {
      kreg0 = *gaddr(t, 0, Kbh, p.ldk); vreg0 = *gaddr(t, 0, Vbh, p.ldv);
      kreg1 = *gaddr(t, 1, Kbh, p.ldk); vreg1 = *gaddr(t, 1, Vbh, p.ldv);
      if constexpr (CH > 2) {
        kreg2 = *gaddr(t, 2, Kbh, p.ldk); vreg2 = *gaddr(t, 2, Vbh, p.ldv);
        kreg3 = *gaddr(t, 3, Kbh, p.ldk); vreg3 = *gaddr(t, 3, Vbh, p.ldv);
      }
    } else {
      kregb0 = *gaddr(t, 0, Kbh, p.ldk); vregb0 = *gaddr(t, 0, Vbh, p.ldv);
      kregb1 = *gaddr(t, 1, Kbh, p.ldk); vregb1 = *gaddr(t, 1, Vbh, p.ldv);
      if constexpr (CH > 2) {
        kregb2 = *gaddr(t, 2, Kbh, p.ldk); vregb2 = *gaddr(t, 2, Vbh, p.ldv);
        kregb3 = *gaddr(t, 3, Kbh, p.ldk); vregb3 = *gaddr(t, 3, Vbh, p.ldv);
      }
    }
  };
  auto lstore1 = [&](int buf, int i, const uint4& kr, const uint4& vr) {
    const int id = tid + i * 256;
    const int key = id / CPR, c = id % CPR;
    *reinterpret_cast<uint4*>(ks_base + buf * KS_BYTES + key * ROWB + (kswz<D>(key, c) << 4)) = kr;
    *reinterpret_cast<uint4*>(vt_base + buf * VT_BYTES + vsub_off<SPLIT_TK>(c >> 1) + key * 32 + (c & 1) * 16) = vr;
  };
  auto lstore = [&](int buf, auto set) {
    if constexpr (decltype(set)::value == 0) {
      lstore1(buf, 0, kreg0, vreg0);
      lstore1(buf, 1, kreg1, vreg1);
      if constexpr (CH > 2) {
        lstore1(buf, 2, kreg2, vreg2);
        lstore1(buf, 3, kreg3, vreg3);
      }
    } else {
      lstore1(buf, 0, kregb0, vregb0);
      lstore1(buf, 1, kregb1, vregb1);
      if constexpr (CH > 2) {
        lstore1(buf, 2, kregb2, vregb2);
        lstore1(buf, 3, kregb3, vregb3);
      }
    }
  };
  // transposing V reads (see attn_mfma4_kernel): 16-lane group g reads sub-tile g&1, keys 4 hi + i/4, column quad i%4
  const int vlane = vsub_off<SPLIT_TK>((lane >> 4) & 1) + (w * 32 + 4 * (lane >> 5) + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;

  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, 1>;
  gload(0, Set0{});
  if (nt > 1) gload(1, Set1{});
  if (km) {
#pragma unroll
    for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(mw0[u]));
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int j = tid + u * 256;
    if (j < nk) madd[j] = j < p.Lk ? ((mw0[u] & 0xffu) ? 0.0f : -FLT_MAX) : -INFINITY;
  }
  for (int j0 = tid + 1024; j0 < nk; j0 += 1024) {             // prompts beyond 1024 keys
    unsigned mw[4] = {1u, 1u, 1u, 1u};
    if (km) {
#pragma unroll
      for (int u = 0; u < 4; ++u) mw[u] = km[j0 + u * 256 < p.Lk ? j0 + u * 256 : p.Lk - 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) asm volatile("" : "+v"(mw[u]));
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int j = j0 + u * 256;
      if (j < nk) madd[j] = j < p.Lk ? ((mw[u] & 0xffu) ? 0.0f : -FLT_MAX) : -INFINITY;
    }
  }
  lstore(0, Set0{});
  __syncthreads();
  // wait for the query fragments ONCE, here: otherwise the wait-count pass waits for them in front of the first MFMAs of every
  // iteration, i.e. for the prefetch loads of the next tile issued just before (see attn_mfma4_kernel)
#pragma unroll
  for (int dd = 0; dd < KD; ++dd) asm volatile("" : "+v"(qf[dd]));
  auto tile = [&](int t, auto cur_set, auto nxt_set) {        // tile t sits in LDS buffer t & 1; tile t+1 is in register set nxt_set
    const int buf = t & 1;
    if (t + 2 < nt) gload(t + 2, cur_set);                     // set of tile t is free (tile t is in LDS)
    const char* ks = ks_base + buf * KS_BYTES;
    const char* vt = vt_base + buf * VT_BYTES + vlane;
    const int kb = t * SPLIT_TK + w * 32;                      // first key of this wave's quarter
    if (kb < p.Lk) {                                           // wave-uniform: quarter not entirely beyond the keys
      f32x16_t s;
#pragma unroll
      for (int r = 0; r < 16; ++r) s[r] = 0.f;
      const int row = w * 32 + l31;
#pragma unroll
      for (int dd = 0; dd < KD; ++dd) {
        const uint4 u = *reinterpret_cast<const uint4*>(ks + row * ROWB + (kswz<D>(row, dd * 2 + hi) << 4));
        s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, u), qf[dd], s, 0, 0, 0);
      }
      // log2-domain softmax with a lazily raised running maximum (see attn_mfma4_kernel)
      float x[16];
      float mt = -INFINITY;
      const float csc = p.scale * kLog2e;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int key = kb + 8 * g + 4 * hi;
        const float4 ma = *reinterpret_cast<const float4*>(madd + key);
        const float mav[4] = {ma.x, ma.y, ma.z, ma.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float v = s[4 * g + e] * csc;
          if (MODE == ATTN_CAUSAL && key + e > qi + p.q_off) v = -1e4f * kLog2e;
          v = v + mav[e];
          x[4 * g + e] = v;
          mt = fmaxf(mt, v);
        }
      }
      mt = fmaxf(mt, __shfl_xor(mt, 32, 64));
      if (__any(mt > m_run + 8.0f)) {
        const float m_new = fmaxf(m_run, mt);                  // finite: key kb is in range
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
        l_run *= alpha;
#pragma unroll
        for (int it = 0; it < OT; ++it)
#pragma unroll
          for (int r = 0; r < 16; ++r) ot[it][r] *= alpha;
        m_run = m_new;
      }
      uint32_t pk[8];
      float rs = 0.f;
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float p0 = __builtin_amdgcn_exp2f(x[r] - m_run);
        const float p1 = __builtin_amdgcn_exp2f(x[r + 1] - m_run);
        pk[r >> 1] = pack2_bf16(p0, p1);
        rs += p0 + p1;
      }
      rs += __shfl_xor(rs, 32, 64);
      l_run += rs;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        uint4 pu;
        pu.x = pk[half * 4 + 0]; pu.y = pk[half * 4 + 1]; pu.z = pk[half * 4 + 2]; pu.w = pk[half * 4 + 3];
        const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pu);
#pragma unroll
        for (int it = 0; it < OT; ++it) {
          const char* vr = vt + vsub_off<SPLIT_TK>(2 * it) + (half * 16) * 32;
          const uint2 a0 = lds_read_tr16(vr);
          const uint2 a1 = lds_read_tr16(vr + 8 * 32);
          uint4 vu;
          vu.x = a0.x; vu.y = a0.y; vu.z = a1.x; vu.w = a1.y;
          ot[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, vu), pf, ot[it], 0, 0, 0);
        }
      }
    }
    if (t + 1 < nt) lstore(buf ^ 1, nxt_set);
    __syncthreads();
  };
  for (int t = 0; t < nt; t += 2) {
    tile(t, Set0{}, Set1{});
    if (t + 1 < nt) tile(t + 1, Set1{}, Set0{});
  }
  // ---- merge the four partial softmax states (stage buffers are free after the loop's last barrier)
  constexpr int NV = OT * 16 + 2;
  float* comb = reinterpret_cast<float*>(smems);               // [4][NV][64]
  float* mine = comb + (w * NV) * 64 + lane;
#pragma unroll
  for (int it = 0; it < OT; ++it)
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[(it * 16 + r) * 64] = ot[it][r];
  mine[(OT * 16) * 64] = m_run;
  mine[(OT * 16 + 1) * 64] = l_run;
  __syncthreads();
  if (w == 0 && qi < p.Lq) {
    float mw[4], m_all = -INFINITY;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      mw[j] = comb[(j * NV + OT * 16) * 64 + lane];
      m_all = fmaxf(m_all, mw[j]);
    }
    float sc[4], l_all = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      sc[j] = __builtin_amdgcn_exp2f(mw[j] - m_all);   // a wave that saw no key has m = -inf -> weight 0
      l_all += comb[(j * NV + OT * 16 + 1) * 64 + lane] * sc[j];
    }
    const float inv = 1.0f / l_all;
    bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + ((long long)b * p.Lq + qi) * p.ldo + h * D;
#pragma unroll
    for (int it = 0; it < OT; ++it)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float o4[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          float acc = 0.f;
#pragma unroll
          for (int j = 0; j < 4; ++j) acc += comb[(j * NV + it * 16 + 4 * g + e) * 64 + lane] * sc[j];
          o4[e] = acc * inv;
        }
        store4(op + it * 32 + 8 * g + 4 * hi, make_float4(o4[0], o4[1], o4[2], o4[3]));
      }
  }
}

inline AttnDev to_dev(const AttnArgs& a) {
  AttnDev d;
  d.q = a.q; d.ldq = a.ldq; d.k = a.k; d.ldk = a.ldk; d.v = a.v; d.ldv = a.ldv; d.out = a.out; d.ldo = a.ldo;
  d.kmask = a.kmask; d.relbias = a.relbias; d.bias_far = (a.mode == ATTN_T5 && a.bias_far > 0 && a.bias_far < a.Lk && a.bias_far < a.Lq) ? a.bias_far : 0; d.B = a.B; d.H = a.H; d.Lq = a.Lq; d.Lk = a.Lk; d.scale = a.scale;
  d.mode = a.mode;
  d.Lkr = a.Lk_rows > 0 ? a.Lk_rows : a.Lk;
  d.k_bs = a.k_bs > 0 ? a.k_bs : (long long)d.Lkr * a.ldk; d.v_bs = a.v_bs > 0 ? a.v_bs : (long long)d.Lkr * a.ldv;
  d.k_hs = a.k_hs > 0 ? a.k_hs : a.D; d.v_hs = a.v_hs > 0 ? a.v_hs : a.D;
  d.q_off = a.q_off;
  d.dbg = a.tune ? a.tune->attn_dbg : nullptr;
  return d;
}

}  // namespace

int launch_vit_attn(const void* qkv, void* out, int M, int S, int W, int heads, bool is_bf16, hipStream_t st, void* out8, float inv8) {
  if (M <= 0) return 0;
  if (S > 16 || W / heads != 32 || W % heads || (out8 && !is_bf16)) return (int)hipErrorInvalidValue;
  static const bool lds_path = [] { const char* e = getenv("VIMA_VIT_ATTN_LDS"); return !(e && e[0] == '0'); }();
  if (is_bf16 && S == VA_S && W == VA_W && heads == VA_H && lds_path) {   // by shape only, never by the number of crops
    hipLaunchKernelGGL(vit_attn_lds_kernel, dim3((unsigned)((M + 1) / 2)), dim3(256), 2 * VA_S * VA_ROWB, st, (const bf16_t*)qkv, (bf16_t*)out, M,
                       (uint8_t*)out8, inv8);
    return (int)hipGetLastError();
  }
  const long long total = (long long)M * S * heads;
  const unsigned g = (unsigned)((total + 255) / 256);
  if (S > 8) {   // 9 .. 16 tokens (the baseline policies' frames: cls + 8 patches): the same body with 16 score registers
    if (is_bf16) hipLaunchKernelGGL((vit_attn_kernel<bf16_t, 16>), dim3(g), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out, total, S, W, heads, (uint8_t*)out8, inv8);
    else hipLaunchKernelGGL((vit_attn_kernel<float, 16>), dim3(g), dim3(256), 0, st, (const float*)qkv, (float*)out, total, S, W, heads);
    return (int)hipGetLastError();
  }
  if (is_bf16) hipLaunchKernelGGL(vit_attn_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)qkv, (bf16_t*)out, total, S, W, heads, (uint8_t*)out8, inv8);
  else hipLaunchKernelGGL(vit_attn_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)qkv, (float*)out, total, S, W, heads);
  return (int)hipGetLastError();
}

int launch_vit_attn_cls(const void* q, const void* kv, void* out, int M, int S, int W, int heads, bool is_bf16, hipStream_t st, void* out8,
                        float inv8) {
  if (M <= 0) return 0;
  if (S > 8 || W / heads != 32 || W % heads || (out8 && !is_bf16)) return (int)hipErrorInvalidValue;
  const long long total = (long long)M * heads;
  const unsigned g = (unsigned)((total + 255) / 256);
  if (is_bf16) hipLaunchKernelGGL(vit_attn_cls_kernel<bf16_t>, dim3(g), dim3(256), 0, st, (const bf16_t*)q, (const bf16_t*)kv, (bf16_t*)out, total, S, W, heads, (uint8_t*)out8, inv8);
  else hipLaunchKernelGGL(vit_attn_cls_kernel<float>, dim3(g), dim3(256), 0, st, (const float*)q, (const float*)kv, (float*)out, total, S, W, heads);
  return (int)hipGetLastError();
}

template <typename T>
static int launch_generic_t(const AttnDev& d, const AttnArgs& a, hipStream_t st) {
  dim3 grid((unsigned)a.Lq, (unsigned)a.H, (unsigned)a.B);
  const size_t sh = (size_t)a.Lk * sizeof(float);
  switch (a.D) {
    case 16: hipLaunchKernelGGL((attn_generic_kernel<T, 16>), grid, dim3(64), sh, st, d); break;
    case 32: hipLaunchKernelGGL((attn_generic_kernel<T, 32>), grid, dim3(64), sh, st, d); break;
    case 64: hipLaunchKernelGGL((attn_generic_kernel<T, 64>), grid, dim3(64), sh, st, d); break;
    case 128: hipLaunchKernelGGL((attn_generic_kernel<T, 128>), grid, dim3(64), sh, st, d); break;
    // the Perceiver of VIMAFlamingoPolicy has 8 heads whatever the width: head dims 40 / 48 / 80 / 96 for E = 320 ... 768
    case 40: hipLaunchKernelGGL((attn_generic_kernel<T, 40>), grid, dim3(64), sh, st, d); break;
    case 48: hipLaunchKernelGGL((attn_generic_kernel<T, 48>), grid, dim3(64), sh, st, d); break;
    case 80: hipLaunchKernelGGL((attn_generic_kernel<T, 80>), grid, dim3(64), sh, st, d); break;
    case 96: hipLaunchKernelGGL((attn_generic_kernel<T, 96>), grid, dim3(64), sh, st, d); break;
    default: return (int)hipErrorInvalidValue;
  }
  return (int)hipGetLastError();
}

// head dims 16 / 32 / 40 / 48 / 64 / 80 / 96 / 128 (the MFMA flash kernels cover 32 and 64; the others run here)
int launch_attn_generic(const AttnArgs& a, bool is_bf16, hipStream_t st) {
  if (a.B <= 0 || a.Lq <= 0 || a.Lk <= 0) return 0;
  if (a.mode == ATTN_T5 && !a.relbias) return (int)hipErrorInvalidValue;
  const AttnDev d = to_dev(a);
  return is_bf16 ? launch_generic_t<bf16_t>(d, a, st) : launch_generic_t<float>(d, a, st);
}


constexpr int kAttn4MinLq = 64;   // default: queries per (batch, head) from which the 4-wave LDS-shared kernel is used

template <int D, int MODE, int QG>
static int launch_mfma4_qg(const AttnDev& d, const AttnArgs& a, hipStream_t st) {
  const int nt = (a.Lk + 63) / 64;
  const int nq = (a.Lq + 128 * QG - 1) / (128 * QG);
  constexpr int NSTG = QG > 1 ? VIMA_ATTN_NSTG2 : VIMA_ATTN_NSTG1;   // as in the kernel
  const size_t sh = NSTG * (64 * D * 2) + NSTG * (D * VT4_STRIDE * 2) + (size_t)nt * 64 * 4 + (size_t)((nt + 3) & ~3) * 4 +
                    (MODE == ATTN_T5 ? (size_t)(128 * QG + nt * 64) * 4 : 0);
  if (sh > 160 * 1024) return (int)hipErrorInvalidValue;
  static PerDeviceOnce attr;   // per instantiation, per device
  {
    const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma4_kernel<D, MODE, QG>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
  }
  const int bh8 = (a.B * a.H + 7) / 8;
  dim3 grid((unsigned)(bh8 * 8 * nq), 1, 1);
  hipLaunchKernelGGL((attn_mfma4_kernel<D, MODE, QG>), grid, dim3(256), sh, st, d);
  return (int)hipGetLastError();
}
template <int D, int MODE>
static int launch_mfma4(const AttnDev& d, const AttnArgs& a, hipStream_t st) {
  // Option attn_qg = 2: 64 queries per wave from 256 queries on (identical results: every query row is processed exactly as
  // before; not for the causal mode, whose D = 64 instantiation would spill). It was the default in round 2 (register-staged K / V:
  // T5 shape 0.418 -> 0.405 ms); with the LDS-DMA ring 32 queries per wave (three workgroups per CU) are 3-4 % faster at the T5 shape
  // on every box measured (0.364 / 0.379, 0.379 / 0.392, 0.388 / 0.398 ms; L = 1024: 1.237 / 1.265 ms) and equal at D = 32.
  const int qg = (a.tune && a.tune->attn_qg > 0) ? a.tune->attn_qg : 1;
  if constexpr (MODE != ATTN_CAUSAL) {
    if (qg == 2 && a.Lq >= 256) return launch_mfma4_qg<D, MODE, 2>(d, a, st);
  }
  return launch_mfma4_qg<D, MODE, 1>(d, a, st);
}

template <int D, int MODE>
static int launch_split(const AttnDev& d, const AttnArgs& a, hipStream_t st) {
  const int nt = (a.Lk + SPLIT_TK - 1) / SPLIT_TK;
  size_t sh = 2 * (SPLIT_TK * D * 2) + 2 * (D * (SPLIT_TK + 4) * 2) + (size_t)nt * SPLIT_TK * 4;
  const size_t comb = (size_t)4 * (D / 32 * 16 + 2) * 64 * 4;
  if (sh < comb) sh = comb;
  if (sh > 160 * 1024) return (int)hipErrorInvalidValue;
  static PerDeviceOnce attr;   // per instantiation, per device
  {
    const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_split_kernel<D, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((attn_split_kernel<D, MODE>), dim3((unsigned)(((a.B * a.H + 15) / 16) * 16)), dim3(256), sh, st, d);
  return (int)hipGetLastError();
}

int launch_attn_mfma(const AttnArgs& a, hipStream_t st) {
  if (a.B <= 0 || a.Lq <= 0 || a.Lk <= 0) return 0;
  if (a.D != 32 && a.D != 64) return (int)hipErrorInvalidValue;
  if (a.mode == ATTN_T5 && !a.relbias) return (int)hipErrorInvalidValue;
  // 16-byte fragment loads: rows and head offsets must be 16-B aligned
  if ((a.ldq % 8) || (a.ldk % 8) || (a.ldv % 8) || (a.ldo % 4)) return (int)hipErrorInvalidValue;
  const AttnDev d = to_dev(a);
  const int use_split = (a.tune && a.tune->attn_split >= 0) ? a.tune->attn_split : 1;   // split-key kernel for Lq <= 32
  const int min_lq4 = (a.tune && a.tune->attn4_min_lq >= 0) ? a.tune->attn4_min_lq : kAttn4MinLq;
  if (use_split && a.Lq <= 32 && a.mode != ATTN_T5 && a.Lk >= 64) {
    if (a.D == 32) return a.mode == ATTN_CROSS ? launch_split<32, ATTN_CROSS>(d, a, st) : launch_split<32, ATTN_CAUSAL>(d, a, st);
    return a.mode == ATTN_CROSS ? launch_split<64, ATTN_CROSS>(d, a, st) : launch_split<64, ATTN_CAUSAL>(d, a, st);
  }
  if (a.Lq >= min_lq4) {
    if (a.D == 32) {
      if (a.mode == ATTN_T5) return launch_mfma4<32, ATTN_T5>(d, a, st);
      if (a.mode == ATTN_CROSS) return launch_mfma4<32, ATTN_CROSS>(d, a, st);
      return launch_mfma4<32, ATTN_CAUSAL>(d, a, st);
    }
    if (a.mode == ATTN_T5) return launch_mfma4<64, ATTN_T5>(d, a, st);
    if (a.mode == ATTN_CROSS) return launch_mfma4<64, ATTN_CROSS>(d, a, st);
    return launch_mfma4<64, ATTN_CAUSAL>(d, a, st);
  }
  dim3 grid((unsigned)((a.Lq + 31) / 32), (unsigned)a.H, (unsigned)a.B);
#define VIMA_ATTN(D_, M_) hipLaunchKernelGGL((attn_mfma_kernel<D_, M_>), grid, dim3(64), 0, st, d)
  if (a.D == 32) {
    if (a.mode == ATTN_T5) VIMA_ATTN(32, ATTN_T5);
    else if (a.mode == ATTN_CROSS) VIMA_ATTN(32, ATTN_CROSS);
    else VIMA_ATTN(32, ATTN_CAUSAL);
  } else {
    if (a.mode == ATTN_T5) VIMA_ATTN(64, ATTN_T5);
    else if (a.mode == ATTN_CROSS) VIMA_ATTN(64, ATTN_CROSS);
    else VIMA_ATTN(64, ATTN_CAUSAL);
  }
#undef VIMA_ATTN
  return (int)hipGetLastError();
}

}  // namespace vima
