// HBM-bound helper kernels of the VIMA policy hot path (gfx950, wave64): normalisation, image patchify,
// token plumbing. Each cites the reference lines it replaces. fp32 statistics everywhere; the operand type T of
// the matrix-core GEMMs (bf16, or fp32 in parity mode) only appears at the GEMM-input boundary.
#include "kernels.h"
#include <stdlib.h>

namespace vima {
namespace {

// ---------------------------------------------------------------------------------------------------------------
// LayerNorm / T5 RMSNorm: one wave per row, row held in registers (E <= 64*4*MAXV).
//   nn.LayerNorm eps 1e-5 (components.py:19,21,128,135; vit.py:164,168,204,214) ; HF T5LayerNorm eps 1e-6 (no mean, no bias)
// ---------------------------------------------------------------------------------------------------------------
constexpr int LN_MAXV = 4;  // float4 per lane -> E <= 1024

template <typename T, typename TIN = float>
__global__ __launch_bounds__(256) void layernorm_kernel(const TIN* __restrict__ in, long long ldin,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float eps, int rms, int rows, int E, float* out32, T* outT,
                                                         uint8_t* out8 = nullptr, float inv8 = 1.0f) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const TIN* x = in + (long long)row * ldin;
  const int nv = E >> 2;  // float4 count (E % 4 == 0)
  float4 v[LN_MAXV];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      v[i] = load4(x + c * 4);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    } else {
      v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  float mean = 0.f;
  if (!rms) mean = wave_sum(s) / (float)E;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
      q += (a * a + b * b) + (cc * cc + d * d);
    }
  }
  const float var = wave_sum(q) / (float)E;
  const float rstd = rsqrtf(var + eps);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      const float4 g = *reinterpret_cast<const float4*>(gamma + c * 4);
      float4 o;
      o.x = (v[i].x - mean) * rstd * g.x;
      o.y = (v[i].y - mean) * rstd * g.y;
      o.z = (v[i].z - mean) * rstd * g.z;
      o.w = (v[i].w - mean) * rstd * g.w;
      if (beta) {
        const float4 bb = *reinterpret_cast<const float4*>(beta + c * 4);
        o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w;
      }
      if (out32) store4(out32 + (long long)row * E + c * 4, o);
      if (outT) store4(outT + (long long)row * E + c * 4, o);
      if (out8) *reinterpret_cast<uint32_t*>(out8 + (long long)row * E + c * 4) = pack4_fp8(o, inv8);   // precision "fp8": e4m3 copy
    }
  }
}

// TWO LayerNorms in a row on fp32 rows: y = LN_a(x) -> out32 / outT (the decoder block's post-LN ln_2, components.py:37), then
// z = LN_b(y) -> out2T (the NEXT layer's XAttention pre-LN of its queries, components.py:166) while the row is still in registers.
// Same lane mapping, same reduction order and same arithmetic as two launches of layernorm_kernel (the second would re-read the fp32 y
// this kernel holds): bit-identical, one launch and one 3-KB row round trip fewer per decoder layer.
template <typename T>
__global__ __launch_bounds__(256) void layernorm2_kernel(const float* __restrict__ in, long long ldin, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, float eps, const float* __restrict__ gamma2,
                                                          const float* __restrict__ beta2, float eps2, int rows, int E, float* out32, T* outT,
                                                          T* out2T) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = in + (long long)row * ldin;
  const int nv = E >> 2;
  float4 v[LN_MAXV];
  auto stats = [&](float& mean, float& rstd, float e) {   // layernorm_kernel's two passes
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (lane + i * 64 < nv) s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    mean = wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i)
      if (lane + i * 64 < nv) {
        const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    rstd = rsqrtf(wave_sum(q) / (float)E + e);
  };
  auto apply = [&](float mean, float rstd, const float* g_, const float* b_) {
#pragma unroll
    for (int i = 0; i < LN_MAXV; ++i) {
      const int c = lane + i * 64;
      if (c < nv) {
        const float4 g = *reinterpret_cast<const float4*>(g_ + c * 4);
        float4 o;
        o.x = (v[i].x - mean) * rstd * g.x;
        o.y = (v[i].y - mean) * rstd * g.y;
        o.z = (v[i].z - mean) * rstd * g.z;
        o.w = (v[i].w - mean) * rstd * g.w;
        if (b_) {
          const float4 bb = *reinterpret_cast<const float4*>(b_ + c * 4);
          o.x += bb.x; o.y += bb.y; o.z += bb.z; o.w += bb.w;
        }
        v[i] = o;
      }
    }
  };
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    v[i] = c < nv ? load4(x + c * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  float mean, rstd;
  stats(mean, rstd, eps);
  apply(mean, rstd, gamma, beta);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) {
      if (out32) store4(out32 + (long long)row * E + c * 4, v[i]);
      if (outT) store4(outT + (long long)row * E + c * 4, v[i]);
    }
  }
  stats(mean, rstd, eps2);
  apply(mean, rstd, gamma2, beta2);
#pragma unroll
  for (int i = 0; i < LN_MAXV; ++i) {
    const int c = lane + i * 64;
    if (c < nv) store4(out2T + (long long)row * E + c * 4, v[i]);
  }
}

// bf16 -> bf16 (+ optional e4m3) LayerNorm over rows of E = 256 * NV elements, the shape of the ViT / decoder stream LayerNorms of a
// big batch (46 launches of 180 MB per B = 256 step). HALF a wave owns a row: every lane loads NV 16-byte chunks (the one-wave-per-row
// kernel above issues 8-byte loads and retires a wave per row: 4.1 TB/s), a wave walks row pairs with a grid-sized stride and has the
// NEXT pair's loads in flight while it reduces and stores the current one; gamma / beta live in registers for the whole launch.
// Same two-pass statistics in fp32 (mean, then centred squares); the sums are taken in this kernel's own fixed order, which does not
// depend on the number of rows (batch-composition invariance).
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NV>
__global__ __launch_bounds__(256) void layernorm_rows2_kernel(const bf16_t* in, long long ldin, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, int rms, int rows, float* out32,
                                                               bf16_t* outT, uint8_t* out8, float inv8) {
  constexpr int E = 256 * NV;
  const int lane = threadIdx.x & 63, l31 = lane & 31, half = lane >> 5;
  const long long nwaves = (long long)gridDim.x * 4;
  long long pair = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long npairs = ((long long)rows + 1) >> 1;
  if (pair >= npairs) return;
  float4 g[NV][2], bt[NV][2];
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const int c = (i * 32 + l31) * 8;
    g[i][0] = load4(gamma + c); g[i][1] = load4(gamma + c + 4);
    if (beta) { bt[i][0] = load4(beta + c); bt[i][1] = load4(beta + c + 4); }
    else { bt[i][0] = make_float4(0.f, 0.f, 0.f, 0.f); bt[i][1] = bt[i][0]; }
  }
  auto fetch = [&](long long pr, uint4 (&u)[NV]) {
    long long r = pr * 2 + half;
    r = r < rows ? r : rows - 1;
    const bf16_t* x = in + r * ldin;
#pragma unroll
    for (int i = 0; i < NV; ++i) u[i] = *reinterpret_cast<const uint4*>(x + (i * 32 + l31) * 8);
  };
  uint4 cur[NV], nxt[NV];
  fetch(pair, cur);
  for (; pair < npairs; pair += nwaves) {
    const bool more = pair + nwaves < npairs;
    if (more) fetch(pair + nwaves, nxt);
    float x[NV][8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const uint32_t w[4] = {cur[i].x, cur[i].y, cur[i].z, cur[i].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        x[i][2 * j] = __uint_as_float(w[j] << 16);
        x[i][2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u);
      }
      s += ((x[i][0] + x[i][1]) + (x[i][2] + x[i][3])) + ((x[i][4] + x[i][5]) + (x[i][6] + x[i][7]));
    }
    float mean = 0.f;
    if (!rms) mean = half_wave_sum(s) / (float)E;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
#pragma unroll
      for (int j = 0; j < 8; ++j) x[i][j] -= mean;
      q += ((x[i][0] * x[i][0] + x[i][1] * x[i][1]) + (x[i][2] * x[i][2] + x[i][3] * x[i][3])) +
           ((x[i][4] * x[i][4] + x[i][5] * x[i][5]) + (x[i][6] * x[i][6] + x[i][7] * x[i][7]));
    }
    const float rstd = rsqrtf(half_wave_sum(q) / (float)E + eps);
    const long long row = pair * 2 + half;
    if (row < rows) {
#pragma unroll
      for (int i = 0; i < NV; ++i) {
        float4 o0, o1;
        o0.x = x[i][0] * rstd * g[i][0].x + bt[i][0].x; o0.y = x[i][1] * rstd * g[i][0].y + bt[i][0].y;
        o0.z = x[i][2] * rstd * g[i][0].z + bt[i][0].z; o0.w = x[i][3] * rstd * g[i][0].w + bt[i][0].w;
        o1.x = x[i][4] * rstd * g[i][1].x + bt[i][1].x; o1.y = x[i][5] * rstd * g[i][1].y + bt[i][1].y;
        o1.z = x[i][6] * rstd * g[i][1].z + bt[i][1].z; o1.w = x[i][7] * rstd * g[i][1].w + bt[i][1].w;
        const long long off = row * E + (i * 32 + l31) * 8;
        if (out32) { store4(out32 + off, o0); store4(out32 + off + 4, o1); }
        if (outT) {
          uint4 o;
          o.x = pack2_bf16(o0.x, o0.y); o.y = pack2_bf16(o0.z, o0.w); o.z = pack2_bf16(o1.x, o1.y); o.w = pack2_bf16(o1.z, o1.w);
          *reinterpret_cast<uint4*>(outT + off) = o;
        }
        if (out8) *reinterpret_cast<uint2*>(out8 + off) = make_uint2(pack4_fp8(o0, inv8), pack4_fp8(o1, inv8));
      }
    }
    if (more) {
#pragma unroll
      for (int i = 0; i < NV; ++i) cur[i] = nxt[i];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ in, T* out, long long n4) {
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long stride = (long long)gridDim.x * 256;
  for (; i < n4; i += stride) store4(out + i * 4, *reinterpret_cast<const float4*>(in + i * 4));
}

// ---------------------------------------------------------------------------------------------------------------
// patchify + normalise: basic_image_tensor_preprocess (preprocess.py:38-43: /255, (x-mean)/std with vit.py:9-10) and the
// im2col of the 16x16 stride-16 conv (vit.py:151-157,172). One thread per 16 contiguous pixels of one patch row.
// out row = crop*4 + (gy*2+gx) ; column k = c*256 + py*16 + px  (conv1.weight [768,3,16,16] flattened)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void patchify_kernel(const uint8_t* __restrict__ crops, T* out, long long total) {
  // total = M * 3 * 32 * 2 segments of 16 pixels
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int gx = (int)(i & 1);
  const int y = (int)((i >> 1) & 31);
  const int c = (int)((i >> 6) % 3);
  const long long m = i / 192;
  const uint4 px = *reinterpret_cast<const uint4*>(crops + ((m * 3 + c) * 32 + y) * 32 + gx * 16);
  const float mean = c == 0 ? 0.3471f : (c == 1 ? 0.3429f : 0.3383f);
  const float sd = c == 0 ? 0.3011f : (c == 1 ? 0.2961f : 0.2956f);
  const int gy = y >> 4, py = y & 15;
  T* o = out + (m * 4 + gy * 2 + gx) * 768 + c * 256 + py * 16;
  const uint32_t wds[4] = {px.x, px.y, px.z, px.w};
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float4 f;
    f.x = ((float)(wds[j] & 0xff) / 255.0f - mean) / sd;
    f.y = ((float)((wds[j] >> 8) & 0xff) / 255.0f - mean) / sd;
    f.z = ((float)((wds[j] >> 16) & 0xff) / 255.0f - mean) / sd;
    f.w = ((float)(wds[j] >> 24) / 255.0f - mean) / sd;
    store4(o + j * 4, f);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// ViT token embed: x[m, t, :] = ln_pre( (t == 0 ? cls : patch[m, t-1]) + pos[t] )   (vit.py:176-180). W = 768 fixed.
// One wave per token row.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vit_embed_kernel(const float* __restrict__ pre, const float* __restrict__ cls,
                                                         const float* __restrict__ pos, const float* __restrict__ g,
                                                         const float* __restrict__ b, float* x, T* xT, long long rows) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long m = row / 5;
  const int t = (int)(row % 5);
  const float* src = t == 0 ? cls : pre + (m * 4 + (t - 1)) * 768;
  float4 v[3];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = (lane + i * 64) * 4;
    const float4 a = *reinterpret_cast<const float4*>(src + c);
    const float4 p = *reinterpret_cast<const float4*>(pos + t * 768 + c);
    v[i] = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
  }
  const float mean = wave_sum(s) / 768.0f;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const float a = v[i].x - mean, bb = v[i].y - mean, c = v[i].z - mean, d = v[i].w - mean;
    q += (a * a + bb * bb) + (c * c + d * d);
  }
  const float rstd = rsqrtf(wave_sum(q) / 768.0f + 1e-5f);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = (lane + i * 64) * 4;
    const float4 gg = *reinterpret_cast<const float4*>(g + c);
    const float4 bb = *reinterpret_cast<const float4*>(b + c);
    float4 o;
    o.x = (v[i].x - mean) * rstd * gg.x + bb.x;
    o.y = (v[i].y - mean) * rstd * gg.y + bb.y;
    o.z = (v[i].z - mean) * rstd * gg.z + bb.z;
    o.w = (v[i].w - mean) * rstd * gg.w + bb.w;
    if (x) *reinterpret_cast<float4*>(x + row * 768 + c) = o;
    if (xT) store4(xT + row * 768 + c, o);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// bbox MLP layer 1 (obj_encoder.py:79-86, nn/utils.py build_mlp): relu(W[N,4] . (bbox / [256,128,128,256]) + b)
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void bbox_l1_kernel(const long long* __restrict__ bbox, const float* __restrict__ W,
                                                       const float* __restrict__ b, T* out, int R, int N) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)R * N) return;
  const int r = (int)(i / N), n = (int)(i % N);
  const float x0 = (float)bbox[r * 4 + 0] / 256.0f;
  const float x1 = (float)bbox[r * 4 + 1] / 128.0f;
  const float x2 = (float)bbox[r * 4 + 2] / 128.0f;
  const float x3 = (float)bbox[r * 4 + 3] / 256.0f;
  const float4 w = *reinterpret_cast<const float4*>(W + n * 4);
  float v = x0 * w.x;
  v = fmaf(x1, w.y, v);
  v = fmaf(x2, w.z, v);
  v = fmaf(x3, w.w, v);
  v += b[n];
  Elem<T>::store(out + (long long)r * N + n, fmaxf(v, 0.f));
}

// ---------------------------------------------------------------------------------------------------------------
// action embedding layer 1 (vima_policy.py:301-322 de-discretise, action_embd.py:40-56): x = idx / bins,
// relu(W[256,K] x + b); bins: position (K=2) -> [50, 100]; rotation (K=4) -> 50
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void action_l1_kernel(const long long* __restrict__ idx, int K,
                                                         const float* __restrict__ W, const float* __restrict__ b, T* out,
                                                         int R, int ldo, int col0) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)R * 256) return;
  const int r = (int)(i >> 8), n = (int)(i & 255);
  float v = 0.f;
  for (int k = 0; k < K; ++k) {
    const float bins = (K == 2 && k == 1) ? 100.0f : 50.0f;
    const float x = (float)idx[(long long)r * K + k] / bins;
    v = k == 0 ? x * W[n * K] : fmaf(x, W[n * K + k], v);
  }
  v += b[n];
  Elem<T>::store(out + (long long)r * ldo + col0 + n, fmaxf(v, 0.f));
}

__global__ __launch_bounds__(256) void add_row_table_kernel(float* out, long long total4, int E4,
                                                             const float* __restrict__ table,
                                                             const long long* __restrict__ sel, int group) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const long long r = i / E4;
  const int c = (int)(i % E4);
  long long s = sel[r / group];
  s = s < 0 ? 0 : (s > 1 ? 1 : s);
  float4 o = *reinterpret_cast<float4*>(out + i * 4);
  const float4 t = *reinterpret_cast<const float4*>(table + s * (E4 * 4) + c * 4);
  o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
  *reinterpret_cast<float4*>(out + i * 4) = o;
}

// ---------------------------------------------------------------------------------------------------------------
// prompt assembly (vima_policy.py:168-233 python loops): row (b,l) source code in tok_src:
//   >= 0 : index into word_ids (token row = word_table[word_ids[i]])   (word_embd.py:18-23)
//   -1   : padding (zeros, mask False)
//   <= -2: object token row  -(v + 2) of obj_tokens / obj_mask
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prompt_assemble_kernel(const int* __restrict__ tok_src,
                                                               const long long* __restrict__ word_ids,
                                                               const float* __restrict__ word_table,
                                                               const float* __restrict__ obj_tokens,
                                                               const uint8_t* __restrict__ obj_mask, float* x,
                                                               uint8_t* mask, int rows, int E) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int s = tok_src[row];
  const float* src = nullptr;
  uint8_t m = 0;
  if (s >= 0) {
    src = word_table + word_ids[s] * E;
    m = 1;
  } else if (s <= -2) {
    const long long o = -(long long)s - 2;
    src = obj_tokens + o * E;
    m = obj_mask[o] ? 1 : 0;
  }
  for (int c = lane * 4; c < E; c += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src) v = *reinterpret_cast<const float4*>(src + c);
    *reinterpret_cast<float4*>(x + (long long)row * E + c) = v;
  }
  if (lane == 0) mask[row] = m;
}

// The same gather writing what the T5 stack's fused-RMSNorm chain starts from -- the operand-type copy of the row and its sum of squares
// (rms_stats_kernel's output, same lane -> element mapping and the same summation order: identical bits) -- instead of the fp32 row: with the
// residual stream carried in the operand type the fp32 prompt is never read again, so its 403-MB write and read (batch 256) disappear.
template <typename T>
__global__ __launch_bounds__(256) void prompt_assemble_stats_kernel(const int* __restrict__ tok_src, const long long* __restrict__ word_ids,
                                                                     const float* __restrict__ word_table, const float* __restrict__ obj_tokens,
                                                                     const uint8_t* __restrict__ obj_mask, T* xT, float* ssq, uint8_t* mask,
                                                                     int rows, int E) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const int s = tok_src[row];
  const float* src = nullptr;
  uint8_t m = 0;
  if (s >= 0) {
    src = word_table + word_ids[s] * E;
    m = 1;
  } else if (s <= -2) {
    const long long o = -(long long)s - 2;
    src = obj_tokens + o * E;
    m = obj_mask[o] ? 1 : 0;
  }
  const int nv = E >> 2;
  float q = 0.f;
  for (int c = lane; c < nv; c += 64) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (src) v = *reinterpret_cast<const float4*>(src + c * 4);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    store4(xT + (long long)row * E + c * 4, v);
  }
  q = wave_sum(q);
  if (lane == 0) { ssq[row] = q; mask[row] = m; }
}

// ---------------------------------------------------------------------------------------------------------------
// decoder input (vima_policy.py:124-146 + xattn_gpt.py:101-106): per batch element b, sequence position l:
//   step t = l / (Q+1), slot s = l % (Q+1); s < Q -> obs token (t,b,s) else action token (t,b); action mask = True
//   position id = cumsum(mask)[l] - 1 ; x = token + positions_embed[pos]
// One workgroup (256 threads) per batch element; Lq <= 512 positions scanned by wave 0 in LDS.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void dec_embed_kernel(const float* __restrict__ obs_tok,
                                                         const uint8_t* __restrict__ obs_mask,
                                                         const float* __restrict__ act_tok,
                                                         const float* __restrict__ pos_table, int n_pos, float* x32,
                                                         T* xT, uint8_t* mask, int Tn, int B, int Q, int L_act, int E) {
  __shared__ int pos_s[512];
  const int b = blockIdx.x;
  const int Lq = Tn * Q + L_act;
  if (threadIdx.x == 0) {
    int run = 0;
    for (int l = 0; l < Lq; ++l) {
      const int t = l / (Q + 1), s = l % (Q + 1);
      const int mk = s < Q ? (obs_mask[((long long)t * B + b) * Q + s] ? 1 : 0) : 1;
      run += mk;
      int p = run - 1;
      p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);   // reference raises on -1 (first token masked); host validates
      pos_s[l] = p;
      mask[(long long)b * Lq + l] = (uint8_t)mk;
    }
  }
  __syncthreads();
  const int e4 = E >> 2;
  for (int i = threadIdx.x; i < Lq * e4; i += 256) {
    const int l = i / e4, c = (i % e4) * 4;
    const int t = l / (Q + 1), s = l % (Q + 1);
    const float* src = s < Q ? obs_tok + (((long long)t * B + b) * Q + s) * E : act_tok + ((long long)t * B + b) * E;
    const float4 a = *reinterpret_cast<const float4*>(src + c);
    const float4 p = *reinterpret_cast<const float4*>(pos_table + (long long)pos_s[l] * E + c);
    const float4 o = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    const long long off = ((long long)b * Lq + l) * E + c;
    store4(x32 + off, o);
    if (xT) store4(xT + off, o);
  }
}

// Incremental form of dec_embed_kernel for ONE env step: the newest tokens of every sample ([a_{t-1},] o_t^1 .. o_t^Q) get
// position ids that continue the sample's running count of valid tokens (cumsum(mask) - 1 over the whole history,
// xattn_gpt.py:96-103), the count and the history key mask are updated in the episode state.
template <typename T>
__global__ __launch_bounds__(256) void dec_embed_step_kernel(const float* __restrict__ obs_tok,
                                                              const uint8_t* __restrict__ obs_mask,
                                                              const float* __restrict__ act_tok,
                                                              const float* __restrict__ pos_table, int n_pos, float* x32,
                                                              T* xT, uint8_t* hist_mask, int* poscnt, int L_hist, int Lmax,
                                                              int Q, int has_act, int E, uint8_t* fresh) {
  __shared__ int pos_s[64];
  const int b = blockIdx.x;
  const int Ln = Q + has_act;
  if (threadIdx.x == 0) {
    int run = poscnt[b];
    // a sample restarted by vima_decode_restart has no previous action: its action slot of this step is a masked (invalid) token
    const int fr = fresh ? fresh[b] : 0;
    if (fr) fresh[b] = 0;
    for (int i = 0; i < Ln; ++i) {
      const int mk = (has_act && i == 0) ? (fr ? 0 : 1) : (obs_mask[(long long)b * Q + (i - has_act)] ? 1 : 0);
      run += mk;
      int p = run - 1;
      p = p < 0 ? 0 : (p >= n_pos ? n_pos - 1 : p);
      pos_s[i] = p;
      hist_mask[(long long)b * Lmax + L_hist + i] = (uint8_t)mk;
    }
    poscnt[b] = run;
  }
  __syncthreads();
  const int e4 = E >> 2;
  for (int i = threadIdx.x; i < Ln * e4; i += 256) {
    const int l = i / e4, c = (i % e4) * 4;
    const float* src = (has_act && l == 0) ? act_tok + (long long)b * E : obs_tok + ((long long)b * Q + (l - has_act)) * E;
    const float4 a = *reinterpret_cast<const float4*>(src + c);
    const float4 pp = *reinterpret_cast<const float4*>(pos_table + (long long)pos_s[l] * E + c);
    const float4 o = make_float4(a.x + pp.x, a.y + pp.y, a.z + pp.z, a.w + pp.w);
    const long long off = ((long long)b * Ln + l) * E + c;
    store4(x32 + off, o);
    if (xT) store4(xT + off, o);
  }
}

// per-sample episode restart (vima_decode_restart): flagged samples forget their history (every cached key masked), their position
// counter restarts at 0 and their next step's action slot is marked absent
__global__ __launch_bounds__(256) void restart_samples_kernel(const uint8_t* __restrict__ flags, uint8_t* hist_mask, int* poscnt, uint8_t* fresh,
                                                               int Lmax) {
  const int b = blockIdx.x;
  if (!flags[b]) return;
  for (int i = threadIdx.x; i < Lmax; i += 256) hist_mask[(long long)b * Lmax + i] = 0;
  if (threadIdx.x == 0) { poscnt[b] = 0; fresh[b] = 1; }
}

// prompt + xattn_positions_embed[cumsum(prompt_mask) - 1]  (vima_policy.py:147, xattn_gpt.py:110-114) -> T [B, Lp, E]
template <typename T>
__global__ __launch_bounds__(256) void prompt_pos_kernel(const float* __restrict__ prompt, long long sb, long long sl,
                                                          const uint8_t* __restrict__ mask,
                                                          const float* __restrict__ pos_table, int n_pos, T* out, int B,
                                                          int Lp, int E, int slab) {
  extern __shared__ int pos_dyn[];
  __shared__ int part[256];
  const int b = blockIdx.x;
  // cumsum(mask) - 1 over the Lp positions of sample b: each thread owns a contiguous run, block-wide exclusive scan of
  // the run totals (Hillis-Steele in LDS), then the local prefix -- no serial chain of dependent global loads
  const int per = (Lp + 255) / 256;
  const int l0 = threadIdx.x * per;
  int cnt = 0;
  for (int j = 0; j < per; ++j) {
    const int l = l0 + j;
    if (l < Lp) cnt += mask[(long long)b * Lp + l] ? 1 : 0;
  }
  part[threadIdx.x] = cnt;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    const int v = threadIdx.x >= off ? part[threadIdx.x - off] : 0;
    __syncthreads();
    part[threadIdx.x] += v;
    __syncthreads();
  }
  int run = part[threadIdx.x] - cnt;   // exclusive prefix of this thread's run
  for (int j = 0; j < per; ++j) {
    const int l = l0 + j;
    if (l < Lp) {
      run += mask[(long long)b * Lp + l] ? 1 : 0;
      int pp = run - 1;
      pp = pp < 0 ? 0 : (pp >= n_pos ? n_pos - 1 : pp);
      pos_dyn[l] = pp;
    }
  }
  __syncthreads();
  // blockIdx.y selects a slab of `slab` positions to copy (the cheap scan above is redone per slab for parallelism): 64, or 8 when the batch is so
  // small that 64-position slabs would leave the copy to a handful of CUs (a CU streams ~25 GB/s: one prompt on 8 workgroups took 29 us)
  const int e4 = E >> 2;
  const int lbeg = blockIdx.y * slab;
  const int lend = lbeg + slab < Lp ? lbeg + slab : Lp;
  for (int i = threadIdx.x + lbeg * e4; i < lend * e4; i += 256) {
    const int l = i / e4, c = (i % e4) * 4;
    const float4 a = *reinterpret_cast<const float4*>(prompt + b * sb + l * sl + c);
    const float4 p = *reinterpret_cast<const float4*>(pos_table + (long long)pos_dyn[l] * E + c);
    store4(out + ((long long)b * Lp + l) * E + c, make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w));
  }
}

// predicted_action_tokens = tokens_out[Q-1 :: Q+1]  (vima_policy.py:158) -> [T, B, E]
__global__ __launch_bounds__(256) void gather_pred_kernel(const float* __restrict__ x, float* out, int Tn, int B, int Q,
                                                           int Lq, int E) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int e4 = E >> 2;
  if (i >= (long long)Tn * B * e4) return;
  const int c = (int)(i % e4) * 4;
  const long long tb = i / e4;
  const int b = (int)(tb % B), t = (int)(tb / B);
  const int l = (Q - 1) + (Q + 1) * t;
  *reinterpret_cast<float4*>(out + tb * E + c) = *reinterpret_cast<const float4*>(x + ((long long)b * Lq + l) * E + c);
}

// Entry of the fused-RMSNorm chain (T5 stack): operand-type copy of the fp32 rows + their sum of squares. One wave per row.
template <typename T>
__global__ __launch_bounds__(256) void rms_stats_kernel(const float* __restrict__ in, int rows, int E, T* outT, float* ssq) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* x = in + (long long)row * E;
  const int nv = E >> 2;
  float q = 0.f;
  for (int c = lane; c < nv; c += 64) {
    const float4 v = *reinterpret_cast<const float4*>(x + c * 4);
    q += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    store4(outT + (long long)row * E + c * 4, v);
  }
  q = wave_sum(q);
  if (lane == 0) ssq[row] = q;
}

// ---- fp8 activations (precision "fp8"): per-tensor static scales. quant: out8 = e4m3(in * inv), saturating at 448 (the
// conversion the GEMM epilogue's fp8 emission uses); amax: slot = max(slot, max |in|) (non-negative floats order like their
// bit patterns, so one unsigned atomicMax per wave)
__global__ __launch_bounds__(256) void quant_fp8_kernel(const bf16_t* __restrict__ in, long long ldin, long long rows, int cols, float inv,
                                                        uint8_t* __restrict__ out, long long ldo) {
  const int cpr = cols >> 3;                                   // 8 elements per thread
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= rows * cpr) return;
  const long long r = i / cpr;
  const int c = (int)(i - r * cpr) * 8;
  const uint4 u = *reinterpret_cast<const uint4*>(in + r * ldin + c);
  auto cl = [](float x) { return __builtin_amdgcn_fmed3f(x, -448.0f, 448.0f); };
  auto lo = [](uint32_t w) { return __uint_as_float(w << 16); };
  auto hi = [](uint32_t w) { return __uint_as_float(w & 0xffff0000u); };
  int w0 = 0, w1 = 0;
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(lo(u.x) * inv), cl(hi(u.x) * inv), w0, false);
  w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(lo(u.y) * inv), cl(hi(u.y) * inv), w0, true);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(lo(u.z) * inv), cl(hi(u.z) * inv), w1, false);
  w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(lo(u.w) * inv), cl(hi(u.w) * inv), w1, true);
  *reinterpret_cast<uint2*>(out + r * ldo + c) = make_uint2((unsigned)w0, (unsigned)w1);
}

__global__ __launch_bounds__(256) void amax_kernel(const bf16_t* __restrict__ in, long long ldin, long long rows, int cols, unsigned* slot) {
  const int cpr = cols >> 3;
  const long long total = rows * cpr;
  float m = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const long long r = i / cpr;
    const int c = (int)(i - r * cpr) * 8;
    const uint4 u = *reinterpret_cast<const uint4*>(in + r * ldin + c);
    const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      m = fmaxf(m, fabsf(__uint_as_float(w[j] << 16)));
      m = fmaxf(m, fabsf(__uint_as_float(w[j] & 0xffff0000u)));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f && m == m) atomicMax(slot, __float_as_uint(m));
}

inline unsigned nblk(long long n, int per) { return (unsigned)((n + per - 1) / per); }

}  // namespace

int launch_quant_fp8(const void* inT, long long ldin, long long rows, int cols, float inv, void* out8, long long ldo, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  if (cols % 8 || ldin % 8 || ldo % 8) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(quant_fp8_kernel, dim3(nblk(rows * (cols >> 3), 256)), dim3(256), 0, st, (const bf16_t*)inT, ldin, rows, cols, inv,
                     (uint8_t*)out8, ldo);
  return (int)hipGetLastError();
}

int launch_amax(const void* inT, long long ldin, long long rows, int cols, float* slot, hipStream_t st) {
  if (rows <= 0 || cols <= 0) return 0;
  if (cols % 8 || ldin % 8) return (int)hipErrorInvalidValue;
  unsigned g = nblk(rows * (cols >> 3), 256 * 8);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(amax_kernel, dim3(g), dim3(256), 0, st, (const bf16_t*)inT, ldin, rows, cols, reinterpret_cast<unsigned*>(slot));
  return (int)hipGetLastError();
}

// A/B switch of the half-wave-per-row LayerNorm (environment VIMA_LN_ROWS2=0 restores the one-wave-per-row kernel)
static bool ln_rows2_enabled() {
  static const bool on = [] { const char* e = getenv("VIMA_LN_ROWS2"); return !(e && e[0] == '0'); }();
  return on;
}

int launch_layernorm(const float* in, long long ldin, const float* gamma, const float* beta, float eps, int rms,
                     int rows, int E, float* out32, void* outT, bool is_bf16, hipStream_t st) {
  if (rows <= 0) return 0;
  if (E % 4 != 0 || E > 64 * 4 * LN_MAXV || ldin % 4 != 0) return (int)hipErrorInvalidValue;
  if (is_bf16)
    hipLaunchKernelGGL(layernorm_kernel<bf16_t>, dim3(nblk(rows, 4)), dim3(256), 0, st, in, ldin, gamma, beta, eps, rms,
                       rows, E, out32, (bf16_t*)outT);
  else
    hipLaunchKernelGGL(layernorm_kernel<float>, dim3(nblk(rows, 4)), dim3(256), 0, st, in, ldin, gamma, beta, eps, rms,
                       rows, E, out32, (float*)outT);
  return (int)hipGetLastError();
}

int launch_layernorm2(const float* in, long long ldin, const float* gamma, const float* beta, float eps, const float* gamma2,
                      const float* beta2, float eps2, int rows, int E, float* out32, void* outT, void* out2T, bool is_bf16, hipStream_t st) {
  if (rows <= 0) return 0;
  if (E % 4 != 0 || E > 64 * 4 * LN_MAXV || ldin % 4 != 0 || !out2T) return (int)hipErrorInvalidValue;
  if (is_bf16)
    hipLaunchKernelGGL(layernorm2_kernel<bf16_t>, dim3(nblk(rows, 4)), dim3(256), 0, st, in, ldin, gamma, beta, eps, gamma2, beta2, eps2,
                       rows, E, out32, (bf16_t*)outT, (bf16_t*)out2T);
  else
    hipLaunchKernelGGL(layernorm2_kernel<float>, dim3(nblk(rows, 4)), dim3(256), 0, st, in, ldin, gamma, beta, eps, gamma2, beta2, eps2,
                       rows, E, out32, (float*)outT, (float*)out2T);
  return (int)hipGetLastError();
}

int launch_layernorm_T(const void* inT, long long ldin, const float* gamma, const float* beta, float eps, int rms, int rows,
                       int E, float* out32, void* outT, bool is_bf16, hipStream_t st, void* out8, float inv8) {
  if (!is_bf16) {
    if (out8) return (int)hipErrorInvalidValue;
    return launch_layernorm((const float*)inT, ldin, gamma, beta, eps, rms, rows, E, out32, outT, false, st);
  }
  if (rows <= 0) return 0;
  if (E % 4 != 0 || E > 64 * 4 * LN_MAXV || ldin % 4 != 0) return (int)hipErrorInvalidValue;
  // the choice depends on the row LENGTH and layout only, never on the number of rows: a row is normalised by the same
  // instruction sequence whatever batch it arrives in
  if (E % 256 == 0 && ldin % 8 == 0 && ln_rows2_enabled()) {
    unsigned g = nblk((rows + 1) / 2, 4);
    if (g > 2048) g = 2048;
    const bf16_t* x = (const bf16_t*)inT;
    switch (E / 256) {
      case 1: hipLaunchKernelGGL(layernorm_rows2_kernel<1>, dim3(g), dim3(256), 0, st, x, ldin, gamma, beta, eps, rms, rows, out32, (bf16_t*)outT, (uint8_t*)out8, inv8); break;
      case 2: hipLaunchKernelGGL(layernorm_rows2_kernel<2>, dim3(g), dim3(256), 0, st, x, ldin, gamma, beta, eps, rms, rows, out32, (bf16_t*)outT, (uint8_t*)out8, inv8); break;
      case 3: hipLaunchKernelGGL(layernorm_rows2_kernel<3>, dim3(g), dim3(256), 0, st, x, ldin, gamma, beta, eps, rms, rows, out32, (bf16_t*)outT, (uint8_t*)out8, inv8); break;
      default: hipLaunchKernelGGL(layernorm_rows2_kernel<4>, dim3(g), dim3(256), 0, st, x, ldin, gamma, beta, eps, rms, rows, out32, (bf16_t*)outT, (uint8_t*)out8, inv8); break;
    }
    return (int)hipGetLastError();
  }
  hipLaunchKernelGGL((layernorm_kernel<bf16_t, bf16_t>), dim3(nblk(rows, 4)), dim3(256), 0, st, (const bf16_t*)inT, ldin, gamma, beta,
                     eps, rms, rows, E, out32, (bf16_t*)outT, (uint8_t*)out8, inv8);
  return (int)hipGetLastError();
}

int launch_rms_stats(const float* in, int rows, int E, void* outT, float* ssq, bool is_bf16, hipStream_t st) {
  if (rows <= 0) return 0;
  if (E % 4 != 0) return (int)hipErrorInvalidValue;
  if (is_bf16) hipLaunchKernelGGL(rms_stats_kernel<bf16_t>, dim3(nblk(rows, 4)), dim3(256), 0, st, in, rows, E, (bf16_t*)outT, ssq);
  else hipLaunchKernelGGL(rms_stats_kernel<float>, dim3(nblk(rows, 4)), dim3(256), 0, st, in, rows, E, (float*)outT, ssq);
  return (int)hipGetLastError();
}

int launch_cast(const float* in, void* outT, long long n, bool is_bf16, hipStream_t st) {
  if (n <= 0) return 0;
  if (n % 4) return (int)hipErrorInvalidValue;
  const long long n4 = n / 4;
  unsigned g = nblk(n4, 256);
  if (g > 8192) g = 8192;
  if (is_bf16) hipLaunchKernelGGL(cast_kernel<bf16_t>, dim3(g), dim3(256), 0, st, in, (bf16_t*)outT, n4);
  else hipLaunchKernelGGL(cast_kernel<float>, dim3(g), dim3(256), 0, st, in, (float*)outT, n4);
  return (int)hipGetLastError();
}

int launch_patchify(const uint8_t* crops, void* outT, int M, bool is_bf16, hipStream_t st) {
  if (M <= 0) return 0;
  const long long total = (long long)M * 192;
  if (is_bf16) hipLaunchKernelGGL(patchify_kernel<bf16_t>, dim3(nblk(total, 256)), dim3(256), 0, st, crops, (bf16_t*)outT, total);
  else hipLaunchKernelGGL(patchify_kernel<float>, dim3(nblk(total, 256)), dim3(256), 0, st, crops, (float*)outT, total);
  return (int)hipGetLastError();
}

int launch_vit_embed(const float* pre, const float* cls, const float* pos, const float* g, const float* b, float* x,
                     void* xT, int M, bool is_bf16, hipStream_t st) {
  if (M <= 0) return 0;
  const long long rows = (long long)M * 5;
  if (is_bf16) hipLaunchKernelGGL(vit_embed_kernel<bf16_t>, dim3(nblk(rows, 4)), dim3(256), 0, st, pre, cls, pos, g, b, x, (bf16_t*)xT, rows);
  else hipLaunchKernelGGL(vit_embed_kernel<float>, dim3(nblk(rows, 4)), dim3(256), 0, st, pre, cls, pos, g, b, x, (float*)xT, rows);
  return (int)hipGetLastError();
}

int launch_bbox_l1(const long long* bbox, const float* W, const float* b, void* outT, int R, int Nout, bool is_bf16,
                   hipStream_t st) {
  if (R <= 0) return 0;
  const long long total = (long long)R * Nout;
  if (is_bf16) hipLaunchKernelGGL(bbox_l1_kernel<bf16_t>, dim3(nblk(total, 256)), dim3(256), 0, st, bbox, W, b, (bf16_t*)outT, R, Nout);
  else hipLaunchKernelGGL(bbox_l1_kernel<float>, dim3(nblk(total, 256)), dim3(256), 0, st, bbox, W, b, (float*)outT, R, Nout);
  return (int)hipGetLastError();
}

int launch_action_l1(const long long* idx, int K, const float* W, const float* b, void* outT, int R, int ldo, int col0,
                     bool is_bf16, hipStream_t st) {
  if (R <= 0) return 0;
  const long long total = (long long)R * 256;
  if (is_bf16) hipLaunchKernelGGL(action_l1_kernel<bf16_t>, dim3(nblk(total, 256)), dim3(256), 0, st, idx, K, W, b, (bf16_t*)outT, R, ldo, col0);
  else hipLaunchKernelGGL(action_l1_kernel<float>, dim3(nblk(total, 256)), dim3(256), 0, st, idx, K, W, b, (float*)outT, R, ldo, col0);
  return (int)hipGetLastError();
}

int launch_add_row_table(float* out, int rows, int E, const float* table, const long long* sel, int group,
                         hipStream_t st) {
  if (rows <= 0) return 0;
  if (E % 4) return (int)hipErrorInvalidValue;
  const long long total4 = (long long)rows * (E / 4);
  hipLaunchKernelGGL(add_row_table_kernel, dim3(nblk(total4, 256)), dim3(256), 0, st, out, total4, E / 4, table, sel, group);
  return (int)hipGetLastError();
}

int launch_prompt_assemble(const int* tok_src, const long long* word_ids, const float* word_table,
                           const float* obj_tokens, const uint8_t* obj_mask, float* x, uint8_t* mask, int rows, int E,
                           hipStream_t st) {
  if (rows <= 0) return 0;
  if (E % 4) return (int)hipErrorInvalidValue;
  hipLaunchKernelGGL(prompt_assemble_kernel, dim3(nblk(rows, 4)), dim3(256), 0, st, tok_src, word_ids, word_table,
                     obj_tokens, obj_mask, x, mask, rows, E);
  return (int)hipGetLastError();
}

// rows [L][N] (row-major) -> [N / D][L][D]: one sample's K | V rows into the head-major prompt cache (vima_decode_restart); 16-byte chunks
template <typename T>
__global__ __launch_bounds__(256) void rows_to_headmajor_kernel(const T* __restrict__ in, T* __restrict__ out, int L, int N, int D) {
  constexpr int EPC = 16 / (int)sizeof(T);
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const int cpr = N / EPC;
  if (i >= (long long)L * cpr) return;
  const int l = (int)(i / cpr), n = (int)(i % cpr) * EPC;
  const uint4 v = *reinterpret_cast<const uint4*>(in + (long long)l * N + n);
  *reinterpret_cast<uint4*>(out + ((long long)(n / D) * L + l) * D + n % D) = v;
}

int launch_rows_to_headmajor(const void* in, void* out, int L, int N, int D, bool is_bf16, hipStream_t st) {
  const int epc = is_bf16 ? 8 : 4;
  if (L <= 0 || N <= 0) return 0;
  if (D % epc || N % D) return (int)hipErrorInvalidValue;
  const long long n = (long long)L * (N / epc);
  if (is_bf16) hipLaunchKernelGGL(rows_to_headmajor_kernel<bf16_t>, dim3(nblk(n, 256)), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, L, N, D);
  else hipLaunchKernelGGL(rows_to_headmajor_kernel<float>, dim3(nblk(n, 256)), dim3(256), 0, st, (const float*)in, (float*)out, L, N, D);
  return (int)hipGetLastError();
}

int launch_prompt_assemble_stats(const int* tok_src, const long long* word_ids, const float* word_table, const float* obj_tokens,
                                 const uint8_t* obj_mask, void* xT, float* ssq, uint8_t* mask, int rows, int E, bool is_bf16, hipStream_t st) {
  if (rows <= 0) return 0;
  if (E % 4) return (int)hipErrorInvalidValue;
  if (is_bf16)
    hipLaunchKernelGGL(prompt_assemble_stats_kernel<bf16_t>, dim3(nblk(rows, 4)), dim3(256), 0, st, tok_src, word_ids, word_table, obj_tokens,
                       obj_mask, (bf16_t*)xT, ssq, mask, rows, E);
  else
    hipLaunchKernelGGL(prompt_assemble_stats_kernel<float>, dim3(nblk(rows, 4)), dim3(256), 0, st, tok_src, word_ids, word_table, obj_tokens,
                       obj_mask, (float*)xT, ssq, mask, rows, E);
  return (int)hipGetLastError();
}

int launch_dec_embed(const float* obs_tok, const uint8_t* obs_mask, const float* act_tok, const float* pos_table,
                     int n_pos, float* x32, void* xT, uint8_t* mask, int T, int B, int Q, int L_act, int E, bool is_bf16,
                     hipStream_t st) {
  if (B <= 0) return 0;
  if (T * Q + L_act > 512 || E % 4) return (int)hipErrorInvalidValue;
  if (is_bf16)
    hipLaunchKernelGGL(dec_embed_kernel<bf16_t>, dim3(B), dim3(256), 0, st, obs_tok, obs_mask, act_tok, pos_table, n_pos,
                       x32, (bf16_t*)xT, mask, T, B, Q, L_act, E);
  else
    hipLaunchKernelGGL(dec_embed_kernel<float>, dim3(B), dim3(256), 0, st, obs_tok, obs_mask, act_tok, pos_table, n_pos,
                       x32, (float*)xT, mask, T, B, Q, L_act, E);
  return (int)hipGetLastError();
}

int launch_dec_embed_step(const float* obs_tok, const uint8_t* obs_mask, const float* act_tok, const float* pos_table, int n_pos,
                          float* x32, void* xT, uint8_t* hist_mask, int* poscnt, int L_hist, int Lmax, int B, int Q,
                          int has_act, int E, bool is_bf16, hipStream_t st, uint8_t* fresh) {
  if (B <= 0) return 0;
  if (Q + has_act > 64 || E % 4 || L_hist + Q + has_act > Lmax) return (int)hipErrorInvalidValue;
  if (is_bf16)
    hipLaunchKernelGGL(dec_embed_step_kernel<bf16_t>, dim3(B), dim3(256), 0, st, obs_tok, obs_mask, act_tok, pos_table, n_pos,
                       x32, (bf16_t*)xT, hist_mask, poscnt, L_hist, Lmax, Q, has_act, E, fresh);
  else
    hipLaunchKernelGGL(dec_embed_step_kernel<float>, dim3(B), dim3(256), 0, st, obs_tok, obs_mask, act_tok, pos_table, n_pos,
                       x32, (float*)xT, hist_mask, poscnt, L_hist, Lmax, Q, has_act, E, fresh);
  return (int)hipGetLastError();
}

int launch_restart_samples(const uint8_t* flags, uint8_t* hist_mask, int* poscnt, uint8_t* fresh, int B, int Lmax, hipStream_t st) {
  if (B <= 0) return 0;
  hipLaunchKernelGGL(restart_samples_kernel, dim3(B), dim3(256), 0, st, flags, hist_mask, poscnt, fresh, Lmax);
  return (int)hipGetLastError();
}

int launch_prompt_pos(const float* prompt, long long sb, long long sl, const uint8_t* mask, const float* pos_table,
                      int n_pos, void* outT, int B, int Lp, int E, bool is_bf16, hipStream_t st) {
  if (B <= 0 || Lp <= 0) return 0;
  if (E % 4 || sb % 4 || sl % 4) return (int)hipErrorInvalidValue;
  const size_t sh = (size_t)Lp * sizeof(int);
  const int slab = (long long)B * ((Lp + 63) / 64) >= 512 ? 64 : 8;
  if (is_bf16)
    hipLaunchKernelGGL(prompt_pos_kernel<bf16_t>, dim3(B, (Lp + slab - 1) / slab), dim3(256), sh, st, prompt, sb, sl, mask, pos_table, n_pos,
                       (bf16_t*)outT, B, Lp, E, slab);
  else
    hipLaunchKernelGGL(prompt_pos_kernel<float>, dim3(B, (Lp + slab - 1) / slab), dim3(256), sh, st, prompt, sb, sl, mask, pos_table, n_pos,
                       (float*)outT, B, Lp, E, slab);
  return (int)hipGetLastError();
}

int launch_gather_pred(const float* x, float* out, int T, int B, int Q, int Lq, int E, hipStream_t st) {
  const long long total = (long long)T * B * (E / 4);
  if (total <= 0) return 0;
  hipLaunchKernelGGL(gather_pred_kernel, dim3(nblk(total, 256)), dim3(256), 0, st, x, out, T, B, Q, Lq, E);
  return (int)hipGetLastError();
}

}  // namespace vima
