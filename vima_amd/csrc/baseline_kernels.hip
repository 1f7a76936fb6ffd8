// Token plumbing of the reference's BASELINE policies (VIMAGPTPolicy / VIMAGatoPolicy / VIMAFlamingoPolicy; SURVEY.md 8(f)
// row 4): whole 64x128 RGB frames through a rectangular ViT with 32x32 patches, and the decoder-only sequence assembly.
// All HBM-bound elementwise work (one pass over the data, 16-byte accesses); the arithmetic of these policies runs on the
// same GEMM / attention / LayerNorm kernels as the VIMA hot path.
#include "kernels.h"

namespace vima {
namespace {

inline unsigned nblk(long long n, int per) { return (unsigned)((n + per - 1) / per); }

// ---------------------------------------------------------------------------------------------------------------
// patchify + normalise for (Gato)VisionTransformerRectangular (vit.py:83-135, 262-329): basic_image_tensor_preprocess
// (preprocess.py:38-43: /255, (x - mean) / std) and the im2col of the PxP stride-P conv. One thread per 16 contiguous
// pixels of an image row. out row = img * (gh*gw) + gy*gw + gx ; column = c*P*P + py*P + px (conv1.weight flattened).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void patchify_rect_kernel(const uint8_t* __restrict__ img, T* out, long long total, int H, int W,
                                                            int P) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int segs = W / 16;
  const int seg = (int)(i % segs);
  const int y = (int)((i / segs) % H);
  const int c = (int)((i / ((long long)segs * H)) % 3);
  const long long m = i / ((long long)segs * H * 3);
  const uint4 px = *reinterpret_cast<const uint4*>(img + ((m * 3 + c) * H + y) * W + seg * 16);
  const float mean = c == 0 ? 0.3471f : (c == 1 ? 0.3429f : 0.3383f);
  const float sd = c == 0 ? 0.3011f : (c == 1 ? 0.2961f : 0.2956f);
  const int gw = W / P;
  const int x0 = seg * 16;
  const int gx = x0 / P, pxo = x0 % P, gy = y / P, py = y % P;
  T* o = out + (m * ((H / P) * gw) + gy * gw + gx) * (3LL * P * P) + (long long)c * P * P + py * P + pxo;
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
// ViT token embed for S tokens per image: x[m, t, :] = ln_pre( src(m, t) + pos[t] ), src = cls for t == 0 when the variant
// has a cls token (vit.py:313-320), else the patch embedding (vit.py:122-127). Width 768; one wave per token row.
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void vit_embed_rect_kernel(const float* __restrict__ pre, const float* __restrict__ cls,
                                                             const float* __restrict__ pos, const float* __restrict__ g,
                                                             const float* __restrict__ b, float* x, T* xT, long long rows, int S,
                                                             int n_patch) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long long m = row / S;
  const int t = (int)(row % S);
  const float* src = cls ? (t == 0 ? cls : pre + (m * n_patch + (t - 1)) * 768) : pre + (m * n_patch + t) * 768;
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

// out[r, :] = src[r % period, :]   (Perceiver latents expanded over the images, modeling_perceiver.py:132-133)
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float* __restrict__ src, float* out, long long total4, int E4,
                                                             int period) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const long long r = i / E4;
  const int c = (int)(i % E4);
  *reinterpret_cast<float4*>(out + i * 4) = *reinterpret_cast<const float4*>(src + ((r % period) * E4 + c) * 4);
}

// ---------------------------------------------------------------------------------------------------------------
// Decoder-only sequence assembly (vima_gpt_policy.py:126-184, vima_gato_policy.py:123-184) + OpenAIGPTModel's position
// embedding (gpt/gpt.py:177-185). Row (b, l) of the [B, L, E] sequence:
//   l <  Lp : prompt token l                                   key mask = prompt mask, position min(l, nv-1)
//   l == Lp : prompt_sep_token                                 key mask 1, position nv
//   else j = l-Lp-1, t = j / (Q+1), s = j % (Q+1): s < Q -> obs token (t, b, s), else action token (t, b); position nv + 1 + j
// with nv = number of valid prompt tokens of sample b. One wave per row; nv is recomputed per wave (Lp <= 512 bytes).
// ---------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void seq_embed_kernel(const float* __restrict__ prompt, long long sb, long long sl,
                                                        const uint8_t* __restrict__ pmask, const float* __restrict__ sep,
                                                        const float* __restrict__ obs_tok, const float* __restrict__ act_tok,
                                                        const float* __restrict__ pos_table, int n_pos, float* x32, T* xT,
                                                        uint8_t* mask, int B, int L, int Lp, int Q, int E) {
  const int lane = threadIdx.x & 63;
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long long)B * L) return;
  const int b = (int)(row / L), l = (int)(row % L);
  float cnt = 0.f;
  for (int j = lane; j < Lp; j += 64) cnt += pmask[(long long)b * Lp + j] ? 1.f : 0.f;
  const int nv = (int)wave_sum(cnt);
  const float* src;
  int pos;
  uint8_t m = 1;
  if (l < Lp) {
    src = prompt + (long long)b * sb + (long long)l * sl;
    pos = l < nv ? l : nv - 1;
    m = pmask[(long long)b * Lp + l] ? 1 : 0;
  } else if (l == Lp) {
    src = sep;
    pos = nv;
  } else {
    const int j = l - Lp - 1;
    const int t = j / (Q + 1), s = j % (Q + 1);
    src = s < Q ? obs_tok + (((long long)t * B + b) * Q + s) * E : act_tok + ((long long)t * B + b) * E;
    pos = nv + 1 + j;
  }
  pos = pos < 0 ? 0 : (pos >= n_pos ? n_pos - 1 : pos);
  for (int c = lane * 4; c < E; c += 256) {
    const float4 a = *reinterpret_cast<const float4*>(src + c);
    const float4 p = *reinterpret_cast<const float4*>(pos_table + (long long)pos * E + c);
    const float4 o = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
    *reinterpret_cast<float4*>(x32 + row * E + c) = o;
    store4(xT + row * E + c, o);
  }
  if (lane == 0) mask[row] = m;
}

__global__ __launch_bounds__(256) void fill_u8_kernel(uint8_t* p, long long n, uint8_t v) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}

}  // namespace

int launch_patchify_rect(const uint8_t* img, void* outT, int M, int H, int W, int P, bool is_bf16, hipStream_t st) {
  if (M <= 0) return 0;
  if (P % 16 || H % P || W % P) return (int)hipErrorInvalidValue;
  const long long total = (long long)M * 3 * H * (W / 16);
  if (is_bf16) hipLaunchKernelGGL(patchify_rect_kernel<bf16_t>, dim3(nblk(total, 256)), dim3(256), 0, st, img, (bf16_t*)outT, total, H, W, P);
  else hipLaunchKernelGGL(patchify_rect_kernel<float>, dim3(nblk(total, 256)), dim3(256), 0, st, img, (float*)outT, total, H, W, P);
  return (int)hipGetLastError();
}

int launch_vit_embed_rect(const float* pre, const float* cls, const float* pos, const float* g, const float* b, float* x, void* xT,
                          int M, int S, int n_patch, bool is_bf16, hipStream_t st) {
  if (M <= 0) return 0;
  const long long rows = (long long)M * S;
  if (is_bf16)
    hipLaunchKernelGGL(vit_embed_rect_kernel<bf16_t>, dim3(nblk(rows, 4)), dim3(256), 0, st, pre, cls, pos, g, b, x, (bf16_t*)xT, rows, S, n_patch);
  else
    hipLaunchKernelGGL(vit_embed_rect_kernel<float>, dim3(nblk(rows, 4)), dim3(256), 0, st, pre, cls, pos, g, b, x, (float*)xT, rows, S, n_patch);
  return (int)hipGetLastError();
}

int launch_broadcast_rows(const float* src, float* out, long long rows, int E, int period, hipStream_t st) {
  if (rows <= 0) return 0;
  if (E % 4 || period <= 0) return (int)hipErrorInvalidValue;
  const long long total4 = rows * (E / 4);
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3(nblk(total4, 256)), dim3(256), 0, st, src, out, total4, E / 4, period);
  return (int)hipGetLastError();
}

int launch_seq_embed(const float* prompt, long long sb, long long sl, const uint8_t* pmask, const float* sep, const float* obs_tok,
                     const float* act_tok, const float* pos_table, int n_pos, float* x32, void* xT, uint8_t* mask, int B, int L, int Lp,
                     int Q, int E, bool is_bf16, hipStream_t st) {
  if (B <= 0 || L <= 0) return 0;
  if (E % 4 || sb % 4 || sl % 4) return (int)hipErrorInvalidValue;
  const long long rows = (long long)B * L;
  if (is_bf16)
    hipLaunchKernelGGL(seq_embed_kernel<bf16_t>, dim3(nblk(rows, 4)), dim3(256), 0, st, prompt, sb, sl, pmask, sep, obs_tok, act_tok, pos_table,
                       n_pos, x32, (bf16_t*)xT, mask, B, L, Lp, Q, E);
  else
    hipLaunchKernelGGL(seq_embed_kernel<float>, dim3(nblk(rows, 4)), dim3(256), 0, st, prompt, sb, sl, pmask, sep, obs_tok, act_tok, pos_table,
                       n_pos, x32, (float*)xT, mask, B, L, Lp, Q, E);
  return (int)hipGetLastError();
}

int launch_fill_u8(uint8_t* p, long long n, uint8_t v, hipStream_t st) {
  if (n <= 0) return 0;
  hipLaunchKernelGGL(fill_u8_kernel, dim3(nblk(n, 256)), dim3(256), 0, st, p, n, v);
  return (int)hipGetLastError();
}

}  // namespace vima
