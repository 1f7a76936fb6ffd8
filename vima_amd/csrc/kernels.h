// Launcher declarations for the hand-written gfx950 kernels (definitions in *.hip).
#pragma once
#include "common.h"

namespace vima {

// Kernel-selection knobs of ONE handle (vima_set_option). -1 = process default (the VIMA_* environment variable named
// beside each field, read once). They travel with every launch (GemmArgs::tune / AttnArgs::tune): two handles in one
// process never see each other's settings, and a handle's captured hipGraphs are keyed on its own generation counter.
struct Tuning {
  int gemm_variant = -1;   // VIMA_GEMM_VARIANT  1 = inline-asm LDS-DMA pipeline (default), 0 = compiler-tracked builtin (TileS)
  int gemm_tile = -1;      // VIMA_GEMM_TILE     0 auto, 1 128x128, 2 256x256 8 waves, 7 32x64, 8 64x64, 10 / 11 / 12 resident-K 32x32 / 64x32 / 64x64
  int gemm_raster = -1;    // VIMA_GEMM_RASTER   tile order: 0 XCD x n-walk, 1 XCD x resident n-group, 2 row-major
  int gemm_epi = -1;       // VIMA_GEMM_EPI      1 = LDS-transposed row-contiguous epilogue (default), 0 = direct
  int gemm_persist = -1;   // VIMA_GEMM_PERSIST  1 = large bf16 GEMMs on the persistent kernel (default)
  int gemm_small = -1;     // VIMA_GEMM_SMALL    1 = 64x64 / 32x64 tiles for underfilled grids (default)
  int gemm_wide = -1;      // VIMA_GEMM_WIDE     1 = 256x384 persistent tile where N % 384 == 0 (bf16-out / bf16-residual epilogues)
  int gemm_pp = -1;        // VIMA_GEMM_PP       1 = ping-pong (8-phase) main loop on the persistent 256x256 kernel (default), 0 = the round-2 loop
  int gemm_q4 = -1;        // VIMA_GEMM_Q4       four-wave (one wave per SIMD) kernel, N % 384 == 0, K % 128 == 0, bf16-out / bf16-stream / head-major epilogues: 1 = its 256x384 tile wherever it fits,
                           //                    2 = its 128x384 tile wherever it fits, 3 = the 128x384 tile where it needs fewer rounds of the chip than 256x256 tiles (batch 32), 6 = the 256x384 tile from 32 768 rows on (default), 0 = off
  int gemm_splitk = -1;    // VIMA_GEMM_SPLITK   1 = two-pass split-K for underfilled grids with K >= 1536 (default 0)
  int gemm_resident = -1;  // VIMA_GEMM_RESIDENT 1 = underfilled grids on gemm_resident_kernel (whole K in flight; default), 0 = the 4-deep ring tiles
  int gemm_res_maxwg = -1; // VIMA_GEMM_RES_MAXWG largest grid (workgroups) that kernel takes at M > 32 (default 256 = one per CU)
  int gemm_skinny = -1;    // VIMA_GEMM_SKINNY   1 = GEMMs of at most 32 rows on gemm_skinny_kernel (K split over the waves of a workgroup; default), 0 = the resident 32x32 tile (bit-identical to every other tile)
  int gemm_flat = -1;        // VIMA_GEMM_FLAT     1: gemm_pp_kernel drops the XCD raster for small grids where it costs a round (default); 0: never
  int gemm_res_nch = -1;   // VIMA_GEMM_RES_NCH  chunk buffers of that kernel's LDS ring (0 = default: 4 / 5 / 4 = up to 128 KiB; max 5 / 6 / 5 = 160 KiB)
  long long* gemm_dbg = nullptr;   // device buffer [blocks*4] of shader-clock stamps (nullptr = off)
  int attn_split = -1;     // 1 = split-key 4-wave kernel for Lq <= 32 (default), 0 = one-wave kernel
  int attn4_min_lq = -1;   // Lq from which the 4-wave LDS-shared flash kernel is used (default 64)
  int attn_qg = -1;        // query groups of 32 per wave in that kernel: 1 (default) or 2 (64 queries per wave, Lq >= 256)
  long long* attn_dbg = nullptr;   // device buffer [workgroups*8] of phase clocks of the 4-wave kernel (nullptr = off)
};

// ---------------------------------------------------------------- GEMM
// C[M,N] = epilogue( A[M,K] . W[N,K]^T ), fp32 accumulation on the matrix cores.
//   epilogue(v) = ((act(v + bias[n])) * mul[r][n]) + res[r][n]
// Operand type T (A, W, mul, outT) is bf16 (mfma_f32_32x32x16_bf16) or float
// (mfma_f32_32x32x2_f32, exact-fp32 parity mode).
struct GemmArgs {
  const void* A = nullptr;   // [M,K] row-major, row stride lda (elements)
  const void* W = nullptr;   // [N,K] row-major (nn.Linear layout), row stride ldw
  int M = 0, N = 0, K = 0;
  int lda = 0, ldw = 0;
  int batch = 1;             // blockIdx.y; strides below in elements
  // GROUPED form (batch = groups; bf16, gemm_resident_kernel only -- gemm_grouped_ok()): group z multiplies A + z * bsA with the rows
  // [grp_col[z], grp_col[z + 1]) of ONE packed W [sum N_z, K] and writes bias + product to the same COLUMNS of out32 (fp32, any
  // alignment: scalar stores); N = the largest group; no activation / mul / residual / outT. grp_col: DEVICE array [batch + 1].
  const int* grp_col = nullptr;
  // DUAL form (bf16, gemm_resident_kernel only -- ask gemm_dual_ok() first): out = act(A . W^T + bias) * bf16(A2 . W2^T), the value /
  // gate pair of a GEGLU in ONE launch (W2 [N, K] row stride ldw2, A2 [M, K] row stride lda2; A2 may be A). Bit-identical to the two
  // launches (gate stored in bf16, then read as `mul`). outT only; act = ACT_GELU; no mul / residual / batch.
  const void* A2 = nullptr; const void* W2 = nullptr; int lda2 = 0, ldw2 = 0;
  long long bsA = 0, bsW = 0, bsBias = 0, bsMul = 0, bsRes = 0, bs32 = 0, bsT = 0;
  const float* bias = nullptr;
  int act = ACT_NONE;
  const void* mul = nullptr;  // T [M, ldmul]
  int ldmul = 0;
  const float* res = nullptr; // fp32 [M, ldres]  (may alias out32: each element is read then written by one thread)
  int ldres = 0;
  const void* resT = nullptr; // the same residual term in the OPERAND type T [M, ldresT] (residual stream carried in T; may
  int ldresT = 0;             // alias outT). At most one of res / resT.
  float* out32 = nullptr;
  int ld32 = 0;
  void* outT = nullptr;
  int ldT = 0;
  // HEAD-MAJOR output (hm_D > 0; bf16-only output of the persistent 256x256 kernels, ask gemm_headmajor_ok() first): rows are hm_L-row
  // batches, columns hm_D-wide heads, and element (r, n) goes to outT[((r / hm_L) * (N / hm_D) + n / hm_D) * hm_L * hm_D + (r % hm_L) * hm_D + n % hm_D]
  // -- [batch][head][row][hm_D], every head's rows contiguous -- instead of outT[r * ldT + n]. The decoder's prompt K / V cache is written
  // this way so that the split-key cross attention streams 32-KB runs per (batch, head) instead of 64-byte slices of 3-KB rows.
  int hm_D = 0, hm_L = 0;
  // GEGLU PAIR (pair32 = 1; bf16, act = ACT_GELU, ask gemm_pair_ok() first): W [N, K] and bias [N] hold TWO layers of N / 2 outputs each, alternating in
  // blocks of 32 rows -- block 2j = rows 32j.. of the GELU'd layer, block 2j + 1 = rows 32j.. of its plain multiplier -- and the output is
  // outT[m][j] = bf16(gelu(A.W1[j] + b1[j]) * bf16(A.Wg[j] + bg[j])), [M, N / 2]: bit-identical to the multiplier GEMM (bf16 output) followed by the
  // GELU GEMM with `mul`, in one launch of the 128x128 ring tile (both factors of an output element sit in one lane's accumulators).
  int pair32 = 0;
  // output-row remap (outputs only): orow = (r / rb) * s_hi + (r % rb) * s_lo + ro ; rb == 0 -> identity
  int rb = 0, s_hi = 0, s_lo = 0, ro = 0;
  // COLUMN-SPLIT output (bf16, outT only, no residual / gate / statistics; split_n % 128 == 0): columns n < split_n are stored to outT_lo[r * ldT_lo + n]
  // (rows NOT remapped), columns n >= split_n to outT[orow * ldT + (n - split_n)]. One launch for the self-attention projection of an incremental
  // decoding step: q of the new tokens to a dense buffer, their k | v rows appended to the episode cache through the row remap (components.py:51-58).
  void* outT_lo = nullptr;
  int ldT_lo = 0, split_n = 0;
  // RMS statistics fused into the GEMMs either side of a T5 RMSNorm (the norm's weight is folded into W at pack time):
  //   producer: ssq_out[r][j] = sum over columns [32j, 32j+32) of out[r][n]^2 (final fp32 values; needs out32, N % 32 == 0,
  //             batch 1). Plain stores of N/32 partials per row -- no atomics, so the result is deterministic and the
  //             same for every tile shape.
  //   consumer: accumulator row r is multiplied by rsqrt((sum_j rs_ssq[r][j], j < rs_parts) * rs_invk + rs_eps) before
  //             bias / activation
  float* ssq_out = nullptr;
  const float* rs_ssq = nullptr;
  int rs_parts = 0;
  float rs_invk = 0.f, rs_eps = 0.f;
  // nn.LayerNorm (mean AND variance, affine) folded the same way -- the pre-LN in front of XAttention's feed-forward (components.py:220-226), whose
  // GEGLU then reads ONE input (the un-normed stream) for both products:
  //   producer (fp32-output residual GEMM, with ssq_out): sum_out[r][j] = sum over columns [32j, 32j+32) of out[r][n]; ask gemm_lnfold_producer_ok()
  //   consumer (the GEGLU pair forms, pair32 or W2, with rs_ssq): mean = sum_j rs_sum[r][j] * rs_invk, var = sum_j rs_ssq[r][j] * rs_invk - mean^2,
  //             and the GELU'd factor's pre-activation becomes rsqrt(var + rs_eps) * (A.W'^T - mean * rs_c[n]) + bias[n], where W' = W diag(gamma) is
  //             what the caller packed, rs_c[n] = sum_k W'[n][k] (of the ROUNDED operand values) and bias[n] = sum_k beta[k] W[n][k] (+ the layer's bias);
  //             the plain multiplier factor is untouched. rs_c is indexed like bias (pair32: the interleaved column space).
  float* sum_out = nullptr;
  const float* rs_sum = nullptr;
  const float* rs_c = nullptr;
  // split-K scratch (fp32 [S][M][N] partial products) for grids that would leave most CUs idle; size it with
  // gemm_splitk_bytes(). Without it the launch is a single pass.
  float* splitk_ws = nullptr;
  size_t splitk_ws_bytes = 0;
  const Tuning* tune = nullptr;   // the calling handle's knobs (nullptr: process defaults)
  // out (profiling): which kernel the launcher chose = kind * 1000 + (act + 1) * 10 + epi ; kind 1 gemm_pp_kernel,
  // 2 gemm_persistent_kernel, 3 gemm_wide_kernel, 4..7 gemm_kernel with the 256x256 / 128x128 / 64x64 / 32x64 tile (epi 0),
  // 8 two-pass split-K, 10 / 11 / 12 gemm_resident_kernel with the 32x32 / 64x32 / 64x64 tile, 15 / 16 its DUAL (GEGLU pair) form on the 32x32 / 64x64 tile,
  // 17 / 18 gemm_skinny_kernel (M <= 32) / its DUAL form
  int* kernel_id = nullptr;
  // fp8 weights (precision "fp8w", bf16 activations): W is [N,K] OCP e4m3 BYTES (ldw / bsW in elements = bytes) and
  // wscale[n] the per-output-channel dequantisation scale; the kernel widens the fragments to bf16 in registers and
  // multiplies accumulator column n by wscale[n] before bias / activation. Needs N % 4 == 0, K % 64 == 0, batch 1.
  int w8 = 0;
  const float* wscale = nullptr;
  // fp8 ACTIVATIONS (precision "fp8"): with a8 = 1, A is [M,K] OCP e4m3 BYTES as well (lda in bytes) with ONE dequantisation scale
  // `ascale` for the tensor; both operands go to v_mfma_scale_f32_32x32x64_f8f6f4 (gemm_pp_kernel<.., F8>: needs w8, K % 256 == 0 and
  // the persistent-kernel shape conditions; anything else fails), the accumulator column is multiplied by wscale[n] * ascale.
  int a8 = 0;
  float ascale = 1.0f;
  // optional fp8 copy of the result for an fp8 consumer: out8[r][n] = e4m3(value * out8_inv) (saturating), row stride ld8 BYTES;
  // bf16-only-output and bf16-stream epilogues of the persistent kernels. With out8 the bf16 output outT may be omitted.
  void* out8 = nullptr;
  int ld8 = 0;
  float out8_inv = 1.0f;
};
// returns hipError_t as int; is_bf16 selects the operand type
int launch_gemm(const GemmArgs& a, bool is_bf16, hipStream_t st);
// Bytes of split-K scratch this problem wants (0: single pass). Deterministic two-pass split-K: pass 1 = the same GEMM
// kernel over S K-ranges writing fp32 partials, pass 2 = gemm_reduce_kernel (sum in split order + the full epilogue).
size_t gemm_splitk_bytes(const GemmArgs& a, bool is_bf16);
int gemm_splitk_enabled(const Tuning* t);   // the effective split-K setting for a handle
int gemm_k_multiple(bool is_bf16);  // K must be a multiple of this
int gemm_pair_large(const Tuning* t, long long M, long long Nout, long long K);   // ... and it will run on the persistent 256x256 kernel's pair epilogue (full tiles, >= 160 of them)
int gemm_pair_ok(const Tuning* t, long long M, long long Nout, long long K);   // the block-interleaved GEGLU pair (GemmArgs::pair32) is the form to use for an [M, Nout] output
int gemm_lnfold_producer_ok(const Tuning* t, long long M, long long N);   // GemmArgs::sum_out is available for an [M, N] fp32 output with this handle's knobs
int gemm_dual_ok(const Tuning* t, int M, int N);   // the DUAL form (GemmArgs::W2) is available for an [M, N] output with this handle's knobs
int gemm_headmajor_ok(const Tuning* t, long long M, long long N, long long K, long long lda, long long ldw, int hm_D, int hm_L, int a8);   // GemmArgs::hm_D / hm_L usable for this problem
int gemm_a8_ok(const Tuning* t, long long M, long long N, long long K, long long lda, long long ldw);   // an fp8-ACTIVATION GEMM (GemmArgs::a8) of this shape is launchable with this handle's knobs
int gemm_grouped_ok(const Tuning* t);   // the grouped form (GemmArgs::grp_col) is available with this handle's knobs

// ---------------------------------------------------------------- normalisation / elementwise
// LayerNorm (rms=0: mean/var, affine) or T5 RMSNorm (rms=1: no mean, no bias). fp32 statistics.
// in: fp32 rows of length E at row stride ldin; outputs optional.
int launch_layernorm(const float* in, long long ldin, const float* gamma, const float* beta, float eps, int rms,
                     int rows, int E, float* out32, void* outT, bool is_bf16, hipStream_t st);
// two LayerNorms in a row: y = LN(in; gamma, beta) -> out32 / outT (both optional), z = LN(y; gamma2, beta2) -> out2T (operand type);
// bit-identical to launch_layernorm twice (the second on the fp32 y)
int launch_layernorm2(const float* in, long long ldin, const float* gamma, const float* beta, float eps, const float* gamma2,
                      const float* beta2, float eps2, int rows, int E, float* out32, void* outT, void* out2T, bool is_bf16, hipStream_t st);
// the same with the input rows in the operand type T (residual stream carried in T)
// out8 (bf16 mode only): additionally (or, with outT == nullptr, only) the fp8 e4m3 copy e4m3(result * inv8), row stride E bytes
int launch_layernorm_T(const void* inT, long long ldin, const float* gamma, const float* beta, float eps, int rms, int rows,
                       int E, float* out32, void* outT, bool is_bf16, hipStream_t st, void* out8 = nullptr, float inv8 = 1.0f);
int launch_cast(const float* in, void* outT, long long n, bool is_bf16, hipStream_t st);
// fp8 activations (precision "fp8"): out8[r][c] = e4m3(in[r][c] * inv) of bf16 rows (saturating at 448; cols, strides % 8 == 0);
// *slot = max(*slot, max |in|) (slot holds a non-negative float, zero it before the first call)
int launch_quant_fp8(const void* inT, long long ldin, long long rows, int cols, float inv, void* out8, long long ldo, hipStream_t st);
int launch_amax(const void* inT, long long ldin, long long rows, int cols, float* slot, hipStream_t st);
// operand-type copy of fp32 rows + their sum of squares (entry point of the fused-RMSNorm chain): outT[r][:] = in[r][:],
// ssq[r] = sum in[r][:]^2
int launch_rms_stats(const float* in, int rows, int E, void* outT, float* ssq, bool is_bf16, hipStream_t st);

// uint8 crops [M,3,32,32] -> normalised patch matrix T [M*4, 768], k = c*256 + py*16 + px (conv1 weight order)
int launch_patchify(const uint8_t* crops, void* outT, int M, bool is_bf16, hipStream_t st);
// tokens = LN_pre(concat(cls, patches) + pos): pre fp32 [M*4,768] -> x fp32 [M*5,768] and / or xT (operand type)
int launch_vit_embed(const float* pre, const float* cls, const float* pos, const float* g, const float* b,
                     float* x, void* xT, int M, bool is_bf16, hipStream_t st);
// first bbox-MLP layer: relu(W[768,4] . (bbox/[256,128,128,256]) + b) -> T [R,768]
int launch_bbox_l1(const long long* bbox, const float* W, const float* b, void* outT, int R, int Nout,
                   bool is_bf16, hipStream_t st);
// first action-embedding layer for one key: x = idx / bins ; relu(W[256,K] x + b) -> T [R, ldo] at column col0
int launch_action_l1(const long long* idx, int K, const float* W, const float* b, void* outT, int R, int ldo,
                     int col0, bool is_bf16, hipStream_t st);
// out[r, :] += table[sel[r / group]][:]   (obs_fusion end-effector term)
int launch_add_row_table(float* out, int rows, int E, const float* table, const long long* sel, int group,
                         hipStream_t st);
// prompt assembly: x[b,l,:] = word table row / object token / 0 ; mask likewise
int launch_prompt_assemble(const int* tok_src, const long long* word_ids, const float* word_table,
                           const float* obj_tokens, const uint8_t* obj_mask, float* x, uint8_t* mask, int rows,
                           int E, hipStream_t st);
// the same gather as the entry of the T5 stack's fused-RMSNorm chain: operand-type rows xT [rows, E] + ssq[r] = sum xT-source[r][:]^2 (what
// launch_rms_stats would produce from the fp32 rows, bit for bit) + mask; the fp32 prompt is not materialised
int launch_prompt_assemble_stats(const int* tok_src, const long long* word_ids, const float* word_table, const float* obj_tokens,
                                 const uint8_t* obj_mask, void* xT, float* ssq, uint8_t* mask, int rows, int E, bool is_bf16, hipStream_t st);
// rows [L][N] of the operand type -> [N / D][L][D] (one sample's block of the head-major prompt K / V cache)
int launch_rows_to_headmajor(const void* in, void* out, int L, int N, int D, bool is_bf16, hipStream_t st);
// decoder input: interleave [o_1..o_Q, a] per step, cumsum position ids, + positions_embed
int launch_dec_embed(const float* obs_tok, const uint8_t* obs_mask, const float* act_tok, const float* pos_table,
                     int n_pos, float* x32, void* xT, uint8_t* mask, int T, int B, int Q, int L_act, int E,
                     bool is_bf16, hipStream_t st);
// one env step of incremental decoding: newest tokens ([prev action,] Q observation tokens) of every sample -> x32 / xT
// [B, Q+has_act, E]; position ids continue poscnt[b]; hist_mask[b][L_hist + i] and poscnt[b] are updated
int launch_dec_embed_step(const float* obs_tok, const uint8_t* obs_mask, const float* act_tok, const float* pos_table, int n_pos,
                          float* x32, void* xT, uint8_t* hist_mask, int* poscnt, int L_hist, int Lmax, int B, int Q,
                          int has_act, int E, bool is_bf16, hipStream_t st, uint8_t* fresh = nullptr);
// per-sample episode restart: for flags[b] != 0 (device array) mask sample b's whole history, reset its position counter, mark its
// next action slot absent (fresh[b] = 1, consumed by launch_dec_embed_step)
int launch_restart_samples(const uint8_t* flags, uint8_t* hist_mask, int* poscnt, uint8_t* fresh, int B, int Lmax, hipStream_t st);
// prompt + xattn_positions_embed[cumsum(mask)-1] -> T [B*Lp, E]; input strides in elements (seq-first views ok)
int launch_prompt_pos(const float* prompt, long long sb, long long sl, const uint8_t* mask, const float* pos_table,
                      int n_pos, void* outT, int B, int Lp, int E, bool is_bf16, hipStream_t st);
// out[t,b,:] = x[b, (Q-1) + (Q+1) t, :]
int launch_gather_pred(const float* x, float* out, int T, int B, int Q, int Lq, int E, hipStream_t st);

// ---- baseline policies (baseline_kernels.hip): whole RGB frames, decoder-only sequences
// uint8 frames [M,3,H,W] -> normalised patch matrix T [M*(H/P)*(W/P), 3*P*P], k = c*P*P + py*P + px
int launch_patchify_rect(const uint8_t* img, void* outT, int M, int H, int W, int P, bool is_bf16, hipStream_t st);
// tokens = LN_pre(concat([cls,] patches) + pos): pre fp32 [M*n_patch,768] -> x fp32 [M*S,768] and / or xT; cls == nullptr: S = n_patch
int launch_vit_embed_rect(const float* pre, const float* cls, const float* pos, const float* g, const float* b, float* x, void* xT,
                          int M, int S, int n_patch, bool is_bf16, hipStream_t st);
// out[r,:] = src[r % period,:] (fp32)
int launch_broadcast_rows(const float* src, float* out, long long rows, int E, int period, hipStream_t st);
// decoder-only input [B,L,E]: [prompt | sep | (Q obs tokens, action)*] + positions_embed, key mask [B,L]
int launch_seq_embed(const float* prompt, long long sb, long long sl, const uint8_t* pmask, const float* sep, const float* obs_tok,
                     const float* act_tok, const float* pos_table, int n_pos, float* x32, void* xT, uint8_t* mask, int B, int L, int Lp,
                     int Q, int E, bool is_bf16, hipStream_t st);
int launch_fill_u8(uint8_t* p, long long n, uint8_t v, hipStream_t st);

// ---------------------------------------------------------------- attention
// ViT: 5-token (S <= 8) multi-head attention on packed qkv T [M*S, 3*W] -> T [M*S, W]; head dim 32
// out8 (bf16 mode): write the result as fp8 e4m3(value * inv8) [M*S, W] bytes INSTEAD of the operand-type output
int launch_vit_attn(const void* qkv, void* out, int M, int S, int W, int heads, bool is_bf16, hipStream_t st, void* out8 = nullptr, float inv8 = 1.0f);
// last ViT block: cls-token query only. q T [M, W], kv T [M*S, 2W] -> out T [M, W]
int launch_vit_attn_cls(const void* q, const void* kv, void* out, int M, int S, int W, int heads, bool is_bf16, hipStream_t st,
                        void* out8 = nullptr, float inv8 = 1.0f);

enum AttnMode : int { ATTN_T5 = 0, ATTN_CROSS = 1, ATTN_CAUSAL = 2 };
struct AttnArgs {
  const void* q = nullptr; int ldq = 0;   // row (b*Lq + i), head h at column h*D
  const void* k = nullptr; int ldk = 0;   // row (b*Lk + j)
  const void* v = nullptr; int ldv = 0;
  void* out = nullptr; int ldo = 0;
  const uint8_t* kmask = nullptr;         // [B, Lk] 1 = attend; masked keys get score finfo(fp32).min
  const float* relbias = nullptr;         // T5: [H][2*Lk-1], index (j - i) + Lk - 1
  int bias_far = 0;                       // T5: > 0 promises that relbias[h] is CONSTANT for j - i >= bias_far and for j - i <= -bias_far (the bucketed
                                          // T5 table is, from |j - i| = 91 on): key tiles wholly beyond it take the constant instead of per-score reads
  int B = 0, H = 0, Lq = 0, Lk = 0, D = 0;
  // K / V addressing: element (b, h, j, d) of K sits at k + b * k_bs + h * k_hs + j * ldk + d (V likewise). 0 = the default layout, rows
  // (b * Lk_rows + j) of width ldk with the heads side by side: k_bs = Lk_rows * ldk, k_hs = D. A head-major cache ([B][heads][Lk][D]: ldk = D,
  // k_hs = Lk * D, k_bs = heads * Lk * D) is read with every (batch, head)'s keys contiguous.
  long long k_bs = 0, v_bs = 0;
  int k_hs = 0, v_hs = 0;
  float scale = 1.0f;
  int mode = ATTN_CROSS;
  // incremental decoding (keys / values / key mask live in an episode cache with room for Lk_rows >= Lk rows per sample,
  // the queries are the newest tokens): K/V row = b * Lk_rows + j, mask = kmask[b * Lk_rows + j] (0 -> Lk), and the causal
  // rule compares key j with the query's GLOBAL position i + q_off.
  int Lk_rows = 0;
  int q_off = 0;
  const Tuning* tune = nullptr;   // the calling handle's knobs (nullptr: process defaults)
};
// generic exact kernel (any T); the MFMA flash kernel (bf16, D in {32,64})
int launch_attn_generic(const AttnArgs& a, bool is_bf16, hipStream_t st);
int launch_attn_mfma(const AttnArgs& a, hipStream_t st);

}  // namespace vima
