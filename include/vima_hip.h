/* vima_hip.h -- C ABI of the MI355X-native (gfx950) VIMA policy forward pass.
 *
 * The reference (vimalabs/VIMA) is pure Python: it has no FFI of its own, its seam is the method surface of
 * `VIMAPolicy` (vima/policy/vima_policy.py:11-322) and the checkpoint contract of `create_policy_from_ckpt`
 * (vima/__init__.py:7-16). This library is what a binding for that seam binds to: one opaque handle per
 * (model, device) and one entry point per policy method. All tensors are plain device pointers owned by the
 * caller (row-major, contiguous unless strides are given); the library never frees caller memory, launches on
 * the caller's hipStream_t and keeps its own workspace. Every function returns 0 on success, otherwise an error
 * code whose text is available from vima_last_error(). The Python host side (vima_amd/policy.py) loads this
 * library with ctypes; INTEGRATION.md shows the binding.
 *
 * Conventions: f32 = float, i64 = int64_t, u8 = uint8_t (also used for torch.bool). Views are ordered
 * sorted(["front","top"]) = {front, top} (obj_encoder.py:31). E = embed_dim.
 */
#ifndef VIMA_HIP_H
#define VIMA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct VimaHandle VimaHandle;
typedef void* vima_stream_t; /* hipStream_t */

/* FP8W (BASELINE.json configs[4]): bf16 activations and bf16 matrix instructions, but the weights of the large Linear
 * layers are stored as OCP FP8 E4M3 with one fp32 scale per output channel (half the weight bytes in HBM / L2 / LDS);
 * the GEMM kernels widen the fp8 fragments to bf16 in registers. Logit error vs the fp32 reference is a few 1e-3 (weights
 * only carry 3 mantissa bits) -- measured and reported by the tests, not covered by the 1e-3 gate of the bf16 mode. */
enum { VIMA_PRECISION_FP32 = 0, VIMA_PRECISION_BF16 = 1, VIMA_PRECISION_FP8W = 2, VIMA_PRECISION_FP8 = 3 };
/* FP8W: bf16 activations and matrix instruction, OCP e4m3 weights (+ one fp32 scale per output channel) for the large Linear layers.
 * FP8 : FP8W plus e4m3 ACTIVATIONS into the large GEMMs of the T5 stack, the ViT and the decoder's prompt K/V projection (one static
 *       scale per layer and site, calibrated by the handle's first pass through each, which runs the FP8W kernels) on
 *       v_mfma_scale_f32_32x32x64_f8f6f4 -- BASELINE.json configs[4]. Shapes the fp8 kernel does not cover, and handles whose kernel-
 *       selection options leave that kernel unreachable (gemm_persist = 0, gemm_tile = 1, gemm_raster != 0, gemm_epi = 0, operands beyond the
 *       32-bit LDS-DMA offsets), keep bf16 activations; no documented option makes a forward fail.
 *       CALIBRATION CONTRACT: the scales are STATIC and come from the first eligible pass of each group, so (i) that first call runs the
 *       FP8W kernels and the SAME input gives (slightly) different outputs on call 1 and on call 2 -- warm a handle up once on
 *       representative data before comparing or serving; (ii) scale = headroom x max |x| / 448 with option "fp8_headroom_pct" (default 125):
 *       later activations up to 1.25x the calibrating maximum stay representable, larger ones SATURATE at +-448 x scale silently (call
 *       "fp8_recalibrate" when the input distribution changes); (iii) a calibrating batch that holds an inf / NaN activation is refused
 *       (the call fails, nothing is frozen); (iv) vima_decode_restart rebuilds the restarted samples' prompt K/V rows with bf16
 *       activations (M = Lp rows is below the fp8 kernel's size), i.e. a restarted sample matches a fresh episode to the fp8 tolerance, not
 *       bit for bit. */

/* Which policy class of vima/policy/ the handle implements. VIMA is the hot path (vima_policy.py); the other three are the
 * reference's baseline policies, which consume whole 64x128 RGB frames instead of object crops:
 *   GPT      VIMAGPTPolicy      (vima_gpt_policy.py): cls feature of a rectangular ViT per view, decoder-only HFGPT over
 *                                [prompt | sep | obs, action, obs, ...]
 *   GATO     VIMAGatoPolicy     (vima_gato_policy.py): the 2 x 8 patch tokens of every frame pair go into the same sequence
 *   FLAMINGO VIMAFlamingoPolicy (vima_flamingo_policy.py): patch tokens resampled to 4 latents by a Perceiver, then XAttnGPT
 * For them xf_n_layers / sattn_n_heads are the constructor's n_layer / n_head (dt_n_layers / dt_n_heads); xattn_n_heads is
 * only used by FLAMINGO (pass n_head otherwise). */
enum { VIMA_POLICY_VIMA = 0, VIMA_POLICY_GPT = 1, VIMA_POLICY_GATO = 2, VIMA_POLICY_FLAMINGO = 3 };

/* Constructor arguments of VIMAPolicy (vima_policy.py:12-19) plus the two table sizes the reference hard-codes
 * (xattn_n_positions=256 at vima_policy.py:30, n_positions=512 at xattn_gpt.py:18). */
typedef struct VimaConfig {
  int32_t embed_dim;
  int32_t xf_n_layers;
  int32_t sattn_n_heads;
  int32_t xattn_n_heads;
  int32_t xattn_n_positions;
  int32_t n_positions;
  int32_t precision; /* VIMA_PRECISION_*: operand type T of the matrix-core GEMMs / attention. Accumulation, LayerNorm /
                        RMSNorm and softmax statistics are always fp32; the DECODER's residual stream is fp32; the residual
                        streams of the T5 stack and of the ViT are carried in T as well when option "stream_T" is 1 (the
                        default: bf16 streams in the BF16 / FP8W precisions, fp32 in FP32) and in fp32 when it is 0. */
  int32_t policy_kind; /* VIMA_POLICY_* (ABI version 3; 0 = VIMAPolicy) */
} VimaConfig;

/* ---- lifetime ------------------------------------------------------------------------------------------------ */
/* replaces VIMAPolicy.__init__ (vima_policy.py:12-114) */
int vima_create(const VimaConfig* cfg, int device, VimaHandle** out);
void vima_destroy(VimaHandle* h);
const char* vima_last_error(void);
/* 2: + vima_crop_objects, vima_comm_*, vima_allgather_logits, vima_prof_read_ex, precision fp8w; 3: + VimaConfig.policy_kind and the
 * baseline-policy entry points; 4: + VIMA_PRECISION_FP8, vima_fp8_act_scales, vima_decode_restart, vima_prof_read_gemm_kernels;
 * 5: + vima_prof_read_gemm_launches, option "fp8_headroom_pct".
 * The ONE place the number lives: the library returns it, the ctypes binding parses it from this header and refuses a mismatch. */
#define VIMA_ABI_VERSION 5
int vima_abi_version(void);

/* ---- weights: the reference checkpoint contract -------------------------------------------------------------- */
/* replaces nn.Module.load_state_dict(strict=True) as used by create_policy_from_ckpt (vima/__init__.py:11-14).
 * `name` is the reference state_dict key, `data` a HOST fp32 array in the reference's own
 * layout (HF Conv1D weights are [in,out], nn.Linear [out,in] -- the library repacks). Integer buffers of the
 * reference state_dict (position_ids, attn.bias, ...) carry no information and are not passed. */
int vima_set_param(VimaHandle* h, const char* name, const float* data, const int64_t* shape, int ndim);
/* Packs / converts / uploads; fails (strict) listing any key that was never set or has the wrong shape. */
int vima_finalize_params(VimaHandle* h);
/* Host-only (no GPU needed): writes the NUL-separated list of state_dict keys vima_finalize_params requires for
 * this config into buf (when it fits) and returns the number of bytes needed. */
int64_t vima_required_params(const VimaConfig* cfg, char* buf, int64_t buflen);

/* ---- policy methods ------------------------------------------------------------------------------------------ */
/* ObjEncoder.forward (obj_encoder.py:66-95): crops[v] u8 [n,Qv,3,32,32], bbox[v] i64 [n,Qv,4] (xc,yc,h,w px)
 * -> out f32 [n, 2*Qv, E] with the object axis ordered [front objs..., top objs...]. */
int vima_obj_encode(VimaHandle* h, const uint8_t* const crops[2], const int64_t* const bbox[2], int n, int qv,
                    float* out, vima_stream_t stream);

/* VIMAPolicy.forward_obs_token (vima_policy.py:242-259): leading dims [T,B] flattened to n = T*B.
 * mask[v] u8 [n,Qv]; ee i64 [n] in {0,1} -> out_tokens f32 [n, 2*Qv, E], out_mask u8 [n, 2*Qv]. */
int vima_obs_encode(VimaHandle* h, const uint8_t* const crops[2], const int64_t* const bbox[2],
                    const uint8_t* const mask[2], const int64_t* ee, int n, int qv, float* out_tokens,
                    uint8_t* out_mask, vima_stream_t stream);

/* VIMAPolicy.forward_prompt_assembly (vima_policy.py:161-240). The python double loop over token types is replaced
 * by an index array built on the host: tok_src i32 [B*Lp] (device), one code per (b, l):
 *     >= 0 : index into word_ids            (word token, mask True)
 *     -1   : padding                        (zeros, mask False)
 *     <= -2: object token -(code+2) = img*2*Qv + q of the encoded prompt images (mask = object mask)
 * word_ids i64 [n_words]; crops/bbox/mask as above with n = n_img.
 * -> out_tokens f32 [B, Lp, E] (batch-first; the reference returns the [Lp,B,E] transposed view of it),
 *    out_mask u8 [B, Lp]. */
int vima_prompt_encode(VimaHandle* h, const int64_t* word_ids, int n_words, const uint8_t* const crops[2],
                       const int64_t* const bbox[2], const uint8_t* const mask[2], int n_img, int qv,
                       const int32_t* tok_src, int B, int Lp, float* out_tokens, uint8_t* out_mask,
                       vima_stream_t stream);

/* T5PromptEncoder.forward (prompt_encoder.py:30-58) on already assembled embeddings x f32 [B,L,768], mask u8 [B,L]
 * -> out f32 [B,L,768] (before t5_prompt_encoder_post_layer). Exposed for operator-level parity tests. */
int vima_t5_encode(VimaHandle* h, const float* x, const uint8_t* mask, int B, int L, float* out,
                   vima_stream_t stream);

/* VIMAPolicy.forward (vima_policy.py:116-159) -> XAttnGPT.forward (xattn_gpt.py:73-139).
 * obs_tok f32 [T,B,Q,E], obs_mask u8 [T,B,Q], act_tok f32 [L_act,B,E] or NULL (L_act in {T-1, T}),
 * prompt f32 element (b,l,e) at prompt[b*stride_b + l*stride_l + e] (so both the [B,Lp,E] buffer and the reference's
 * sequence-first [Lp,B,E] layout can be passed), prompt_mask u8 [B,Lp] -> out f32 [T,B,E] (sequence-first).
 * kv_cache_mode: the per-layer prompt K/V projection (components.py:175) is loop-invariant over the env steps of an
 * episode. 0 = stateless (recompute, like the reference); 1 = compute it into the handle's
 * cache and use it; 2 = reuse the cache of the last mode-1 call (same B, Lp; `prompt` is not read). Outputs are
 * bit-identical in all three modes. */
int vima_decode(VimaHandle* h, const float* obs_tok, const uint8_t* obs_mask, const float* act_tok, int T, int B,
                int Q, int L_act, const float* prompt, int64_t stride_b, int64_t stride_l,
                const uint8_t* prompt_mask, int Lp, int kv_cache_mode, float* out, vima_stream_t stream);

/* Incremental form of vima_decode for the env-step loop of an episode (scripts/example.py:135-190 re-feeds the whole
 * history every step): call with step = 0, 1, 2, ... and only the NEWEST tokens -- obs_tok f32
 * [B,Q,E] / obs_mask u8 [B,Q] of env step `step` and, for step > 0, act_tok f32 [B,E] = the embedded action of step-1.
 * The handle keeps the per-layer prompt K/V (built at step 0, so `prompt` must be valid there) and the self-attention
 * K/V, key mask and position counters of the history. out f32 [B,E] = the predicted action token of this step, equal
 * (up to floating-point reassociation) to row `step` of vima_decode on the full history. step 0 starts a new
 * episode; any other step must follow step-1 with the same B, Q, Lp. */
int vima_decode_step(VimaHandle* h, const float* obs_tok, const uint8_t* obs_mask, const float* act_tok, int step, int B,
                     int Q, const float* prompt, int64_t stride_b, int64_t stride_l, const uint8_t* prompt_mask,
                     int Lp, float* out, vima_stream_t stream);
/* Per-sample episode restart inside a running batch of incremental decoding (no reference counterpart: scripts/example.py:97-110 runs
 * one episode at a time; batched environments end their episodes at different steps). For every sample with restart[b] != 0 (HOST
 * array of B bytes): its cached history is masked out, its position counter restarts at 0, its next step has no previous action, and
 * its rows of the per-layer prompt K/V cache are rebuilt from ITS row of `prompt` / `prompt_mask` (the new episode's prompt; layout and
 * strides as in vima_decode_step; rows of the other samples are not read). The batch then keeps calling vima_decode_step with
 * step + 1 (pass any act_tok row for restarted samples: it is ignored). n_positions still bounds the steps since the batch's step 0. */
int vima_decode_restart(VimaHandle* h, const uint8_t* restart, int B, const float* prompt, int64_t stride_b, int64_t stride_l,
                        const uint8_t* prompt_mask, int Lp, vima_stream_t stream);

/* VIMAPolicy.forward_action_decoder (vima_policy.py:264-265 -> action_decoder.py:51-52,165-166): tokens f32 [R,E]
 * -> raw logits f32 [R,700] = concat over keys (pose0_position, pose0_rotation, pose1_position, pose1_rotation) of
 * the 12 MLP outputs; the MultiCategorical wrapper (dists.py) stays on the host side. */
int vima_action_head(VimaHandle* h, const float* tokens, int R, float* out_logits, vima_stream_t stream);

/* VIMAPolicy.forward_action_token (vima_policy.py:261-262 -> :301-322 -> action_embd.py:29-56): discrete bin
 * indices i64, keys in sorted order: pose0_position [R,2], pose0_rotation [R,4], pose1_position [R,2],
 * pose1_rotation [R,4] -> out f32 [R,E]. */
int vima_action_embed(VimaHandle* h, const int64_t* const idx[4], int R, float* out, vima_stream_t stream);

/* ---- baseline policies (policy_kind != VIMA_POLICY_VIMA) ---------------------------------------- */
/* Tokens one frame pair contributes: Q = 1 (GPT), 16 (GATO: 8 patches x 2 views), 4 (FLAMINGO: Perceiver latents); feature
 * width of obj_encoder's output Eo = 2E (GPT: views concatenated on the feature axis, obj_encoder.py:232-245) or E. */
int vima_rgb_tokens_per_image(const VimaConfig* cfg);
/* obj_encoder.forward of the handle's policy (MultiViewRGBEncoder obj_encoder.py:207-246, GatoMultiViewRGBEncoder :98-145,
 * MultiViewRGBPerceiverEncoder :148-204): rgb[v] u8 [n,3,64,128] -> out f32 [n, Q, Eo]. */
int vima_rgb_encode(VimaHandle* h, const uint8_t* const rgb[2], int n, float* out, vima_stream_t stream);
/* forward_obs_token (vima_gpt_policy.py:249-259, vima_gato_policy.py:253-264, vima_flamingo_policy.py:216-227): leading dims
 * [T,B] flattened to n; ee i64 [n] in {0,1} -> out f32 [n, Q, E] (GPT: Q = 1, the reference returns [T,B,E]). */
int vima_rgb_obs_encode(VimaHandle* h, const uint8_t* const rgb[2], const int64_t* ee, int n, float* out, vima_stream_t stream);
/* forward_prompt_assembly (vima_gpt_policy.py:189-247, vima_gato_policy.py:190-251, vima_flamingo_policy.py:156-214): like
 * vima_prompt_encode with whole frames: tok_src codes >= 0 word index, -1 padding, <= -2 image token -(code+2) = img*Q + q;
 * every assembled token is valid (these policies have no object masks). -> out_tokens f32 [B,Lp,E], out_mask u8 [B,Lp]. */
int vima_rgb_prompt_encode(VimaHandle* h, const int64_t* word_ids, int n_words, const uint8_t* const rgb[2], int n_img,
                           const int32_t* tok_src, int B, int Lp, float* out_tokens, uint8_t* out_mask, vima_stream_t stream);
/* VIMAGPTPolicy.forward / VIMAGatoPolicy.forward (vima_gpt_policy.py:118-187, vima_gato_policy.py:115-188) -> HFGPT.forward
 * (gpt/gpt.py:45-80): sequence [prompt (Lp) | prompt_sep_token | o_1..o_Q, a, o_1..o_Q, a, ...] per sample, key mask = prompt
 * mask then ones, position ids continuing after the sample's valid prompt tokens. obs_tok f32 [T,B,Q,E], act_tok f32
 * [L_act,B,E] or NULL (L_act in {T-1, T}), prompt/strides/mask as in vima_decode -> out f32 [T,B,E].
 * (VIMAFlamingoPolicy.forward is vima_decode with an all-ones obs_mask.) */
int vima_seq_decode(VimaHandle* h, const float* obs_tok, const float* act_tok, int T, int B, int L_act, const float* prompt,
                    int64_t stride_b, int64_t stride_l, const uint8_t* prompt_mask, int Lp, float* out, vima_stream_t stream);

/* ---- image preprocessing in front of the policy ------------------------------------------- */
/* The per-object work of prepare_obs / prepare_prompt (/root/reference/scripts/example.py:374-473 and :243-371; numpy +
 * cv2 per object on the host there): for every frame and every object id, segmentation mask -> pixel bbox ->
 * inclusive crop -> zero-pad to a square -> cv2.resize(32x32, INTER_AREA) -> uint8, with the reference's slot order
 * (objects covering >= 2 pixels first, in obj_ids order; the rest zero rows with mask 0).
 * rgb u8 [n_frames,3,H,W], segm [n_frames,H,W] of uint8 (segm_elem_bytes 1) or int32 (4), obj_ids i32 [n_obj <= 64]
 * (all device pointers; H, W <= 320) -> crops u8 [n_frames,n_obj,3,32,32], bbox i64 [n_frames,n_obj,4] = (int x-centre,
 * int y-centre, ymax-ymin, xmax-xmin), mask u8 [n_frames,n_obj]: exactly the cropped_img / bbox / mask arrays of ONE view
 * that vima_obs_encode / vima_prompt_encode take. No handle: there are no weights involved. */
int vima_crop_objects(const uint8_t* rgb, const void* segm, int segm_elem_bytes, const int32_t* obj_ids, int n_frames,
                      int n_obj, int H, int W, uint8_t* crops, int64_t* bbox, uint8_t* mask, vima_stream_t stream);

/* ---- multi-GPU: the one exchange step of the data-parallel path ----------------------------------- */
/* The reference has no distributed code (no torch.distributed / NCCL call site anywhere under vima/); batched episodes
 * are independent, so the path shards over one process per GPU with a full weight replica and ONE collective per
 * step: an all-gather of the raw action logits (the output of vima_action_head) over RCCL / xGMI.
 * vima_comm_unique_id: rank 0 creates the 128-byte rendezvous id (ncclGetUniqueId) and hands it to the other ranks
 * out of band (the Python host broadcasts it through the torch.distributed store);
 * vima_comm_create: every rank joins (ncclCommInitRank) on its own `device`;
 * vima_allgather_logits: local f32 [rows_per_rank, width] -> global f32 [world * rows_per_rank, width] on every
 * rank, enqueued on `stream` (no host synchronisation). Equal shard sizes (the host pads ragged tails). */
#define VIMA_COMM_ID_BYTES 128
typedef struct VimaComm VimaComm;
int vima_comm_unique_id(uint8_t id[VIMA_COMM_ID_BYTES]);
int vima_comm_create(const uint8_t id[VIMA_COMM_ID_BYTES], int world, int rank, int device, VimaComm** out);
int vima_comm_world(const VimaComm* c);
int vima_comm_rank(const VimaComm* c);
int vima_allgather_logits(VimaComm* c, const float* local, float* global, int64_t rows_per_rank, int width,
                          vima_stream_t stream);
void vima_comm_destroy(VimaComm* c);

/* ---- operator-level entry points (parity tests / microbenchmarks) ---------------------------------------------- */
/* out = epilogue(A[M,K] . W[N,K]^T) in the handle's precision; fp32 host-visible device buffers in/out, the
 * operand conversion is done internally. act: 0 none, 1 relu, 2 gelu(erf), 3 quickgelu. bias/mul/res may be NULL. */
int vima_op_linear(VimaHandle* h, const float* A, const float* W, const float* bias, const float* mul,
                   const float* res, int M, int N, int K, int act, float* out, vima_stream_t stream);
/* LayerNorm (rms=0) / T5 RMSNorm (rms=1) over rows of length E. */
int vima_op_layernorm(VimaHandle* h, const float* x, const float* gamma, const float* beta, float eps, int rms,
                      int rows, int E, float* out, vima_stream_t stream);
/* attention on fp32 buffers q [B,Lq,H*D], k/v [B,Lk,H*D]; mode 0 T5 (relbias [H][2Lk-1]), 1 cross, 2 causal;
 * impl 0 = exact generic kernel, 1 = MFMA flash kernel (bf16 precision only). */
int vima_op_attention(VimaHandle* h, const float* q, const float* k, const float* v, const uint8_t* kmask,
                      const float* relbias, int B, int H, int Lq, int Lk, int D, float scale, int mode, int impl,
                      float* out, vima_stream_t stream);
/* precision FP8: the calibrated activation scales (dequantisation scale = headroom x max |x| / 448, see VIMA_PRECISION_FP8) of group 0 the T5 stack [12 layers][4 sites:
 * stream before qkv, attention context, stream before wi, ReLU hidden], 1 the ViT [4 blocks][4 sites: ln_1 output, attention output,
 * ln_2 output, QuickGELU hidden], 2 the decoder's prompt K/V projection [1]; returns their number (0 before that group's calibrating
 * pass, < 0 on error). Option "fp8_recalibrate" makes the next pass of every group measure them again. */
int vima_fp8_act_scales(VimaHandle* h, int group, float* out, int max_n);
/* host-only: the fp32 -> OCP FP8 E4M3 (round to nearest even, saturating at 448) encoder the FP8W weight packing uses */
void vima_fp8_e4m3_encode(const float* src, uint8_t* dst, int64_t n);
/* host-only: HF T5 bidirectional relative-position bucket (32 buckets, max distance 128) of (key_pos - query_pos) */
int vima_t5_bucket(int relative_position);

/* ---- tuning / instrumentation ---------------------------------------------------------------------------------- */
/* Per-handle options (every key vima_set_option accepts; unknown keys fail). Defaults in brackets.
 *   kernel selection, GEMM:  "gemm_variant" [1] 1 inline-asm LDS-DMA pipeline, 0 compiler-tracked builtin (128x128 tile only)
 *                            "gemm_tile"    [0] 0 auto, 1 force 128x128, 2 force 256x256, 7 force 32x64, 8 force 64x64,
 *                                               10 / 11 / 12 force the resident-K kernel's 32x32 / 64x32 / 64x64 tile
 *                            "gemm_persist" [1] large bf16 GEMMs on the persistent 256x256 kernels
 *                            "gemm_pp"      [1] ping-pong (8-phase) main loop of the persistent kernel (0: the round-2 loop)
 *                            "gemm_wide"    [0] 256x384 persistent tile where N % 384 == 0
 *                            "gemm_q4"      [6] gemm_q4_kernel, FOUR waves (one per SIMD) with a 384-column tile, where N % 384 == 0, K % 128 == 0 (bf16-output /
 *                                               bf16-stream / head-major epilogues): 1 its 256x384 tile wherever it fits, 2 its 128x384 tile wherever it fits,
 *                                               3 the 128x384 tile where it needs fewer rounds of the chip than the 256x256 tiling (batch 32), 6 (default) the 256x384 tile
 *                                               for GEMMs of >= 32 768 rows (wall clock of the two-stream model: -1.4 % on the headline step), 0 off; bit-identical results in every mode
 *                            "gemm_small"   [1] 64x64 / 32x64 tiles for grids that would leave most CUs idle
 *                            "gemm_resident" [1] underfilled grids on gemm_resident_kernel ((almost) the whole K extent in flight, one
 *                                               barrier per chunk of K-slices; bit-identical to the ring tiles), 0: the 4-deep ring tiles
 *                            "gemm_res_maxwg" [256] largest grid (workgroups) gemm_resident_kernel takes for M > 32
 *                            "gemm_skinny"  [1] GEMMs of at most 32 rows (one env step at batch <= 3, the action head ...) on gemm_skinny_kernel: K split
 *                                               over the 4 / 8 / 16 waves of a workgroup, operands straight from global memory, 8 / 16 / 32-column tiles.
 *                                               A DIFFERENT summation order than every other tile (which all agree bit for bit): a sample evaluated alone and the
 *                                               same sample inside a large batch then agree to bf16 rounding (~2e-4 on logits of 0.08), not bit for bit;
 *                                               0 = the resident 32x32 tile for these shapes
 *                            "gemm_flat"    [1] gemm_pp_kernel enumerates its tiles plainly (no XCD raster) where the raster's padding of the A panels to a
 *                                               multiple of 8 would cost a round and the grid is at most two tiles per workgroup (M = 2304: the GEGLU pair
 *                                               of an incremental env step at batch 256); bit-identical, 0 = always the raster
 *                            "gemm_res_nch" [0] chunk buffers of its LDS ring: 0 = default (4 / 5 / 4 for the 32x32 / 64x32 / 64x64 tile: 128 KiB),
 *                                               up to 5 / 6 / 5 (160 KiB)
 *                            "gemm_splitk"  [0] deterministic two-pass split-K for underfilled grids with K >= 1536
 *                            "gemm_raster"  [0] tile order of the one-tile-per-workgroup kernels: 0 XCD x n-walk, 1 XCD x
 *                                               resident n-group, 2 row-major
 *                            "gemm_epi"     [1] LDS-transposed row-contiguous epilogue (0: direct per-lane stores)
 *   kernel selection, attention: "attn_impl" [1] 1 MFMA flash kernels, 0 exact generic kernel
 *                            "attn_split"   [1] split-key 4-wave kernel for <= 32 queries
 *                            "attn4_min_lq" [64] query count from which the 4-wave LDS-shared flash kernel is used
 *                            "attn_qg"      [1] 32-query groups per wave in that kernel (2: 64 queries per wave from 256 queries on)
 *   numerics / fusion:       "stream_T"     [1] T5 / ViT residual streams carried in the operand type (see VimaConfig)
 *                            "t5_fuse_rms"  [1] T5 RMSNorms folded into the neighbouring GEMMs
 *                            "vit_prune_last" [1] last ViT block evaluated for the cls row only (identical values)
 *                            "fp8_recalibrate" (any value) VIMA_PRECISION_FP8: the next pass of every group measures the activation scales again
 *                            "fp8_headroom_pct" [125] VIMA_PRECISION_FP8: scale = pct/100 x max |x| / 448 (>= 100); setting it re-calibrates
 *                            "kv_headmajor" [1] decoder prompt K / V written head-major ([B][2 heads][Lp][head dim]) where the projection runs on the
 *                                               persistent 256x256 GEMM (batch x prompt large enough): same values, contiguous reads in the cross attention
 *                            "geglu_pair"   [1] a GEGLU whose two products read the same input (the decoder blocks' MLP) as ONE GEMM launch over
 *                                               block-interleaved weights (128x128 ring tile, or the persistent 256x256 kernel's pair epilogue from 160 full
 *                                               tiles of the interleaved [M, 8E] problem on: 2048 rows at E = 768) where the grid is beyond the
 *                                               dual-accumulator form: same values
 *                            "ln_fuse"      [1] decoder LayerNorms (bf16 precisions): ln_2 of layer i and the query pre-LN of XAttention in layer i + 1 as ONE
 *                                               launch (bit-identical); the pre-LN in front of XAttention's feed-forward folded into the GEMMs either side of
 *                                               it (sums / sums of squares per 32 columns from attention_out's epilogue, mean / rstd applied to the GELU'd factor
 *                                               in the GEGLU pair's epilogue, gamma folded into the weight) where a pair form exists for the row count --
 *                                               the operand is then bf16(a) instead of bf16(LN(a)): same precision class, different rounding point
 *   scheduling:              "dual_stream"  [1] independent halves of the work on an auxiliary HIP stream
 *                            "graphs"       [0] replay the per-step entry points as captured hipGraphs
 *                            "vit_chunk"    [16384] crops per ViT pass
 *                            "dual_t5_rows" [0] 0 = automatic (two streams unless the batch is one nearly full round of 256x256 tiles, 14.4 k .. 21.8 k rows), > 0: batch x prompt length from which dual_stream splits the T5 stack over two streams; "dual_vit_crops" [8192] the same for the ViT's chunks (crops)
 *                            "t5_pad"       [1] the T5 stack is computed on the next multiple of 256 rows when batch x prompt length (>= 2048 rows per stream) is not one (zero pad rows, never read)
 *                            "vit_pad"      [1] ViT passes of >= 1024 crops are computed on a multiple of 256 crops (zero-image pad crops, never read) so their GEMMs stay on the 256-row tile kernels at any crop count
 *   test / instrumentation:  "op_bf16_out", "op_stream_T" (route vima_op_linear through the bf16-output / bf16-residual
 *                            epilogues; op_stream_T also feeds vima_op_layernorm a bf16 input), "op_bias_far" (vima_op_attention, T5 mode: promise
 *                            that the relbias table is constant from that |key - query| on, as the bucketed T5 table is from 91; 0 = no promise),
 *                            "gemm_dbg_ptr", "attn_dbg_ptr" (device buffers for shader-clock stamps, 0 = off)
 * Process-wide A/B switches read once from the environment (results do not depend on them beyond fp32 summation order inside
 * a LayerNorm row): VIMA_LN_ROWS2=0 (one-wave-per-row LayerNorm instead of the half-wave kernel), VIMA_VIT_ATTN_LDS=0 (per-thread
 * ViT attention instead of the LDS-staged kernel; identical bits), VIMA_GEMM_* (defaults of the GEMM options above),
 * VIMA_GEMM_NGROUP_KB (n-group size of the persistent GEMM's raster, default 2560). */
int vima_set_option(VimaHandle* h, const char* key, int64_t value);
/* When enabled every kernel launch is bracketed by HIP events on the launch stream and attributed to a class:
 * 0 = GEMM, 1 = attention, 2 = other. vima_prof_read synchronises and returns per class
 * {milliseconds, launches, algorithmic FLOPs (2*M*N*K for GEMMs, 4*B*H*Lq*Lk*D for attention)}; then resets. */
int vima_prof_enable(VimaHandle* h, int on);
int vima_prof_read(VimaHandle* h, double out_ms[3], int64_t out_launches[3], double out_flops[3]);
/* The same with the GEMM class split in two and the ALGORITHMIC HBM bytes of every GEMM launch (each operand, output and
 * epilogue input counted once): class 0 = GEMMs without, class 3 = GEMMs with an fp32-residual epilogue (read fp32
 * residual, write the fp32 stream [+ operand-type copy + RMS partials]: the HBM-heavy ones), 1 = attention, 2 = other. */
int vima_prof_read_ex(VimaHandle* h, double out_ms[4], int64_t out_launches[4], double out_flops[4], double out_bytes[4]);
/* The GEMM launches recorded since vima_prof_enable, grouped by the KERNEL the launcher chose: ids[i] = kind * 1000 +
 * (activation + 1) * 10 + epilogue (1 bf16 output, 2 GEGLU gate, 3 fp32 output +- residual, 4 bf16 residual stream, 5 bf16 output written head-major, 6 GEGLU pair over block-interleaved weights),
 * kind 1 gemm_pp_kernel<ACT, EPI>, 2 gemm_persistent_kernel<ACT, EPI>, 3 gemm_wide_kernel,
 * 4..7 gemm_kernel with the 256x256 / 128x128 / 64x64 / 32x64 tile (kind 5 with N = 8 x embed_dim: the block-interleaved GEGLU pair, two products per launch), 8 two-pass split-K, 10..12 gemm_resident_kernel with the
 * 32x32 / 64x32 / 64x64 tile (also the grouped launch of the action head's last layers), 15 / 16 its GEGLU-pair form (two products per launch:
 * 32x32 / 64x64 tile), 17 / 18 gemm_skinny_kernel (at most 32 rows) / its GEGLU-pair form; per kernel the summed milliseconds,
 * launches, algorithmic FLOPs and algorithmic HBM bytes. Returns the number of kernels (<= max_n) or a negative error; does
 * NOT reset the records (call it before vima_prof_read / vima_prof_read_ex). */
int vima_prof_read_gemm_kernels(VimaHandle* h, int max_n, int32_t* ids, double* ms, int64_t* launches, double* flops, double* bytes);
/* The same records ONE BY ONE in launch order (ABI version 5): ids[i] as above, mnk[3 i .. 3 i + 2] = M, N, K of launch i, us[i] (may be
 * NULL) its HIP-event duration in microseconds. With "dual_stream" 0 the launch order is the dispatch order of a rocprofv3 trace of the
 * same call sequence, which is how scripts/pmc_summary.py attributes per-dispatch counter values to GEMM SHAPES. Returns the number of
 * recorded GEMM launches (it may exceed max_n: only the first max_n are written) or a negative error; does not reset the records. */
int vima_prof_read_gemm_launches(VimaHandle* h, int max_n, int32_t* ids, int32_t* mnk, float* us);
/* bytes currently held by the workspace arena */
int64_t vima_workspace_bytes(VimaHandle* h);
/* hipGraph replay (vima_set_option(h, "graphs", 1)): the per-env-step entry points (vima_obs_encode, vima_decode,
 * vima_decode_step, vima_action_head, vima_action_embed) capture their launch sequence the second time the same call
 * (shapes, pointers, options) is seen and replay it afterwards -- at small batch a step is ~1300 microsecond kernels and
 * the host launch rate is the bound. Results are bit-identical to eager execution. Counts since handle creation. */
int vima_graph_stats(VimaHandle* h, int64_t* replays, int64_t* captures);

#ifdef __cplusplus
}
#endif
#endif /* VIMA_HIP_H */
