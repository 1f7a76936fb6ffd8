// MFMA GEMM for gfx950:  C[M,N] = epilogue(A[M,K] . W[N,K]^T)
//
// Replaces every nn.Linear / HF Conv1D / `@ projection` / patch-embed conv of the reference hot path
// (e.g. components.py:167,175,217,221-226, vit.py:172,189, prompt_encoder.py q/k/v/o,
// T5DenseActDense wi/wo, nn/utils.py build_mlp Linear layers).
//
// Design (CDNA4, wave64):
//  * Tile shapes from one template (every shape accumulates K in the same order: results do not depend on the choice):
//      TileL 256x256, 8 waves (2x4, 128x64 per wave), 128 KiB ring -> 1 workgroup/CU: the big bf16 GEMMs (half the LDS
//            and L2 bytes per FLOP of TileS). Full-tile problems with one of the specialised epilogues run on the
//            PERSISTENT kernel (gemm_persistent_kernel: one workgroup per CU, the K-slices of all its tiles form one
//            LDS-DMA stream, epilogue through a private 32 KiB of LDS), the rest one tile per workgroup.
//      TileS 128x128, 4 waves (2x2, 64x64 per wave), 64 KiB -> 2 workgroups/CU: mid-size problems, fp32 mode
//      Tile64 64x64 (4 waves) / TileXS 32x64 (2 waves), 4-deep rings: grids that would leave most CUs idle (batch 1-32)
//    Measured (profiles/r01_gemm_ablation.md): the 256x256 main loop is bound by the L2->LDS operand path (~19-25
//    B/clk/CU with all CUs active), not by the matrix pipe; small grids by how fast one workgroup walks K.
//  * K is consumed in 128-BYTE row slices (64 bf16 / 32 fp32): each tile row is 8 x 16 B chunks.
//  * global -> LDS with `global_load_lds_dwordx4` (no VGPR round trip). The LDS image is lane-linear, so the
//    bank-conflict swizzle is applied on the SOURCE address: LDS slot (row r, position p) receives global chunk
//    c = p ^ ((r >> 1) & 7); fragments are read back with ds_read_b128 at position c ^ ((r >> 1) & 7).
//    With 128-B rows every 16-lane ds_read_b128 group then touches 16 distinct 16-B slots of the 256-B bank row.
//  * 2 (4 for the small tiles) LDS stages; the LDS-DMA of slice k+1 is issued (inline asm, hidden from hipcc's waitcnt bookkeeping so it is
//    NOT drained in front of the ds_reads) while slice k is multiplied; fragments of step kk+1 are prefetched into a
//    second register set while the MFMAs of step kk run; ONE counted `s_waitcnt vmcnt(n)` + `s_barrier` per K-slice.
//  * operands are fed SWAPPED to the matrix core (W fragment as A-operand, activation fragment as B-operand) so a
//    lane ends up holding 4 CONSECUTIVE output columns of one output row -> 8/16-byte vector epilogue
//    (bias / activation / GEGLU gate multiply / residual / dual fp32+bf16 store) instead of scalar stores.
//  * blockIdx -> tile mapping is XCD-aware: consecutive workgroups on one XCD (blockIdx % 8) walk the n-tiles of
//    the same A row panel, so the panel is fetched from HBM once per XCD L2.
#include "kernels.h"
#include <stdlib.h>
#include <type_traits>

namespace vima {

namespace {

// The main-loop ablation builds behind DESIGN.md 4.2 (timing only, wrong results) are NOT part of this source:
// scripts/ablate/gemm_ablate.patch re-creates them on a scratch copy (scripts/build_ablate.sh), and so does scripts/ablate/gemm_lab_variants.patch
// for round 4's epilogue / cache-policy experiments (-DVIMA_LAB_NORES | _NOSTORE | _SKEW=n timing only,
// -DVIMA_LAB_NT_A | _NT_ST non-temporal A loads / output stores; scripts/micro/build_gemm_lab.sh applies it). The shipped kernels have ONE schedule.
#ifdef VIMA_GEMM_LAB
constexpr bool kLab = true;    // scripts/micro/gemm_lab.hip: only the 256x256 bf16 kernels are instantiated (compile time)
#else
constexpr bool kLab = false;
#endif
constexpr bool kDephase = true;          // waves sharing a SIMD prefetch fragments at different points of a step
constexpr bool kInterleaveDma = true;    // DMA pieces issued between the MFMAs of the last k-step

template <int BM_, int BN_, int WM_, int WN_, int RB_, int NS_, int MINW_ = 2>
struct Tile {
  static constexpr int MINW = MINW_;             // __launch_bounds__ waves per SIMD (1 -> the wave may use all 512 registers)
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
  static constexpr int RB = RB_;                 // bytes of K per tile row (one K-slice): 128 or 64
  static constexpr int NS = NS_;                 // LDS stages
  static constexpr int CPR = RB / 16;            // 16-B chunks per row
  static constexpr int NW = WM * WN;             // waves
  static constexpr int THREADS = NW * 64;
  static constexpr int MI = BM / WM / 32;        // 32x32 MFMA tiles per wave along m
  static constexpr int NI = BN / WN / 32;        // ... along n
  static constexpr int PA = BM * RB / 1024 / NW; // 1-KiB LDS-DMA pieces per wave for the A tile of one K-slice
  static constexpr int PW = BN * RB / 1024 / NW;
  static constexpr int A_BYTES = BM * RB;
  static constexpr int STAGE_BYTES = (BM + BN) * RB;
  static constexpr int SMEM_BYTES = NS * STAGE_BYTES;
  // L2 prefetch of the A stream with 4-byte LDS-DMA touches, 2 slices ahead of the DMA. Measured on MI355X: -3 % (the
  // loop is not bound by the HBM latency of the A stream although a cache-resident A runs +25 %), so it is compiled out.
  static constexpr bool PREFETCH = false;
  static constexpr int SMEM_ALLOC = SMEM_BYTES + (PREFETCH ? NW * 256 : 0);
};
using TileS = Tile<128, 128, 2, 2, 128, 2>;   // 64 KiB, 2 workgroups / CU
using Tile64 = Tile<64, 64, 2, 2, 128, 4>;    // 64 KiB ring of four 16-KiB slices, 4 waves of 32x32: underfilled grids (M <= 512 or so)
using TileXS = Tile<32, 64, 1, 2, 128, 4>;    // 48 KiB ring of four 12-KiB slices, 2 waves of 32x32: M <= 32 (one env step at batch <= 4)
using TileL = Tile<256, 256, 2, 4, 128, 2>;   // 128 KiB, 1 workgroup / CU, 8 waves
using Tile64x128 = Tile<64, 128, 2, 2, 128, 4>;   // 96 KiB ring of four 24-KiB slices, 4 waves of 32x64 (two independent accumulators per wave)
using Tile128x64 = Tile<128, 64, 2, 2, 128, 4>;   // the transposed shape: 4 waves of 64x32

template <int RB> __device__ __forceinline__ int swz(int r) { return RB == 128 ? ((r >> 1) & 7) : ((r >> 2) & 3); }

template <typename T> struct KCfg;
template <> struct KCfg<bf16_t> { static constexpr int EPC = 8; };  // elems / 16-B chunk
template <> struct KCfg<float> { static constexpr int EPC = 4; };

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_dst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_dst, 16, 0, 0);
}

// Same LDS-DMA hidden from hipcc's waitcnt bookkeeping: the compiler otherwise drains vmcnt(0) in front of the first
// ds_read of every K-slice (it cannot prove the DMA target stage and the stage being read are disjoint), which
// serialises load and MFMA inside a wave. M0 carries the wave-uniform LDS byte address and is written in the same
// statement that uses it (hipcc reserves M0 and does not preserve it across statements).
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_byte_addr) : "memory");
}

// SADDR form: uniform 64-bit base in SGPRs + per-lane unsigned 32-bit BYTE offset (half the address registers and VALU)
__device__ __forceinline__ void glds16_asm_s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}
// the same with the non-temporal cache policy: an operand every line of which is consumed once per XCD (the A stream of a short-K GEMM)
// should not displace the re-used W panels from the XCD's L2
__device__ __forceinline__ void glds16_asm_s_nt(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_byte_addr) : "memory");
}

// wait for this wave's outstanding LDS-DMA + LDS reads, then workgroup barrier (compiler memory barrier too)
// The vmcnt wait is inline asm (the DMA is invisible to hipcc); the lgkmcnt wait uses the BUILTIN so that hipcc's own
// waitcnt scoreboard knows the earlier ds_reads have completed -- otherwise it re-waits for them (and, in order, for
// every ds_read issued since) in front of the next MFMA, which defeats the prefetch across the barrier.
__device__ __forceinline__ void wait_all_and_barrier() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0)  (vmcnt = 63, expcnt = 7: no wait)
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}
template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  uint4 v;
  template <int RB> __device__ __forceinline__ void load(const char* tile, int r, int kk, int hi) {
    const int c = kk * 2 + hi;
    v = *reinterpret_cast<const uint4*>(tile + r * RB + ((c ^ swz<RB>(r)) << 4));
  }
  // W operand: bf16 rows like A (W8 = false), or FP8 e4m3 rows of RBW = 64 bytes per 64-element K-slice (W8 = true): the
  // lane's 8 consecutive k of step kk are 8 BYTES at chunk kk (16 B = both lane halves), half `hi`; they are widened
  // to bf16 in registers (v_cvt_pk_f32_fp8 + v_perm_b32: exact, every e4m3 value is a bf16 value) so that the matrix
  // instruction and everything behind it stay the bf16 path. 8 VALU ops per fragment beside 8 MFMAs per k-step.
  template <int RBW, bool W8> __device__ __forceinline__ void loadw(const char* tile, int r, int kk, int hi) {
    if constexpr (!W8) {
      load<RBW>(tile, r, kk, hi);
    } else {
      typedef __attribute__((ext_vector_type(2))) float f2_t;
      const uint2 u = *reinterpret_cast<const uint2*>(tile + r * RBW + ((kk ^ swz<RBW>(r)) << 4) + hi * 8);
      const f2_t a = __builtin_amdgcn_cvt_pk_f32_fp8((int)u.x, false), b = __builtin_amdgcn_cvt_pk_f32_fp8((int)u.x, true);
      const f2_t c = __builtin_amdgcn_cvt_pk_f32_fp8((int)u.y, false), d = __builtin_amdgcn_cvt_pk_f32_fp8((int)u.y, true);
      v.x = __builtin_amdgcn_perm(__float_as_uint(a.y), __float_as_uint(a.x), 0x07060302u);   // high halves: (bf16(a.x), bf16(a.y))
      v.y = __builtin_amdgcn_perm(__float_as_uint(b.y), __float_as_uint(b.x), 0x07060302u);
      v.z = __builtin_amdgcn_perm(__float_as_uint(c.y), __float_as_uint(c.x), 0x07060302u);
      v.w = __builtin_amdgcn_perm(__float_as_uint(d.y), __float_as_uint(d.x), 0x07060302u);
    }
  }
};
template <> struct Frag<float> {
  float4 v0, v1;
  template <int RBW, bool W8> __device__ __forceinline__ void loadw(const char* tile, int r, int kk, int hi) {
    static_assert(!W8, "fp8 weights exist for the bf16 path only");
    load<RBW>(tile, r, kk, hi);
  }
  template <int RB> __device__ __forceinline__ void load(const char* tile, int r, int kk, int hi) {
    static_assert(RB == 128, "fp32 operands use 128-byte K-slices");
    const int c = kk * 4 + hi * 2;
    const int sw = swz<RB>(r);
    v0 = *reinterpret_cast<const float4*>(tile + r * RB + ((c ^ sw) << 4));
    v1 = *reinterpret_cast<const float4*>(tile + r * RB + (((c + 1) ^ sw) << 4));
  }
};

__device__ __forceinline__ f32x16_t mma(const Frag<bf16_t>& w, const Frag<bf16_t>& a, f32x16_t acc) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, w.v), __builtin_bit_cast(bf16x8_t, a.v),
                                                 acc, 0, 0, 0);
}
__device__ __forceinline__ f32x16_t mma(const Frag<float>& w, const Frag<float>& a, f32x16_t acc) {
  // lane (i = lane&31, hi = lane>>5) holds k = hi*8 + e, e = 0..7 ; MFMA e contracts k in {e, 8+e}
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v0.x, a.v0.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v0.y, a.v0.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v0.z, a.v0.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v0.w, a.v0.w, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v1.x, a.v1.x, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v1.y, a.v1.y, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v1.z, a.v1.z, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w.v1.w, a.v1.w, acc, 0, 0, 0);
  return acc;
}

// FP8 (OCP e4m3) operands for v_mfma_scale_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate, K = 64 per instruction): a lane
// (i = lane & 31, hi = lane >> 5) holds row i's 32 consecutive k of half `hi` = two 16-B chunks of the 128-byte K-tile row
// (chunks ks*4 + hi*2 + {0,1} for the K = 64 step ks). Both operands are read with the same pattern, so the product does
// not depend on how the instruction orders k inside a lane's 32 bytes.
typedef __attribute__((ext_vector_type(8))) int i32x8_t;
typedef __attribute__((ext_vector_type(4))) int i32x4_t;
struct Frag8 {
  i32x8_t v;      // ONE 8-register tuple (the MFMA operand): built from two 16-byte LDS reads without copies
  __device__ __forceinline__ void load(const char* tile, int r, int ks, int hi) {
    const int c = ks * 4 + hi * 2;
    const int sw = swz<128>(r);
    const i32x4_t a = *reinterpret_cast<const i32x4_t*>(tile + r * 128 + ((c ^ sw) << 4));
    const i32x4_t b = *reinterpret_cast<const i32x4_t*>(tile + r * 128 + (((c + 1) ^ sw) << 4));
    v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
  }
};
__device__ __forceinline__ f32x16_t mma8(const Frag8& w, const Frag8& a, f32x16_t acc) {
  // cbsz = blgp = 0: both operands e4m3; scales E8M0 127 = 2^0 (per-tensor / per-channel scales are applied in the epilogue)
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w.v, a.v, acc, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
}

// fused RMSNorm row scale: rsqrt(mean of squares + eps) from the producer's per-64-column partial sums (fixed order)
__device__ __forceinline__ float rms_row_scale(const float* ssq, int parts, int row, float invk, float eps) {
  const float* q = ssq + (long long)row * parts;
  float t;
  if (parts == 24) {   // E = 768: six independent 16-byte loads (a scalar loop serialises the load latencies); pairs of
    // 32-column partials are summed first -- the same add every producer shape would do for a 64-column group
    const float4 a = load4(q), b = load4(q + 4), c = load4(q + 8), d = load4(q + 12), e = load4(q + 16), f = load4(q + 20);
    const float s0 = a.x + a.y, s1 = a.z + a.w, s2 = b.x + b.y, s3 = b.z + b.w, s4 = c.x + c.y, s5 = c.z + c.w;
    const float s6 = d.x + d.y, s7 = d.z + d.w, s8 = e.x + e.y, s9 = e.z + e.w, s10 = f.x + f.y, s11 = f.z + f.w;
    t = (((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))) + ((s8 + s9) + (s10 + s11));
  } else {
    t = 0.f;
    for (int j = 0; j < parts; ++j) t += q[j];
  }
  return __builtin_amdgcn_rsqf(t * invk + eps);
}

// fused LayerNorm (mean AND variance): mean and 1 / sqrt(var + eps) of row `row` from the producer's per-32-column partial sums and
// partial sums of squares (GemmArgs::sum_out / ssq_out), summed in rms_row_scale's fixed order; var = E[x^2] - mean^2 in fp32
__device__ __forceinline__ float ln_tree24(const float* q) {
  const float4 a = load4(q), b = load4(q + 4), c = load4(q + 8), d = load4(q + 12), e = load4(q + 16), f = load4(q + 20);
  const float s0 = a.x + a.y, s1 = a.z + a.w, s2 = b.x + b.y, s3 = b.z + b.w, s4 = c.x + c.y, s5 = c.z + c.w;
  const float s6 = d.x + d.y, s7 = d.z + d.w, s8 = e.x + e.y, s9 = e.z + e.w, s10 = f.x + f.y, s11 = f.z + f.w;
  return (((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7))) + ((s8 + s9) + (s10 + s11));
}
__device__ __forceinline__ void ln_row_stats(const float* sum, const float* ssq, int parts, int row, float invk, float eps, float& mean, float& rstd) {
  const float* qs = sum + (long long)row * parts;
  const float* qq = ssq + (long long)row * parts;
  float ts, tq;
  if (parts == 24) { ts = ln_tree24(qs); tq = ln_tree24(qq); }
  else {
    ts = 0.f; tq = 0.f;
    for (int j = 0; j < parts; ++j) { ts += qs[j]; tq += qq[j]; }
  }
  mean = ts * invk;
  const float var = fmaxf(__builtin_fmaf(-mean, mean, tq * invk), 0.0f);
  rstd = __builtin_amdgcn_rsqf(var + eps);
}

// sum of squares of 4 consecutive output columns; one fixed fma chain so that every kernel rounds identically (the
// fused-RMSNorm statistics must not depend on which tile shape produced the row: batch-composition invariance)
__device__ __forceinline__ float sumsq4(const float4& v) {
  return __builtin_fmaf(v.w, v.w, __builtin_fmaf(v.z, v.z, __builtin_fmaf(v.y, v.y, v.x * v.x)));
}

// sum of squares of 8 consecutive STORED bf16 values (the residual stream carried in bf16: its RMS statistics are those of
// the rounded values the consumer multiplies with); fixed order, shared by every kernel that produces such partials
__device__ __forceinline__ float sumsq8_bf16(const uint4& o) {
  const float a0 = __uint_as_float(o.x << 16), a1 = __uint_as_float(o.x & 0xffff0000u);
  const float a2 = __uint_as_float(o.y << 16), a3 = __uint_as_float(o.y & 0xffff0000u);
  const float a4 = __uint_as_float(o.z << 16), a5 = __uint_as_float(o.z & 0xffff0000u);
  const float a6 = __uint_as_float(o.w << 16), a7 = __uint_as_float(o.w & 0xffff0000u);
  return sumsq4(make_float4(a0, a1, a2, a3)) + sumsq4(make_float4(a4, a5, a6, a7));
}

struct GemmDev {
  const void* A; const void* W;
  int M, N, K, lda, ldw;
  long long bsA, bsW, bsBias, bsMul, bsRes, bs32, bsT;
  const float* bias; int act;
  const void* mul; int ldmul;
  const float* res; int ldres;
  const void* resT; int ldresT;
  float* out32; int ld32;
  void* outT; int ldT;
  int rb, s_hi, s_lo, ro;
  void* outT_lo; int ldT_lo, split_n;   // GemmArgs::split_n: columns < split_n go to outT_lo[m][n] (no row remap), the others to outT[orow][n - split_n]
  int hm_D, hm_L;   // head-major output of the 256x256 bf16-output epilogue (GemmArgs::hm_D / hm_L; 0 = row-major)
  float* ssq_out; const float* rs_ssq; int rs_parts; float rs_invk, rs_eps;
  float* sum_out; const float* rs_sum; const float* rs_c;   // fused LayerNorm (GemmArgs::sum_out / rs_sum / rs_c)
  const float* wscale;   // fp8 weights: per-output-channel dequantisation scale [N], applied to the accumulator column
  float ascale;          // fp8 ACTIVATIONS (gemm_pp_kernel<.., F8>): the A operand's per-tensor dequantisation scale (x wscale[n])
  void* out8; int ld8; float out8_inv;   // optional fp8 e4m3 copy of the bf16 output: e4m3(value * out8_inv), saturating at 448
  int mtiles, ntiles;
  int vtotal;   // persistent kernel: number of virtual tile ids = ceil8(mtiles) * ntiles
  int flat;     // gemm_pp_kernel: 1 = virtual tile v is tile (v % mtiles, v / mtiles), no XCD raster (small grids, see launch_pp)
  int raster;   // 0: XCD walks the n-tiles of one A panel; 1: XCD keeps a group of `ngroup` n-tiles (W panels) resident
  int ngroup;   //    and walks its A panels through it; 2: plain row-major (no XCD awareness)
  int epi_lds;  // 1 = LDS-transposed (row-contiguous) vector epilogue, 0 = direct per-lane epilogue
  int wide8;    // bf16-only output with 8-column alignment: 16-byte stores in the LDS epilogue
  int res_nch;  // gemm_resident_kernel: chunk buffers in LDS
  int sk_cols;  // gemm_skinny_kernel: output columns per workgroup (32 / 16 / 8)
  const int* grp_col;   // gemm_resident_kernel, grouped form (GemmArgs::grp_col): column starts of the groups, or nullptr
  const void* A2; const void* W2; int lda2, ldw2;   // gemm_resident_kernel<RTile<.., DUAL>>: the gate product's operands
  long long* dbg;   // optional: 8 debug slots per workgroup (shader-clock stamps of the 4 phases, real time, placement)
};

// ACT >= 0: compile-time activation; ACT == -1: runtime p.act. VEC: 4-wide vector epilogue. ASMLDS: inline-asm LDS-DMA.
template <typename T, typename TL, int ACT, bool VEC, bool ASMLDS, bool W8 = false>
__global__ __launch_bounds__(TL::THREADS, TL::MINW) void gemm_kernel(const GemmDev p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = TL::RB, NS = TL::NS, CPR = TL::CPR;
  // W operand geometry: bf16 / fp32 rows like A, or fp8 rows of half the bytes (half the LDS-DMA pieces per K-slice)
  constexpr int RBW = W8 ? RB / 2 : RB, CPRW = RBW / 16, PWN = W8 ? TL::PW / 2 : TL::PW;
  constexpr int ESW = W8 ? 1 : (int)sizeof(T);
  static_assert(!W8 || (sizeof(T) == 2 && ASMLDS && VEC && TL::PW % 2 == 0), "fp8 weights: bf16 path, asm LDS-DMA, vector epilogue");
  constexpr int BK = RB / (int)sizeof(T);
  constexpr int EPC = KCfg<T>::EPC;
  constexpr int KSTEPS = BK / 16;
  constexpr int MI = TL::MI, NI = TL::NI, NW = TL::NW;
  static_assert(KSTEPS >= 2 && (KSTEPS % 2) == 0, "fragment double buffer assumes an even number of k-steps");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5;
  const int l31 = lane & 31;

  // XCD-aware tile mapping (block b runs on XCD b % 8): one XCD walks all n-tiles of its A panel back to back
  const int bid = blockIdx.x;
  int tm, tn;
  if (p.raster == 2) {
    tn = bid % p.ntiles;
    tm = bid / p.ntiles;
  } else {
    const int xcd = bid & 7;
    const int idx = bid >> 3;
    if (p.raster == 0) {
      tn = idx % p.ntiles;
      tm = (idx / p.ntiles) * 8 + xcd;
    } else {
      const int mtx = (p.mtiles + 7) >> 3;                 // A panels per XCD
      const int full = p.ntiles / p.ngroup;                // complete n-groups
      const int per_group = mtx * p.ngroup;
      int g = idx / per_group, rem, ng;
      if (g < full) { rem = idx - g * per_group; ng = p.ngroup; }
      else { g = full; rem = idx - full * per_group; ng = p.ntiles - full * p.ngroup; }
      tn = g * p.ngroup + rem % ng;
      tm = (rem / ng) * 8 + xcd;
    }
  }
  if (tm >= p.mtiles) return;
  const int z = blockIdx.y;
  const int m0 = tm * TL::BM, n0 = tn * TL::BN;

  const T* A = reinterpret_cast<const T*>(p.A) + (long long)z * p.bsA;
  const char* W = reinterpret_cast<const char*>(p.W) + (long long)z * p.bsW * ESW;
  // 8 slots per workgroup: 0-3 shader-clock stamps, 4/5 constant-rate (100 MHz) real-time at start/end, 6 HW_ID, 7 XCC_ID
  auto stamp = [&](int slot) {
    if (p.dbg && tid == 0) {
      long long* d = p.dbg + (long long)blockIdx.x * 8;
      d[slot] = (long long)__builtin_readcyclecounter();
      if (slot == 0) {
        d[4] = (long long)__builtin_amdgcn_s_memrealtime();
        d[6] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
        d[7] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);   // HW_REG_XCC_ID
      }
      if (slot == 3) d[5] = (long long)__builtin_amdgcn_s_memrealtime();
    }
  };
  stamp(0);

  // per-lane source pointers of this wave's LDS-DMA pieces of one K-slice (piece = 1 KiB = 1024/RB tile rows)
  const T* srcA[TL::PA];
  const char* srcW[PWN];
#pragma unroll
  for (int i = 0; i < TL::PA; ++i) {
    const int s = (i * NW + w) * 64 + lane;
    const int r = s / CPR, pp = s % CPR;
    const int c = pp ^ swz<RB>(r);
    int ra = m0 + r; ra = ra < p.M ? ra : p.M - 1;
    srcA[i] = A + (long long)ra * p.lda + c * EPC;
  }
#pragma unroll
  for (int i = 0; i < PWN; ++i) {
    const int s = (i * NW + w) * 64 + lane;
    const int r = s / CPRW, pp = s % CPRW;
    const int c = pp ^ swz<RBW>(r);
    int rw = n0 + r; rw = rw < p.N ? rw : p.N - 1;
    srcW[i] = W + ((long long)rw * p.ldw) * ESW + c * 16;
  }

  f32x16_t acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

  const int wm = w / TL::WN, wn = w % TL::WN;
  const int arow = wm * (MI * 32) + l31;   // + mi*32
  const int wrow = wn * (NI * 32) + l31;   // + ni*32
  const int nk = p.K / BK;
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;

  // one 1-KiB LDS-DMA piece j of K-slice kt into `stage` (j < PA: A tile rows, else W tile rows)
  constexpr int NP = TL::PA + PWN;
  auto issue_piece = [&](int stage, int kt, int j) {
    const int i = j < TL::PA ? j : j - TL::PA;
    const int off = stage * TL::STAGE_BYTES + (j < TL::PA ? 0 : TL::A_BYTES) + (i * NW + w) * 1024;
    const void* src = j < TL::PA ? (const void*)(srcA[i] + kt * BK) : (const void*)(srcW[i] + kt * RBW);
    if constexpr (ASMLDS) glds16_asm(src, smem_base + off);
    else glds16(src, smem + off);
  };

  // Software pipeline (NS LDS stages in a ring, fragments double-buffered in registers):
  //   slice kt, steps kk = 0 .. KSTEPS-2 : prefetch fragments of step kk+1 (same stage)      | MFMAs of step kk
  //   last step                          : counted vmcnt + lgkmcnt(0), s_barrier -> every wave has finished READING
  //                                        stage kt % NS and the DMA of slice kt+1 has landed; then prefetch the
  //                                        fragments of slice kt+1 / step 0 and issue the DMA of slice kt+NS into the
  //                                        stage just freed, one piece between two MFMAs   | MFMAs of the last step
  // so the barrier, the LDS refill latency after it and the VMEM issue are covered by the last step's MFMAs (whose
  // operands are already in registers). VMEM loads retire in order, so `vmcnt(n * NP)` = "all but the n youngest slices".
  auto wait_slices_and_barrier = [&](int younger) {   // `younger` slices of DMA may stay in flight (wave-uniform)
    if constexpr (ASMLDS) {
      if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NP) : "memory");
      else if (younger == 2 || NS <= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(2 * NP) : "memory");
      else if (younger == 3 || NS <= 4) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NS > 3 ? 3 * NP : 0) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NS > 4 ? 4 * NP : 0) : "memory");
      __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0) via the builtin: keeps hipcc's scoreboard exact
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    } else {
      wait_all_and_barrier();
    }
  };
  static_assert(NS >= 2 && NS <= 5 && (NS - 1) * NP <= 63, "ring depth (vmcnt is a 6-bit counter)");
  const int npro = nk < NS ? nk : NS;
  for (int t = 0; t < npro; ++t) {
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(t, t, j);
  }
  wait_slices_and_barrier(npro - 1);   // start as soon as slice 0 has landed
  stamp(1);
  Frag<T> fa[2][MI], fw[2][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) fa[0][mi].template load<RB>(smem, arow + mi * 32, 0, hi);
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) fw[0][ni].template loadw<RBW, W8>(smem + TL::A_BYTES, wrow + ni * 32, 0, hi);
  // The two waves that share a SIMD (w and w + NW/2 of an 8-wave workgroup) run the SAME instruction stream in lock
  // step after every barrier; if both fetch fragments at the same moment the matrix pipe idles, then both compete
  // for it. The second half of the waves therefore issues its fragment prefetch in the MIDDLE of each step's MFMAs.
  auto main_loop = [&](auto late_tag) {
    constexpr bool LATE = decltype(late_tag)::value;
    constexpr int PPM = (NP + MI * NI - 1) / (MI * NI);
    // MFMAs of k-step buffer cb for mi in [mi0, mi1); when `dma`, the pieces of `slice` are issued into `stage` one
    // (PPM) per MFMA, in MFMA order
    auto mma_block = [&](int mi0, int mi1, int cb, bool dma, int stage, int slice) {
#pragma unroll
      for (int mi = mi0; mi < mi1; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = mma(fw[cb][ni], fa[cb][mi], acc[mi][ni]);
          if (dma) {
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = (mi * NI + ni) * PPM; j < (mi * NI + ni + 1) * PPM && j < NP; ++j) issue_piece(stage, slice, j);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
    };
    int cur = 0;
    for (int kt = 0; kt < nk; ++kt) {
      const int nxt = cur + 1 == NS ? 0 : cur + 1;
      const char* sA = smem + cur * TL::STAGE_BYTES;
      const char* sW = sA + TL::A_BYTES;
      const char* nA = smem + nxt * TL::STAGE_BYTES;
      const char* nW = nA + TL::A_BYTES;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const int cb = kk & 1, nb = cb ^ 1;
        if (kk + 1 < KSTEPS) {   // prefetch the next step's fragments while this step's MFMAs run
          if constexpr (LATE) {
            __builtin_amdgcn_sched_barrier(0);
            mma_block(0, MI / 2, cb, false, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
          }
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) fa[nb][mi].template load<RB>(sA, arow + mi * 32, kk + 1, hi);
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) fw[nb][ni].template loadw<RBW, W8>(sW, wrow + ni * 32, kk + 1, hi);
          // pin the order: [ds_reads] then [MFMAs]; without this hipcc re-serialises read -> wait -> 2 MFMAs
          __builtin_amdgcn_sched_barrier(0);
          mma_block(LATE ? MI / 2 : 0, MI, cb, false, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        } else {
          // slices issued so far: up to min(nk-1, kt+NS-1); younger than kt+1 may stay in flight
          int last = kt + NS - 1;
          last = last < nk - 1 ? last : nk - 1;
          wait_slices_and_barrier(last - (kt + 1));
          if (kt + 1 < nk) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[nb][mi].template load<RB>(nA, arow + mi * 32, 0, hi);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fw[nb][ni].template loadw<RBW, W8>(nW, wrow + ni * 32, 0, hi);
          }
          __builtin_amdgcn_sched_barrier(0);
          // the DMA pieces of slice kt+NS go into the stage just freed, ONE BY ONE BETWEEN the MFMAs
          const bool more = kt + NS < nk;
          mma_block(0, MI, cb, more, cur, kt + NS);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      cur = nxt;
    }
  };
  if (kDephase && NW == 8 && w >= NW / 2) main_loop(std::true_type{});
  else main_loop(std::false_type{});
  stamp(2);
  // (the barrier inside the last slice already guarantees that no wave reads the stage buffers any more, so the LDS
  // epilogue below may reuse them)

  // ------------------------------------------------------------------ epilogue
  // acc[mi][ni][4q+e] = C[m = m0 + wm*MI*32 + mi*32 + l31][n = n0 + wn*NI*32 + ni*32 + 8q + 4hi + e]
  const float* bias = p.bias ? p.bias + (long long)z * p.bsBias : nullptr;
  const T* mul = p.mul ? reinterpret_cast<const T*>(p.mul) + (long long)z * p.bsMul : nullptr;
  const float* res = p.res ? p.res + (long long)z * p.bsRes : nullptr;
  // residual in the operand type (batch 1 only): residual GEMMs have no activation, so the activation instantiations
  // (GELU's erf is register-hungry: the extra live values spilled there) do not carry this code
  constexpr bool kStreamEpi = ACT == ACT_NONE || ACT == -1;
  const T* resT = kStreamEpi ? reinterpret_cast<const T*>(p.resT) : nullptr;
  float* out32 = p.out32 ? p.out32 + (long long)z * p.bs32 : nullptr;
  T* outT = p.outT ? reinterpret_cast<T*>(p.outT) + (long long)z * p.bsT : nullptr;
  const int act = ACT >= 0 ? ACT : p.act;

  if constexpr (VEC) {
    if (p.epi_lds) {
      // LDS-transposed epilogue. The MFMA layout gives a lane 4 consecutive columns of ONE row, i.e. a store
      // instruction touches 32 different rows with 16-32 B each (32 partial cache lines per instruction: the write
      // path, not HBM, then bounds the epilogue, ~7 us per 256x256 tile). Each wave therefore bounces its
      // 32-row x 64-column slabs through a private 16 KiB LDS region (free after the main loop) and finishes the
      // epilogue row-contiguously: 16 lanes cover one 64-column row segment, so every global access (gate `mul`,
      // residual, fp32 / bf16 stores) is a full 128/256-byte line.
      constexpr int WCOLS = NI * 32;                        // columns of the wave tile
      constexpr int LDE = WCOLS + 4;                        // padded fp32 row stride: conflict-free ds_write_b128
      constexpr int LPR = WCOLS / 4;                        // lanes per row in the read-back phase
      constexpr int RPI = 64 / LPR;                         // rows per wave-instruction
      static_assert(TL::SMEM_BYTES / NW >= 32 * LDE * 4, "per-wave LDS slab for the epilogue");
      float* stage = reinterpret_cast<float*>(smem + w * (TL::SMEM_BYTES / NW));
      if constexpr (ACT == ACT_GELU && sizeof(T) == 2 && NI == 2 && !W8) {
        if (p.epi_lds == 2) {
          // GEGLU PAIR (GemmArgs::pair32): W's rows alternate in blocks of 32 between the GELU'd layer and its plain multiplier, so a lane's
          // acc[mi][0][r] and acc[mi][1][r] are the two factors of ONE output element: out[m][j] = bf16(gelu(a + bias) * bf16(g + bias)), the
          // very operations (and roundings) of the multiplier GEMM's bf16 store followed by the GELU GEMM's `mul` epilogue, without the
          // second launch, the multiplier's store and the per-iteration `mul` loads. A 32-row x 32-column product slab goes through LDS.
          constexpr int LDP = 32 + 4;
          const int nout = (n0 >> 1) + wn * 32 + (lane & 3) * 8;
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            float mu = 0.f, rsd = 1.f;   // fused LayerNorm of the GELU'd factor's input (GemmArgs::rs_sum): a = rstd * (A.W1' - mean * c) + d
            if (p.rs_sum) {
              int mr = m0 + wm * (MI * 32) + mi * 32 + l31; mr = mr < p.M ? mr : p.M - 1;
              ln_row_stats(p.rs_sum, p.rs_ssq, p.rs_parts, mr, p.rs_invk, p.rs_eps, mu, rsd);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int nl = 8 * q + 4 * hi;
              const int nb = n0 + wn * WCOLS + nl;
              float4 a = make_float4(acc[mi][0][4 * q], acc[mi][0][4 * q + 1], acc[mi][0][4 * q + 2], acc[mi][0][4 * q + 3]);
              float4 g = make_float4(acc[mi][1][4 * q], acc[mi][1][4 * q + 1], acc[mi][1][4 * q + 2], acc[mi][1][4 * q + 3]);
              if (p.rs_sum) {
                const float4 c4 = load4(p.rs_c + nb);
                a.x = __builtin_fmaf(-mu, c4.x, a.x) * rsd; a.y = __builtin_fmaf(-mu, c4.y, a.y) * rsd;
                a.z = __builtin_fmaf(-mu, c4.z, a.z) * rsd; a.w = __builtin_fmaf(-mu, c4.w, a.w) * rsd;
              }
              if (bias) {
                const float4 ba = load4(bias + nb), bg = load4(bias + nb + 32);
                a.x += ba.x; a.y += ba.y; a.z += ba.z; a.w += ba.w; g.x += bg.x; g.y += bg.y; g.z += bg.z; g.w += bg.w;
              }
              a.x = apply_act_t<T>(a.x, ACT_GELU); a.y = apply_act_t<T>(a.y, ACT_GELU); a.z = apply_act_t<T>(a.z, ACT_GELU); a.w = apply_act_t<T>(a.w, ACT_GELU);
              const uint32_t g01 = pack2_bf16(g.x, g.y), g23 = pack2_bf16(g.z, g.w);   // the multiplier as the separate launch stores it
              a.x *= __uint_as_float(g01 << 16); a.y *= __uint_as_float(g01 & 0xffff0000u);
              a.z *= __uint_as_float(g23 << 16); a.w *= __uint_as_float(g23 & 0xffff0000u);
              *reinterpret_cast<float4*>(stage + l31 * LDP + nl) = a;
            }
#pragma unroll
            for (int it = 0; it < 2; ++it) {
              const int r = it * 16 + (lane >> 2);
              const float4 v0 = *reinterpret_cast<const float4*>(stage + r * LDP + (lane & 3) * 8);
              const float4 v1 = *reinterpret_cast<const float4*>(stage + r * LDP + (lane & 3) * 8 + 4);
              const int m = m0 + wm * (MI * 32) + mi * 32 + r;
              if (m < p.M) {
                uint4 o;
                o.x = pack2_bf16(v0.x, v0.y); o.y = pack2_bf16(v0.z, v0.w); o.z = pack2_bf16(v1.x, v1.y); o.w = pack2_bf16(v1.z, v1.w);
                *reinterpret_cast<uint4*>(outT + (long long)m * p.ldT + nout) = o;
              }
            }
          }
          stamp(3);
          return;
        }
      }
      const int rr = lane / LPR, cc = (lane % LPR) * 4;
      const int n = n0 + wn * WCOLS + cc;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        float rsc = 1.0f;   // fused RMSNorm: scale of this lane's accumulator row
        if (p.rs_ssq) {
          int mr = m0 + wm * (MI * 32) + mi * 32 + l31; mr = mr < p.M ? mr : p.M - 1;
          rsc = rms_row_scale(p.rs_ssq, p.rs_parts, mr, p.rs_invk, p.rs_eps);
        }
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int nl = ni * 32 + 8 * q + 4 * hi;
            float4 v = make_float4(acc[mi][ni][4 * q] * rsc, acc[mi][ni][4 * q + 1] * rsc, acc[mi][ni][4 * q + 2] * rsc, acc[mi][ni][4 * q + 3] * rsc);
            const int nb = n0 + wn * WCOLS + nl;
            if (p.wscale && nb < p.N) { const float4 sc = load4(p.wscale + nb); v.x *= sc.x; v.y *= sc.y; v.z *= sc.z; v.w *= sc.w; }
            if (bias && nb < p.N) { const float4 b = load4(bias + nb); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
            if (act != ACT_NONE) { v.x = apply_act_t<T>(v.x, act); v.y = apply_act_t<T>(v.y, act); v.z = apply_act_t<T>(v.z, act); v.w = apply_act_t<T>(v.w, act); }
            *reinterpret_cast<float4*>(stage + l31 * LDE + nl) = v;
          }
        if constexpr (sizeof(T) == 2) {
          if (p.wide8) {   // bf16-only output: 8 columns per lane -> 16-byte stores, half the store instructions
            constexpr int LPR8 = WCOLS / 8, RPI8 = 64 / LPR8;
            const int rr8 = lane / LPR8, cc8 = (lane % LPR8) * 8;
            const int n8 = n0 + wn * WCOLS + cc8;
#pragma unroll
            for (int it = 0; it < 32 / RPI8; ++it) {
              const int r = it * RPI8 + rr8;
              float4 v0 = *reinterpret_cast<const float4*>(stage + r * LDE + cc8);
              float4 v1 = *reinterpret_cast<const float4*>(stage + r * LDE + cc8 + 4);
              const int m = m0 + wm * (MI * 32) + mi * 32 + r;
              float sq8 = 0.f;
              if (m < p.M && n8 < p.N) {
                long long orow = m;
                if (p.rb > 0) orow = (long long)(m / p.rb) * p.s_hi + (long long)(m % p.rb) * p.s_lo + p.ro;
                if (mul) {
                  const float4 g0 = load4(mul + (long long)m * p.ldmul + n8), g1 = load4(mul + (long long)m * p.ldmul + n8 + 4);
                  v0.x *= g0.x; v0.y *= g0.y; v0.z *= g0.z; v0.w *= g0.w; v1.x *= g1.x; v1.y *= g1.y; v1.z *= g1.z; v1.w *= g1.w;
                }
                if (res) {
                  const float4 r0 = load4(res + (long long)m * p.ldres + n8), r1 = load4(res + (long long)m * p.ldres + n8 + 4);
                  v0.x += r0.x; v0.y += r0.y; v0.z += r0.z; v0.w += r0.w; v1.x += r1.x; v1.y += r1.y; v1.z += r1.z; v1.w += r1.w;
                }
                if (resT) {
                  const float4 r0 = load4(resT + (long long)m * p.ldresT + n8), r1 = load4(resT + (long long)m * p.ldresT + n8 + 4);
                  v0.x += r0.x; v0.y += r0.y; v0.z += r0.z; v0.w += r0.w; v1.x += r1.x; v1.y += r1.y; v1.z += r1.z; v1.w += r1.w;
                }
                uint4 o;
                o.x = pack2_bf16(v0.x, v0.y); o.y = pack2_bf16(v0.z, v0.w); o.z = pack2_bf16(v1.x, v1.y); o.w = pack2_bf16(v1.z, v1.w);
                if (p.split_n) {
                  if (n8 < p.split_n) *reinterpret_cast<uint4*>(reinterpret_cast<T*>(p.outT_lo) + (long long)m * p.ldT_lo + n8) = o;
                  else *reinterpret_cast<uint4*>(outT + orow * p.ldT + (n8 - p.split_n)) = o;
                } else
                *reinterpret_cast<uint4*>(outT + orow * p.ldT + n8) = o;
                if (kStreamEpi) sq8 = sumsq8_bf16(o);
              }
              if (kStreamEpi && p.ssq_out) {   // statistics of the STORED (rounded) stream values; 4 lanes hold the 32 columns of a partial
                sq8 += __shfl_xor(sq8, 1, 64); sq8 += __shfl_xor(sq8, 2, 64);
                if ((lane & 3) == 0 && m < p.M && n8 < p.N) p.ssq_out[(long long)m * (p.N >> 5) + (n8 >> 5)] = sq8;
              }
            }
            continue;
          }
        }
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
          const int r = it * RPI + rr;
          float4 v = *reinterpret_cast<const float4*>(stage + r * LDE + cc);
          const int m = m0 + wm * (MI * 32) + mi * 32 + r;
          float sq = 0.f, sm = 0.f;
          if (m < p.M && n < p.N) {
            long long orow = m;
            if (p.rb > 0) orow = (long long)(m / p.rb) * p.s_hi + (long long)(m % p.rb) * p.s_lo + p.ro;
            if (mul) { const float4 g = load4(mul + (long long)m * p.ldmul + n); v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w; }
            if (res) { const float4 r4 = load4(res + (long long)m * p.ldres + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
            if (resT) { const float4 r4 = load4(resT + (long long)m * p.ldresT + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
            if (out32) store4(out32 + orow * p.ld32 + n, v);
            if (outT) {
              if (p.split_n) {
                if (n < p.split_n) store4(reinterpret_cast<T*>(p.outT_lo) + (long long)m * p.ldT_lo + n, v);
                else store4(outT + orow * p.ldT + (n - p.split_n), v);
              } else store4(outT + orow * p.ldT + n, v);
            }
            sq = sumsq4(v);
            sm = (v.x + v.y) + (v.z + v.w);
          }
          if (p.ssq_out) {   // wave-uniform; 8 consecutive lanes hold 32 columns of one row: butterfly, one partial per 32 columns
            sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
            if ((lane & 7) == 0 && m < p.M && n < p.N) p.ssq_out[(long long)m * (p.N >> 5) + (n >> 5)] = sq;
            if (p.sum_out) {   // fused LayerNorm: the row's partial SUMS as well (same columns, same tree)
              sm += __shfl_xor(sm, 1, 64); sm += __shfl_xor(sm, 2, 64); sm += __shfl_xor(sm, 4, 64);
              if ((lane & 7) == 0 && m < p.M && n < p.N) p.sum_out[(long long)m * (p.N >> 5) + (n >> 5)] = sm;
            }
          }
        }
      }
      stamp(3);
      return;
    }
  }

#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + wm * (MI * 32) + mi * 32 + l31;
    if (m >= p.M) continue;
    long long orow = m;
    if (p.rb > 0) orow = (long long)(m / p.rb) * p.s_hi + (long long)(m % p.rb) * p.s_lo + p.ro;
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = n0 + wn * (NI * 32) + ni * 32 + 8 * q + 4 * hi;
        if (n >= p.N) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = acc[mi][ni][4 * q + e];
        if (p.rs_ssq) {
          const float rsc = rms_row_scale(p.rs_ssq, p.rs_parts, m, p.rs_invk, p.rs_eps);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= rsc;
        }
        if constexpr (VEC) {   // N % 4 == 0 => n + 3 < N
          if (p.wscale) { const float4 sc = load4(p.wscale + n); v[0] *= sc.x; v[1] *= sc.y; v[2] *= sc.z; v[3] *= sc.w; }
          if (bias) { const float4 b = load4(bias + n); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
          if (act != ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act_t<T>(v[e], act);
          }
          if (mul) { const float4 g = load4(mul + (long long)m * p.ldmul + n); v[0] *= g.x; v[1] *= g.y; v[2] *= g.z; v[3] *= g.w; }
          if (res) { const float4 r4 = load4(res + (long long)m * p.ldres + n); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
          if (resT) { const float4 r4 = load4(resT + (long long)m * p.ldresT + n); v[0] += r4.x; v[1] += r4.y; v[2] += r4.z; v[3] += r4.w; }
          const float4 o = make_float4(v[0], v[1], v[2], v[3]);
          if (out32) store4(out32 + orow * p.ld32 + n, o);
          if (outT) {
            if (p.split_n) {
              if (n < p.split_n) store4(reinterpret_cast<T*>(p.outT_lo) + (long long)m * p.ldT_lo + n, o);
              else store4(outT + orow * p.ldT + (n - p.split_n), o);
            } else store4(outT + orow * p.ldT + n, o);
          }
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int ne = n + e;
            if (ne >= p.N) break;
            float x = v[e];
            if (bias) x += bias[ne];
            x = apply_act_t<T>(x, act);
            if (mul) x *= Elem<T>::load(mul + (long long)m * p.ldmul + ne);
            if (res) x += res[(long long)m * p.ldres + ne];
            if (resT) x += Elem<T>::load(resT + (long long)m * p.ldresT + ne);
            if (out32) out32[orow * p.ld32 + ne] = x;
            if (outT) Elem<T>::store(outT + orow * p.ldT + ne, x);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------ 256x256 tile epilogue
// Shared by the persistent kernels (gemm_persistent_kernel, gemm_pp_kernel): the wave's 128x64 accumulator block goes
// through a wave-private 4 KiB LDS slab (32x32 fp32, XOR-swizzled) and is finished row-contiguously.
// HAS_BIAS / HAS_RS (fused-RMSNorm row scale) are compile-time here: as runtime flags hipcc turned the conditional adds into
// v_pk_add + one v_cndmask per value and kept the x rsc multiply -- 256 of the 321 VALU instructions per wave of a
// bias-less, scale-less epilogue (every T5 GEMM) did nothing, and the epilogue is VALU-issue-bound (2 waves per SIMD).
// NI / MI: 32-column / 32-row blocks per wave (2 / 4: the 256x256 kernels' 128x64 block; 6 / 4 and 6 / 2: gemm_q4_kernel's 128x192 and 64x192)
template <int ACT, int EPI, bool W8, bool HAS_BIAS, bool HAS_RS, bool OUT8, int NI = 2, int MI = 4>
__device__ __forceinline__ void tile_epilogue_256_impl(const GemmDev& p, const f32x16_t (&acc)[MI][NI], const float (&rscv)[MI], char* slab,
                                                       int lane, int m0, int n0, int wm, int wn) {
  using T = bf16_t;
  const int act = ACT >= 0 ? ACT : p.act;
  // ---------------------------------------------------------------- epilogue (32x32 fp32 slabs, private LDS region)
  // acc[mi][ni][4q+e] = C[m0 + wm*128 + mi*32 + l31][n0 + wn*64 + ni*32 + 8q + 4hi + e]
  // slab row r keeps its eight 16-B chunks at slot c ^ fsw(r), fsw = r&7 with its two low bits swapped
  auto fsw = [](int r) { return ((r >> 1) & 1) | ((r & 1) << 1) | (r & 4); };
  float* stg = reinterpret_cast<float*>(slab);
  constexpr bool WIDE8 = EPI == 1 || EPI == 2 || EPI == 4 || EPI == 5;      // bf16-only output, 16-byte stores
  // the fp8 copy of the output (GemmDev::out8) as a compile-time capability: only the fp8-operand kernels' consumers ask for it, and as run-time
  // branches per store (pointer tests that also fence the scheduler) it cost the bf16 launches 1.3-1.6 % (profiles/r04_gemm_epilogue_branches.txt)
  constexpr bool O8 = OUT8 && (EPI == 1 || EPI == 4 || EPI == 5);
  const T* mul = EPI == 2 ? reinterpret_cast<const T*>(p.mul) : nullptr;
  const float* res = EPI == 3 ? p.res : nullptr;
  float* out32 = EPI == 3 ? p.out32 : nullptr;
  T* outT = reinterpret_cast<T*>(p.outT);
  float* ssq_out = (EPI == 3 || EPI == 4) ? p.ssq_out : nullptr;
  // the epilogue's per-lane address arithmetic is tile-invariant: hipcc would hoist it out of the tile loop and keep
  // (spill) it across the main loop. An opaque copy of the lane id pins it here.
  int elane = lane;
  asm volatile("" : "+v"(elane));
  const int el31 = elane & 31, ehi = elane >> 5;
  const int fw_ = fsw(el31);
  // Two phases per 32x32 slab: (1) the accumulators (x fused-RMSNorm row scale) go to the wave's LDS slab in MFMA
  // layout; (2) they are read back ROW-CONTIGUOUSLY -- a lane owns 8 (bf16-only output) or 4 consecutive columns of a
  // row, the SAME columns for every row and every mi -- and finished there: x weight scale (fp8w), + bias, activation,
  // x gate, + residual, stores. Everything that depends on the column only (bias, fp8 scales) is therefore loaded once
  // per tile. The per-row operands (gate / residual) are prefetched ONE SLAB AHEAD with inline-asm loads and counted
  // `vmcnt` waits: left to hipcc, every row group waited `vmcnt(0)`, i.e. for the previous group's STORES as well (32
  // serialised store round trips per tile, ~25 us of a 50 us tile at K = 768). VMEM operations of a wave retire in issue
  // order; the launcher only sends M % 256 == 0, N % 256 == 0 here, so there are no bounds checks.
  constexpr int CPL = WIDE8 ? 8 : 4;                            // columns per lane in the read-back layout
  constexpr int LPR = 32 / CPL;                                 // lanes per slab row
  constexpr int RPI = 64 / LPR;                                 // rows per wave-instruction (16 / 8)
  constexpr int NIT = 32 / RPI;                                 // row groups per slab (2 / 4)
  const int ccol = (elane % LPR) * CPL;                         // first column of this lane inside a slab
  const int crow = elane / LPR;                                 // row of this lane inside a row group
  const int ncol0 = n0 + wn * (NI * 32) + ccol;                 // + ni * 32
  const int mrow0 = m0 + wm * (MI * 32) + crow;                 // + mi * 32 + it * RPI
  float4 bcol[NI][CPL / 4], scol[NI][CPL / 4];
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int j = 0; j < CPL / 4; ++j) {
      bcol[ni][j] = make_float4(0.f, 0.f, 0.f, 0.f);
      scol[ni][j] = make_float4(1.f, 1.f, 1.f, 1.f);
      if constexpr (HAS_BIAS) bcol[ni][j] = load4(p.bias + ncol0 + ni * 32 + 4 * j);
      if (W8) {
        scol[ni][j] = load4(p.wscale + ncol0 + ni * 32 + 4 * j);
        scol[ni][j].x *= p.ascale; scol[ni][j].y *= p.ascale; scol[ni][j].z *= p.ascale; scol[ni][j].w *= p.ascale;   // 1 unless the A operand is fp8
      }
    }
  // the column constants are needed (waited for) HERE, before the per-row prefetch starts: hipcc would otherwise wait for
  // them at their first use with `vmcnt(0)`, i.e. for the prefetched loads issued in between as well
#pragma unroll
  for (int ni = 0; ni < NI; ++ni)
#pragma unroll
    for (int j = 0; j < CPL / 4; ++j) {
      if constexpr (HAS_BIAS) asm volatile("" : "+v"(bcol[ni][j].x), "+v"(bcol[ni][j].y), "+v"(bcol[ni][j].z), "+v"(bcol[ni][j].w));
      if (W8) asm volatile("" : "+v"(scol[ni][j].x), "+v"(scol[ni][j].y), "+v"(scol[ni][j].z), "+v"(scol[ni][j].w));
    }
  // per-row operand of slab s = mi * NI + ni, row group it: 16 bytes per lane (8 bf16 gate values / 4 fp32 residuals)
  constexpr bool AUX = EPI == 2 || EPI == 3 || EPI == 4;
  static_assert(WIDE8 || AUX, "the fp32-output epilogue (EPI 3) stores through emit_stores");
  const bool aux_on = AUX && !(EPI == 3 && p.res == nullptr);   // EPI 3 also serves fp32 outputs WITHOUT a residual (kernel argument: wave-uniform)
  const char* auxp = nullptr;
  long long aux_ld = 0;                                         // bytes per row
  if (EPI == 2) { auxp = reinterpret_cast<const char*>(mul + (long long)mrow0 * p.ldmul + ncol0); aux_ld = (long long)p.ldmul * 2; }
  if (EPI == 3) { auxp = reinterpret_cast<const char*>(res + (long long)mrow0 * p.ldres + ncol0); aux_ld = (long long)p.ldres * 4; }
  if (EPI == 4) { auxp = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.resT) + (long long)mrow0 * p.ldresT + ncol0); aux_ld = (long long)p.ldresT * 2; }
  // prefetch distance of the per-row operand in slabs. One slab ahead (rounds 2-5) the loads of slab sl are issued one slab period (~2 k clocks of a wave that
  // shares its SIMD) before their wait -- less than the latency of a load that misses L2 under load (the bf16 residual stream is 200 MB: HBM), and the
  // stream epilogue ran 17 k clocks against 3.5 k for the bf16-output one. The fragments are dead here, so the extra buffers cost no registers.
  // (three slabs ahead where a slab's operand is 2 registers per lane and row group -- the bf16 stream and the GEGLU gate of the 8-wave kernels; the fp32
  // residual (4 row groups x 4 registers per slab) and gemm_q4_kernel (128 of its accumulators are VGPRs) keep one: deeper they spill)
  constexpr int AUXD = (NIT == 2 && NI == 2) ? 3 : 1;
  f32x4_t aux[AUXD + 1][NIT];
  auto issue_aux = [&](int sl, f32x4_t (&dst)[NIT]) {
    const int mi = sl / NI, ni = sl % NI;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const char* q = auxp + (long long)(mi * 32 + it * RPI) * aux_ld + ni * 32 * (EPI == 3 ? 4 : 2);
      asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[it]) : "v"(q) : "memory");
    }
  };
  if (aux_on) {
#pragma unroll
    for (int d = 0; d < AUXD && d < MI * NI; ++d) issue_aux(d, aux[d]);
  }
  // Epilogues with a per-row operand (AUX) hold a slab's finished values in registers and issue its STORES one slab late, behind the next slab's
  // operand wait -- see the note at that wait. h_*: the held slab (which members are live depends on EPI; the others are dead code).
  uint4 h_o[NIT];
  uint2 h_o8[NIT];
  float4 h_v[NIT];
  float h_sq[NIT];
  auto emit_stores = [&](int sl) {   // the stores of slab sl from the held registers (compile-time sl after unrolling)
    const int mi = sl / NI, ni = sl % NI;
    const int n = ncol0 + ni * 32;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const long long m = mrow0 + mi * 32 + it * RPI;
      if constexpr (WIDE8) {
        if (!O8 || outT) *reinterpret_cast<uint4*>(outT + m * p.ldT + n) = h_o[it];
        if (EPI == 4 && O8 && p.out8) *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.out8) + m * p.ld8 + n) = h_o8[it];
        if (EPI == 4 && ssq_out && (elane & 3) == 0) ssq_out[m * (p.N >> 5) + ((n0 + wn * (NI * 32) + ni * 32) >> 5)] = h_sq[it];
      } else {
        store4(out32 + m * p.ld32 + n, h_v[it]);
        if (outT) store4(outT + m * p.ldT + n, h_v[it]);
        if (ssq_out && (elane & 7) == 0) ssq_out[m * (p.N >> 5) + ((n0 + wn * (NI * 32) + ni * 32) >> 5)] = h_sq[it];   // plain store: deterministic
      }
    }
  };
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const float rsc = rscv[mi];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      constexpr int NSLAB = MI * NI;
      const int sl = mi * NI + ni;
      // ---- (1) stage
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(acc[mi][ni][4 * q], acc[mi][ni][4 * q + 1], acc[mi][ni][4 * q + 2], acc[mi][ni][4 * q + 3]);
        if constexpr (HAS_RS) { v.x *= rsc; v.y *= rsc; v.z *= rsc; v.w *= rsc; }
        *reinterpret_cast<float4*>(stg + el31 * 32 + (((2 * q + ehi) ^ fw_) << 2)) = v;
      }
      // ---- prefetch the next slab's per-row operand, wait for this slab's (issued one slab ago), THEN issue the previous slab's stores.
      // gfx950 retires loads in order and stores in order, but NOT loads and stores against each other (measured in round 4,
      // profiles/r04_gemm_epilogue_ablation.txt: counting the previous slab's stores as "younger than these loads" gave 2-5 % wrong output
      // elements -- store acknowledgements overtake older loads). A counted wait for LOADS may therefore only count younger LOADS: vmcnt(NIT)
      // (the NIT loads just issued for the next slab; 0 behind the last slab) is the largest count that guarantees all NIT loads of this slab,
      // whatever the stores do. It also waits for every store still in flight -- which is why a slab's stores are issued one slab LATE, from
      // registers (emit_stores): the stores in flight at this wait are those of slab sl - 2, issued a whole slab ago, not those of the slab just
      // finished. Rounds 2-3 stored right away and waited vmcnt(2 NIT) here: correct only while the youngest store was still unacknowledged
      // whenever a load was -- a timing margin that was never seen to fail (bit-exact cross-checks over 1e8-4e8 elements per run), but no
      // guarantee. Cost of the guarantee, same box, pp kernel: T5 o +2 %, wo +1 %, fp32-residual and GEGLU epilogues -2...+3 %. The "memory"
      // clobber keeps hipcc from moving the C++ stores across the wait.
      if (AUX) {
       if (aux_on) {
        if (sl + AUXD < NSLAB) issue_aux(sl + AUXD, aux[(sl + AUXD) % (AUXD + 1)]);
        f32x4_t(&a)[NIT] = aux[sl % (AUXD + 1)];
        // younger LOADS at this point: those of slabs sl + 1 .. min(sl + AUXD, NSLAB - 1)
        const int younger = (NSLAB - 1 - sl < AUXD ? NSLAB - 1 - sl : AUXD) * NIT;   // compile-time after unrolling
        if constexpr (NIT == 2) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "n"(younger) : "memory");
        else asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "n"(younger) : "memory");
       }
        if (sl > 0) emit_stores(sl - 1);
      }
      // ---- (2) read back and finish
      const int n = ncol0 + ni * 32;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int r = it * RPI + crow;
        const int f = fsw(r);
        const long long m = mrow0 + mi * 32 + it * RPI;
        float4 v[CPL / 4];
#pragma unroll
        for (int j = 0; j < CPL / 4; ++j) {
          v[j] = *reinterpret_cast<const float4*>(stg + r * 32 + ((((ccol >> 2) + j) ^ f) << 2));
          if (W8) { v[j].x *= scol[ni][j].x; v[j].y *= scol[ni][j].y; v[j].z *= scol[ni][j].z; v[j].w *= scol[ni][j].w; }
          if constexpr (HAS_BIAS) { v[j].x += bcol[ni][j].x; v[j].y += bcol[ni][j].y; v[j].z += bcol[ni][j].z; v[j].w += bcol[ni][j].w; }
          if (act != ACT_NONE) { v[j].x = apply_act_t<bf16_t>(v[j].x, act); v[j].y = apply_act_t<bf16_t>(v[j].y, act); v[j].z = apply_act_t<bf16_t>(v[j].z, act); v[j].w = apply_act_t<bf16_t>(v[j].w, act); }
        }
        if constexpr (EPI == 2) {        // x gate: 8 bf16 values
          const f32x4_t g = aux[sl % (AUXD + 1)][it];
          const uint32_t g0 = __float_as_uint(g[0]), g1 = __float_as_uint(g[1]), g2 = __float_as_uint(g[2]), g3 = __float_as_uint(g[3]);
          v[0].x *= __uint_as_float(g0 << 16); v[0].y *= __uint_as_float(g0 & 0xffff0000u);
          v[0].z *= __uint_as_float(g1 << 16); v[0].w *= __uint_as_float(g1 & 0xffff0000u);
          v[1].x *= __uint_as_float(g2 << 16); v[1].y *= __uint_as_float(g2 & 0xffff0000u);
          v[1].z *= __uint_as_float(g3 << 16); v[1].w *= __uint_as_float(g3 & 0xffff0000u);
        }
        if constexpr (EPI == 4) {        // + residual carried in bf16: 8 values
          const f32x4_t g = aux[sl % (AUXD + 1)][it];
          const uint32_t g0 = __float_as_uint(g[0]), g1 = __float_as_uint(g[1]), g2 = __float_as_uint(g[2]), g3 = __float_as_uint(g[3]);
          v[0].x += __uint_as_float(g0 << 16); v[0].y += __uint_as_float(g0 & 0xffff0000u);
          v[0].z += __uint_as_float(g1 << 16); v[0].w += __uint_as_float(g1 & 0xffff0000u);
          v[1].x += __uint_as_float(g2 << 16); v[1].y += __uint_as_float(g2 & 0xffff0000u);
          v[1].z += __uint_as_float(g3 << 16); v[1].w += __uint_as_float(g3 & 0xffff0000u);
        }
        if constexpr (EPI == 3) {        // + residual: 4 fp32 values
          if (aux_on) {
            const f32x4_t r4 = aux[sl % (AUXD + 1)][it];
            v[0].x += r4[0]; v[0].y += r4[1]; v[0].z += r4[2]; v[0].w += r4[3];
          }
        }
        if constexpr (WIDE8) {
          uint4 o;
          o.x = pack2_bf16(v[0].x, v[0].y); o.y = pack2_bf16(v[0].z, v[0].w); o.z = pack2_bf16(v[1].x, v[1].y); o.w = pack2_bf16(v[1].z, v[1].w);
          if constexpr (AUX) h_o[it] = o;
          else {
          if constexpr (EPI == 5) {   // head-major: [batch][head][row][hm_D]; a tile's 256 rows lie in one batch (hm_L % 256 == 0), the lane's 8 columns in one head
            const int lg = 31 - __builtin_clz((unsigned)p.hm_D);
            const long long bq = m0 / p.hm_L;
            const long long o_ = bq * (long long)p.hm_L * (p.N - p.hm_D) + m * p.hm_D + ((long long)(n >> lg) * p.hm_L << lg) + (n & (p.hm_D - 1));
            *reinterpret_cast<uint4*>(outT + o_) = o;
          } else
          if (!O8 || outT) *reinterpret_cast<uint4*>(outT + m * p.ldT + n) = o;
          }
          if (O8 && p.out8) {   // fp8 e4m3 copy for an fp8 consumer GEMM: e4m3(v * out8_inv), saturating
            const float q = p.out8_inv;
            auto cl = [](float x) { return __builtin_amdgcn_fmed3f(x, -448.0f, 448.0f); };
            int w0 = 0, w1 = 0;
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v[0].x * q), cl(v[0].y * q), w0, false);
            w0 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v[0].z * q), cl(v[0].w * q), w0, true);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v[1].x * q), cl(v[1].y * q), w1, false);
            w1 = __builtin_amdgcn_cvt_pk_fp8_f32(cl(v[1].z * q), cl(v[1].w * q), w1, true);
            if constexpr (AUX) h_o8[it] = make_uint2((unsigned)w0, (unsigned)w1);
            else *reinterpret_cast<uint2*>(reinterpret_cast<char*>(p.out8) + m * p.ld8 + n) = make_uint2((unsigned)w0, (unsigned)w1);
          }
          if (EPI == 4 && ssq_out) {   // RMS partials of the stored (rounded) stream: 4 lanes hold the 32 columns of a slab row
            float sq = sumsq8_bf16(o);
            sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64);
            h_sq[it] = sq;
          }
        } else {   // EPI 3 (always AUX): fp32 (+ operand-type) output, stored by emit_stores
          h_v[it] = v[0];
          if (ssq_out) {   // the row's 8 column groups of this 32-column slab: butterfly, one partial per row and slab
            float sq = sumsq4(v[0]);
            sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64); sq += __shfl_xor(sq, 4, 64);
            h_sq[it] = sq;
          }
        }
      }
    }
  }
  if (AUX) emit_stores(MI * NI - 1);
}

template <int ACT, int EPI, bool W8, bool OUT8 = true, int NI = 2, int MI = 4>
__device__ __forceinline__ void tile_epilogue_256(const GemmDev& p, const f32x16_t (&acc)[MI][NI], const float (&rscv)[MI], char* slab,
                                                  int lane, int m0, int n0, int wm, int wn) {
  const bool hb = p.bias != nullptr;                                   // kernel arguments: wave-uniform branches
  const bool hr = (EPI == 0 || EPI == 1 || EPI == 5) && p.rs_ssq != nullptr;
  if constexpr (EPI == 0 || EPI == 1 || EPI == 5) {
    if (hb) {
      if (hr) tile_epilogue_256_impl<ACT, EPI, W8, true, true, OUT8, NI, MI>(p, acc, rscv, slab, lane, m0, n0, wm, wn);
      else tile_epilogue_256_impl<ACT, EPI, W8, true, false, OUT8, NI, MI>(p, acc, rscv, slab, lane, m0, n0, wm, wn);
    } else {
      if (hr) tile_epilogue_256_impl<ACT, EPI, W8, false, true, OUT8, NI, MI>(p, acc, rscv, slab, lane, m0, n0, wm, wn);
      else tile_epilogue_256_impl<ACT, EPI, W8, false, false, OUT8, NI, MI>(p, acc, rscv, slab, lane, m0, n0, wm, wn);
    }
  } else {
    if (hb) tile_epilogue_256_impl<ACT, EPI, W8, true, false, OUT8, NI, MI>(p, acc, rscv, slab, lane, m0, n0, wm, wn);
    else tile_epilogue_256_impl<ACT, EPI, W8, false, false, OUT8, NI, MI>(p, acc, rscv, slab, lane, m0, n0, wm, wn);
  }
}

// GEGLU PAIR epilogue of the 256x256 tile (GemmArgs::pair32 on gemm_pp_kernel, EPI 6): the weight's rows alternate in blocks of 32 between the GELU'd
// layer and its plain multiplier, so a lane's acc[mi][0][r] / acc[mi][1][r] are the two factors of ONE output element (the wave's 128x64 block
// becomes 128x32 outputs). Same operations and roundings as gemm_kernel's pair epilogue (and hence as the two-launch form): out[m][j] =
// bf16(gelu(a + bias) * bf16(g + bias_g)), with the fused LayerNorm of the GELU'd factor's input (GemmArgs::rs_sum) applied first when HAS_LN.
// The 32x32 product slab goes through the wave's private LDS slab (the XOR-swizzled layout of tile_epilogue_256_impl) and is stored
// row-contiguously, 8 bf16 per lane.
template <bool HAS_LN>
__device__ __forceinline__ void tile_epilogue_256_pair(const GemmDev& p, const f32x16_t (&acc)[4][2], char* slab, int lane, int m0, int n0, int wm, int wn) {
  using T = bf16_t;
  auto fsw = [](int r) { return ((r >> 1) & 1) | ((r & 1) << 1) | (r & 4); };
  float* stg = reinterpret_cast<float*>(slab);
  int elane = lane;
  asm volatile("" : "+v"(elane));
  const int el31 = elane & 31, ehi = elane >> 5;
  const int fw_ = fsw(el31);
  const int ccol = (elane & 3) * 8, crow = elane >> 2;     // read-back: 8 columns per lane, 4 lanes per row, 16 rows per instruction
  T* outT = reinterpret_cast<T*>(p.outT);
  const int nb0 = n0 + wn * 64 + 4 * ehi;                 // + 8 q: this lane's columns of the GELU'd block (interleaved space); multiplier block: + 32
  const int nout = (n0 >> 1) + wn * 32 + ccol;
  // (the column constants -- bias of both factors, the LayerNorm's c -- are re-read per 32-row block from L1: held for the whole tile they are 48
  // registers on top of the 128 accumulators and the kernel spills; the row statistics are fetched here, not at the start of the tile, for the
  // same reason -- the pair form runs where a CU owns one or two tiles, so there is no next tile's operand stream to queue behind)
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) {
    float mu = 0.f, rsd = 1.f;
    if constexpr (HAS_LN) ln_row_stats(p.rs_sum, p.rs_ssq, p.rs_parts, m0 + wm * 128 + mi * 32 + el31, p.rs_invk, p.rs_eps, mu, rsd);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float4 a = make_float4(acc[mi][0][4 * q], acc[mi][0][4 * q + 1], acc[mi][0][4 * q + 2], acc[mi][0][4 * q + 3]);
      float4 g = make_float4(acc[mi][1][4 * q], acc[mi][1][4 * q + 1], acc[mi][1][4 * q + 2], acc[mi][1][4 * q + 3]);
      if constexpr (HAS_LN) {
        const float4 c4 = load4(p.rs_c + nb0 + 8 * q);
        a.x = __builtin_fmaf(-mu, c4.x, a.x) * rsd; a.y = __builtin_fmaf(-mu, c4.y, a.y) * rsd;
        a.z = __builtin_fmaf(-mu, c4.z, a.z) * rsd; a.w = __builtin_fmaf(-mu, c4.w, a.w) * rsd;
      }
      if (p.bias) {
        const float4 ba = load4(p.bias + nb0 + 8 * q), bg = load4(p.bias + nb0 + 8 * q + 32);
        a.x += ba.x; a.y += ba.y; a.z += ba.z; a.w += ba.w; g.x += bg.x; g.y += bg.y; g.z += bg.z; g.w += bg.w;
      }
      a.x = apply_act_t<T>(a.x, ACT_GELU); a.y = apply_act_t<T>(a.y, ACT_GELU); a.z = apply_act_t<T>(a.z, ACT_GELU); a.w = apply_act_t<T>(a.w, ACT_GELU);
      const uint32_t g01 = pack2_bf16(g.x, g.y), g23 = pack2_bf16(g.z, g.w);   // the multiplier as the separate launch stores it
      a.x *= __uint_as_float(g01 << 16); a.y *= __uint_as_float(g01 & 0xffff0000u);
      a.z *= __uint_as_float(g23 << 16); a.w *= __uint_as_float(g23 & 0xffff0000u);
      *reinterpret_cast<float4*>(stg + el31 * 32 + (((2 * q + ehi) ^ fw_) << 2)) = a;
    }
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int r = it * 16 + crow;
      const int f = fsw(r);
      const float4 v0 = *reinterpret_cast<const float4*>(stg + r * 32 + ((((ccol >> 2) + 0) ^ f) << 2));
      const float4 v1 = *reinterpret_cast<const float4*>(stg + r * 32 + ((((ccol >> 2) + 1) ^ f) << 2));
      const long long m = m0 + wm * 128 + mi * 32 + r;
      uint4 o;
      o.x = pack2_bf16(v0.x, v0.y); o.y = pack2_bf16(v0.z, v0.w); o.z = pack2_bf16(v1.x, v1.y); o.w = pack2_bf16(v1.z, v1.w);
      *reinterpret_cast<uint4*>(outT + m * p.ldT + nout) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------ persistent GEMM
// Measured on MI355X (scripts/gpu_job_ablate.sh, profiles/r01_gemm_ablation.md): the 256x256 main loop is NOT bound by
// the matrix pipe but by the global->LDS path -- with the MFMAs removed it runs no faster, ~19-25 B/clk/CU however
// the 64 KiB per K-slice are fetched (LDS-DMA, or global_load + ds_write). At K = 768 a tile therefore spends 40 k
// clocks streaming operands and another 20 k in a prologue (first slices' latency) and an epilogue (stores) during
// which that path idles. The persistent kernel keeps ONE workgroup per CU and treats the K-slices of all its tiles as
// one continuous LDS-DMA stream: the first two slices of tile i+1 are requested during the last two k-slices of tile
// i and land while tile i's epilogue runs. The epilogue stages through its own 32 KiB of LDS (32x32 fp32 slabs per
// wave, XOR-swizzled: conflict-free for ds_write_b128 and both read-back shapes), so both ring stages stay free.
// EPI specialises the epilogue at compile time (fewer live scalars / registers than the all-runtime form, which spilled):
//   (0 = every feature a runtime flag: not instantiated) 1 bf16-only output, optional bias / fused-RMSNorm row scale
//   2 bf16-only output x gate (`mul`: GEGLU)      3 fp32 (+ optional bf16) output, with or without an fp32 residual, optional RMS partials
//   4 residual stream carried in bf16 (`resT`): out = bf16(acc + bias + bf16 residual), optional RMS partials of the stored values
template <int ACT, int EPI, bool W8 = false>
__global__ __launch_bounds__(TileL::THREADS, 2) void gemm_persistent_kernel(const GemmDev p) {
  using T = bf16_t;
  using TL = TileL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = TL::RB, NS = TL::NS, CPR = TL::CPR;
  constexpr int BK = RB / (int)sizeof(T), EPC = KCfg<T>::EPC, KSTEPS = BK / 16;
  // W8: fp8 e4m3 weight rows (64 B per K-slice, half the LDS-DMA pieces), widened to bf16 in registers (Frag::loadw)
  constexpr int RBW = W8 ? RB / 2 : RB, CPRW = RBW / 16, PWN = W8 ? TL::PW / 2 : TL::PW, ESW = W8 ? 1 : (int)sizeof(T);
  constexpr int MI = TL::MI, NI = TL::NI, NW = TL::NW, NP = TL::PA + PWN;
  constexpr int EPI_OFF = NS * TL::STAGE_BYTES;     // epilogue slabs live behind the ring: NW x 4 KiB
  static_assert(NS == 2 && RB == 128 && NI == 2 && MI == 4 && NW == 8, "written for TileL");
  static_assert(NP <= MI * NI, "at most one LDS-DMA piece per MFMA of the last k-step");

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int G = gridDim.x;
  const int nk = p.K / BK;
  const T* A = reinterpret_cast<const T*>(p.A);
  const char* W = reinterpret_cast<const char*>(p.W);

  // virtual tile id v -> (tm, tn): workgroup b only ever takes v = b, b + G, ... (G % 8 == 0), so v & 7 is its XCD and
  // one XCD walks the n-tiles of one A panel back to back, as in the one-tile-per-workgroup kernel
  auto tile_at = [&](int v, int& tm, int& tn) {
    const int idx = v >> 3;
    const int q = idx / p.ntiles;
    tn = idx - q * p.ntiles;
    tm = q * 8 + (v & 7);
  };
  auto next_valid = [&](int v) {
    while (v < p.vtotal) {
      int tm, tn;
      tile_at(v, tm, tn);
      if (tm < p.mtiles) return v;
      v += G;
    }
    return -1;
  };

  // ---- issue cursor: the (tile, K-slice) the next LDS-DMA slice belongs to; runs up to two slices ahead of the MFMAs
  // per-lane BYTE offsets from A / W (SADDR-form LDS-DMA: uniform 64-bit base + unsigned 32-bit lane offset; the
  // launcher guarantees they fit); the k-slice offset is added at issue. The launcher only sends problems with
  // M % 256 == 0 and N % 256 == 0 here, so no row needs clamping.
  unsigned offA[TL::PA], offW[PWN];
  int iv, ikt = 0;
  const int r0 = (w * 64 + lane) / CPR;                 // row of this lane within a 64-row piece group
  const int c0 = ((lane % CPR) ^ swz<RB>(r0)) * 16;     // swizzled 16-B chunk (the same for every piece: swz ignores r / 64)

  auto set_ptrs = [&](int v) {
    int tm, tn;
    tile_at(v, tm, tn);
#pragma unroll
    for (int i = 0; i < TL::PA; ++i)
      offA[i] = (unsigned)(tm * TL::BM + i * (NW * 64 / CPR) + r0) * (unsigned)(p.lda * (int)sizeof(T)) + c0;
    const int r0w = W8 ? (w * 64 + lane) / CPRW : r0;     // W tile rows of 64 B (fp8): 128-row piece groups; recomputed
    const int c0w = W8 ? ((lane % CPRW) ^ swz<RBW>(r0w)) * 16 : c0;   // per tile so that nothing extra stays live in the main loop
#pragma unroll
    for (int i = 0; i < PWN; ++i)
      offW[i] = (unsigned)(tn * TL::BN + i * (NW * 64 / CPRW) + r0w) * (unsigned)(p.ldw * ESW) + c0w;
  };
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto issue_piece = [&](int stage, int j) {
    const int i = j < TL::PA ? j : j - TL::PA;
    const int off = stage * TL::STAGE_BYTES + (j < TL::PA ? 0 : TL::A_BYTES) + (i * NW + w) * 1024;
    if (j < TL::PA) glds16_asm_s(A, offA[i] + (unsigned)(ikt * RB), smem_base + off);
    else glds16_asm_s(W, offW[i] + (unsigned)(ikt * RBW), smem_base + off);
  };
  auto advance_issue = [&]() {   // after the last piece of a slice
    if (++ikt == nk) {
      ikt = 0;
      iv = next_valid(iv + G);
      if (iv >= 0) set_ptrs(iv);
    }
  };

  int cv = next_valid(blockIdx.x);
  if (cv < 0) return;
  iv = cv;
  set_ptrs(iv);
  for (int t = 0; t < NS; ++t) {   // nk >= NS (launcher): both slices belong to the first tile
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(t, j);
    advance_issue();
  }
  asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NP) : "memory");   // slice 0 landed, slice 1 may be in flight
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int wm = w / TL::WN, wn = w % TL::WN;
  const int arow = wm * (MI * 32) + l31;
  const int wrow = wn * (NI * 32) + l31;
  const int act = ACT >= 0 ? ACT : p.act;
  int cur = 0;

  while (true) {
    int tm, tn;
    tile_at(cv, tm, tn);
    const int m0 = tm * TL::BM, n0 = tn * TL::BN;
    auto stamp = [&](int slot) {
      if (p.dbg && tid == 0) {
        long long* d = p.dbg + (long long)cv * 8;
        d[slot] = (long long)__builtin_readcyclecounter();
        if (slot == 0) {
          d[4] = (long long)__builtin_amdgcn_s_memrealtime();
          d[6] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
          d[7] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
        }
        if (slot == 3) d[5] = (long long)__builtin_amdgcn_s_memrealtime();
      }
    };
    stamp(0);
    // fused RMSNorm: scales of this lane's four accumulator rows, fetched at the START of the tile (the wave is about to
    // wait for its first K-slice anyway; in the epilogue the same loads would queue behind the next tile's LDS-DMA)
    float rscv[MI];
    int tl31 = l31;                     // opaque per tile: the row indices below are tile-invariant up to m0 and would
    asm volatile("" : "+v"(tl31));      // otherwise be hoisted out of the tile loop and held (spilled) across the main loop
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      rscv[mi] = 1.0f;
      if ((EPI == 0 || EPI == 1 || EPI == 5) && p.rs_ssq) {
        int mr = m0 + wm * (MI * 32) + mi * 32 + tl31; mr = mr < p.M ? mr : p.M - 1;
        rscv[mi] = rms_row_scale(p.rs_ssq, p.rs_parts, mr, p.rs_invk, p.rs_eps);
      }
    }
    f32x16_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    Frag<T> fa[2][MI], fw[2][NI];
    {
      const char* sA = smem + cur * TL::STAGE_BYTES;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[0][mi].template load<RB>(sA, arow + mi * 32, 0, hi);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fw[0][ni].template loadw<RBW, W8>(sA + TL::A_BYTES, wrow + ni * 32, 0, hi);
    }
    stamp(1);
    auto main_loop = [&](auto late_tag) {
      constexpr bool LATE = decltype(late_tag)::value;
      for (int kt = 0; kt < nk; ++kt) {
        const int nxt = cur ^ 1;
        const char* sA = smem + cur * TL::STAGE_BYTES;
        const char* sW = sA + TL::A_BYTES;
        const char* nA = smem + nxt * TL::STAGE_BYTES;
        const char* nW = nA + TL::A_BYTES;
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
          const int cb = kk & 1, nb = cb ^ 1;
          if (kk + 1 < KSTEPS) {
            if constexpr (LATE) {
              __builtin_amdgcn_sched_barrier(0);
#pragma unroll
              for (int mi = 0; mi < MI / 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mma(fw[cb][ni], fa[cb][mi], acc[mi][ni]);
              __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) fa[nb][mi].template load<RB>(sA, arow + mi * 32, kk + 1, hi);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) fw[nb][ni].template loadw<RBW, W8>(sW, wrow + ni * 32, kk + 1, hi);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = (LATE ? MI / 2 : 0); mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mma(fw[cb][ni], fa[cb][mi], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
          } else {
            // the next slice of the stream (slice kt+1, or slice 0 of the NEXT tile) has landed; every wave is done
            // reading stage `cur`
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (kt + 1 < nk) {
#pragma unroll
              for (int mi = 0; mi < MI; ++mi) fa[nb][mi].template load<RB>(nA, arow + mi * 32, 0, hi);
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) fw[nb][ni].template loadw<RBW, W8>(nW, wrow + ni * 32, 0, hi);
            }
            __builtin_amdgcn_sched_barrier(0);
            const bool more = iv >= 0;   // wave-uniform: the stream has another slice (this tile's or a later tile's)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
              for (int ni = 0; ni < NI; ++ni) {
                acc[mi][ni] = mma(fw[cb][ni], fa[cb][mi], acc[mi][ni]);
                __builtin_amdgcn_sched_barrier(0);
                if (more && mi * NI + ni < NP) issue_piece(cur, mi * NI + ni);
                __builtin_amdgcn_sched_barrier(0);
              }
            if (more) advance_issue();
          }
        }
        cur = nxt;
      }
    };
    if (kDephase && w >= NW / 2) main_loop(std::true_type{});
    else main_loop(std::false_type{});
    stamp(2);

    tile_epilogue_256<ACT, EPI, W8>(p, acc, rscv, smem + EPI_OFF + w * 4096, lane, m0, n0, wm, wn);
    stamp(3);
    cv = next_valid(cv + G);
    if (cv < 0) break;
  }
}

// ------------------------------------------------------------------------------------------------ ping-pong persistent GEMM
// Same 256x256 tile, same 32x32x16 MFMAs, same K order per accumulator (bit-identical results) and the same persistent
// stream / epilogue as gemm_persistent_kernel, with the main loop re-scheduled after the guide's 8-phase template
// (cdna_hip_programming.md "The 256^2 8-phase template"):
//   * the two waves of a SIMD (w and w + 4 = the two 128-row halves wm = 0 / 1 of the tile) run ONE BARRIER APART: while
//     one issues 8 MFMAs (a 64x32 quadrant of its 128x64 block x the whole 64-deep K-tile) under s_setprio 1, the other
//     reads the fragments of its next quadrant and issues LDS-DMA -- the matrix pipe of a SIMD always has exactly one
//     wave feeding it and the LDS / VMEM issue never sits in front of an MFMA;
//   * a K-tile is staged as four 16-KiB HALF-TILES, each the set of rows ONE phase reads: B0 (W rows wn*64 + [0,32) of
//     every wn), A0 (A rows wm*128 + [0,64)), B1, A1; quadrant order (A0,B0) (A0,B1) (A1,B1) (A1,B0) with both B
//     fragments kept in registers, so phases 1-3 read 12 / 4 / 8 ds_read_b128 and phase 4 none, and a half-tile slot is
//     free one phase after it was read (the reading waves wait lgkmcnt(0) BEFORE their barrier);
//   * ONE half-tile (2 LDS-DMA per wave) is requested per phase: phases 2-5 re-request the slots of the even K-tile (B0,
//     A0, B1, A1 of the K-tile two ahead), phases 6-8 and 1 those of the odd one; `s_waitcnt vmcnt(6)` in phases 4 and 8
//     only (never 0 while the stream lasts) retires the buffer that is read from the next phase on, three half-tiles stay
//     in flight. Measured A/B in one process (scripts/micro/gemm_lab.hip): waiting per half-tile as late as possible
//     (vmcnt(10) every phase, five half-tiles in flight) is 5-10 % SLOWER -- the template's throttle is the optimum; and
//     requesting the next tile's last half-tile before the epilogue + not waiting for the epilogue's store
//     acknowledgements in the next tile's first iteration changes nothing (+-1 %): both removed again.
// LDS: slot(par, h) = (par * 4 + h) * 16 KiB, h = 0 B0, 1 A0, 2 B1, 3 A1; rows of 128 B with the usual XOR chunk swizzle
// (slot rows are 32-aligned blocks of consecutive tile rows, so the conflict-free read pattern is unchanged); epilogue
// slabs behind the ring at 128 KiB.
// Measured on MI355X, uniform random [-1,1) operands (scripts/micro/gemm_lab.hip, profiles/r03_gemm_lab.txt): 8192^3 1.35
// PFLOP/s against 1.10-1.13 for gemm_persistent_kernel; 131072 x 2304 x 768 bf16-out 980 vs 836 TFLOP/s.
template <int P> struct PhaseTag { static constexpr int value = P; };

// F8: both operands fp8 e4m3 (A [M,K] and W [N,K] BYTES; lda / ldw in elements = bytes), K-tile = 128 elements = the same 128-B
// rows, 4 v_mfma_scale_f32_32x32x64_f8f6f4 per phase instead of 8 bf16 MFMAs (same 256 matrix-pipe cycles, twice the FLOPs, half
// the operand bytes per FLOP); the accumulator column is multiplied by wscale[n] * ascale in the epilogue.
template <int ACT, int EPI, bool F8 = false>
__global__ __launch_bounds__(TileL::THREADS, 2) void gemm_pp_kernel(const GemmDev p) {
  using T = bf16_t;
  using TL = TileL;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = 128, BK = F8 ? 128 : 64, ES = F8 ? 1 : 2, MI = 4, NI = 2;
  constexpr int SLOT = 16384, EPI_OFF = 8 * SLOT;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int wm = w >> 2, wn = w & 3;
  const int G = gridDim.x;
  const int nk = p.K / BK;              // even, >= 2 (launcher)
  const char* A = reinterpret_cast<const char*>(p.A);
  const char* W = reinterpret_cast<const char*>(p.W);

  // virtual tile id v -> (tm, tn); workgroup b takes v = b, b + G, ... (G % 8 == 0), so v & 7 is its XCD. Within an XCD the
  // n-tiles are walked in groups of `ngroup` (= ntiles: one group): the XCD takes ALL its A panels through one group of W
  // panels -- small enough to stay in its 4-MiB L2 -- before it moves to the next group (the A panels are then fetched once
  // per group; with one group the W panels of a wide N are re-streamed from beyond L2 for every A panel).
  auto tile_at = [&](int v, int& tm, int& tn) {
    if (p.flat) { tn = v / p.mtiles; tm = v - tn * p.mtiles; return; }
    const int idx = v >> 3;
    const int mtx = (p.mtiles + 7) >> 3;                // A panels per XCD (incl. padding panels)
    const int per_group = mtx * p.ngroup;
    const int g = idx / per_group;
    const int rem = idx - g * per_group;
    const int ng = min(p.ngroup, p.ntiles - g * p.ngroup);   // the last group may be smaller
    const int q = rem / ng;
    tn = g * p.ngroup + (rem - q * ng);
    tm = q * 8 + (v & 7);
  };
  auto next_valid = [&](int v) {
    while (v < p.vtotal) {
      int tm, tn;
      tile_at(v, tm, tn);
      if (tm < p.mtiles) return v;
      v += G;
    }
    return -1;
  };

  // ---- stream cursor: (tile iv, K-tile ikt) of the next half-tile request. Piece i (0 / 1) of a half-tile = slot rows
  // [(i * 8 + w) * 8, + 8); lane -> slot row rs, 16-B position pp, global chunk pp ^ ((rs >> 1) & 7).
  unsigned offA[2], offW[2];
  int iv, ikt = 0;
  auto set_ptrs = [&](int v) {
    int tm, tn;
    tile_at(v, tm, tn);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int rs = (i * 8 + w) * 8 + (lane >> 3);
      const int c = ((lane & 7) ^ ((rs >> 1) & 7)) * 16;
      offA[i] = (unsigned)(tm * 256 + (rs >> 6) * 128 + (rs & 63)) * (unsigned)(p.lda * ES) + c;    // A0 rows; A1 = + 64 rows
      offW[i] = (unsigned)(tn * 256 + (rs >> 5) * 64 + (rs & 31)) * (unsigned)(p.ldw * ES) + c;     // B0 rows; B1 = + 32 rows
    }
  };
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  const unsigned stepA1 = 64u * (unsigned)(p.lda * ES), stepB1 = 32u * (unsigned)(p.ldw * ES);
  // request half-tile h of the cursor's K-tile into slot(par, h)
  auto stage_half = [&](int par, int h) {
    if (iv < 0) return;
    const unsigned dst = smem_base + (unsigned)((par * 4 + h) * SLOT + w * 1024);
    const unsigned koff = (unsigned)(ikt * RB);
    if (h & 1) {
      const unsigned s = koff + (h == 3 ? stepA1 : 0u);
      glds16_asm_s(A, offA[0] + s, dst);
      glds16_asm_s(A, offA[1] + s, dst + 8 * 1024);
    } else {
      const unsigned s = koff + (h == 2 ? stepB1 : 0u);
      glds16_asm_s(W, offW[0] + s, dst);
      glds16_asm_s(W, offW[1] + s, dst + 8 * 1024);
    }
  };
  auto advance = [&]() {   // after the last half-tile (A1) of a K-tile
    if (iv < 0) return;
    if (++ikt == nk) {
      ikt = 0;
      iv = next_valid(iv + G);
      if (iv >= 0) set_ptrs(iv);
    }
  };

  int cv = next_valid(blockIdx.x);
  if (cv < 0) return;
  iv = cv;
  set_ptrs(iv);
  // prologue: K-tile 0 completely, K-tile 1 up to B1 (its A1 is phase 1's request)
  stage_half(0, 0); stage_half(0, 1); stage_half(0, 2); stage_half(0, 3); advance();
  stage_half(1, 0); stage_half(1, 1); stage_half(1, 2);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");    // K-tile 0 has landed
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  const int arow = wm * 64 + l31;     // slot row of A fragment mi2 = 0 (+ 32 for mi2 = 1)
  const int brow = wn * 32 + l31;     // slot row of the B fragment

  while (true) {
    int tm, tn;
    tile_at(cv, tm, tn);
    const int m0 = tm * TL::BM, n0 = tn * TL::BN;
    auto stamp = [&](int slot) {
      if (p.dbg && tid == 0) {
        long long* d = p.dbg + (long long)cv * 8;
        d[slot] = (long long)__builtin_readcyclecounter();
        if (slot == 0) {
          d[4] = (long long)__builtin_amdgcn_s_memrealtime();
          d[6] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
          d[7] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 20);
        }
        if (slot == 3) d[5] = (long long)__builtin_amdgcn_s_memrealtime();
      }
    };
    stamp(0);
    float rscv[MI];
    int tl31 = l31;
    asm volatile("" : "+v"(tl31));
    // Fused-RMSNorm row statistics (E = 768: 24 partial sums of squares per row, 96 contiguous bytes). Loaded into registers here they cost every
    // tile an exposed round trip to L2 / HBM (the consumer GEMMs of the T5 stack ran 2-2.5 % slower for it). Instead wave w requests the partials of
    // tile rows [32 w, 32 w + 32) -- 3 KiB, contiguous -- by LDS-DMA into ITS epilogue slab (idle until the epilogue); they land behind the main
    // loop's own counted waits (these requests are OLDER than every half-tile request of this tile, VMEM retires in order, and the launcher
    // guarantees nk >= 2, i.e. at least one vmcnt wait + barrier per tile) and are reduced after the loop, from LDS, in rms_row_scale's order.
    const bool rs_dma = (EPI == 1 || EPI == 5) && p.rs_ssq && p.rs_parts == 24;
    if (rs_dma) {
      const unsigned so = (unsigned)(m0 + w * 32) * 96u + (unsigned)lane * 16u;
      const unsigned dst = smem_base + (unsigned)(EPI_OFF + w * 4096);
      glds16_asm_s(p.rs_ssq, so, dst);
      glds16_asm_s(p.rs_ssq, so + 1024u, dst + 1024u);
      glds16_asm_s(p.rs_ssq, so + 2048u, dst + 2048u);
    }
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      rscv[mi] = 1.0f;
      if ((EPI == 1 || EPI == 5) && p.rs_ssq && !rs_dma) {
        const int mr = m0 + wm * (MI * 32) + mi * 32 + tl31;
        rscv[mi] = rms_row_scale(p.rs_ssq, p.rs_parts, mr, p.rs_invk, p.rs_eps);
      }
    }
    f32x16_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    constexpr int KS = F8 ? 2 : 4;  // MFMA k-steps per K-tile
    typename std::conditional<F8, Frag8, Frag<T>>::type fa[2][KS], fb[2][KS];   // A sub-tile [mi2][kk] (single buffer), B sub-tiles [b][kk] (both kept)
    auto ldf = [&](auto& f, const char* slot, int row, int kk) {
      if constexpr (F8) f.load(slot, row, kk, hi);
      else f.template load<RB>(slot, row, kk, hi);
    };
    stamp(1);
    // the 128-row halves run one barrier apart (re-joined at the end of the tile so that both run their epilogue together)
    if (wm == 1) __builtin_amdgcn_s_barrier();
#ifdef VIMA_PP_PHASE_STAMPS
    int lab_it = 0;
#endif

    auto phase = [&](auto tag) {
      constexpr int P = decltype(tag)::value;          // 1 .. 8
      constexpr int par = (P - 1) / 4, q = (P - 1) % 4 + 1;
      const char* sl = smem + par * 4 * SLOT;
      __builtin_amdgcn_sched_barrier(0);
      // ---- memory segment: fragments of this phase's quadrant, half-tile request(s)
      if constexpr (q == 1) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) ldf(fb[0][kk], sl + 0 * SLOT, brow, kk);
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          ldf(fa[0][kk], sl + 1 * SLOT, arow, kk);
          ldf(fa[1][kk], sl + 1 * SLOT, arow + 32, kk);
        }
      } else if constexpr (q == 2) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) ldf(fb[1][kk], sl + 2 * SLOT, brow, kk);
      } else if constexpr (q == 3) {
#pragma unroll
        for (int kk = 0; kk < KS; ++kk) {
          ldf(fa[0][kk], sl + 3 * SLOT, arow, kk);
          ldf(fa[1][kk], sl + 3 * SLOT, arow + 32, kk);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      // requests: P = 2..5 -> slots (0, 0..3) ; P = 6, 7, 8, 1 -> slots (1, 0..3)
      if constexpr (P >= 2 && P <= 5) stage_half(0, P - 2);
      if constexpr (P == 6 || P == 7) stage_half(1, P - 6);
      if constexpr (P == 8) stage_half(1, 2);
      if constexpr (P == 1) stage_half(1, 3);
      if constexpr (P == 5 || P == 1) advance();
      // waits (counted vmcnt, never 0 while the stream lasts): phase 4 retires the odd buffer (read in phases 5-7), phase 8
      // the even one; the three half-tiles requested last (6 LDS-DMA) stay in flight. VMEM operations retire in issue order.
      if constexpr (q == 4) {
        if (iv >= 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);              // lgkmcnt(0): this wave's fragment reads are done -> their slot is free
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      // ---- compute segment: quadrant (mi pair, ni) x K = 64
      constexpr int mi0 = (q <= 2) ? 0 : 2;
      constexpr int ni = (q == 1 || q == 4) ? 0 : 1;
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) {
        if constexpr (F8) {
          acc[mi0][ni] = mma8(fb[ni][kk], fa[0][kk], acc[mi0][ni]);
          acc[mi0 + 1][ni] = mma8(fb[ni][kk], fa[1][kk], acc[mi0 + 1][ni]);
          // hipcc sinks these (register-only) instructions out of their phase -- all 32 of an iteration ended up behind phase
          // 8, with the fragments spilled to scratch; an empty asm that consumes the accumulators pins them here
          if (kk == KS - 1) asm volatile("" : "+v"(acc[mi0][ni]), "+v"(acc[mi0 + 1][ni]));
        } else {
          acc[mi0][ni] = mma(fb[ni][kk], fa[0][kk], acc[mi0][ni]);
          acc[mi0 + 1][ni] = mma(fb[ni][kk], fa[1][kk], acc[mi0 + 1][ni]);
        }
      }
      __builtin_amdgcn_s_setprio(0);
      __builtin_amdgcn_sched_barrier(0);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#ifdef VIMA_PP_PHASE_STAMPS   // clocks at the end of every phase of iterations 0 and 1 (slots 0 .. 15 behind the standard slots)
      if (p.dbg && tid == 0 && lab_it < 4) p.dbg[(long long)p.vtotal * 8 + (long long)cv * 32 + lab_it * 4 + P - 1] = (long long)__builtin_readcyclecounter();
#endif
    };
    for (int it = 0; it < nk; it += 2) {
#ifdef VIMA_PP_PHASE_STAMPS
      lab_it = it;
#endif
      phase(PhaseTag<1>{}); phase(PhaseTag<2>{}); phase(PhaseTag<3>{}); phase(PhaseTag<4>{});
      phase(PhaseTag<5>{}); phase(PhaseTag<6>{}); phase(PhaseTag<7>{}); phase(PhaseTag<8>{});
#ifdef VIMA_PP_PHASE_STAMPS   // clocks at the end of iterations 2 .. 17 (slots 16 .. 31)
      if (p.dbg && tid == 0 && it >= 4 && it < 36) p.dbg[(long long)p.vtotal * 8 + (long long)cv * 32 + 14 + (it >> 1)] = (long long)__builtin_readcyclecounter();
#endif
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if constexpr (EPI == 1 || EPI == 5) {
      if (rs_dma) {   // rows wm * 128 + mi * 32 + l31 of the tile: slab of wave wm * 4 + mi, row l31
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
          const float* q = reinterpret_cast<const float*>(smem + EPI_OFF + (wm * 4 + mi) * 4096) + tl31 * 24;
          rscv[mi] = __builtin_amdgcn_rsqf(ln_tree24(q) * p.rs_invk + p.rs_eps);
        }
        // every wave of the workgroup has read its rows before any wave's epilogue overwrites a slab (one more workgroup barrier per tile; the two
        // 128-row halves have re-joined above, so it pairs up)
        __builtin_amdgcn_s_waitcnt(0xC07F);
        asm volatile("" : "+v"(rscv[0]), "+v"(rscv[1]), "+v"(rscv[2]), "+v"(rscv[3]));
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    stamp(2);
    if constexpr (EPI == 6) {
      if (p.rs_sum) tile_epilogue_256_pair<true>(p, acc, smem + EPI_OFF + w * 4096, lane, m0, n0, wm, wn);
      else tile_epilogue_256_pair<false>(p, acc, smem + EPI_OFF + w * 4096, lane, m0, n0, wm, wn);
    } else
    tile_epilogue_256<ACT, EPI, F8, F8>(p, acc, rscv, smem + EPI_OFF + w * 4096, lane, m0, n0, wm, wn);   // fp8 copy of the output: fp8-operand instantiations only
    stamp(3);
    cv = next_valid(cv + G);
    if (cv < 0) break;
  }
}

// ------------------------------------------------------------------------------------------------ 256 x 384 persistent GEMM
// The 256x256 loop period is the time the L2 -> LDS fabric needs for one K-slice, not the matrix pipe's: skipping a quarter
// of the LDS-DMA pieces (timing-only experiment) shortens it by 18 %, half of them by 23 % (then the MFMA
// floor). A 256 x 384 tile moves 80 KiB per 12.6 MFLOP instead of 64 KiB per 8.4 (154 instead of 128 FLOP per operand
// byte). Same stream structure as gemm_persistent_kernel -- one workgroup per CU, the K-slices of all its tiles one
// LDS-DMA stream, two stages -- with what the larger tile forces:
//   * 8 waves of 128 x 96 (MI = 4, NI = 3): 192 accumulator registers per wave, 64 left -> fragments are SINGLE-buffered and
//     reloaded as they retire (A fragment mi right after its last MFMA of the k-step, the W fragments between the MFMAs of
//     the last mi); DMA source offsets are one register per operand plus scalar strides;
//   * two 80-KiB stages fill the LDS: the epilogue slabs overlay the stage the tile's last slice was read from, so the
//     slice that would be requested into it during that last slice (slice 1 of the next tile) is requested right after the
//     epilogue instead (slice 0 of the next tile is in flight during the epilogue as before);
//   * the epilogue walks the slabs column-group-major (ni outer) so that only one group of column constants is live.
// EPI 1 (bf16 output, bias / activation / fused-RMSNorm row scale) and EPI 4 (bf16 residual stream, RMS partials).
using TileW = Tile<256, 384, 2, 4, 128, 2>;

template <int ACT, int EPI>
__global__ __launch_bounds__(TileW::THREADS, 2) void gemm_wide_kernel(const GemmDev p) {
  using T = bf16_t;
  using TL = TileW;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int RB = TL::RB, CPR = TL::CPR, BK = RB / 2, KSTEPS = BK / 16;
  constexpr int MI = TL::MI, NI = TL::NI, NW = TL::NW, NP = TL::PA + TL::PW;
  static_assert(TL::NS == 2 && RB == 128 && MI == 4 && NI == 3 && NW == 8 && NP <= MI * NI && EPI != 2 && EPI != 3, "written for TileW");
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, l31 = lane & 31;
  const int G = gridDim.x;
  const int nk = p.K / BK;
  const T* A = reinterpret_cast<const T*>(p.A);
  const char* W = reinterpret_cast<const char*>(p.W);
  auto tile_at = [&](int v, int& tm, int& tn) {
    const int idx = v >> 3;
    const int q = idx / p.ntiles;
    tn = idx - q * p.ntiles;
    tm = q * 8 + (v & 7);
  };
  auto next_valid = [&](int v) {
    while (v < p.vtotal) {
      int tm, tn;
      tile_at(v, tm, tn);
      if (tm < p.mtiles) return v;
      v += G;
    }
    return -1;
  };
  // LDS-DMA sources: piece i of an operand = 64 tile rows further down -> one per-lane byte offset per operand + a scalar stride
  unsigned offA0, offW0;
  const unsigned stepA = 64u * (unsigned)(p.lda * 2), stepW = 64u * (unsigned)(p.ldw * 2);
  int iv, ikt = 0;
  const int r0 = (w * 64 + lane) / CPR;
  const int c0 = ((lane % CPR) ^ swz<RB>(r0)) * 16;
  auto set_ptrs = [&](int v) {
    int tm, tn;
    tile_at(v, tm, tn);
    offA0 = (unsigned)(tm * TL::BM + r0) * (unsigned)(p.lda * 2) + c0;
    offW0 = (unsigned)(tn * TL::BN + r0) * (unsigned)(p.ldw * 2) + c0;
  };
  const unsigned smem_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)smem;
  auto issue_piece = [&](int stage, int j) {
    const int i = j < TL::PA ? j : j - TL::PA;
    const int off = stage * TL::STAGE_BYTES + (j < TL::PA ? 0 : TL::A_BYTES) + (i * NW + w) * 1024;
    if (j < TL::PA) glds16_asm_s(A, offA0 + (unsigned)i * stepA + (unsigned)(ikt * RB), smem_base + off);
    else glds16_asm_s(W, offW0 + (unsigned)i * stepW + (unsigned)(ikt * RB), smem_base + off);
  };
  auto advance_issue = [&]() {
    if (++ikt == nk) {
      ikt = 0;
      iv = next_valid(iv + G);
      if (iv >= 0) set_ptrs(iv);
    }
  };
  int cv = next_valid(blockIdx.x);
  if (cv < 0) return;
  iv = cv;
  set_ptrs(iv);
  for (int t = 0; t < 2; ++t) {   // nk >= 2 (launcher): both slices belong to the first tile
#pragma unroll
    for (int j = 0; j < NP; ++j) issue_piece(t, j);
    advance_issue();
  }
  const int wm = w / TL::WN, wn = w % TL::WN;
  const int arow = wm * (MI * 32) + l31;
  const int wrow = wn * (NI * 32) + l31;
  const int act = ACT >= 0 ? ACT : p.act;
  int cur = 0;
  bool pending = true;            // a slice younger than slice 0 of the coming tile is in flight (wave-uniform)

  while (true) {
    int tm, tn;
    tile_at(cv, tm, tn);
    const int m0 = tm * TL::BM, n0 = tn * TL::BN;
    auto stamp = [&](int slot) {
      if (p.dbg && tid == 0) {
        long long* d = p.dbg + (long long)cv * 8;
        d[slot] = (long long)__builtin_readcyclecounter();
        if (slot == 0) d[4] = (long long)__builtin_amdgcn_s_memrealtime();
        if (slot == 3) d[5] = (long long)__builtin_amdgcn_s_memrealtime();
      }
    };
    stamp(0);
    // slice 0 of this tile has landed (its pieces are older than everything except the NP pieces of the slice after it)
    if (pending) asm volatile("s_waitcnt vmcnt(%0)" ::"i"(NP) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float rscv[MI];
    int tl31 = l31;
    asm volatile("" : "+v"(tl31));
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      rscv[mi] = 1.0f;
      if ((EPI == 1 || EPI == 5) && p.rs_ssq) {
        const int mr = m0 + wm * (MI * 32) + mi * 32 + tl31;
        rscv[mi] = rms_row_scale(p.rs_ssq, p.rs_parts, mr, p.rs_invk, p.rs_eps);
      }
    }
    f32x16_t acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;
    Frag<T> fa[MI], fw[NI];
    {
      const char* sA = smem + cur * TL::STAGE_BYTES;
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) fa[mi].template load<RB>(sA, arow + mi * 32, 0, hi);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) fw[ni].template load<RB>(sA + TL::A_BYTES, wrow + ni * 32, 0, hi);
    }
    stamp(1);
    for (int kt = 0; kt < nk; ++kt) {
      const int nxt = cur ^ 1;
      const char* sA = smem + cur * TL::STAGE_BYTES;
      const char* nA = smem + nxt * TL::STAGE_BYTES;
      const bool last_slice = kt + 1 == nk;
#pragma unroll
      for (int kk = 0; kk < KSTEPS; ++kk) {
        const bool last = kk + 1 == KSTEPS;
        if (last) {
          // slice kt+1 of the stream (or slice 0 of the next tile) has landed; every wave is done READING stage `cur` (the
          // fragments of this last k-step were fetched during the previous one)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_waitcnt(0xC07F);
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
        const char* fA = last ? nA : sA;                    // where the NEXT k-step's fragments live
        const int fk = last ? 0 : kk + 1;
        const bool reload = !(last && last_slice);          // after the tile's last k-step the next fragments come at the next tile
        const bool more = last && !last_slice && iv >= 0;   // request the slice two ahead into the stage just freed
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            acc[mi][ni] = mma(fw[ni], fa[mi], acc[mi][ni]);
            __builtin_amdgcn_sched_barrier(0);
            if (more && mi * NI + ni < NP) issue_piece(cur, mi * NI + ni);
            if (mi == MI - 1 && reload) fw[ni].template load<RB>(fA + TL::A_BYTES, wrow + ni * 32, fk, hi);   // retired by this MFMA
            __builtin_amdgcn_sched_barrier(0);
          }
          if (reload) fa[mi].template load<RB>(fA, arow + mi * 32, fk, hi);   // its three MFMAs of this k-step are issued
          __builtin_amdgcn_sched_barrier(0);
        }
        if (more) advance_issue();
      }
      cur = nxt;
    }
    stamp(2);
    // ---------------------------------------------------------------- epilogue: slabs overlay the stage of the last slice
    auto fsw = [](int r) { return ((r >> 1) & 1) | ((r & 1) << 1) | (r & 4); };
    float* stg = reinterpret_cast<float*>(smem + (cur ^ 1) * TL::STAGE_BYTES + w * 4096);
    T* outT = reinterpret_cast<T*>(p.outT);
    float* ssq_out = EPI == 4 ? p.ssq_out : nullptr;
    int elane = lane;
    asm volatile("" : "+v"(elane));
    const int el31 = elane & 31, ehi = elane >> 5;
    const int fw_ = fsw(el31);
    constexpr int CPL = 8, LPR = 4, RPI = 16, NIT = 2;
    const int ccol = (elane % LPR) * CPL;
    const int crow = elane / LPR;
    const int ncol0 = n0 + wn * (NI * 32) + ccol;
    const int mrow0 = m0 + wm * (MI * 32) + crow;
    constexpr bool AUX = EPI == 4;
    const char* auxp = nullptr;
    long long aux_ld = 0;
    if (EPI == 4) { auxp = reinterpret_cast<const char*>(reinterpret_cast<const T*>(p.resT) + (long long)mrow0 * p.ldresT + ncol0); aux_ld = (long long)p.ldresT * 2; }
    f32x4_t aux[2][NIT];
    auto issue_aux = [&](int sl, f32x4_t (&dst)[NIT]) {   // slab sl = ni * MI + mi (column-group-major)
      const int ni = sl / MI, mi = sl % MI;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const char* q = auxp + (long long)(mi * 32 + it * RPI) * aux_ld + ni * 32 * 2;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[it]) : "v"(q) : "memory");
      }
    };
    if (AUX) issue_aux(0, aux[0]);
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      float4 bcol[2];
      bcol[0] = bcol[1] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p.bias) {
        bcol[0] = load4(p.bias + ncol0 + ni * 32);
        bcol[1] = load4(p.bias + ncol0 + ni * 32 + 4);
        if (!AUX) asm volatile("" : "+v"(bcol[0].x), "+v"(bcol[0].y), "+v"(bcol[0].z), "+v"(bcol[0].w), "+v"(bcol[1].x), "+v"(bcol[1].y), "+v"(bcol[1].z), "+v"(bcol[1].w));
      }
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        constexpr int NSLAB = MI * NI;
        const int sl = ni * MI + mi;
        const float rsc = rscv[mi];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = make_float4(acc[mi][ni][4 * q] * rsc, acc[mi][ni][4 * q + 1] * rsc, acc[mi][ni][4 * q + 2] * rsc, acc[mi][ni][4 * q + 3] * rsc);
          *reinterpret_cast<float4*>(stg + el31 * 32 + (((2 * q + ehi) ^ fw_) << 2)) = v;
        }
        if (AUX) {
          if (sl + 1 < NSLAB) issue_aux(sl + 1, aux[(sl + 1) & 1]);
          f32x4_t(&a)[NIT] = aux[sl & 1];
          // the bias loads of this column group (mi == 0) sit between the prefetch and this wait: they are older than the
          // prefetch of the NEXT slab, so the same counts cover them
          // loads-only count (see tile_epilogue_256_impl: loads and stores do not retire in order against each other); this opt-in kernel
          // issues a slab's stores right away, so the wait also covers every store issued so far
          if (sl + 1 < NSLAB) asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a[0]), "+v"(a[1]) : "i"(NIT) : "memory");
          else asm volatile("s_waitcnt vmcnt(0)" : "+v"(a[0]), "+v"(a[1]) : : "memory");
        }
        const int n = ncol0 + ni * 32;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const int r = it * RPI + crow;
          const int f = fsw(r);
          const long long m = mrow0 + mi * 32 + it * RPI;
          float4 v[2];
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            v[j] = *reinterpret_cast<const float4*>(stg + r * 32 + ((((ccol >> 2) + j) ^ f) << 2));
            if (p.bias) { v[j].x += bcol[j].x; v[j].y += bcol[j].y; v[j].z += bcol[j].z; v[j].w += bcol[j].w; }
            if (act != ACT_NONE) { v[j].x = apply_act_t<bf16_t>(v[j].x, act); v[j].y = apply_act_t<bf16_t>(v[j].y, act); v[j].z = apply_act_t<bf16_t>(v[j].z, act); v[j].w = apply_act_t<bf16_t>(v[j].w, act); }
          }
          if constexpr (EPI == 4) {
            const f32x4_t g = aux[sl & 1][it];
            const uint32_t g0 = __float_as_uint(g[0]), g1 = __float_as_uint(g[1]), g2 = __float_as_uint(g[2]), g3 = __float_as_uint(g[3]);
            v[0].x += __uint_as_float(g0 << 16); v[0].y += __uint_as_float(g0 & 0xffff0000u);
            v[0].z += __uint_as_float(g1 << 16); v[0].w += __uint_as_float(g1 & 0xffff0000u);
            v[1].x += __uint_as_float(g2 << 16); v[1].y += __uint_as_float(g2 & 0xffff0000u);
            v[1].z += __uint_as_float(g3 << 16); v[1].w += __uint_as_float(g3 & 0xffff0000u);
          }
          uint4 o;
          o.x = pack2_bf16(v[0].x, v[0].y); o.y = pack2_bf16(v[0].z, v[0].w); o.z = pack2_bf16(v[1].x, v[1].y); o.w = pack2_bf16(v[1].z, v[1].w);
          *reinterpret_cast<uint4*>(outT + m * p.ldT + n) = o;
          if (EPI == 4 && ssq_out) {
            float sq = sumsq8_bf16(o);
            sq += __shfl_xor(sq, 1, 64); sq += __shfl_xor(sq, 2, 64);
            if ((elane & 3) == 0) ssq_out[m * (p.N >> 5) + ((n0 + wn * (NI * 32) + ni * 32) >> 5)] = sq;
          }
        }
      }
    }
    stamp(3);
    // every wave is done with its slab before the deferred slice (slice 1 of the next tile) is requested into that stage
    __builtin_amdgcn_s_waitcnt(0xC07F);
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    pending = iv >= 0;
    if (pending) {
#pragma unroll
      for (int j = 0; j < NP; ++j) issue_piece(cur ^ 1, j);
      advance_issue();
    }
    cv = next_valid(cv + G);
    if (cv < 0) break;
  }
}

#include "gemm_q4.inc"      // gemm_q4_kernel: 256x384 tile, four waves (one per SIMD), Gray-code quadrant phases with rolling fragment reloads

#ifndef VIMA_GEMM_LAB
#include "gemm_small.inc"   // gemm_resident_kernel: underfilled grids (batch 1 .. 32, one env step), whole K in flight
#include "gemm_skinny.inc"  // gemm_skinny_kernel: M <= 32 (one env step at batch <= 3), K split over the waves of a workgroup, operands straight from global memory
#endif

// Knob resolution: the handle's Tuning value when set (>= 0), otherwise the process default from the environment
// (read once; never written afterwards, so it is safe to share between handles).
inline int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return (e && e[0]) ? atoi(e) : dflt;
}
inline int env_cached(const char* name, int& cache, int dflt) {
  if (cache < 0) cache = env_int(name, dflt);
  return cache;
}
int g_env_variant = -1, g_env_tile = -1, g_env_raster = -1, g_env_epi = -1, g_env_persist = -1, g_env_small = -1, g_env_splitk = -1;
#define VIMA_KNOB(fn, field, env, cache, dflt)                                   \
  inline int fn(const Tuning* t) { return (t && t->field >= 0) ? t->field : env_cached(env, cache, dflt); }
VIMA_KNOB(gemm_variant, gemm_variant, "VIMA_GEMM_VARIANT", g_env_variant, 1)
VIMA_KNOB(gemm_tile, gemm_tile, "VIMA_GEMM_TILE", g_env_tile, 0)
VIMA_KNOB(gemm_raster, gemm_raster, "VIMA_GEMM_RASTER", g_env_raster, 0)
VIMA_KNOB(gemm_epi, gemm_epi, "VIMA_GEMM_EPI", g_env_epi, 1)
VIMA_KNOB(gemm_persist, gemm_persist, "VIMA_GEMM_PERSIST", g_env_persist, 1)
VIMA_KNOB(gemm_small, gemm_small, "VIMA_GEMM_SMALL", g_env_small, 1)
VIMA_KNOB(gemm_splitk, gemm_splitk, "VIMA_GEMM_SPLITK", g_env_splitk, 0)
int g_env_wide = -1;
VIMA_KNOB(gemm_wide, gemm_wide, "VIMA_GEMM_WIDE", g_env_wide, 0)
int g_env_pp = -1;
VIMA_KNOB(gemm_pp, gemm_pp, "VIMA_GEMM_PP", g_env_pp, 1)
int g_env_q4 = -1;
VIMA_KNOB(gemm_q4, gemm_q4, "VIMA_GEMM_Q4", g_env_q4, 6)
int g_env_resident = -1, g_env_res_maxwg = -1;
VIMA_KNOB(gemm_resident, gemm_resident, "VIMA_GEMM_RESIDENT", g_env_resident, 1)
VIMA_KNOB(gemm_res_maxwg, gemm_res_maxwg, "VIMA_GEMM_RES_MAXWG", g_env_res_maxwg, 256)
int g_env_res_nch = -1;
VIMA_KNOB(gemm_res_nch, gemm_res_nch, "VIMA_GEMM_RES_NCH", g_env_res_nch, 0)
int g_env_skinny = -1;
VIMA_KNOB(gemm_skinny, gemm_skinny, "VIMA_GEMM_SKINNY", g_env_skinny, 1)
int g_env_flat = -1;
VIMA_KNOB(gemm_flat, gemm_flat, "VIMA_GEMM_FLAT", g_env_flat, 1)
#undef VIMA_KNOB

inline bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

template <typename T, typename TL, int ACT, bool VEC, bool ASMLDS, bool W8 = false>
int launch_inst(const GemmDev& d, dim3 grid, hipStream_t st) {
  static PerDeviceOnce attr;   // per instantiation, per device
  {
    const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_kernel<T, TL, ACT, VEC, ASMLDS, W8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, TL::SMEM_ALLOC); });
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((gemm_kernel<T, TL, ACT, VEC, ASMLDS, W8>), grid, dim3(TL::THREADS), TL::SMEM_ALLOC, st, d);
  return (int)hipGetLastError();
}

template <typename T, typename TL, bool ASMLDS>
int launch_tile(GemmDev d, const GemmArgs& a, bool vec, hipStream_t st) {
  d.mtiles = (d.M + TL::BM - 1) / TL::BM;
  d.ntiles = (d.N + TL::BN - 1) / TL::BN;
  const int groups = (d.mtiles + 7) / 8;
  d.raster = gemm_raster(a.tune);
  d.epi_lds = a.pair32 ? 2 : (gemm_epi(a.tune) || a.ssq_out != nullptr);   // the RMS partial sums exist only in the LDS epilogue; 2 = GEGLU pair
  d.dbg = a.tune ? a.tune->gemm_dbg : nullptr;
  {   // n-group: W panels of ~1.5 MB stay resident in one XCD's 4 MiB L2 while its A panels stream through
    const long long panel = (long long)TL::BN * a.K * (long long)sizeof(T);
    long long ng = (3LL << 19) / (panel > 0 ? panel : 1);
    if (ng < 1) ng = 1;
    if (ng > d.ntiles) ng = d.ntiles;
    d.ngroup = (int)ng;
  }
  dim3 grid((unsigned)(d.raster == 2 ? d.mtiles * d.ntiles : groups * 8 * d.ntiles), (unsigned)(a.batch > 0 ? a.batch : 1), 1);
#ifndef VIMA_GEMM_LAB
  if constexpr (sizeof(T) == 2) {
    if (a.w8) {   // fp8 weights: always the asm LDS-DMA pipeline and the vector epilogue (checked by launch_t)
      switch (a.act) {
        case ACT_NONE: return launch_inst<T, TL, ACT_NONE, true, true, true>(d, grid, st);
        case ACT_RELU: return launch_inst<T, TL, ACT_RELU, true, true, true>(d, grid, st);
        case ACT_GELU: return launch_inst<T, TL, ACT_GELU, true, true, true>(d, grid, st);
        case ACT_QUICKGELU: return launch_inst<T, TL, ACT_QUICKGELU, true, true, true>(d, grid, st);
        default: return (int)hipErrorInvalidValue;
      }
    }
  }
  if (!vec) return launch_inst<T, TL, -1, false, ASMLDS>(d, grid, st);
#endif
  switch (a.act) {
    case ACT_NONE: return launch_inst<T, TL, ACT_NONE, true, ASMLDS>(d, grid, st);
    case ACT_RELU: return launch_inst<T, TL, ACT_RELU, true, ASMLDS>(d, grid, st);
    case ACT_GELU: return launch_inst<T, TL, ACT_GELU, true, ASMLDS>(d, grid, st);
    case ACT_QUICKGELU: return launch_inst<T, TL, ACT_QUICKGELU, true, ASMLDS>(d, grid, st);
    default: return (int)hipErrorInvalidValue;
  }
}

int g_num_cu = 0;

template <int ACT, int EPI, bool W8>
int launch_persistent_w(const GemmDev& d, int grid, hipStream_t st) {
  constexpr int SMEM = TileL::SMEM_BYTES + TileL::NW * 4096;   // ring + epilogue slabs = 160 KiB
  static PerDeviceOnce attr;   // per instantiation, per device
  {
    const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_persistent_kernel<ACT, EPI, W8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); });
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((gemm_persistent_kernel<ACT, EPI, W8>), dim3((unsigned)grid), dim3(TileL::THREADS), SMEM, st, d);
  return (int)hipGetLastError();
}
template <int ACT, int EPI>
int launch_persistent_inst(const GemmDev& d, int grid, hipStream_t st) {
#ifdef VIMA_GEMM_LAB
  return launch_persistent_w<ACT, EPI, false>(d, grid, st);
#else
  return d.wscale ? launch_persistent_w<ACT, EPI, true>(d, grid, st) : launch_persistent_w<ACT, EPI, false>(d, grid, st);
#endif
}

int launch_persistent(GemmDev d, const GemmArgs& a, hipStream_t st) {
  if (a.split_n) return -1;   // (the column-split output is not part of the 256x256 epilogues)
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cu = n / 8 * 8;
  }
  d.mtiles = (d.M + TileL::BM - 1) / TileL::BM;
  d.ntiles = (d.N + TileL::BN - 1) / TileL::BN;
  d.vtotal = (d.mtiles + 7) / 8 * 8 * d.ntiles;
  d.raster = 0; d.ngroup = 1; d.epi_lds = 1;
  d.dbg = a.tune ? a.tune->gemm_dbg : nullptr;
  const int grid = d.vtotal < g_num_cu ? d.vtotal : g_num_cu;
  // specialised epilogues for the combinations the policy uses; everything else takes the generic instantiation
  int epi = 0;
  if (a.rb == 0) {
    if (d.wide8 && !a.mul && !a.res && !a.resT && !a.ssq_out) epi = 1;
    else if (d.wide8 && a.mul && a.outT && !a.out8 && !a.res && !a.resT && !a.rs_ssq && !a.ssq_out) epi = 2;
    else if (!d.wide8 && a.out32 && !a.mul && !a.rs_ssq && !a.resT) epi = 3;   // fp32 (+ operand-type) output, with or without a residual
    else if (d.wide8 && a.resT && !a.mul && !a.res && !a.rs_ssq) epi = 4;
  }
  if (a.resT && epi != 4) return -1;   // one-tile-per-workgroup kernel
  if (a.hm_D) {   // head-major output: its own instantiation (as a run-time branch of EPI 1 it cost every row-major launch ~9 %)
    if (epi != 1 || a.act != ACT_NONE) return (int)hipErrorInvalidValue;
    return launch_persistent_inst<ACT_NONE, 5>(d, grid, st);
  }
  switch (a.act * 8 + epi) {
    case ACT_NONE * 8 + 1: return launch_persistent_inst<ACT_NONE, 1>(d, grid, st);
    case ACT_NONE * 8 + 3: return launch_persistent_inst<ACT_NONE, 3>(d, grid, st);
    case ACT_NONE * 8 + 4: return launch_persistent_inst<ACT_NONE, 4>(d, grid, st);
    case ACT_RELU * 8 + 1: return launch_persistent_inst<ACT_RELU, 1>(d, grid, st);
    case ACT_GELU * 8 + 2: return launch_persistent_inst<ACT_GELU, 2>(d, grid, st);
    case ACT_QUICKGELU * 8 + 1: return launch_persistent_inst<ACT_QUICKGELU, 1>(d, grid, st);
    default: return -1;   // no specialised instantiation (the all-runtime form spills): one-tile-per-workgroup kernel
  }
}

template <int ACT, int EPI, bool F8>
int launch_pp_inst2(const GemmDev& d, int grid, hipStream_t st) {
  constexpr int SMEM = 8 * 16384 + TileL::NW * 4096;   // eight half-tile slots + epilogue slabs = 160 KiB
  static PerDeviceOnce attr;   // per instantiation, per device
  {
    const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pp_kernel<ACT, EPI, F8>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, SMEM); });
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((gemm_pp_kernel<ACT, EPI, F8>), dim3((unsigned)grid), dim3(TileL::THREADS), SMEM, st, d);
  return (int)hipGetLastError();
}
template <int ACT, int EPI>
int launch_pp_inst(const GemmDev& d, int grid, hipStream_t st) {
  if constexpr (EPI == 1 || EPI == 4 || EPI == 5) {     // the fp8-activation path exists for the bf16-output and bf16-stream epilogues
    if (d.raster == 8) return launch_pp_inst2<ACT, EPI, true>(d, grid, st);
  } else {
    if (d.raster == 8) return (int)hipErrorInvalidValue;
  }
  return launch_pp_inst2<ACT, EPI, false>(d, grid, st);
}

// ping-pong persistent kernel (option gemm_pp): same eligibility as launch_persistent plus an even number of K-tiles and
// bf16 weights; returns -1 when the problem does not fit it (caller falls back)
int launch_pp(GemmDev d, const GemmArgs& a, hipStream_t st) {
  if (a.split_n) return -1;
  if (a.a8) { if (!a.w8 || a.K % 256 != 0) return -1; }
  else if (a.w8 || (a.K / 64) % 2 != 0 || a.out8) return -1;   // (an fp8 copy of a bf16-operand GEMM's output: launch_persistent)
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cu = n / 8 * 8;
    if (kLab) { const int lim = env_int("VIMA_GEMM_LAB_CUS", 0); if (lim >= 8 && lim < g_num_cu) g_num_cu = lim / 8 * 8; }   // lab only: run the persistent grid on a part of the chip
  }
  d.mtiles = (d.M + TileL::BM - 1) / TileL::BM;
  d.ntiles = (d.N + TileL::BN - 1) / TileL::BN;
  d.vtotal = (d.mtiles + 7) / 8 * 8 * d.ntiles;
  d.raster = a.a8 ? 8 : 0; d.epi_lds = 1;     // raster 8: marker for the fp8-operand instantiation (the kernel ignores `raster`)
  d.ngroup = d.ntiles;
  {   // W panels of one n-group <= `VIMA_GEMM_NGROUP_KB` (default 2560 KiB) so that they stay resident in an XCD's L2
    static int kb = -1;
    if (kb < 0) kb = env_int("VIMA_GEMM_NGROUP_KB", 2560);
    const long long panel = (long long)TileL::BN * a.K * (a.a8 ? 1 : 2);
    // only for short K: every extra group re-reads the whole A matrix (measured: 131072 x 768 x 3072 in three groups
    // 1167 -> 999 TFLOP/s; 131072 x 3072 x 768 in two groups: same time, W no longer re-streamed from beyond L2)
    if (kb > 0 && a.K <= 1536 && (long long)d.ntiles * panel > (long long)kb * 1024) {
      int ng = (int)((long long)kb * 1024 / panel);
      if (ng < 1) ng = 1;
      const int groups = (d.ntiles + ng - 1) / ng;
      d.ngroup = (d.ntiles + groups - 1) / groups;    // equal groups
    }
  }
  d.dbg = a.tune ? a.tune->gemm_dbg : nullptr;
  // The XCD raster pads the A panels to a multiple of 8 (panel tm lives on XCD tm % 8): with mtiles % 8 small and a grid of about one tile per CU
  // the panels of the last group pile up on a few XCDs -- M = 2304 (an incremental env step at batch 256: 9 panels) x 24 n-tiles gave XCD 0 48 tiles
  // for its 32 workgroups, i.e. TWO rounds for 216 tiles (49 us instead of 30). Where the plain enumeration needs fewer rounds and the problem is
  // small enough for L2 placement not to matter (<= 2 tiles per workgroup) the raster is dropped.
  d.flat = 0;
  {
    const long long tiles = (long long)d.mtiles * d.ntiles;
    const long long per_xcd = (long long)((d.mtiles + 7) / 8) * d.ntiles, wg_xcd = g_num_cu / 8;
    const long long rounds_raster = (per_xcd + wg_xcd - 1) / wg_xcd, rounds_flat = (tiles + g_num_cu - 1) / g_num_cu;
    if (tiles <= 2LL * g_num_cu && rounds_flat < rounds_raster && gemm_flat(a.tune)) { d.flat = 1; d.vtotal = (int)tiles; }
  }
  const int grid = d.vtotal < g_num_cu ? d.vtotal : g_num_cu;
  int epi = 0;
  if (a.rb == 0) {
    if (d.wide8 && !a.mul && !a.res && !a.resT && !a.ssq_out) epi = 1;
    else if (d.wide8 && a.mul && a.outT && !a.out8 && !a.res && !a.resT && !a.rs_ssq && !a.ssq_out) epi = 2;
    else if (!d.wide8 && a.out32 && !a.mul && !a.rs_ssq && !a.resT) epi = 3;   // fp32 (+ operand-type) output, with or without a residual
    else if (d.wide8 && a.resT && !a.mul && !a.res && !a.rs_ssq) epi = 4;
  }
  if (a.resT && epi != 4) return -1;
  if (a.pair32) {   // GEGLU pair over block-interleaved weights (launch_t has validated the form): its own epilogue
    if (a.a8 || a.act != ACT_GELU) return -1;
    return launch_pp_inst<ACT_GELU, 6>(d, grid, st);
  }
  if (a.hm_D) {   // head-major output: its own instantiation (see launch_persistent)
    if (epi != 1 || a.act != ACT_NONE) return (int)hipErrorInvalidValue;
    return launch_pp_inst<ACT_NONE, 5>(d, grid, st);
  }
  switch (a.act * 8 + epi) {
    case ACT_NONE * 8 + 1: return launch_pp_inst<ACT_NONE, 1>(d, grid, st);
    case ACT_NONE * 8 + 3: return launch_pp_inst<ACT_NONE, 3>(d, grid, st);
    case ACT_NONE * 8 + 4: return launch_pp_inst<ACT_NONE, 4>(d, grid, st);
    case ACT_RELU * 8 + 1: return launch_pp_inst<ACT_RELU, 1>(d, grid, st);
    case ACT_GELU * 8 + 2: return launch_pp_inst<ACT_GELU, 2>(d, grid, st);
    case ACT_QUICKGELU * 8 + 1: return launch_pp_inst<ACT_QUICKGELU, 1>(d, grid, st);
    default: return -1;
  }
}

template <int ACT, int EPI>
int launch_wide_inst(const GemmDev& d, int grid, hipStream_t st) {
  static PerDeviceOnce attr;   // per instantiation, per device
  {
    const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wide_kernel<ACT, EPI>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, TileW::SMEM_BYTES); });
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((gemm_wide_kernel<ACT, EPI>), dim3((unsigned)grid), dim3(TileW::THREADS), TileW::SMEM_BYTES, st, d);
  return (int)hipGetLastError();
}

// 256 x 384 persistent kernel (option gemm_wide): returns -1 when the problem does not fit it (caller falls back)
int launch_wide(GemmDev d, const GemmArgs& a, hipStream_t st) {
  if (a.w8 || a.rb != 0 || a.batch > 1 || a.M % TileW::BM || a.N % TileW::BN || a.K < 2 * 64 || a.K % 64 || !d.wide8) return -1;
  if ((long long)a.M * a.lda * 2 >= (1LL << 32) || (long long)a.N * a.ldw * 2 >= (1LL << 32)) return -1;
  if (a.mul || a.res || a.out32 || a.hm_D || a.split_n) return -1;
  int epi = 0;
  if (!a.resT && !a.ssq_out) epi = 1;
  else if (a.resT && !a.rs_ssq && a.act == ACT_NONE) epi = 4;
  if (!epi) return -1;
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cu = n / 8 * 8;
  }
  d.mtiles = d.M / TileW::BM;
  d.ntiles = d.N / TileW::BN;
  if ((long long)d.mtiles * d.ntiles < 128) return -1;
  d.vtotal = (d.mtiles + 7) / 8 * 8 * d.ntiles;
  d.raster = 0; d.ngroup = 1; d.epi_lds = 1;
  d.dbg = a.tune ? a.tune->gemm_dbg : nullptr;
  const int grid = d.vtotal < g_num_cu ? d.vtotal : g_num_cu;
  switch (a.act * 8 + epi) {
    case ACT_NONE * 8 + 1: return launch_wide_inst<ACT_NONE, 1>(d, grid, st);
    case ACT_RELU * 8 + 1: return launch_wide_inst<ACT_RELU, 1>(d, grid, st);
    case ACT_QUICKGELU * 8 + 1: return launch_wide_inst<ACT_QUICKGELU, 1>(d, grid, st);
    case ACT_NONE * 8 + 4: return launch_wide_inst<ACT_NONE, 4>(d, grid, st);
    default: return -1;
  }
}

template <int ACT, int EPI, int MIH>
int launch_q4_inst(const GemmDev& d, int grid, hipStream_t st) {
  static PerDeviceOnce attr;   // per instantiation, per device
  {
    const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_q4_kernel<ACT, EPI, MIH>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, Q4T<MIH>::SMEM_BYTES); });
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((gemm_q4_kernel<ACT, EPI, MIH>), dim3((unsigned)grid), dim3(Q4::THREADS), Q4T<MIH>::SMEM_BYTES, st, d);
  return (int)hipGetLastError();
}

// four-wave kernel (option gemm_q4: 1 = 256 x 384 tile, 2 = 128 x 384 tile): returns -1 when the problem does not fit it (caller falls back to the 256x256 kernels)
template <int MIH>
int launch_q4(GemmDev d, const GemmArgs& a, hipStream_t st) {
  using QC = Q4T<MIH>;
  if (a.w8 || a.a8 || a.rb != 0 || a.batch > 1 || a.M % QC::BM || a.N % QC::BN || a.K < 128 || a.K % 128 || !d.wide8) return -1;
  if ((long long)a.M * a.lda * 2 >= (1LL << 32) || (long long)a.N * a.ldw * 2 >= (1LL << 32)) return -1;
  if (a.mul || a.res || a.out32 || a.out8 || a.split_n || a.pair32 || a.sum_out || a.rs_sum) return -1;
  if (a.rs_ssq && (MIH != 1 || a.rs_parts != 24 || a.resT)) return -1;   // fused-RMSNorm consumer: the 128 x 384 tile only (E = 768: 24 partials per row), bf16-output epilogues
  int epi = 0;
  if (!a.resT && !a.ssq_out) epi = a.hm_D ? 5 : 1;
  else if (a.resT && a.act == ACT_NONE && !a.hm_D) epi = 4;
  if (!epi) return -1;
  if (epi == 5 && (a.act != ACT_NONE || a.hm_L % QC::BM != 0)) return -1;
  if (g_num_cu == 0) {
    int dev = 0, n = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
      n = 256;
    g_num_cu = n / 8 * 8;
    if (kLab) { const int lim = env_int("VIMA_GEMM_LAB_CUS", 0); if (lim >= 8 && lim < g_num_cu) g_num_cu = lim / 8 * 8; }
  }
  d.mtiles = d.M / QC::BM;
  d.ntiles = d.N / QC::BN;
  if ((long long)d.mtiles * d.ntiles < 128) return -1;
  d.vtotal = (d.mtiles + 7) / 8 * 8 * d.ntiles;
  d.raster = 0; d.epi_lds = 1; d.flat = 0;
  d.ngroup = d.ntiles;
  {   // n-groups whose W panels stay in an XCD's L2 (see launch_pp)
    static int kb = -1;
    if (kb < 0) kb = env_int("VIMA_GEMM_NGROUP_KB", 2560);
    const long long panel = (long long)QC::BN * a.K * 2;
    if (kb > 0 && a.K <= 1536 && (long long)d.ntiles * panel > (long long)kb * 1024) {
      int ng = (int)((long long)kb * 1024 / panel);
      if (ng < 1) ng = 1;
      const int groups = (d.ntiles + ng - 1) / ng;
      d.ngroup = (d.ntiles + groups - 1) / groups;
    }
  }
  d.dbg = a.tune ? a.tune->gemm_dbg : nullptr;
  const int grid = d.vtotal < g_num_cu ? d.vtotal : g_num_cu;
  if (epi == 5) return launch_q4_inst<ACT_NONE, 5, MIH>(d, grid, st);
  switch (a.act * 8 + epi) {
    case ACT_NONE * 8 + 1: return launch_q4_inst<ACT_NONE, 1, MIH>(d, grid, st);
    case ACT_RELU * 8 + 1: return launch_q4_inst<ACT_RELU, 1, MIH>(d, grid, st);
    case ACT_QUICKGELU * 8 + 1: return launch_q4_inst<ACT_QUICKGELU, 1, MIH>(d, grid, st);
    case ACT_NONE * 8 + 4: return launch_q4_inst<ACT_NONE, 4, MIH>(d, grid, st);
    default: return -1;
  }
}

#ifndef VIMA_GEMM_LAB
// ------------------------------------------------------------------------------------------------ resident-K tiles
template <typename RT, int ACT>
int launch_resident_inst(const GemmDev& d, dim3 grid, hipStream_t st) {
  static PerDeviceOnce attr;   // per instantiation, per device
  {
    const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_resident_kernel<RT, ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, RT::NCH * RT::CHUNK); });
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((gemm_resident_kernel<RT, ACT>), grid, dim3(RT::THREADS), (size_t)d.res_nch * RT::CHUNK, st, d);
  return (int)hipGetLastError();
}
template <typename RT>
dim3 resident_geometry(GemmDev& d, const GemmArgs& a);
template <typename RT, int ACT>
int launch_resident_act(GemmDev d, const GemmArgs& a, hipStream_t st) {   // one activation only (the GEGLU pair: GELU)
  const dim3 grid = resident_geometry<RT>(d, a);
  return launch_resident_inst<RT, ACT>(d, grid, st);
}
template <typename RT>
int launch_resident_tile(GemmDev d, const GemmArgs& a, hipStream_t st) {
  const dim3 grid = resident_geometry<RT>(d, a);
  switch (a.act) {
    case ACT_NONE: return launch_resident_inst<RT, ACT_NONE>(d, grid, st);
    case ACT_RELU: return launch_resident_inst<RT, ACT_RELU>(d, grid, st);
    case ACT_GELU: return launch_resident_inst<RT, ACT_GELU>(d, grid, st);
    case ACT_QUICKGELU: return launch_resident_inst<RT, ACT_QUICKGELU>(d, grid, st);
    default: return (int)hipErrorInvalidValue;
  }
}
template <typename RT>
dim3 resident_geometry(GemmDev& d, const GemmArgs& a) {
  d.mtiles = (d.M + RT::BM - 1) / RT::BM;
  d.ntiles = (d.N + RT::BN - 1) / RT::BN;
  {   // chunk buffers: the option when set (2 .. what the LDS holds), else the default; never more than the problem has chunks
    int nch = gemm_res_nch(a.tune);
    if (nch <= 0) nch = RT::NCH_DEFAULT;
    nch = nch < 2 ? 2 : (nch > RT::NCH ? RT::NCH : nch);
    const int chunks = (a.K / 64 + RT::CS - 1) / RT::CS;
    if (nch > chunks) nch = chunks < 1 ? 1 : chunks;
    d.res_nch = nch;
  }
  d.dbg = a.tune ? a.tune->gemm_dbg : nullptr;
  return dim3((unsigned)(d.mtiles * d.ntiles), (unsigned)(a.batch > 0 ? a.batch : 1), 1);
}
// bf16 problems with a vector-aligned epilogue and no fp8 operands; < 0 = not taken. Tile (`force`: gemm_tile 10 / 11 / 12 = 32x32 /
// 64x32 / 64x64 whatever the grid): the smallest one whose grid still fits the chip once (`gemm_res_maxwg`, default 256 = one
// workgroup per CU; M <= 32 always takes 32x32) -- a lone 32x32 accumulator per wave is a dependent MFMA chain, so what counts is
// how many of them run side by side, not the operand bytes per FLOP (measured, profiles/r03_small_m.txt).
int launch_resident(const GemmDev& d, const GemmArgs& a, int force, hipStream_t st) {
  if (a.w8 || a.a8 || a.out8 || a.K % 64 != 0 || (a.N % 4 != 0 && !a.grp_col)) return -1;
  if (a.ssq_out && a.act != ACT_NONE) return -1;
  if (a.resT && a.act != ACT_NONE) return -1;   // the kernel applies the bf16 residual only without an activation (launch_t refuses the pair anyway)
  const long long nb = a.batch > 0 ? a.batch : 1;
  const long long maxwg = gemm_res_maxwg(a.tune);
  const long long m32 = (a.M + 31) / 32, m64 = (a.M + 63) / 64, n32 = (a.N + 31) / 32, n64 = (a.N + 63) / 64;
  int tile = 0;
  if (a.W2) {   // GEGLU pair: 32x32 for M <= 32, else 64x64 while its grid fits the chip once (the caller asks gemm_dual_ok() first)
    tile = a.M <= 32 ? 15 : (m64 * n64 <= maxwg ? 16 : 0);
    if (!tile || a.act != ACT_GELU) return -1;
    if (a.kernel_id) *a.kernel_id = tile * 1000 + (a.act + 1) * 10;
    return tile == 15 ? launch_resident_act<RT32D, ACT_GELU>(d, a, st) : launch_resident_act<RT64D, ACT_GELU>(d, a, st);
  }
  if (force) tile = force;
  else if (a.M <= 32) tile = (n32 * nb <= 4 * maxwg || a.grp_col) ? 10 : 0;
  else if (m32 * n32 * nb <= maxwg) tile = 10;
  else if (m64 * n32 * nb <= maxwg) tile = 11;
  else if (m64 * n64 * nb <= maxwg || a.grp_col) tile = 12;
  if (!tile) return -1;
  if (a.kernel_id) *a.kernel_id = tile * 1000 + (a.act + 1) * 10;
  switch (tile) {
    case 10: return launch_resident_tile<RT32>(d, a, st);
    case 11: return launch_resident_tile<RT64x32>(d, a, st);
    default: return launch_resident_tile<RT64>(d, a, st);
  }
}
#endif


// ------------------------------------------------------------------------------------------------ skinny (M <= 32)
#ifndef VIMA_GEMM_LAB
template <typename ST, int ACT>
int launch_skinny_inst(const GemmDev& d, dim3 grid, hipStream_t st) {
  static PerDeviceOnce attr;   // per instantiation, per device
  if (ST::SMEM > 48 * 1024) {
    const hipError_t e = attr.ensure([&] { return hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_kernel<ST, ACT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, ST::SMEM); });
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL((gemm_skinny_kernel<ST, ACT>), grid, dim3(ST::THREADS), (size_t)ST::SMEM, st, d);
  return (int)hipGetLastError();
}
template <typename ST>
int launch_skinny_act(const GemmDev& d, const GemmArgs& a, dim3 grid, hipStream_t st) {
  switch (a.act) {
    case ACT_NONE: return launch_skinny_inst<ST, ACT_NONE>(d, grid, st);
    case ACT_RELU: return launch_skinny_inst<ST, ACT_RELU>(d, grid, st);
    case ACT_GELU: return launch_skinny_inst<ST, ACT_GELU>(d, grid, st);
    case ACT_QUICKGELU: return launch_skinny_inst<ST, ACT_QUICKGELU>(d, grid, st);
    default: return (int)hipErrorInvalidValue;
  }
}
// bf16 problems of at most 32 rows with a vector-aligned epilogue and no fp8 operands; < 0 = not taken. The number of waves (= the K split)
// depends on K only.
int launch_skinny(GemmDev d, const GemmArgs& a, hipStream_t st) {
  if (a.M > 32 || a.w8 || a.a8 || a.out8 || a.K % 64 != 0 || a.N % 4 != 0 || a.grp_col || a.hm_D || a.pair32) return -1;
  if ((a.ssq_out || a.resT) && a.act != ACT_NONE) return -1;
  if (a.W2 && (a.act != ACT_GELU || a.K > 2048)) return -1;
  if ((long long)a.M * a.lda * 2 >= (1LL << 31) || (long long)a.N * a.ldw * 2 >= (1LL << 31)) return -1;   // buffer descriptors with 31-bit byte ranges
  if (a.W2 && ((long long)a.M * a.lda2 * 2 >= (1LL << 31) || (long long)a.N * a.ldw2 * 2 >= (1LL << 31))) return -1;
  // columns per workgroup: the narrowest of 32 / 16 / 8 that still leaves whole tiles and at most ~256 workgroups -- what bounds a launch is the
  // bytes ONE workgroup pulls (a CU streams ~24 GB/s), so N is spread over as many CUs as there are; partial sums for a downstream fused norm are
  // per 32 columns and need the full-width tile
  const long long nb = a.batch > 0 ? a.batch : 1;
  int cols = 32;
  static int force_cols = -1;
  if (force_cols < 0) force_cols = env_int("VIMA_SKINNY_COLS", 0);
  if (!a.ssq_out) {
    while (cols > 8 && a.N % (cols / 2) == 0 && (long long)(a.N / (cols / 2)) * nb <= 400) cols /= 2;
    if (force_cols == 8 || force_cols == 16 || force_cols == 32) cols = force_cols;
  }
  d.sk_cols = cols;
  d.dbg = a.tune ? a.tune->gemm_dbg : nullptr;
  const dim3 grid((unsigned)((a.N + cols - 1) / cols), (unsigned)nb, 1);
  const int ks = a.K / 16;
  if (a.kernel_id) *a.kernel_id = (a.W2 ? 18 : 17) * 1000 + (a.act + 1) * 10;
  if (a.W2) return ks <= 64 ? launch_skinny_inst<SkTile<4, 12, true>, ACT_GELU>(d, grid, st) : launch_skinny_inst<SkTile<8, 8, true>, ACT_GELU>(d, grid, st);
  if (ks <= 64) return launch_skinny_act<SkTile<4, 12, false>>(d, a, grid, st);
  if (ks <= 128) return launch_skinny_act<SkTile<8, 12, false>>(d, a, grid, st);
  return launch_skinny_act<SkTile<16, 12, false>>(d, a, grid, st);
}
#endif

// ------------------------------------------------------------------------------------------------ split-K
// With M = 8 .. 512 rows (batch 1 .. 32 decoder / prompt GEMMs) a 128x128 grid has 6 .. 100 workgroups, each walking
// its K dimension serially at the per-CU operand-path rate (~23 B/clk): 8 us at K = 768, 30 us at K = 3072, most CUs
// idle. Such problems are split along K into S ranges (pass 1: the same kernel, batched over the ranges, raw fp32
// partials) and finished by an elementwise pass that sums the partials in split order -- deterministic -- and applies
// the whole epilogue. The summation order differs from the single-pass kernels, i.e. results would depend (at fp32
// rounding level) on whether a problem was small enough to be split -- batch-composition and chunking invariance would
// only hold to bf16 tolerance. Measured gain: 35 -> 25 us at M = 8, K = 3072; nothing at K = 768 (a second launch costs
// as much as the saved slices); batch-1 step 6.2 -> 5.5 ms. It is therefore OPT-IN (option gemm_splitk / VIMA_GEMM_SPLITK).
struct SplitPlan { int S; int Ks; };
inline SplitPlan splitk_plan(const GemmArgs& a, bool is_bf16) {
  SplitPlan p{1, a.K};
  if (!gemm_splitk(a.tune) || a.w8) return p;
  const int bk = is_bf16 ? 64 : 32;
  if (a.batch > 1 || a.M <= 0 || a.N <= 0 || a.K % bk || a.N % 4 || a.ssq_out || a.rb > 0 || a.resT) return p;
  if (a.pair32 || a.hm_D || a.W2 || a.grp_col || a.rs_sum || a.sum_out) return p;   // forms whose epilogue the reduce pass does not have
  const long long tiles = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128);
  const int slices = a.K / bk;
  if (tiles >= 128 || slices < 24) return p;   // measured: the second pass only pays from K = 1536 (bf16) on
  const int want = (int)((256 + tiles - 1) / tiles);
  int best = 1;
  for (int s = 2; s <= want && s * 2 <= slices; ++s)
    if (slices % s == 0) best = s;
  if (best > 1) { p.S = best; p.Ks = slices / best * bk; }
  return p;
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_reduce_kernel(const float* __restrict__ part, int S, long long MN, int M, int N,
                                                          const float* bias, int act, const T* mul, int ldmul,
                                                          const float* res, int ldres, float* out32, int ld32, T* outT, int ldT,
                                                          const float* rs_ssq, int rs_parts, float rs_invk, float rs_eps) {
  const int n4 = N >> 2;
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long long)M * n4) return;
  const int m = (int)(i / n4), n = (int)(i % n4) * 4;
  const float* q = part + (long long)m * N + n;
  float4 v = load4(q);
  for (int s = 1; s < S; ++s) {   // fixed order: deterministic
    const float4 t = load4(q + s * MN);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  if (rs_ssq) {
    const float r = rms_row_scale(rs_ssq, rs_parts, m, rs_invk, rs_eps);
    v.x *= r; v.y *= r; v.z *= r; v.w *= r;
  }
  if (bias) { const float4 b = load4(bias + n); v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w; }
  if (act != ACT_NONE) { v.x = apply_act_t<T>(v.x, act); v.y = apply_act_t<T>(v.y, act); v.z = apply_act_t<T>(v.z, act); v.w = apply_act_t<T>(v.w, act); }
  if (mul) { const float4 g = load4(mul + (long long)m * ldmul + n); v.x *= g.x; v.y *= g.y; v.z *= g.z; v.w *= g.w; }
  if (res) { const float4 r4 = load4(res + (long long)m * ldres + n); v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w; }
  if (out32) store4(out32 + (long long)m * ld32 + n, v);
  if (outT) store4(outT + (long long)m * ldT + n, v);
}

template <typename T>
int launch_t(const GemmArgs& a, hipStream_t st) {
  constexpr int BK = 128 / (int)sizeof(T);   // K granularity of the widest K-slice (TileS / TileL)
  if (a.M <= 0 || a.N <= 0) return 0;
  if (a.K <= 0 || a.K % BK != 0) return (int)hipErrorInvalidValue;
  // LDS-DMA reads 16-B chunks: rows must start 16-B aligned
  const size_t es = sizeof(T);
  const size_t esw = a.w8 ? 1 : es;   // fp8 weights: ldw / bsW count bytes
  if (a.w8 && (sizeof(T) != 2 || !a.wscale || a.N % 4 != 0 || a.K % 64 != 0 || a.batch > 1 || !aligned_to(a.wscale, 16)))
    return (int)hipErrorInvalidValue;
  if (!aligned_to(a.A, 16) || !aligned_to(a.W, 16) || (a.lda * es) % 16 || (a.ldw * esw) % 16 ||
      (a.bsA * es) % 16 || (a.bsW * esw) % 16)
    return (int)hipErrorInvalidValue;
  if (a.splitk_ws) {   // underfilled grid: two deterministic passes (see splitk_plan)
    const SplitPlan sp = splitk_plan(a, sizeof(T) == 2);
    const long long MN = (long long)a.M * a.N;
    const bool ok4 = (!a.bias || aligned_to(a.bias, 16)) && (!a.mul || (aligned_to(a.mul, 4 * es) && a.ldmul % 4 == 0)) &&
                     (!a.res || (aligned_to(a.res, 16) && a.ldres % 4 == 0)) && (!a.out32 || (aligned_to(a.out32, 16) && a.ld32 % 4 == 0)) &&
                     (!a.outT || (aligned_to(a.outT, 4 * es) && a.ldT % 4 == 0));
    if (sp.S > 1 && ok4 && a.splitk_ws_bytes >= (size_t)sp.S * MN * sizeof(float) && aligned_to(a.splitk_ws, 16)) {
      GemmArgs p1;
      p1.A = a.A; p1.lda = a.lda; p1.W = a.W; p1.ldw = a.ldw; p1.M = a.M; p1.N = a.N; p1.K = sp.Ks;
      if (a.kernel_id) *a.kernel_id = 8000 + (a.act + 1) * 10;
      p1.tune = a.tune; p1.batch = sp.S; p1.bsA = sp.Ks; p1.bsW = sp.Ks; p1.out32 = a.splitk_ws; p1.ld32 = a.N; p1.bs32 = MN;
      if (int e = launch_t<T>(p1, st)) return e;
      const long long work = (long long)a.M * (a.N / 4);
      hipLaunchKernelGGL(gemm_reduce_kernel<T>, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, a.splitk_ws, sp.S, MN, a.M,
                         a.N, a.bias, a.act, reinterpret_cast<const T*>(a.mul), a.ldmul, a.res, a.ldres, a.out32, a.ld32,
                         reinterpret_cast<T*>(a.outT), a.ldT, a.rs_ssq, a.rs_parts, a.rs_invk, a.rs_eps);
      return (int)hipGetLastError();
    }
  }
  if (a.grp_col && (sizeof(T) != 2 || a.w8 || a.a8 || a.act != ACT_NONE || a.mul || a.res || a.resT || a.outT || a.out8 || a.ssq_out || a.rs_ssq ||
                    a.rb > 0 || !a.out32 || a.K % 64 != 0 || !gemm_grouped_ok(a.tune)))
    return (int)hipErrorInvalidValue;
  if (a.W2 && (sizeof(T) != 2 || a.w8 || a.a8 || a.mul || a.res || a.resT || a.out32 || a.out8 || a.ssq_out || (a.rs_ssq && !a.rs_sum) || a.rb > 0 || a.batch > 1 ||
               a.grp_col || !a.outT || !a.A2 || a.K % 64 != 0 || a.N % 4 != 0 || !aligned_to(a.A2, 16) || !aligned_to(a.W2, 16) || (a.lda2 * es) % 16 ||
               (a.ldw2 * es) % 16 || !gemm_dual_ok(a.tune, a.M, a.N)))
    return (int)hipErrorInvalidValue;
  GemmDev d;
  d.grp_col = a.grp_col;
  d.A2 = a.A2; d.W2 = a.W2; d.lda2 = a.lda2; d.ldw2 = a.ldw2;
  d.A = a.A; d.W = a.W; d.M = a.M; d.N = a.N; d.K = a.K; d.lda = a.lda; d.ldw = a.ldw;
  d.bsA = a.bsA; d.bsW = a.bsW; d.bsBias = a.bsBias; d.bsMul = a.bsMul; d.bsRes = a.bsRes; d.bs32 = a.bs32; d.bsT = a.bsT;
  d.bias = a.bias; d.act = a.act; d.mul = a.mul; d.ldmul = a.ldmul; d.res = a.res; d.ldres = a.ldres;
  d.resT = a.resT; d.ldresT = a.ldresT;
  if (a.resT && (a.res || a.batch > 1 || a.act != ACT_NONE)) return (int)hipErrorInvalidValue;
  d.out32 = a.out32; d.ld32 = a.ld32; d.outT = a.outT; d.ldT = a.ldT;
  d.rb = a.rb; d.s_hi = a.s_hi; d.s_lo = a.s_lo; d.ro = a.ro;
  d.hm_D = a.hm_D; d.hm_L = a.hm_L;
  d.outT_lo = a.outT_lo; d.ldT_lo = a.ldT_lo; d.split_n = a.split_n;
  if (a.hm_D && (sizeof(T) != 2 || !a.outT || a.out32 || a.mul || a.res || a.resT || a.out8 || a.ssq_out || a.rb > 0 || a.batch > 1 || a.grp_col || a.W2 ||
                 !gemm_headmajor_ok(a.tune, a.M, a.N, a.K, a.lda, a.ldw, a.hm_D, a.hm_L, a.a8)))
    return (int)hipErrorInvalidValue;
  d.ssq_out = a.ssq_out; d.rs_ssq = a.rs_ssq; d.rs_parts = a.rs_parts; d.rs_invk = a.rs_invk; d.rs_eps = a.rs_eps;
  d.sum_out = a.sum_out; d.rs_sum = a.rs_sum; d.rs_c = a.rs_c;
  // fused LayerNorm: partial sums ride with the partial sums of squares of an fp32-output producer; the consumers are the two GEGLU-pair forms
  if (a.sum_out && (sizeof(T) != 2 || !a.ssq_out || !a.out32 || !gemm_lnfold_producer_ok(a.tune, a.M, a.N))) return (int)hipErrorInvalidValue;
  if (a.rs_sum && (sizeof(T) != 2 || !a.rs_ssq || !a.rs_c || !(a.pair32 || a.W2) || !aligned_to(a.rs_c, 16))) return (int)hipErrorInvalidValue;
  d.wscale = a.w8 ? a.wscale : nullptr;
  d.ascale = a.a8 ? a.ascale : 1.0f;
  d.out8 = a.out8; d.ld8 = a.ld8; d.out8_inv = a.out8_inv;
  if (a.ssq_out && ((!a.out32 && !a.outT) || a.batch > 1 || a.N % 32 != 0)) return (int)hipErrorInvalidValue;
  if (a.rs_ssq && a.rs_parts <= 0) return (int)hipErrorInvalidValue;
  d.mtiles = d.ntiles = 0;
  bool v = (a.N % 4 == 0);
  if (a.bias) v = v && aligned_to(a.bias, 16) && (a.bsBias % 4 == 0);
  if (a.mul) v = v && aligned_to(a.mul, 4 * es) && (a.ldmul % 4 == 0) && (a.bsMul % 4 == 0);
  if (a.res) v = v && aligned_to(a.res, 16) && (a.ldres % 4 == 0) && (a.bsRes % 4 == 0);
  if (a.resT) v = v && aligned_to(a.resT, 4 * es) && (a.ldresT % 4 == 0);
  if (a.out32) v = v && aligned_to(a.out32, 16) && (a.ld32 % 4 == 0) && (a.bs32 % 4 == 0);
  if (a.outT) v = v && aligned_to(a.outT, 4 * es) && (a.ldT % 4 == 0) && (a.bsT % 4 == 0);
  if (a.w8 && !v) return (int)hipErrorInvalidValue;   // fp8 weights need the vector epilogue (16-byte aligned outputs)
  // column-split output: the VECTOR epilogues of the one-tile-per-workgroup / resident / skinny kernels only (the scalar epilogue and the persistent
  // kernels' epilogues do not know split_n: a launch that could reach them is refused, never mis-stored)
  if (a.split_n && (sizeof(T) != 2 || !v || !a.outT || !a.outT_lo || a.out32 || a.out8 || a.mul || a.res || a.resT || a.ssq_out || a.hm_D || a.pair32 || a.W2 || a.grp_col ||
                    a.batch > 1 || a.split_n <= 0 || a.split_n >= a.N || a.split_n % 128 != 0 || a.N % 8 != 0 || a.ldT_lo % 8 != 0 || !aligned_to(a.outT_lo, 16) ||
                    a.w8 || a.a8 || gemm_splitk(a.tune)))
    return (int)hipErrorInvalidValue;
  d.wide8 = 0;
  if constexpr (sizeof(T) == 2) {
    d.wide8 = (v && (a.outT || a.out8) && !a.out32 && a.N % 8 == 0 && a.ldT % 8 == 0 && a.bsT % 8 == 0 && aligned_to(a.outT, 16) &&
               (!a.out8 || (a.ld8 % 8 == 0 && aligned_to(a.out8, 8))) &&
               (!a.mul || (a.ldmul % 8 == 0 && a.bsMul % 8 == 0 && aligned_to(a.mul, 16))) &&
               (!a.resT || (a.ldresT % 8 == 0 && aligned_to(a.resT, 16)))) ? 1 : 0;
    // RMS partials without the fp32 stream exist only in the 8-column layout (statistics of the stored bf16 values)
    if (a.ssq_out && !a.out32 && !d.wide8) return (int)hipErrorInvalidValue;
  }
#ifndef VIMA_GEMM_LAB
  if constexpr (sizeof(T) == 2) {
    if (a.grp_col || a.W2) {
      if (a.W2 && !v) return (int)hipErrorInvalidValue;
      if (a.W2 && a.M <= 32 && gemm_skinny(a.tune)) {
        const int e = launch_skinny(d, a, st);
        if (e >= 0) return e;
      }
      const int e = launch_resident(d, a, 0, st);
      return e >= 0 ? e : (int)hipErrorInvalidValue;
    }
    if (a.pair32) {   // GEGLU pair over block-interleaved weights: the 128x128 ring tile's pair epilogue (the caller asks gemm_pair_ok() first)
      if (!v || a.w8 || a.a8 || a.act != ACT_GELU || a.mul || a.res || a.resT || a.out32 || a.out8 || a.ssq_out || (a.rs_ssq && !a.rs_sum) || a.rb > 0 ||
          a.batch > 1 || a.hm_D || !a.outT || a.N % TileS::BN != 0 || a.K % 64 != 0 || a.ldT % 8 != 0 || !aligned_to(a.outT, 16) ||
          !gemm_pair_ok(a.tune, a.M, a.N / 2, a.K))
        return (int)hipErrorInvalidValue;
      if (gemm_pair_large(a.tune, a.M, a.N / 2, a.K) && (long long)a.M * a.lda * 2 < (1LL << 32) && (long long)a.N * a.ldw * 2 < (1LL << 32)) {
        d.wide8 = 1;
        const int e = launch_pp(d, a, st);   // >= 160 full 256x256 tiles of the interleaved [M, 2 Nout] problem: the persistent kernel's pair epilogue
        if (e >= 0) { if (a.kernel_id) *a.kernel_id = 1000 + (a.act + 1) * 10 + 6; return e; }
      }
      if (a.kernel_id) *a.kernel_id = 5000 + (a.act + 1) * 10;
      return launch_tile<T, TileS, true>(d, a, v, st);
    }
  }
#endif
  if (a.pair32) return (int)hipErrorInvalidValue;
  if constexpr (sizeof(T) == 2) {
    // TileL when the 256x256 grid still fills the chip (>= ~1 workgroup per CU) and padding waste is small
    const long long mt = (a.M + 255) / 256, nt = (a.N + 255) / 256;
    const double waste = (double)(mt * 256) * (double)(nt * 256) / ((double)a.M * (double)a.N);
    bool large = v && (mt * nt * (a.batch > 0 ? a.batch : 1) >= 160) && waste < 1.15;   // measured: 192 tiles of 256x256 beat 768 of 128x128 by 10-28 %
    if (gemm_tile(a.tune) == 1) large = false;
    if (gemm_tile(a.tune) >= 2 && gemm_tile(a.tune) < 7) large = v;
    if (gemm_tile(a.tune) >= 7) large = false;
    if (a.split_n) large = false;   // after every override: the column-split output exists in the one-tile-per-workgroup / resident / skinny epilogues only
    if (large && gemm_q4(a.tune) && gemm_tile(a.tune) == 0 && gemm_persist(a.tune) && gemm_raster(a.tune) == 0 && gemm_epi(a.tune)) {
      // gemm_q4: 1 = the 256 x 384 tile wherever it fits, 2 = the 128 x 384 tile wherever it fits, 3 = the 128 x 384 tile where it needs fewer ROUNDS of the chip than
      // the 256 x 256 tiling. A round of 128 x 384 tiles is 0.75 of a round of 256 x 256 tiles in FLOPs and measures ~0.8 of it in time (12 MFMAs per phase and barrier
      // instead of 24); the margin keeps the choice away from ties. M = 16 384 (batch 32): N = 768 one round instead of one of 192 tiles, N = 1536 / 2304 2 / 3 rounds
      // instead of 2 / 3 larger ones; the batch-256 shapes (M = 131 072, 81 920) never qualify. The choice depends on the SHAPE only, and the tile is bit-identical anyway.
      int mode = gemm_q4(a.tune);
      // 6 (the default since the end of round 6) = the 256 x 384 tile for GEMMs of at least 32 768 rows. Kernel for kernel the two tilings take the same time
      // (power-bound: DESIGN.md 4.2c) -- but with the model's TWO streams the four-wave kernel's 7 % fewer busy cycles are left to whatever runs beside it:
      // wall clock of the headline step 53.63 -> 52.89 ms, 1024-token prompt 109.8 -> 108.2, T = 8 65.6 -> 65.0, batch 128 27.85 -> 27.58 (alternating, same box:
      // profiles/r06_q4_wall_clock.txt); below that size it loses (batch 32 +5 %, the env steps +1 %, VIMA-20M batch 32 +3 %), hence the row threshold.
      if (mode == 6) mode = a.M >= 32768 ? 1 : 0;
      if (mode == 4) mode = a.N >= 1536 ? 1 : 0;   // 4 / 5 (lab): the 256 x 384 tile for the wide-N GEMMs only / for N = 768 only
      if (mode == 5) mode = a.N == 768 ? 1 : 0;
      if (mode == 3) {
        mode = 0;
        if (a.M % 128 == 0 && a.N % 384 == 0 && a.M % 256 == 0 && a.N % 256 == 0) {
          const long long ncu = 256;
          const long long r_pp = ((long long)(a.M / 256) * (a.N / 256) + ncu - 1) / ncu;
          const long long r_h = ((long long)(a.M / 128) * (a.N / 384) + ncu - 1) / ncu;
          if (r_h * 80 <= r_pp * 93) mode = 2;
        }
      }
      if (mode) {
        const bool half = mode == 2;
        const int e = half ? launch_q4<1>(d, a, st) : launch_q4<2>(d, a, st);
        if (e >= 0) { if (a.kernel_id) *a.kernel_id = (half ? 20000 : 19000) + (a.act + 1) * 10 + (a.hm_D ? 5 : (a.resT ? 4 : 1)); return e; }
      }
    }
    if (large && gemm_wide(a.tune) && gemm_tile(a.tune) == 0 && gemm_persist(a.tune) && gemm_raster(a.tune) == 0 && gemm_epi(a.tune)) {
      const int e = launch_wide(d, a, st);
      if (e >= 0) { if (a.kernel_id) *a.kernel_id = 3000 + (a.act + 1) * 10 + (a.resT ? 4 : 1); return e; }
    }
    if (large && (gemm_tile(a.tune) == 0 || gemm_tile(a.tune) == 2) && gemm_persist(a.tune) && a.batch <= 1 &&
        a.K >= 2 * 64 && gemm_raster(a.tune) == 0 && gemm_epi(a.tune) &&
        a.M % TileL::BM == 0 && a.N % TileL::BN == 0 && (long long)a.M * a.lda * 2 < (1LL << 32) &&
        (long long)a.N * a.ldw * (long long)esw < (1LL << 32)) {
      const int epi_id = a.hm_D ? 5 : (a.resT ? 4 : ((a.res || a.out32) ? 3 : (a.mul ? 2 : 1)));
      if (gemm_pp(a.tune) || a.a8) {
        const int e = launch_pp(d, a, st);
        if (e >= 0) { if (a.kernel_id) *a.kernel_id = (a.a8 ? 9000 : 1000) + (a.act + 1) * 10 + epi_id; return e; }
      }
      if (a.a8) return (int)hipErrorInvalidValue;   // fp8 activations exist on the ping-pong kernel only
      const int e = launch_persistent(d, a, st);
      if (e >= 0) { if (a.kernel_id) *a.kernel_id = 2000 + (a.act + 1) * 10 + epi_id; return e; }
    }
    if (a.a8 || a.hm_D) return (int)hipErrorInvalidValue;   // (head-major output exists in the persistent kernels' epilogue only)
    if (large) { if (a.kernel_id) *a.kernel_id = 4000 + (a.act + 1) * 10; return launch_tile<T, TileL, true>(d, a, v, st); }
  }
  if (a.a8 || a.hm_D) return (int)hipErrorInvalidValue;
#ifdef VIMA_GEMM_LAB
  return (int)hipErrorInvalidValue;
#else
  if constexpr (sizeof(T) == 2) {
    // Underfilled 128x128 grids (batch 1 .. 32: M = 8 .. 512) are bound by how fast ONE workgroup walks its K dimension
    // (a 32-KiB slice per ~1400 clocks, most of the A tile being padding rows): smaller tiles move fewer bytes per slice
    // and spread the problem over more CUs. Every tile shape accumulates K in the same order, so results do not depend
    // on the choice.
    const long long t128 = (long long)((a.M + 127) / 128) * ((a.N + 127) / 128) * (a.batch > 0 ? a.batch : 1);
    {   // the same class with (almost) the whole K extent in flight (gemm_small.inc); bit-identical to the ring tiles
      const int gt = gemm_tile(a.tune);
      const int force = (gt >= 10 && gt <= 12) ? gt : 0;
      if (v && gt == 0 && a.M <= 32 && gemm_small(a.tune) && gemm_resident(a.tune) && gemm_skinny(a.tune)) {   // at most 32 rows: K split over the waves
        const int e = launch_skinny(d, a, st);
        if (e >= 0) return e;
      }
      if (v && (force || (gt == 0 && gemm_small(a.tune) && gemm_resident(a.tune) && t128 < 128))) {
        const int e = launch_resident(d, a, force, st);
        if (e >= 0) return e;
      }
    }
    if (v && gemm_tile(a.tune) == 0 && gemm_small(a.tune) && t128 < 128) {
      if (a.kernel_id) *a.kernel_id = (a.M <= 32 ? 7000 : 6000) + (a.act + 1) * 10;
      if (a.M <= 32) return launch_tile<T, TileXS, true>(d, a, v, st);
      return launch_tile<T, Tile64, true>(d, a, v, st);
    }
    if (gemm_tile(a.tune) == 13 && v) return launch_tile<T, Tile64x128, true>(d, a, v, st);
    if (gemm_tile(a.tune) == 14 && v) return launch_tile<T, Tile128x64, true>(d, a, v, st);
    if (gemm_tile(a.tune) == 7 && v) return launch_tile<T, TileXS, true>(d, a, v, st);
    if (gemm_tile(a.tune) == 8 && v) return launch_tile<T, Tile64, true>(d, a, v, st);
  }
  if (a.kernel_id) *a.kernel_id = 5000 + (a.act + 1) * 10;
  if (gemm_variant(a.tune) == 1 || a.w8) return launch_tile<T, TileS, true>(d, a, v, st);
  return launch_tile<T, TileS, false>(d, a, v, st);
#endif
}

}  // namespace

int launch_gemm(const GemmArgs& a, bool is_bf16, hipStream_t st) {
#ifdef VIMA_GEMM_LAB
  return is_bf16 ? launch_t<bf16_t>(a, st) : (int)hipErrorInvalidValue;
#else
  return is_bf16 ? launch_t<bf16_t>(a, st) : launch_t<float>(a, st);
#endif
}
size_t gemm_splitk_bytes(const GemmArgs& a, bool is_bf16) {
  const SplitPlan sp = splitk_plan(a, is_bf16);
  return sp.S > 1 ? (size_t)sp.S * a.M * a.N * sizeof(float) : 0;
}
int gemm_splitk_enabled(const Tuning* t) { return gemm_splitk(t); }
int gemm_k_multiple(bool is_bf16) { return is_bf16 ? 64 : 32; }
// GEGLU pair over block-interleaved weights (GemmArgs::pair32): for [M, Nout] outputs that would otherwise be two ring-tile launches --
// neither the dual-accumulator resident form (small grids) nor the persistent 256x256 kernels (large ones)
// the interleaved [M, 2 Nout] problem fills the chip with full 256x256 tiles: the pair runs on gemm_pp_kernel<ACT_GELU, 6>
int gemm_pair_large(const Tuning* t, long long M, long long Nout, long long K) {
#ifdef VIMA_GEMM_LAB
  return 0;
#else
  if (M <= 0 || Nout <= 0 || M % 256 != 0 || Nout % 128 != 0 || K < 128 || (K / 64) % 2 != 0) return 0;
  if (gemm_tile(t) != 0 || !gemm_persist(t) || !gemm_pp(t) || gemm_raster(t) != 0 || !gemm_epi(t) || gemm_wide(t) || gemm_variant(t) != 1) return 0;
  return (M / 256) * (2 * Nout / 256) >= 160 ? 1 : 0;
#endif
}
int gemm_pair_ok(const Tuning* t, long long M, long long Nout, long long K) {
#ifdef VIMA_GEMM_LAB
  return 0;
#else
  if (M <= 0 || Nout <= 0 || Nout % 64 != 0 || K < 64 || K % 64 != 0) return 0;
  if (gemm_tile(t) != 0 || gemm_variant(t) != 1 || !gemm_epi(t) || gemm_raster(t) != 0) return 0;
  if (gemm_dual_ok(t, (int)M, (int)Nout)) return 0;
  if (gemm_pair_large(t, M, Nout, K)) return 1;
  if (((M + 255) / 256) * ((Nout + 255) / 256) >= 160) return 0;   // `large` without the pair epilogue's shape conditions: the GELU GEMM takes the persistent kernel's gate epilogue
  if (gemm_small(t) && ((M + 127) / 128) * ((Nout + 127) / 128) < 128) return 0;   // small grids: 64x64 ring tiles / resident kernel
  return 1;
#endif
}
// Fused-LayerNorm PRODUCER (GemmArgs::sum_out): the partial sums exist in the resident kernel's epilogue and in the ring tiles' LDS epilogue, not
// in the persistent 256x256 kernels' -- i.e. for every [M, N] output that does not take the `large` path
int gemm_lnfold_producer_ok(const Tuning* t, long long M, long long N) {
#ifdef VIMA_GEMM_LAB
  return 0;
#else
  if (M <= 0 || N <= 0 || N % 32 != 0) return 0;
  const int gt = gemm_tile(t);
  if (gt >= 2 && gt < 7) return 0;
  if (gt == 0 && ((M + 255) / 256) * ((N + 255) / 256) >= 160) return 0;
  return 1;
#endif
}
int gemm_dual_ok(const Tuning* t, int M, int N) {
  if (!gemm_grouped_ok(t) || M <= 0 || N <= 0 || N % 4 != 0) return 0;
  if (M <= 32) return 1;
  return (long long)((M + 63) / 64) * ((N + 63) / 64) <= gemm_res_maxwg(t) ? 1 : 0;
}
// Head-major output (GemmArgs::hm_D / hm_L) exists in the persistent 256x256 kernels' bf16-output epilogue: the problem must take that path
// (same conditions as launch_t's: the large-tile rule, full tiles, the persistent / raster / epilogue knobs, 32-bit LDS-DMA offsets; fp8 operands: gemm_a8_ok),
// a batch of hm_L rows must be whole tiles and a head a power of two >= 8 columns.
int gemm_headmajor_ok(const Tuning* t, long long M, long long N, long long K, long long lda, long long ldw, int hm_D, int hm_L, int a8) {
  if (hm_D < 8 || (hm_D & (hm_D - 1)) || hm_L <= 0 || hm_L % TileL::BM != 0 || M % hm_L != 0 || N % hm_D != 0) return 0;
  if (gemm_wide(t)) return 0;   // the opt-in 256x384 kernel has its own epilogue
  if (a8) return gemm_a8_ok(t, M, N, K, lda, ldw);
  if (M <= 0 || N <= 0 || K < 128 || K % 64 != 0 || M % TileL::BM != 0 || N % TileL::BN != 0 || N % 8 != 0 || lda % 8 != 0 || ldw % 8 != 0) return 0;
  const int gt = gemm_tile(t);
  if (!(gt == 0 || gt == 2) || !gemm_persist(t) || gemm_raster(t) != 0 || !gemm_epi(t)) return 0;
  if (gt == 0 && (M / TileL::BM) * (N / TileL::BN) < 160) return 0;
  if (M * lda * 2 >= (1LL << 32) || N * ldw * 2 >= (1LL << 32)) return 0;
  return 1;
}
// fp8 ACTIVATIONS (GemmArgs::a8) exist on gemm_pp_kernel<.., F8> only: mirrors, condition for condition, the path launch_t takes
// for an a8 problem (anything else makes launch_gemm return hipErrorInvalidValue), so that callers can decide BEFORE quantising.
int gemm_a8_ok(const Tuning* t, long long M, long long N, long long K, long long lda, long long ldw) {
  if (M <= 0 || N <= 0 || K < 256 || K % 256 != 0 || M % TileL::BM != 0 || N % TileL::BN != 0) return 0;
  if (N % 8 != 0 || lda % 16 != 0 || ldw % 16 != 0) return 0;
  const int gt = gemm_tile(t);
  if (!(gt == 0 || gt == 2) || !gemm_persist(t) || gemm_raster(t) != 0 || !gemm_epi(t)) return 0;
  if (gt == 0 && (M / TileL::BM) * (N / TileL::BN) < 160) return 0;        // `large`: the 256x256 grid fills the chip
  if (M * lda * 2 >= (1LL << 32) || N * ldw >= (1LL << 32)) return 0;      // 32-bit LDS-DMA offsets (launch_t prices A at 2 bytes)
  return 1;
}
int gemm_grouped_ok(const Tuning* t) {
#ifdef VIMA_GEMM_LAB
  return 0;
#else
  return gemm_tile(t) == 0 && gemm_small(t) && gemm_resident(t);
#endif
}

}  // namespace vima
