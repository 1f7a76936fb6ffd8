// GEMM lab (experiment harness, not part of the product): times the GEMM kernels of vima_amd/csrc/gemm.hip on random
// bf16 data WITHOUT python / torch (a fresh GPU box pays 1-2 minutes for the first `import torch`), compares kernel
// variants bit for bit and against a host fp64 reference on sampled entries, and prints per-tile phase stamps.
//   build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVIMA_GEMM_LAB -I vima_amd/csrc -o scripts/micro/gemm_lab scripts/micro/gemm_lab.hip
//   run:   scripts/micro/gemm_lab M N K [epi=1|3|4] [act=0..3] [rounds=5] [variants=persist,pp]
// Variants: persist (gemm_persistent_kernel), pp (gemm_pp_kernel), tile (one 256x256 tile per workgroup), wide (256x384)
#include "../../vima_amd/csrc/gemm.hip"

#include <algorithm>
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static uint32_t g_rng = 12345u;
static inline uint32_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 17; g_rng ^= g_rng << 5; return g_rng; }
static inline float urand() { return (float)(rnd() >> 8) * (2.0f / 16777216.0f) - 1.0f; }   // uniform [-1, 1)
static inline uint16_t f2bf(float f) {
  uint32_t u; memcpy(&u, &f, 4);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
static inline float bf2f(uint16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }

__global__ void fill_bf16(uint16_t* p, long long n, uint32_t seed, float scale) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)(i * 2654435761u) ^ seed;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5; x *= 0x9E3779B1u; x ^= x >> 15;
    const float f = ((float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f) * scale;
    uint32_t u = __float_as_uint(f);
    u += 0x7fffu + ((u >> 16) & 1u);
    p[i] = (uint16_t)(u >> 16);
  }
}

__global__ void fill_f32(float* p, long long n, uint32_t seed) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    uint32_t x = (uint32_t)(i * 2654435761u) ^ seed;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5; x *= 0x9E3779B1u; x ^= x >> 15;
    p[i] = (float)(x >> 8) * (2.0f / 16777216.0f) - 1.0f;
  }
}

struct Variant { std::string name; vima::Tuning t; };

int main(int argc, char** argv) {
  if (argc < 4) { printf("usage: gemm_lab M N K [epi] [act] [rounds] [variants]\n"); return 1; }
  const int M = atoi(argv[1]), N = atoi(argv[2]), K = atoi(argv[3]);
  const int epi = argc > 4 ? atoi(argv[4]) : 1;
  const int act = argc > 5 ? atoi(argv[5]) : 0;
  const int rounds = argc > 6 ? atoi(argv[6]) : 5;
  const std::string vlist = argc > 7 ? argv[7] : "persist,pp";
  const bool stamps = getenv("STAMPS") != nullptr;
  std::vector<Variant> vars;
  {
    size_t pos = 0;
    while (pos <= vlist.size()) {
      size_t c = vlist.find(',', pos);
      if (c == std::string::npos) c = vlist.size();
      const std::string nm = vlist.substr(pos, c - pos);
      pos = c + 1;
      if (nm.empty()) continue;
      Variant v; v.name = nm;
      v.t.gemm_variant = 1; v.t.gemm_tile = 0; v.t.gemm_raster = 0; v.t.gemm_epi = 1; v.t.gemm_persist = 1; v.t.gemm_small = 1;
      v.t.gemm_wide = 0; v.t.gemm_pp = 0; v.t.gemm_splitk = 0; v.t.gemm_q4 = 0;
      if (nm == "persist") {}
      else if (nm == "pp") v.t.gemm_pp = 1;
      else if (nm == "tile") v.t.gemm_persist = 0;
      else if (nm == "fp8") v.t.gemm_pp = 1;     // fp8 e4m3 A and W, v_mfma_scale_f32_32x32x64_f8f6f4 (own operands / reference)
      else if (nm == "wide") v.t.gemm_wide = 1;
      else if (nm == "q4") { v.t.gemm_pp = 1; v.t.gemm_q4 = 1; }   // 256x384, four waves (one per SIMD)
      else if (nm == "q4h") { v.t.gemm_pp = 1; v.t.gemm_q4 = 2; }  // 128x384, four waves
      else { printf("unknown variant %s\n", nm.c_str()); return 1; }
      vars.push_back(v);
    }
  }
  uint16_t *A, *W, *res0;
  float* bias = nullptr;
  CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&W, (size_t)N * K * 2)); CK(hipMalloc(&res0, (size_t)M * N * 4));
  fill_bf16<<<2048, 256>>>(A, (long long)M * K, 0x1234567u, 1.0f);
  fill_bf16<<<2048, 256>>>(W, (long long)N * K, 0x7654321u, 1.0f);
  if (epi == 3) fill_f32<<<2048, 256>>>((float*)res0, (long long)M * N, 0x2468aceu);       // fp32 residual stream
  else fill_bf16<<<2048, 256>>>(res0, (long long)M * N * 2, 0x2468aceu, 1.0f);             // bf16 residual stream / gate
  if (getenv("BIAS")) {
    CK(hipMalloc(&bias, (size_t)N * 4));
    std::vector<float> hb(N);
    for (int i = 0; i < N; ++i) hb[i] = urand();
    CK(hipMemcpy(bias, hb.data(), (size_t)N * 4, hipMemcpyHostToDevice));
  }
  const size_t out_bytes = (size_t)M * N * (epi == 3 ? 4 : 2);
  std::vector<void*> outs(vars.size());
  for (size_t i = 0; i < vars.size(); ++i) CK(hipMalloc(&outs[i], out_bytes));
  float* ssq = nullptr;
  if (getenv("SSQ")) CK(hipMalloc(&ssq, (size_t)M * (N / 32) * 4));
  long long* dbg = nullptr;
  const int nblk = ((M + 255) / 256 + 7) / 8 * 8 * ((N + 255) / 256);
  if (stamps) { CK(hipMalloc(&dbg, (size_t)nblk * 40 * 8)); }
  CK(hipDeviceSynchronize());

  // fp8 operands: random finite e4m3 bytes (exponent field < 15 or mantissa < 7), per-channel weight scales, one activation scale
  uint8_t *A8 = nullptr, *W8 = nullptr;
  float* wsc = nullptr;
  const float asc = 0.037f;
  bool any8 = false;
  for (auto& v : vars) any8 |= v.name == "fp8";
  std::vector<uint8_t> hA8, hW8;
  std::vector<float> hws;
  if (any8) {
    hA8.resize((size_t)M * K); hW8.resize((size_t)N * K); hws.resize(N);
    auto fill8 = [&](std::vector<uint8_t>& v) {
      for (auto& b : v) { uint8_t x = (uint8_t)(rnd() >> 11); if ((x & 0x7f) == 0x7f) x ^= 1; if ((x & 0x78) == 0x78) x &= 0xbf; b = x; }   // |x| <= 240: no NaN
    };
    fill8(hA8); fill8(hW8);
    for (auto& x : hws) x = 0.01f + 0.02f * (float)(rnd() >> 8) / 16777216.0f;
    CK(hipMalloc(&A8, hA8.size())); CK(hipMalloc(&W8, hW8.size())); CK(hipMalloc(&wsc, (size_t)N * 4));
    CK(hipMemcpy(A8, hA8.data(), hA8.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(W8, hW8.data(), hW8.size(), hipMemcpyHostToDevice));
    CK(hipMemcpy(wsc, hws.data(), (size_t)N * 4, hipMemcpyHostToDevice));
  }
  auto make_args = [&](size_t vi) {
    vima::GemmArgs a;
    a.A = A; a.W = W; a.M = M; a.N = N; a.K = K; a.lda = K; a.ldw = K; a.bias = bias; a.act = act;
    if (vars[vi].name == "fp8") { a.A = A8; a.W = W8; a.a8 = 1; a.w8 = 1; a.wscale = wsc; a.ascale = asc; }
    a.tune = &vars[vi].t;
    if (epi == 1) { a.outT = outs[vi]; a.ldT = N; }
    else if (epi == 4) { a.outT = outs[vi]; a.ldT = N; a.resT = outs[vi]; a.ldresT = N; a.ssq_out = ssq; }   // in place, like the model's stream (SSQ=1: + RMS partials)
    else if (epi == 3) { a.out32 = (float*)outs[vi]; a.ld32 = N; if (!getenv("NORES3")) { a.res = (float*)outs[vi]; a.ldres = N; } }   // NORES3=1: fp32 output without a residual
    else if (epi == 2) { a.outT = outs[vi]; a.ldT = N; a.mul = res0; a.ldmul = N; }
    return a;
  };
  auto reset_out = [&](size_t vi) {   // residual epilogues update in place: start every measured launch from the same stream
    if (epi == 3 || epi == 4) CK(hipMemcpyAsync(outs[vi], res0, out_bytes, hipMemcpyDeviceToDevice, 0));
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<std::vector<float>> ms(vars.size());
  const int inner = getenv("INNER") ? atoi(getenv("INNER")) : 3;
  for (int r = 0; r < rounds + 1; ++r) {
    for (size_t vi = 0; vi < vars.size(); ++vi) {
      vima::GemmArgs a = make_args(vi);
      reset_out(vi);
      CK(hipEventRecord(e0, 0));
      for (int it = 0; it < inner; ++it) {
        const int e = vima::launch_gemm(a, true, 0);
        if (e) { printf("launch_gemm(%s) failed: %d\n", vars[vi].name.c_str(), e); return 1; }
      }
      CK(hipEventRecord(e1, 0));
      CK(hipEventSynchronize(e1));
      float t; CK(hipEventElapsedTime(&t, e0, e1));
      if (r > 0) ms[vi].push_back(t / inner);
    }
  }
  // one clean launch per variant for the comparisons
  for (size_t vi = 0; vi < vars.size(); ++vi) {
    vima::GemmArgs a = make_args(vi);
    reset_out(vi);
    if (stamps) { CK(hipMemsetAsync(dbg, 0, (size_t)nblk * 320, 0)); vars[vi].t.gemm_dbg = dbg; }
    const int e = vima::launch_gemm(a, true, 0);
    if (e) { printf("launch failed %d\n", e); return 1; }
    CK(hipDeviceSynchronize());
    if (stamps) {
      std::vector<long long> h((size_t)nblk * 40);
      CK(hipMemcpy(h.data(), dbg, h.size() * 8, hipMemcpyDeviceToHost));
      if (vars[vi].name == "pp" || vars[vi].name == "pp0") {   // mean clocks of the 16 phases of iterations 0 / 1, then of iterations 2 .. 17
        const int nit = std::min(K / 128, 18);
        std::vector<double> acc(32, 0.0); int c2 = 0;
        for (int b = 0; b < nblk; ++b) {
          const long long* d = &h[(size_t)b * 8];
          const long long* it = &h[(size_t)nblk * 8 + (size_t)b * 32];
          if (d[3] <= 0) continue;
          long long prev = d[1];
          for (int i = 0; i < 14 + nit; ++i) { if (i >= 32 || it[i] == 0) break; acc[i] += (double)(it[i] - prev); prev = it[i]; }
          ++c2;
        }
        printf("  [pp] clocks per phase, iterations 0 and 1:");
        for (int i = 0; i < 16 && i < 8 * nit; ++i) printf(" %.0f", acc[i] / c2);
        printf("\n  [pp] clocks per iteration 2 ..:");
        for (int i = 16; i < 14 + nit; ++i) printf(" %.0f", acc[i] / c2);
        printf("\n");
      }
      if ((vars[vi].name == "q4" || vars[vi].name == "q4h") && getenv("KTSTAMPS")) {   // -DVIMA_Q4_KT_STAMPS builds: mean clocks of each K-tile of a tile (first 32), from the start of its main loop
        const int nkt = std::min(K / 64, 32);
        const int nb4 = ((M + 255) / 256 + 7) / 8 * 8 * (N / 384);
        std::vector<double> acc(32, 0.0); int c2 = 0;
        for (int b = 0; b < nb4; ++b) {
          const long long* d = &h[(size_t)b * 8];
          const long long* it = &h[(size_t)nb4 * 8 + (size_t)b * 32];
          if (d[3] <= 0) continue;
          long long prev = d[1];
          for (int i = 0; i < nkt; ++i) { acc[i] += (double)(it[i] - prev); prev = it[i]; }
          ++c2;
        }
        printf("  [q4] clocks per K-tile:");
        for (int i = 0; i < nkt; ++i) printf(" %.0f", acc[i] / std::max(c2, 1));
        printf("\n");
      }
      double pro = 0, mainl = 0, epil = 0, rt = 0; int cnt = 0;
      for (int b = 0; b < nblk; ++b) {
        const long long* d = &h[(size_t)b * 8];
        if (d[3] <= 0) continue;
        pro += (double)(d[1] - d[0]); mainl += (double)(d[2] - d[1]); epil += (double)(d[3] - d[2]); rt += (double)(d[5] - d[4]) / 100.0; ++cnt;
      }
      if (cnt) printf("  [%s] stamps over %d tiles (shader clocks): prologue %.0f  main loop %.0f (%.0f per K-tile)  epilogue %.0f ; %.2f us per tile, clock %.3f GHz\n",
                      vars[vi].name.c_str(), cnt, pro / cnt, mainl / cnt, mainl / cnt / (K / 64), epil / cnt, rt / cnt, (pro + mainl + epil) / cnt / (rt / cnt * 1e3));
      vars[vi].t.gemm_dbg = nullptr;
    }
  }
  // bitwise comparison against variant 0
  std::vector<uint8_t> h0(out_bytes), h1;
  CK(hipMemcpy(h0.data(), outs[0], out_bytes, hipMemcpyDeviceToHost));
  for (size_t vi = 1; vi < vars.size(); ++vi) {
    if (vars[vi].name == "fp8") continue;        // different operands
    h1.resize(out_bytes);
    CK(hipMemcpy(h1.data(), outs[vi], out_bytes, hipMemcpyDeviceToHost));
    long long diff = 0, first = -1;
    const size_t es = epi == 3 ? 4 : 2;
    for (size_t i = 0; i < out_bytes / es; ++i)
      if (memcmp(&h0[i * es], &h1[i * es], es)) { if (first < 0) first = (long long)i; ++diff; }
    printf("  %s vs %s: %lld of %lld elements differ%s", vars[vi].name.c_str(), vars[0].name.c_str(), diff, (long long)(out_bytes / es), diff ? "" : " (bit-identical)\n");
    if (diff) printf(" (first at row %lld col %lld)\n", first / N, first % N);
    if (diff && getenv("DIFFMAP")) {   // where inside a 256 x 384 tile the differing elements sit: 8 x 12 blocks of 32 x 32, then by tile row
      long long hist[8][12] = {}, byrow[32] = {};
      for (size_t i = 0; i < out_bytes / es; ++i)
        if (memcmp(&h0[i * es], &h1[i * es], es)) { const long long r = i / N, c = i % N; ++hist[(r % 256) / 32][(c % 384) / 32]; ++byrow[r % 32]; }
      for (int a = 0; a < 8; ++a) { printf("    mi-block %d:", a); for (int b = 0; b < 12; ++b) printf(" %8lld", hist[a][b]); printf("\n"); }
      printf("    by row %% 32:"); for (int a = 0; a < 32; ++a) printf(" %lld", byrow[a]); printf("\n");
    }
  }
  // host fp64 reference on sampled entries (epi 1 / act 0 / no bias only: the raw product), transpose-detecting (random data)
  if (epi == 1 && act == 0 && !bias) {
    std::vector<uint16_t> hA((size_t)M * K), hW((size_t)N * K);
    CK(hipMemcpy(hA.data(), A, hA.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hW.data(), W, hW.size() * 2, hipMemcpyDeviceToHost));
    for (size_t vi = 0; vi < vars.size(); ++vi) {
      std::vector<uint16_t> ho((size_t)M * N);
      CK(hipMemcpy(ho.data(), outs[vi], ho.size() * 2, hipMemcpyDeviceToHost));
      double maxerr = 0, maxref = 0; int bad = 0;
      const int samples = 20000;
      for (int s = 0; s < samples; ++s) {
        int m, n;
        if (s < 4096) { m = (s * 37) % M; n = (s * 101 + (s >> 6)) % N; }       // spread over tiles
        else { m = rnd() % M; n = rnd() % N; }
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += (double)bf2f(hA[(size_t)m * K + k]) * (double)bf2f(hW[(size_t)n * K + k]);
        const double got = bf2f(ho[(size_t)m * N + n]);
        const double err = fabs(got - acc);
        if (err > maxerr) maxerr = err;
        if (fabs(acc) > maxref) maxref = fabs(acc);
        if (err > 0.01 * fabs(acc) + 0.05 * sqrt((double)K) * 0.02) ++bad;
      }
      printf("  [%s] host fp64 check on %d sampled entries: max |err| %.4g (max |ref| %.4g), %d outside bf16 rounding\n", vars[vi].name.c_str(), samples, maxerr, maxref, bad);
    }
  }
  if (any8 && epi == 1 && act == 0 && !bias) {     // host fp64 reference of the fp8 product on sampled entries
    auto dec = [](uint8_t b) {
      const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
      const double v = e == 0 ? ldexp((double)m, -9) : ldexp(1.0 + m / 8.0, e - 7);
      return s ? -v : v;
    };
    for (size_t vi = 0; vi < vars.size(); ++vi) {
      if (vars[vi].name != "fp8") continue;
      std::vector<uint16_t> ho((size_t)M * N);
      CK(hipMemcpy(ho.data(), outs[vi], ho.size() * 2, hipMemcpyDeviceToHost));
      double maxrel = 0; int bad = 0;
      const int samples = 20000;
      for (int smp = 0; smp < samples; ++smp) {
        int m, n;
        if (smp < 4096) { m = (smp * 37) % M; n = (smp * 101 + (smp >> 6)) % N; }
        else { m = rnd() % M; n = rnd() % N; }
        double acc = 0;
        for (int k = 0; k < K; ++k) acc += dec(hA8[(size_t)m * K + k]) * dec(hW8[(size_t)n * K + k]);
        acc *= (double)hws[n] * (double)asc;
        const double got = bf2f(ho[(size_t)m * N + n]);
        const double tol = 0.008 * fabs(acc) + 1e-3 * sqrt((double)K) * 240.0 * 240.0 * 0.03 * 0.037;
        if (fabs(got - acc) > tol) ++bad;
        if (fabs(acc) > 1.0) maxrel = std::max(maxrel, fabs(got - acc) / fabs(acc));
      }
      printf("  [fp8] host fp64 check of the e4m3 product on %d sampled entries: max rel err %.3g (|ref| > 1), %d outside bf16 rounding\n", samples, maxrel, bad);
    }
  }
  for (size_t vi = 0; vi < vars.size(); ++vi) {
    std::vector<float> v = ms[vi];
    std::sort(v.begin(), v.end());
    const double med = v[v.size() / 2], mn = v[0];
    const double fl = 2.0 * M * N * K;
    printf("M%d N%d K%d epi%d act%d [%s]: median %.4f ms = %.1f TFLOP/s ; best %.4f ms = %.1f TFLOP/s\n", M, N, K, epi, act, vars[vi].name.c_str(), med,
           fl / med / 1e9, mn, fl / mn / 1e9);
  }
  return 0;
}
