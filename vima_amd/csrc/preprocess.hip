// Image preprocessing in front of the policy, on the GPU: the per-object work of `prepare_obs` /
// `prepare_prompt` (/root/reference/scripts/example.py:243-473) -- segmentation mask -> pixel bbox -> inclusive crop ->
// zero-pad to a square -> cv2.resize(32x32, INTER_AREA) -> uint8 -- for every (frame, object id) at once.
// Integer / byte work, HBM-trivial (a 128x256 frame is 128 KiB): one workgroup per (frame, output slot) scans the mask (LDS
// min/max/count per object id; the scan is repeated by the n_obj workgroups of a frame -- 32 KiB each, cheaper than a
// second launch), orders the objects like the reference (present ones first, missing ones zero-padded at the end, mask
// False) and resamples ITS crop straight from the frame. The three resize regimes follow OpenCV's
// documented arithmetic exactly (oracle/preprocess_oracle.py restates them; results are compared bit for bit):
//   S % 32 == 0 : block sums; 2x2 -> (sum + 2) >> 2, else rint(sum * (1.f / f^2))   (resizeAreaFast_)
//   S > 32      : separable fp32 area weights, adds in table order, no fma          (computeResizeAreaTab / ResizeArea_Invoker)
//   S < 32      : 11-bit fixed-point bilinear with "area" source coordinates        (INTER_AREA up-scaling branch of cv::resize)
#include "../../include/vima_hip.h"

#include <hip/hip_runtime.h>
#include <string>

#include "common.h"   // PerDeviceOnce

namespace vima {
int api_fail(const std::string& m);
}

namespace {

constexpr int kOut = 32;
constexpr int kMaxObj = 64;
constexpr int kMaxSide = 320;   // largest frame side accepted: crops of S <= 320 px -> scale <= 10 -> at most 12 area taps
constexpr int kMaxTaps = 12;

struct Taps {               // area taps of one destination index
  int n;
  int si[kMaxTaps];
  float a[kMaxTaps];
};

// computeResizeAreaTab for destination index d (double arithmetic like OpenCV, alpha rounded to float)
__device__ void area_taps(int ssize, double scale, int d, Taps& t) {
  const double fsx1 = d * scale, fsx2 = fsx1 + scale;
  const double cell = fmin(scale, (double)ssize - fsx1);
  int sx1 = (int)ceil(fsx1), sx2 = (int)floor(fsx2);
  sx2 = sx2 < ssize - 1 ? sx2 : ssize - 1;
  sx1 = sx1 < sx2 ? sx1 : sx2;
  int n = 0;
  if (sx1 - fsx1 > 1e-3) { t.si[n] = sx1 - 1; t.a[n] = (float)((sx1 - fsx1) / cell); ++n; }
  for (int sx = sx1; sx < sx2 && n < kMaxTaps - 1; ++sx) { t.si[n] = sx; t.a[n] = (float)(1.0 / cell); ++n; }
  if (fsx2 - sx2 > 1e-3 && n < kMaxTaps) { t.si[n] = sx2; t.a[n] = (float)(fmin(fmin(fsx2 - sx2, 1.0), cell) / cell); ++n; }
  t.n = n;
}

__device__ __forceinline__ int sat_u8(int v) { return v < 0 ? 0 : (v > 255 ? 255 : v); }

template <typename SegT>
__global__ __launch_bounds__(256) void crop_objects_kernel(const uint8_t* __restrict__ rgb, const SegT* __restrict__ segm,
                                                           const int* __restrict__ obj_ids, int n_obj, int H, int W,
                                                           uint8_t* __restrict__ crops, long long* __restrict__ bbox,
                                                           uint8_t* __restrict__ mask, int lds_bytes) {
  extern __shared__ uint8_t s_crop[];   // the crop's source pixels [3][hc][wc] (when they fit): every tap then reads LDS
  __shared__ int s_id[kMaxObj];
  __shared__ volatile int s_xmin[kMaxObj], s_xmax[kMaxObj], s_ymin[kMaxObj], s_ymax[kMaxObj], s_cnt[kMaxObj];
  __shared__ int s_slot_obj[kMaxObj];   // slot -> object index (present objects in order), -1 = padding
  __shared__ Taps s_taps[kOut];
  __shared__ int s_lofs[kOut], s_lc0[kOut], s_lc1[kOut];
  const int tid = threadIdx.x;
  const long long f = blockIdx.x;
  const uint8_t* img = rgb + f * 3LL * H * W;
  const SegT* sg = segm + f * (long long)H * W;
  if (tid < n_obj) {
    s_id[tid] = obj_ids[tid];
    s_xmin[tid] = W; s_xmax[tid] = -1; s_ymin[tid] = H; s_ymax[tid] = -1; s_cnt[tid] = 0;
  }
  __syncthreads();
  // ---- pass 1: per-object pixel extent and count (np.nonzero(segm == obj_id), example.py:400). Only a handful of
  // workgroups run, so nothing hides a load's latency: every thread fetches 8 pixels (independent loads) before it looks at them
  for (int p0 = tid * 8; p0 < H * W; p0 += 256 * 8) {
    int vals[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) vals[j] = p0 + j < H * W ? (int)sg[p0 + j] : -1;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int v = vals[j], p = p0 + j;
      if (p >= H * W) break;
      for (int k = 0; k < n_obj; ++k) {
        if (v == s_id[k]) {   // ids may repeat in obj_ids: every matching entry sees the pixel
          const int y = p / W, x = p - y * W;
          // plain reads only FILTER the atomics (a stale value costs one redundant atomic, never a wrong extent); the count
          // is only ever compared with 2
          if (x < s_xmin[k]) atomicMin((int*)&s_xmin[k], x);
          if (x > s_xmax[k]) atomicMax((int*)&s_xmax[k], x);
          if (y < s_ymin[k]) atomicMin((int*)&s_ymin[k], y);
          if (y > s_ymax[k]) atomicMax((int*)&s_ymax[k], y);
          if (s_cnt[k] < 2) atomicAdd((int*)&s_cnt[k], 1);
        }
      }
    }
  }
  __syncthreads();
  // ---- slot order: present objects (>= 2 pixels, example.py:401) first, in obj_ids order; the rest is zero padding
  if (tid == 0) {
    int slot = 0;
    for (int k = 0; k < n_obj; ++k)
      if (s_cnt[k] >= 2) s_slot_obj[slot++] = k;
    for (; slot < n_obj; ++slot) s_slot_obj[slot] = -1;
  }
  __syncthreads();
  {   // one workgroup per (frame, slot): every workgroup of a frame repeats the (cheap) mask scan, then resamples its own crop
    const int slot = blockIdx.y;
    const int k = s_slot_obj[slot];   // workgroup-uniform
    uint8_t* out = crops + (f * n_obj + slot) * 3LL * kOut * kOut;
    long long* bb = bbox + (f * n_obj + slot) * 4;
    if (k < 0) {
      for (int i = tid; i < 3 * kOut * kOut; i += 256) out[i] = 0;
      if (tid < 4) bb[tid] = 0;
      if (tid == 0) mask[f * n_obj + slot] = 0;
      return;
    }
    const int xmin = s_xmin[k], xmax = s_xmax[k], ymin = s_ymin[k], ymax = s_ymax[k];
    const int hc = ymax - ymin + 1, wc = xmax - xmin + 1;
    if (tid == 0) {
      bb[0] = (xmin + xmax) / 2;   // int((xmin + xmax) / 2): non-negative, so truncation == floor
      bb[1] = (ymin + ymax) / 2;
      bb[2] = ymax - ymin;
      bb[3] = xmax - xmin;
      mask[f * n_obj + slot] = 1;
    }
    const int S = hc > wc ? hc : wc;
    const int padx = hc > wc ? (hc - wc) / 2 : 0;   // zeros BEFORE the crop on the shorter axis (example.py:411-413)
    const int pady = wc > hc ? (wc - hc) / 2 : 0;
    // stage the crop's pixels in LDS (coalesced row reads); a crop of a frame larger than the LDS budget is read in place
    const bool staged = 3 * hc * wc <= lds_bytes;
    if (staged) {
      // row-wise copy (no per-byte index divisions): wave w takes crop rows w, w + 4, ... of the 3 * hc (channel, row) pairs,
      // four rows per trip so that up to 16 independent loads are in flight per lane
      const int wv = tid >> 6, ln = tid & 63;
      for (int r0 = wv * 4; r0 < 3 * hc; r0 += 16) {
        uint8_t px[4][4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = r0 + j;
          if (row < 3 * hc) {
            const int c = row / hc, yy = row - c * hc;
            const uint8_t* g = img + ((long long)c * H + (ymin + yy)) * W + xmin;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (ln + q * 64 < wc) px[j][q] = g[ln + q * 64];
          }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = r0 + j;
          if (row < 3 * hc) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (ln + q * 64 < wc) s_crop[row * wc + ln + q * 64] = px[j][q];
          }
        }
        for (int xx = 256 + ln; xx < wc; xx += 64)      // frames wider than 256 px (up to 320)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (r0 + j < 3 * hc) {
              const int row = r0 + j, c = row / hc, yy = row - c * hc;
              s_crop[row * wc + xx] = img[((long long)c * H + (ymin + yy)) * W + xmin + xx];
            }
      }
    }
    // square-source pixel (c, sy, sx) -> crop pixel or the zero padding
    auto src = [&](int c, int sy, int sx) -> int {
      const int yy = sy - pady, xx = sx - padx;
      if (yy < 0 || yy >= hc || xx < 0 || xx >= wc) return 0;
      if (staged) return (int)s_crop[(c * hc + yy) * wc + xx];
      return (int)img[((long long)c * H + (ymin + yy)) * W + (xmin + xx)];
    };
    const bool integer_scale = S >= kOut && S % kOut == 0;
    const bool up = S < kOut;
    const bool tabbed = !integer_scale && !up;
    if (tid < kOut) {
      if (up) {
        const double scale = (double)S / kOut, inv = 1.0 / scale;
        int s = (int)floor(tid * scale);
        float fx = (float)((tid + 1) - (s + 1) * inv);
        fx = fx <= 0.f ? 0.f : fx - floorf(fx);
        if (s < 0) { fx = 0.f; s = 0; }
        if (s >= S - 1) { fx = 0.f; s = S - 1; }
        s_lofs[tid] = s;
        int c0 = (int)rintf((1.f - fx) * 2048.f), c1 = (int)rintf(fx * 2048.f);   // saturate_cast<short>(cbuf * 2048)
        s_lc0[tid] = c0 > 32767 ? 32767 : c0;
        s_lc1[tid] = c1 > 32767 ? 32767 : c1;
      } else if (tabbed) {
        area_taps(S, (double)S / kOut, tid, s_taps[tid]);
      }
    }
    __syncthreads();
    for (int i = tid; i < 3 * kOut * kOut; i += 256) {
      const int c = i / (kOut * kOut), dy = (i / kOut) % kOut, dx = i % kOut;
      int v;
      if (integer_scale) {
        const int fct = S / kOut;
        int sum = 0;
        for (int y = 0; y < fct; ++y)
          for (int x = 0; x < fct; ++x) sum += src(c, dy * fct + y, dx * fct + x);
        if (fct == 1) v = sum;
        else if (fct == 2) v = (sum + 2) >> 2;
        else v = sat_u8((int)rintf(__fmul_rn((float)sum, __fdiv_rn(1.f, (float)(fct * fct)))));
      } else if (up) {
        const int sx = s_lofs[dx], sy = s_lofs[dy];
        const int sx1 = sx + 1 < S ? sx + 1 : S - 1, sy1 = sy + 1 < S ? sy + 1 : S - 1;
        const int r0 = src(c, sy, sx) * s_lc0[dx] + src(c, sy, sx1) * s_lc1[dx];
        const int r1 = src(c, sy1, sx) * s_lc0[dx] + src(c, sy1, sx1) * s_lc1[dx];
        v = sat_u8((((s_lc0[dy] * (r0 >> 4)) >> 16) + ((s_lc1[dy] * (r1 >> 4)) >> 16) + 2) >> 2);
      } else {
        const Taps* ty = &s_taps[dy];
        const Taps* tx = &s_taps[dx];
        float sum = 0.f;
        for (int j = 0; j < ty->n; ++j) {
          float buf = 0.f;
          for (int q = 0; q < tx->n; ++q) buf = __fadd_rn(buf, __fmul_rn((float)src(c, ty->si[j], tx->si[q]), tx->a[q]));
          const float t = __fmul_rn(ty->a[j], buf);
          sum = j == 0 ? t : __fadd_rn(sum, t);
        }
        v = sat_u8((int)rintf(sum));
      }
      out[i] = (uint8_t)v;
    }
  }
}

}  // namespace

extern "C" int vima_crop_objects(const uint8_t* rgb, const void* segm, int segm_elem_bytes, const int32_t* obj_ids, int n_frames,
                                 int n_obj, int H, int W, uint8_t* crops, int64_t* bbox, uint8_t* mask, vima_stream_t stream) {
  if (n_frames <= 0 || n_obj <= 0) return 0;
  if (!rgb || !segm || !obj_ids || !crops || !bbox || !mask) return vima::api_fail("vima_crop_objects: null argument");
  if (n_obj > kMaxObj) return vima::api_fail("vima_crop_objects: at most 64 object ids per call");
  if (H <= 0 || W <= 0 || H > kMaxSide || W > kMaxSide)
    return vima::api_fail("vima_crop_objects: frame sides must be in [1, 320] (VIMA-Bench frames are 128 x 256)");
  if (segm_elem_bytes != 1 && segm_elem_bytes != 4) return vima::api_fail("vima_crop_objects: segm must be uint8 or int32");
  hipStream_t st = (hipStream_t)stream;
  long long need = 3LL * H * W;
  const int lds = (int)(need < 96 * 1024 ? need : 96 * 1024);   // a whole 128 x 256 frame fits; larger frames: crops up to 96 KiB
  // the launch goes to the device that owns the output tensors (not necessarily the current one: the call has no handle)
  int cur_dev = 0, out_dev = 0;
  hipPointerAttribute_t pa;
  if (hipGetDevice(&cur_dev) != hipSuccess) return vima::api_fail("vima_crop_objects: hipGetDevice failed");
  out_dev = cur_dev;
  if (hipPointerGetAttributes(&pa, crops) == hipSuccess) out_dev = pa.device;
  struct DeviceGuard {
    int back; bool active;
    ~DeviceGuard() { if (active) (void)hipSetDevice(back); }
  } guard{cur_dev, false};
  if (out_dev != cur_dev) {
    if (hipSetDevice(out_dev) != hipSuccess) return vima::api_fail("vima_crop_objects: hipSetDevice failed");
    guard.active = true;
  }
  static vima::PerDeviceOnce attr;
  if (attr.ensure([&] {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&crop_objects_kernel<uint8_t>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        if (e != hipSuccess) return e;
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&crop_objects_kernel<int>), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      }) != hipSuccess)
    return vima::api_fail("vima_crop_objects: hipFuncSetAttribute failed");
  if (segm_elem_bytes == 1)
    hipLaunchKernelGGL(crop_objects_kernel<uint8_t>, dim3((unsigned)n_frames, (unsigned)n_obj), dim3(256), (size_t)lds, st, rgb, (const uint8_t*)segm, obj_ids,
                       n_obj, H, W, crops, (long long*)bbox, mask, lds);
  else
    hipLaunchKernelGGL(crop_objects_kernel<int>, dim3((unsigned)n_frames, (unsigned)n_obj), dim3(256), (size_t)lds, st, rgb, (const int*)segm, obj_ids, n_obj, H,
                       W, crops, (long long*)bbox, mask, lds);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vima::api_fail(std::string("vima_crop_objects launch failed: ") + hipGetErrorString(e));
  return 0;
}
