// Per-frame detection post-processing in ONE pass over all classes:
// box decode + clip + /scale (lib/bbox/bbox_transform.py:103-140, :45-60, tester.py:148-152),
// per-class score threshold + greedy NMS (tester.py:265-272, lib/nms/nms.py:37-74) and the
// max_per_image cap (tester.py:274-281).  The reference runs 30 numpy NMS calls per frame on
// the host after a blocking D2H copy; here one workgroup per class does filter -> decode ->
// rank-sort -> 64-bit IoU mask in LDS -> sweep, then a single small workgroup applies the cap.
// Arithmetic is float64 like the numpy reference (fp64 is full rate on CDNA4), following
// oracle orc_det_postprocess / orc_nms_f64 / orc_bbox_pred_clip operation for operation.
#include "common.h"

using namespace lsfa;

namespace {

constexpr int kThreads = 256;
constexpr int kMaxR = 512;

__device__ __forceinline__ void decode_box(const float* __restrict__ roi, const float* __restrict__ d, double im_h,
                                           double im_w, double scale, double* o) {
  const double x1 = roi[1], y1 = roi[2], x2 = roi[3], y2 = roi[4];
  const double widths = x2 - x1 + 1.0, heights = y2 - y1 + 1.0;
  const double ctr_x = x1 + 0.5 * (widths - 1.0), ctr_y = y1 + 0.5 * (heights - 1.0);
  const double pcx = (double)d[0] * widths + ctr_x;
  const double pcy = (double)d[1] * heights + ctr_y;
  const double pw = (double)expf_cr(d[2]) * widths;
  const double ph = (double)expf_cr(d[3]) * heights;
  double bx1 = pcx - 0.5 * (pw - 1.0), by1 = pcy - 0.5 * (ph - 1.0);
  double bx2 = pcx + 0.5 * (pw - 1.0), by2 = pcy + 0.5 * (ph - 1.0);
  bx1 = fmax(fmin(bx1, im_w - 1), 0.0); by1 = fmax(fmin(by1, im_h - 1), 0.0);
  bx2 = fmax(fmin(bx2, im_w - 1), 0.0); by2 = fmax(fmin(by2, im_h - 1), 0.0);
  o[0] = bx1 / scale; o[1] = by1 / scale; o[2] = bx2 / scale; o[3] = by2 / scale;
}

__global__ __launch_bounds__(kThreads) void bbox_pred_clip_kernel(const float* __restrict__ rois,
                                                                  const float* __restrict__ deltas, int R, int nreg,
                                                                  double im_h, double im_w, double scale,
                                                                  double* __restrict__ pred) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= R * nreg) return;
  const int r = i / nreg;
  decode_box(rois + (size_t)r * 5, deltas + (size_t)i * 4, im_h, im_w, scale, pred + (size_t)i * 4);
}

// grid (ncls); block 256.  LDS carve (all sized by R, offsets multiples of 16):
//   box[R][4] f64 | score[R] f64 | mask[R][Wd] u64 | src[R] i32 | order[R] i32 | kept[R] i32 | misc
__global__ __launch_bounds__(kThreads) void det_class_kernel(
    const float* __restrict__ rois, const float* __restrict__ deltas, const float* __restrict__ probs, int R,
    int ncls, int nreg, int class_agnostic, double im_h, double im_w, double scale, double score_thresh,
    double nms_thresh, double* __restrict__ dets, int* __restrict__ counts, int* __restrict__ keep_idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int j = blockIdx.x;
  const int tid = threadIdx.x;
  if (j == 0) { if (tid == 0) counts[0] = 0; return; }
  const int Wd = (R + 63) / 64;
  double* box = reinterpret_cast<double*>(smem);
  double* score = box + (size_t)R * 4;
  uint64_t* mask = reinterpret_cast<uint64_t*>(score + R);
  int* src = reinterpret_cast<int*>(mask + (size_t)R * Wd);
  int* order = src + R;
  int* kept = order + R;
  int* misc = kept + R;  // [0] running count, [1..4] wave sums

  // 1. threshold + in-order compaction (np.where(scores[:, j] > thresh), tester.py:267)
  if (tid == 0) misc[0] = 0;
  __syncthreads();
  const int lane = tid & 63, wid = tid >> 6;
  for (int r0 = 0; r0 < R; r0 += kThreads) {
    const int r = r0 + tid;
    const double s = r < R ? (double)probs[(size_t)r * ncls + j] : 0.0;
    const bool flag = r < R && s > score_thresh;
    const unsigned long long bal = __ballot(flag);
    if (lane == 0) misc[1 + wid] = __popcll(bal);
    __syncthreads();
    int base = misc[0];
    for (int w = 0; w < wid; ++w) base += misc[1 + w];
    if (flag) {
      const int pos = base + __popcll(bal & ((1ULL << lane) - 1ULL));
      src[pos] = r;
      score[pos] = s;
      const int col = class_agnostic ? 1 : j;
      decode_box(rois + (size_t)r * 5, deltas + ((size_t)r * nreg + col) * 4, im_h, im_w, scale, box + (size_t)pos * 4);
    }
    __syncthreads();
    if (tid == 0) misc[0] += misc[1] + misc[2] + misc[3] + misc[4];
    __syncthreads();
  }
  const int m = misc[0];
  if (m == 0) { if (tid == 0) counts[j] = 0; return; }

  // 2. rank sort: score descending, ties by ascending candidate (= roi) index
  for (int i = tid; i < m; i += kThreads) {
    const double si = score[i];
    int rank = 0;
    for (int k = 0; k < m; ++k) {
      const double sk = score[k];
      rank += (sk > si) || (sk == si && k < i);
    }
    order[rank] = i;
  }
  __syncthreads();

  // 3. suppression mask over sorted positions: bit b of mask[a][w] <=> position 64w+b > a and
  //    NOT (ovr <= thresh)  (nms.py:71 keeps `ovr <= thresh`)
  for (int t = tid; t < m * Wd; t += kThreads) {
    const int a = t / Wd, w = t - a * Wd;
    uint64_t bits = 0;
    if (64 * w + 63 > a) {
      const int ia = order[a];
      const double ax1 = box[ia * 4], ay1 = box[ia * 4 + 1], ax2 = box[ia * 4 + 2], ay2 = box[ia * 4 + 3];
      const double area_a = (ax2 - ax1 + 1) * (ay2 - ay1 + 1);
      const int b0 = max(64 * w, a + 1), b1 = min(64 * w + 64, m);
      for (int b = b0; b < b1; ++b) {
        const int ib = order[b];
        const double bx1 = box[ib * 4], by1 = box[ib * 4 + 1], bx2 = box[ib * 4 + 2], by2 = box[ib * 4 + 3];
        const double xx1 = fmax(ax1, bx1), yy1 = fmax(ay1, by1);
        const double xx2 = fmin(ax2, bx2), yy2 = fmin(ay2, by2);
        const double ww = fmax(0.0, xx2 - xx1 + 1), hh = fmax(0.0, yy2 - yy1 + 1);
        const double inter = ww * hh;
        const double area_b = (bx2 - bx1 + 1) * (by2 - by1 + 1);
        const double ovr = inter / (area_a + area_b - inter);
        if (!(ovr <= nms_thresh)) bits |= 1ULL << (b - 64 * w);
      }
    }
    mask[(size_t)a * Wd + w] = bits;
  }
  __syncthreads();

  // 4. sweep by wave 0: lane w holds word w of the removed set
  int nk = 0;
  if (wid == 0) {
    uint64_t remv = 0;
    for (int a = 0; a < m; ++a) {
      const int w = a >> 6, b = a & 63;
      const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)remv, w);
      const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(remv >> 32), w);
      const bool removed = b < 32 ? ((lo >> b) & 1u) : ((hi >> (b - 32)) & 1u);
      if (!removed) {
        if (lane == 0) kept[nk] = order[a];
        ++nk;
        if (lane < Wd) remv |= mask[(size_t)a * Wd + lane];
      }
    }
    if (lane == 0) { misc[0] = nk; counts[j] = nk; }
  }
  __syncthreads();
  nk = misc[0];

  // 5. survivors in NMS order
  for (int k = tid; k < nk; k += kThreads) {
    const int i = kept[k];
    double* o = dets + ((size_t)j * R + k) * 5;
    o[0] = box[i * 4]; o[1] = box[i * 4 + 1]; o[2] = box[i * 4 + 2]; o[3] = box[i * 4 + 3];
    o[4] = score[i];
    if (keep_idx) keep_idx[(size_t)j * R + k] = src[i];
  }
}

__device__ __forceinline__ uint32_t desc_key(float score) {
  const uint32_t u = __float_as_uint(score + 0.0f);
  const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~asc;
}

// max_per_image cap (tester.py:274-281).  The scores are float32 probabilities widened to
// float64, so a 32-bit radix select on their float image is exact.  Single workgroup.
__global__ __launch_bounds__(1024) void det_cap_kernel(double* __restrict__ dets, int* __restrict__ counts,
                                                       int* __restrict__ keep_idx, int R, int ncls,
                                                       int max_per_image) {
  __shared__ uint32_t hist[256];
  __shared__ int misc[4];
  const int tid = threadIdx.x;
  int total = 0;
  for (int j = 1; j < ncls; ++j) total += counts[j];
  if (total <= max_per_image) return;
  // image_thresh = np.sort(image_scores)[-max_per_image]  == the max_per_image-th largest
  uint32_t prefix = 0;
  int remaining = max_per_image;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int j = 1; j < ncls; ++j) {
      const int cj = counts[j];
      for (int k = tid; k < cj; k += 1024) {
        const uint32_t key = desc_key((float)dets[((size_t)j * R + k) * 5 + 4]);
        const bool match = (shift == 24) || (((key ^ prefix) >> (shift + 8)) == 0);
        if (match) atomicAdd(&hist[(key >> shift) & 255u], 1u);
      }
    }
    __syncthreads();
    if (tid == 0) {
      int before = 0, bucket = 0;
      for (int q = 0; q < 256; ++q) {
        if (before + (int)hist[q] >= remaining) { bucket = q; break; }
        before += (int)hist[q];
      }
      misc[0] = bucket;
      misc[1] = remaining - before;
    }
    __syncthreads();
    prefix |= (uint32_t)misc[0] << shift;
    remaining = misc[1];
    __syncthreads();
  }
  const uint32_t T = prefix;  // keep score >= image_thresh  <=>  key <= T
  const int lane = tid & 63, wid = tid >> 6;
  for (int j = 1 + wid; j < ncls; j += 16) {
    const int cj = counts[j];
    int m = 0;
    for (int k0 = 0; k0 < cj; k0 += 64) {
      const int k = k0 + lane;
      double v[5] = {0, 0, 0, 0, 0};
      int ki = -1;
      bool flag = false;
      if (k < cj) {
        const double* srcp = dets + ((size_t)j * R + k) * 5;
        v[0] = srcp[0]; v[1] = srcp[1]; v[2] = srcp[2]; v[3] = srcp[3]; v[4] = srcp[4];
        if (keep_idx) ki = keep_idx[(size_t)j * R + k];
        flag = desc_key((float)v[4]) <= T;
      }
      const unsigned long long bal = __ballot(flag);
      if (flag) {
        const int pos = m + __popcll(bal & ((1ULL << lane) - 1ULL));
        double* dst = dets + ((size_t)j * R + pos) * 5;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3]; dst[4] = v[4];
        if (keep_idx) keep_idx[(size_t)j * R + pos] = ki;
      }
      m += __popcll(bal);
    }
    if (lane == 0) counts[j] = m;
  }
}

size_t class_lds_bytes(int R) {
  const size_t Wd = (size_t)(R + 63) / 64;
  return (size_t)R * 4 * 8 + (size_t)R * 8 + (size_t)R * Wd * 8 + (size_t)R * 4 * 3 + 64;
}

}  // namespace

extern "C" size_t lsfa_det_workspace_bytes(int R, int ncls) {
  (void)R; (void)ncls;
  return 256;  // everything lives in LDS; kept in the ABI so callers never need to change
}

extern "C" int lsfa_bbox_pred_clip(const float* rois, const float* deltas, int R, int nreg, double im_h, double im_w,
                                   double scale, double* pred_boxes, void* stream) {
  LSFA_REQUIRE(R >= 0 && nreg > 0, "lsfa_bbox_pred_clip: bad shape");
  if (R == 0) return LSFA_OK;
  LSFA_REQUIRE(rois && deltas && pred_boxes, "lsfa_bbox_pred_clip: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(LSFA_OP_DET, s);
  hipLaunchKernelGGL(bbox_pred_clip_kernel, dim3(ceil_div(R * nreg, kThreads)), dim3(kThreads), 0, s, rois, deltas, R,
                     nreg, im_h, im_w, scale, pred_boxes);
  LSFA_LAUNCH_CHECK("lsfa_bbox_pred_clip");
  return LSFA_OK;
}

extern "C" int lsfa_det_postprocess(const float* rois, const float* deltas, const float* probs, int R, int ncls,
                                    int nreg, int class_agnostic, double im_h, double im_w, double scale,
                                    double score_thresh, double nms_thresh, int max_per_image, double* dets,
                                    int* counts, int* keep_idx, void* ws, size_t ws_bytes, void* stream) {
  (void)ws; (void)ws_bytes;
  LSFA_REQUIRE(rois && deltas && probs && dets && counts, "lsfa_det_postprocess: NULL argument");
  LSFA_REQUIRE(R > 0 && ncls > 1 && nreg > 0, "lsfa_det_postprocess: bad shape R=%d ncls=%d nreg=%d", R, ncls, nreg);
  LSFA_REQUIRE(class_agnostic ? nreg >= 2 : nreg >= ncls, "lsfa_det_postprocess: nreg=%d too small", nreg);
  if (R > kMaxR) { set_error("lsfa_det_postprocess: R=%d exceeds %d", R, kMaxR); return LSFA_ENOTSUP; }
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = class_lds_bytes(R);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)det_class_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  ProfScope prof(LSFA_OP_DET, s);
  hipLaunchKernelGGL(det_class_kernel, dim3(ncls), dim3(kThreads), lds, s, rois, deltas, probs, R, ncls, nreg,
                     class_agnostic, im_h, im_w, scale, score_thresh, nms_thresh, dets, counts, keep_idx);
  if (max_per_image > 0)
    hipLaunchKernelGGL(det_cap_kernel, dim3(1), dim3(1024), 0, s, dets, counts, keep_idx, R, ncls, max_per_image);
  LSFA_LAUNCH_CHECK("lsfa_det_postprocess");
  return LSFA_OK;
}
