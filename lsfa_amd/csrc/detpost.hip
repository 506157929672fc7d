// Per-frame detection post-processing in ONE pass over all classes:
// box decode + clip + /scale (lib/bbox/bbox_transform.py:103-140, :45-60, tester.py:148-152),
// per-class score threshold + greedy NMS (tester.py:265-272, lib/nms/nms.py:37-74) and the
// max_per_image cap (tester.py:274-281).  The reference runs 30 numpy NMS calls per frame on
// the host after a blocking D2H copy; here one workgroup per class does filter -> decode ->
// rank -> 64-bit suppression mask in LDS (no division) -> ballot-fixpoint sweep, then a single small
// workgroup applies the cap.
// Arithmetic is float64 like the numpy reference (fp64 is full rate on CDNA4), following
// oracle orc_det_postprocess / orc_nms_f64 / orc_bbox_pred_clip operation for operation.
#include "common.h"
#include "nms_block.h"

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

// grid (ncls); block kClassThreads (16 waves).  LDS carve (all sized by R, offsets multiples of 8):
//   box[R][4] f64 (threshold order) | score[R] f64 | sbox[R][4] f64 (score order) | sarea[R] f64 |
//   mask[R][Wd] u64 | colw[R] u64 | src[R] i32 | order[R] i32 | rank[R] i32 | kept[R] i32 | misc
//
// Phases: threshold + ordered compaction -> rank by (score desc, roi index asc): the m^2 comparisons are
// split 4 ways per candidate over all 16 waves and summed with LDS atomics -> boxes re-laid in score order
// -> suppression mask over sorted positions, all 16 waves, WITHOUT the float64 division (exact: see
// suppresses()) and with the transposed words of the diagonal 64x64 blocks built alongside -> one wave
// sweeps the 64-position blocks, each resolved as a ballot fixpoint (a handful of steps) instead of a
// 64-step scalar loop.

constexpr int kClassThreads = 1024;
constexpr int kClassWaves = kClassThreads / 64;
constexpr int kRankSplit = 4;

// numpy keeps `ovr <= thresh` (lib/nms/nms.py:71) with ovr = fl64(inter / uni): the pair suppresses when
// NOT (ovr <= thresh).  Decided without dividing: v = inter - thresh*uni and d = fma(-thresh, uni, inter)
// = fl(v) have the same sign; v <= 0 => quotient <= thresh (rounding is monotone); v >= thresh*uni*2^-52
// => the exact quotient is at least the double after thresh => suppresses; d >= thresh*uni*2^-51
// implies that.  The sliver in between and non-positive / NaN unions take the division.
struct NmsTest64 { double thresh, thresh_eps; int fast; };

__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
  return __longlong_as_double((long long)readlane64((uint64_t)__double_as_longlong(v), src_lane));
}

__device__ __forceinline__ bool suppresses(double inter, double uni, const NmsTest64& t) {
  const double d = fma(-t.thresh, uni, inter);
  const bool pos = d > 0.0, clear = d >= t.thresh_eps * uni;
  const bool unsure = !t.fast || !(uni > 0.0) || (pos && !clear);
  bool r = pos && clear;
  if (__builtin_expect(unsure, 0)) r = !(inter / uni <= t.thresh);
  return r;
}

// (second launch bound, minimum waves per SIMD: 4 = one 16-wave workgroup per CU, 72 registers; 8 = two per CU, 63 registers + 8 spilled - an A/B
//  of tools/lab/build_variant.sh, profiles/r6)
#ifndef LSFA_DET_CLASS_MIN_WAVES
#define LSFA_DET_CLASS_MIN_WAVES 4
#endif
__global__ __launch_bounds__(kClassThreads, LSFA_DET_CLASS_MIN_WAVES) void det_class_kernel(
    const float* __restrict__ rois, const float* __restrict__ deltas, const float* __restrict__ probs, int R,
    int ncls, int nreg, int class_agnostic, double im_h, double im_w, double scale, double score_thresh,
    NmsTest64 nms, double* __restrict__ dets, int* __restrict__ counts, int* __restrict__ keep_idx) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int j = blockIdx.x;
  const int tid = threadIdx.x;
  {   // image blockIdx.y of a batch: its R rois (MultiProposal's layout: image b's rows are [b*R, (b+1)*R)) and its own outputs
    const size_t b = blockIdx.y;
    rois += b * R * 5; deltas += b * (size_t)R * nreg * 4; probs += b * (size_t)R * ncls;
    dets += b * (size_t)ncls * R * 5; counts += b * ncls;
    if (keep_idx) keep_idx += b * (size_t)ncls * R;
  }
  if (j == 0) { if (tid == 0) counts[0] = 0; return; }
  const int Wd = (R + 63) / 64;
  double* box = reinterpret_cast<double*>(smem);
  double* score = box + (size_t)R * 4;
  double* sbox = score + R;
  double* sarea = sbox + (size_t)R * 4;
  uint64_t* mask = reinterpret_cast<uint64_t*>(sarea + R);
  uint64_t* colw = mask + (size_t)R * Wd;
  int* src = reinterpret_cast<int*>(colw + R);
  int* order = src + R;
  int* rank = order + R;
  int* kept = rank + R;
  int* misc = kept + R;  // [0] running count, [1..16] wave sums

  // 1. threshold + in-order compaction (np.where(scores[:, j] > thresh), tester.py:267)
  if (tid == 0) misc[0] = 0;
  for (int i = tid; i < R; i += kClassThreads) rank[i] = 0;
  __syncthreads();
  const int lane = tid & 63, wid = tid >> 6;
  for (int r0 = 0; r0 < R; r0 += kClassThreads) {
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
    if (tid == 0) {
      int t = 0;
      for (int w = 0; w < kClassWaves; ++w) t += misc[1 + w];
      misc[0] += t;
    }
    __syncthreads();
  }
  const int m = misc[0];
  if (m == 0) { if (tid == 0) counts[j] = 0; return; }

  // 2. rank: score descending, ties by ascending candidate (= roi) index; candidate i's comparisons are cut
  //    into kRankSplit slices so that m * kRankSplit threads work
  const int slice = (m + kRankSplit - 1) / kRankSplit;
  for (int t = tid; t < m * kRankSplit; t += kClassThreads) {
    const int i = t / kRankSplit, q = t - i * kRankSplit;
    const double si = score[i];
    const int k0 = q * slice, k1 = min(k0 + slice, m);
    int before = 0;
    for (int k = k0; k < k1; ++k) {
      const double sk = score[k];
      before += (sk > si) || (sk == si && k < i);
    }
    if (before) atomicAdd(&rank[i], before);
  }
  __syncthreads();
  for (int i = tid; i < m; i += kClassThreads) {
    const int p = rank[i];
    order[p] = i;
    const double x1 = box[i * 4], y1 = box[i * 4 + 1], x2 = box[i * 4 + 2], y2 = box[i * 4 + 3];
    sbox[p * 4] = x1; sbox[p * 4 + 1] = y1; sbox[p * 4 + 2] = x2; sbox[p * 4 + 3] = y2;
    sarea[p] = (x2 - x1 + 1) * (y2 - y1 + 1);
  }
  __syncthreads();

  // 3. suppression mask over sorted positions, one wave per 64x64 tile of the upper triangle: bit c of
  //    mask[a][w] <=> position 64w+c > a and the pair suppresses.  The lane owns row a (box in registers), the
  //    column boxes are read from LDS at wave-uniform addresses, the word is built in a register: no
  //    atomics, no zero fill (words left of the diagonal are never read).  Diagonal tiles also produce
  //    colw[c] = the earlier positions of c's own block that suppress c (64 ballots).
  const int nblk = (m + 63) / 64;
  const int ntiles = nblk * (nblk + 1) / 2;
  for (int t = __builtin_amdgcn_readfirstlane(wid); t < ntiles; t += kClassWaves) {
    int cb = 0;
    while ((cb + 1) * (cb + 2) / 2 <= t) ++cb;
    const int rb = t - cb * (cb + 1) / 2;
    const int a = rb * 64 + lane;
    const bool a_ok = a < m;
    const int ar = a_ok ? a : 0;
    const double ax1 = sbox[ar * 4], ay1 = sbox[ar * 4 + 1], ax2 = sbox[ar * 4 + 2], ay2 = sbox[ar * 4 + 3];
    const double area_a = sarea[ar];
    const int c0 = cb * 64, ncol = min(64, m - c0);
    const bool diag = rb == cb;
    // column boxes are read from LDS at wave-uniform addresses (a broadcast), four columns per step so that the reads
    // of a step are in flight together; the "cannot tell without dividing" case is collected per lane and, if any lane
    // of the wave met it, the tile is redone with the division (rare)
    uint64_t bits = 0, tbits = 0;
    bool unsure = !nms.fast;
    for (int c4 = 0; c4 < ncol; c4 += 4) {
      double bx1[4], by1[4], bx2[4], by2[4], bar[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int q = min(c0 + c4 + u, m - 1);
        bx1[u] = sbox[q * 4]; by1[u] = sbox[q * 4 + 1]; bx2[u] = sbox[q * 4 + 2]; by2[u] = sbox[q * 4 + 3];
        bar[u] = sarea[q];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c4 + u;
        const double ww = fmax(0.0, fmin(ax2, bx2[u]) - fmax(ax1, bx1[u]) + 1), hh = fmax(0.0, fmin(ay2, by2[u]) - fmax(ay1, by1[u]) + 1);
        const double inter = ww * hh;
        const double uni = area_a + bar[u] - inter;
        const double d = fma(-nms.thresh, uni, inter);
        const bool pos = d > 0.0, clear = d >= nms.thresh_eps * uni;
        unsure = unsure || ((!(uni > 0.0) || (pos && !clear)) && c < ncol);
        const bool p = pos && clear && a_ok && c < ncol && (!diag || c > lane);
        bits |= (uint64_t)p << c;
        if (diag) {
          const uint64_t cw = __ballot(p);
          if (lane == c) tbits = cw;
        }
      }
    }
    if (__builtin_expect(__any(unsure), 0)) {       // the sliver around the threshold, or a degenerate union: divide
      bits = 0; tbits = 0;
      for (int c = 0; c < ncol; ++c) {
        const double* q = sbox + (size_t)(c0 + c) * 4;
        const double ww = fmax(0.0, fmin(ax2, q[2]) - fmax(ax1, q[0]) + 1), hh = fmax(0.0, fmin(ay2, q[3]) - fmax(ay1, q[1]) + 1);
        const double inter = ww * hh;
        const bool p = !(inter / (area_a + sarea[c0 + c] - inter) <= nms.thresh) && a_ok && (!diag || c > lane);
        bits |= (uint64_t)p << c;
        if (diag) {
          const uint64_t cw = __ballot(p);
          if (lane == c) tbits = cw;
        }
      }
    }
    if (a_ok) {
      mask[(size_t)a * Wd + cb] = bits;
      if (diag) colw[a] = tbits;
    }
  }
  __syncthreads();

  // 4. sweep by wave 0 in 64-position blocks: lane w (< Wd) holds word w of the removed set; a block is
  //    resolved by resolve_block (nms_block.h: ballot fixpoint, scalar scan over the survivors when the
  //    suppression chains are long), then the survivors' rows are ORed in.
  int nk = 0;
  if (wid == 0) {
    uint64_t remv = 0;
    for (int b = 0; b < nblk; ++b) {
      const int base = b * 64, nb = min(64, m - base);
      const uint64_t cw = lane < nb ? colw[base + lane] : 0ULL;
      const uint64_t rowd = lane < nb ? mask[(size_t)(base + lane) * Wd + b] : 0ULL;
      const uint64_t cur = readlane64(remv, b);
      const bool alive = lane < nb && !((cur >> lane) & 1ULL);
      const uint64_t G = resolve_block(__ballot(alive), alive, cw, rowd, 64);
      if ((G >> lane) & 1ULL) kept[nk + __popcll(G & ((1ULL << lane) - 1ULL))] = order[base + lane];
      nk += __popcll(G);
      uint64_t rem = G;
      while (rem) {
        int ks[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          if (rem) { ks[u] = __builtin_ctzll(rem); rem &= rem - 1; } else ks[u] = -1;
        }
        uint64_t vv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) vv[u] = (ks[u] >= 0 && lane > b && lane < Wd) ? mask[(size_t)(base + ks[u]) * Wd + lane] : 0ULL;
#pragma unroll
        for (int u = 0; u < 8; ++u) remv |= vv[u];
      }
    }
    if (lane == 0) { misc[0] = nk; counts[j] = nk; }
  }
  __syncthreads();
  nk = misc[0];

  // 5. survivors in NMS order
  for (int k = tid; k < nk; k += kClassThreads) {
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

// max_per_image cap (tester.py:274-281).  The scores are float32 probabilities widened to float64, so a
// select on the 32-bit order keys of their float image is exact.  Single workgroup; the keys of all survivors are
// staged in LDS once (ncls*R <= 16K keys).  image_thresh = the max_per_image-th best key, found in one pass:
// a 4096-bin histogram over (exponent, 7 mantissa bits) of the score, the bin the max_per_image-th best falls
// in, then the exact order inside that bin by pairwise counting among its (few) members.  A bin holding more
// than kCapListMax keys (thousands of equal scores) falls back to an 8-bit radix select over the staged keys.
constexpr int kCapMaxKeys = 16384;
constexpr int kCapBins = 4096;
constexpr int kCapListMax = 1024;

// monotone 12-bit image of a descending order key: larger score -> larger bin; 128 bins per octave over [2^-32, 1)
__device__ __forceinline__ int cap_bin(uint32_t key) {
  const uint32_t asc = ~key;
  if (!(asc & 0x80000000u)) return 0;
  const int b = (int)((asc & 0x7fffffffu) >> 16) - ((127 - 32) << 7);
  return b < 0 ? 0 : (b > kCapBins - 1 ? kCapBins - 1 : b);
}

__global__ __launch_bounds__(1024) void det_cap_kernel(double* __restrict__ dets, int* __restrict__ counts,
                                                       int* __restrict__ keep_idx, int R, int ncls,
                                                       int max_per_image) {
  __shared__ uint32_t keys[kCapMaxKeys];
  __shared__ uint32_t hist[kCapBins];
  __shared__ uint32_t list[kCapListMax];
  __shared__ int misc[40];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  dets += (size_t)blockIdx.x * ncls * R * 5; counts += (size_t)blockIdx.x * ncls;      // image blockIdx.x
  if (keep_idx) keep_idx += (size_t)blockIdx.x * ncls * R;
  if (tid == 0) {
    int t = 0;
    for (int j = 1; j < ncls; ++j) t += counts[j];
    misc[0] = t;
    misc[3] = 0;      // members of the threshold bin collected so far
  }
  for (int i = tid; i < kCapBins; i += 1024) hist[i] = 0;
  __syncthreads();
  const int total = misc[0];
  if (total <= max_per_image) return;
  // stage the survivors' keys at flat position j*R + k: a wave per class walks only the k < counts[j] that exist
  const int span = ncls * R;
  for (int j = 1 + wid; j < ncls; j += 16) {
    const int cj = counts[j];
    for (int k = lane; k < cj; k += 64) {
      const uint32_t key = desc_key((float)dets[((size_t)j * R + k) * 5 + 4]);
      keys[j * R + k] = key;
      atomicAdd(&hist[cap_bin(key)], 1u);
    }
  }
  __syncthreads();
  // thread t owns bins 4095-4t .. 4092-4t (best scores first); `above` = survivors in better bins
  {
    uint32_t c[4];
    int mine = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) { c[k] = hist[kCapBins - 1 - (4 * tid + k)]; mine += (int)c[k]; }
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const int o = __shfl_up(incl, d, 64);
      if (lane >= d) incl += o;
    }
    if (lane == 63) misc[8 + wid] = incl;
    __syncthreads();
    int above = incl - mine;
    for (int w = 0; w < wid; ++w) above += misc[8 + w];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (above < max_per_image && above + (int)c[k] >= max_per_image) {
        misc[1] = kCapBins - 1 - (4 * tid + k);       // the threshold bin
        misc[2] = max_per_image - above;              // image_thresh is its misc[2]-th best member (1-based)
        misc[4] = (int)c[k];
      }
      above += (int)c[k];
    }
  }
  __syncthreads();
  const int tbin = misc[1], want = misc[2], members = misc[4];
  uint32_t T = 0;
  if (members <= kCapListMax) {
    for (int j = 1 + wid; j < ncls; j += 16) {
      const int cj = counts[j];
      for (int k = lane; k < cj; k += 64) {
        const uint32_t key = keys[j * R + k];
        if (cap_bin(key) == tbin) list[atomicAdd(&misc[3], 1)] = key;
      }
    }
    __syncthreads();
    for (int i = tid; i < members; i += 1024) {
      const uint32_t ki = list[i];
      int better = 0, not_worse = 0;
      for (int k = 0; k < members; ++k) {
        const uint32_t kk = list[k];
        better += kk < ki;
        not_worse += kk <= ki;
      }
      if (better < want && want <= not_worse) misc[5] = (int)ki;    // every writer writes the same key
    }
    __syncthreads();
    T = (uint32_t)misc[5];
  } else {
    for (int i = tid; i < span; i += 1024) {          // the radix passes walk the whole span: absent slots get the largest key
      const int j = i / R, k = i - j * R;
      if (!(j >= 1 && k < counts[j])) keys[i] = 0xFFFFFFFFu;
    }
    __syncthreads();
    uint32_t* rhist = list;       // 1024 words = 4 x 256 replicated radix histograms
    uint32_t prefix = 0;
    int remaining = max_per_image;
    for (int shift = 24; shift >= 0; shift -= 8) {
      for (int i = tid; i < 4 * 256; i += 1024) rhist[i] = 0;
      __syncthreads();
      uint32_t* myhist = rhist + (lane & 3) * 256;
      for (int i = tid; i < span; i += 1024) {
        const uint32_t key = keys[i];
        const bool match = (shift == 24) || (((key ^ prefix) >> (shift + 8)) == 0);
        if (match) atomicAdd(&myhist[(key >> shift) & 255u], 1u);
      }
      __syncthreads();
      int c = 0, incl = 0;
      if (tid < 256) {
        c = (int)(rhist[tid] + rhist[256 + tid] + rhist[512 + tid] + rhist[768 + tid]);
        incl = c;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
          const int o = __shfl_up(incl, d, 64);
          if (lane >= d) incl += o;
        }
        if (lane == 63) misc[8 + wid] = incl;
      }
      __syncthreads();
      if (tid < 256) {
        for (int w = 0; w < wid; ++w) incl += misc[8 + w];
        if (incl >= remaining && incl - c < remaining) { misc[1] = tid; misc[2] = remaining - (incl - c); }
      }
      __syncthreads();
      prefix |= (uint32_t)misc[1] << shift;
      remaining = misc[2];
      __syncthreads();
    }
    T = prefix;
  }
  // keep score >= image_thresh  <=>  key <= T
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
        flag = keys[j * R + k] <= T;
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
  return (size_t)R * 4 * 8 * 2 + (size_t)R * 8 * 2 + (size_t)R * Wd * 8 + (size_t)R * 8 + (size_t)R * 4 * 4 + 128;
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
  return lsfa_det_postprocess_batch(rois, deltas, probs, 1, R, ncls, nreg, class_agnostic, im_h, im_w, scale, score_thresh, nms_thresh,
                                    max_per_image, dets, counts, keep_idx, ws, ws_bytes, stream);
}

extern "C" int lsfa_det_postprocess_batch(const float* rois, const float* deltas, const float* probs, int B, int R, int ncls,
                                          int nreg, int class_agnostic, double im_h, double im_w, double scale,
                                          double score_thresh, double nms_thresh, int max_per_image, double* dets,
                                          int* counts, int* keep_idx, void* ws, size_t ws_bytes, void* stream) {
  (void)ws; (void)ws_bytes;
  LSFA_REQUIRE(rois && deltas && probs && dets && counts, "lsfa_det_postprocess: NULL argument");
  LSFA_REQUIRE(B > 0 && B <= 65535 && R > 0 && ncls > 1 && nreg > 0, "lsfa_det_postprocess: bad shape B=%d R=%d ncls=%d nreg=%d", B, R, ncls, nreg);
  LSFA_REQUIRE(class_agnostic ? nreg >= 2 : nreg >= ncls, "lsfa_det_postprocess: nreg=%d too small", nreg);
  if (R > kMaxR || (long)R * ncls > kCapMaxKeys) {
    set_error("lsfa_det_postprocess: R=%d (max %d) or R*ncls=%ld (max %d) unsupported", R, kMaxR, (long)R * ncls, kCapMaxKeys);
    return LSFA_ENOTSUP;
  }
  hipStream_t s = (hipStream_t)stream;
  const size_t lds = class_lds_bytes(R);
  static PerDeviceOnce lds_attr;
  lds_attr.run([] {
    (void)hipFuncSetAttribute((const void*)det_class_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  });
  ProfScope prof(LSFA_OP_DET, s);
  NmsTest64 nms;
  nms.thresh = nms_thresh;
  nms.thresh_eps = nms_thresh * 4.440892098500626e-16;   // 2^-51
  nms.fast = (nms_thresh > 1e-300 && nms_thresh < 1e300) ? 1 : 0;
  hipLaunchKernelGGL(det_class_kernel, dim3(ncls, B), dim3(kClassThreads), lds, s, rois, deltas, probs, R, ncls, nreg,
                     class_agnostic, im_h, im_w, scale, score_thresh, nms, dets, counts, keep_idx);
  if (max_per_image > 0)
    hipLaunchKernelGGL(det_cap_kernel, dim3(B), dim3(1024), 0, s, dets, counts, keep_idx, R, ncls, max_per_image);
  LSFA_LAUNCH_CHECK("lsfa_det_postprocess");
  return LSFA_OK;
}
