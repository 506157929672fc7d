// Device kernels shared by nms.hip (lsfa_nms_sorted, _nms) and proposal.hip.
//
// Greedy NMS over score-sorted boxes, wave64-native (replaces nms_kernel + D2H of the 4.5 MB mask +
// serial host sweep + H2D of the keep list, lib/nms/nms_kernel.cu:40-150, multi_proposal.cu:262-357):
//
//   1. nms_mask_kernel — one 64-lane WAVE per 64x64 tile of the upper triangle (4 tiles per 256-thread
//      workgroup, tiles enumerated linearly so no workgroup is launched for the lower triangle).  A
//      lane owns one ROW box in registers; the 64 COLUMN boxes sit one per lane and are broadcast with
//      v_readlane (no LDS tile, no barrier).  Bit k of word (i, cb) is set iff IoU(box_i, box_{64cb+k})
//      > thresh and 64cb+k > i, as in the reference.  On diagonal tiles the same predicate, balloted
//      across the wave, is the TRANSPOSED word (the rows that suppress column k): it is written to
//      diagT so the sweep can resolve a block without a 64-step serial loop.
//      IoU > thresh is decided WITHOUT the division, bit for bit (IouTest below).
//   2. nms_sweep_kernel — ONE wave walks the 64-box blocks in score order.  A block is resolved as a
//      fixpoint on ballots (G(k) = alive(k) && no earlier member of G suppresses k: a handful of
//      iterations, not 64 scalar steps); the survivors' mask rows are ORed into the removed-set in
//      LDS, 8 rows per batch of loads; it stops at `max_keep` survivors.  With `rois` set it also
//      writes the Proposal output (first post_n survivors, cyclic pad, multi_proposal.cu:363-388).
// devIoU arithmetic = oracle dev_iou (nms_kernel.cu:30-38), -ffp-contract=off.
#pragma once
#include <math.h>

#include "common.h"
#include "nms_block.h"

namespace lsfa {

// fl32(inter / uni) > thresh, decided exactly without dividing.  With uni > 0 and 0 < thresh:
//   v = inter - thresh*uni (exact) and d = fmaf(-thresh, uni, inter) = fl(v) have the same sign;
//   v <= 0            =>  inter/uni <= thresh           => the rounded quotient is <= thresh (rounding is monotone)
//   v >= thresh*uni*u =>  inter/uni >= thresh*(1 + u)   => >= the float after thresh (u = 2^-23), so the quotient is > thresh
// and d >= thresh*uni*2^-22 implies the second case.  The sliver in between (and non-positive or NaN
// unions, and thresholds outside (1e-30, 1e30)) takes the division.  Callers test `unsure` wave-wide.
struct IouTest {
  float thresh;
  float thresh_eps;   // thresh * 2^-22
  int fast;
};

static inline IouTest make_iou_test(float thresh) {
  IouTest t;
  t.thresh = thresh;
  t.thresh_eps = thresh * 2.384185791015625e-07f;
  t.fast = (thresh > 1e-30f && thresh < 1e30f) ? 1 : 0;
  return t;
}

__device__ __forceinline__ float box_area(const float4& b) { return (b.z - b.x + 1) * (b.w - b.y + 1); }

__device__ __forceinline__ bool iou_exceeds_fast(const float4& a, float Sa, const float4& b, float Sb, const IouTest& t,
                                                 bool& unsure) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  const float interS = width * height;
  const float uni = Sa + Sb - interS;
  const float d = fmaf(-t.thresh, uni, interS);
  const bool pos = d > 0.f, clear = d >= t.thresh_eps * uni;
  unsure = unsure || !(uni > 0.f) || (pos && !clear);
  return pos && clear;
}
__device__ __forceinline__ bool iou_exceeds_div(const float4& a, float Sa, const float4& b, float Sb, const IouTest& t) {
  const float left = fmaxf(a.x, b.x), right = fminf(a.z, b.z);
  const float top = fmaxf(a.y, b.y), bottom = fminf(a.w, b.w);
  const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  const float interS = width * height;
  return interS / (Sa + Sb - interS) > t.thresh;
}

__device__ __forceinline__ float4 bcast4(const float4& v, int src_lane) {
  return make_float4(__uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v.x), src_lane)),
                     __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v.y), src_lane)),
                     __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v.z), src_lane)),
                     __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v.w), src_lane)));
}
__device__ __forceinline__ float bcast1(float v, int src_lane) {
  return __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src_lane));
}

// number of upper-triangle tiles (diagonal included) of a col_blocks x col_blocks tile grid
__host__ __device__ static inline int nms_tile_count(int col_blocks) { return col_blocks * (col_blocks + 1) / 2; }

// grid (ceil(tiles / 4), 1, images); block 256 = 4 waves = 4 tiles.
// boxes: (images, n_total, box_dim) floats; `order` (images, n) int32 or NULL: row p of the sorted list is
// boxes[order[p]] (Proposal: candidates stay where the decode kernel put them).  mask: (images, n, col_blocks);
// diagT: (images, n).
// PACKED4: boxes are a plain float4 array (Proposal's sorted boxes) and are loaded as one 16-byte access.
// The 64 column boxes of a tile are staged once in LDS (one per lane) and read back at wave-uniform addresses
// (a broadcast read per column, issued ahead by the unrolled loop); the loop body is the IoU test and a bit
// insert only — row / column validity and the j > i rule of diagonal tiles are applied to the finished words,
// and the transposed diagonal words come from 64 ballots over the finished rows (diagonal tiles only).
template <bool PACKED4>
static __global__ __launch_bounds__(256) void nms_mask_kernel(const float* __restrict__ boxes_all, long boxes_img_stride,
                                                              int box_dim, const int* __restrict__ order_all, int n,
                                                              IouTest t, uint64_t* __restrict__ mask_all,
                                                              uint64_t* __restrict__ diagT_all, int col_blocks, int diag_only) {
  __shared__ float4 cbox[4][64];
  __shared__ float carea[4][64];
  const int lane = threadIdx.x & 63;
  // wave-uniform, and the compiler is told so: the tile indices and the `diag` branch stay scalar
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int tile = blockIdx.x * 4 + wv;
  const bool tile_ok = tile < (diag_only ? col_blocks : nms_tile_count(col_blocks));
  // tile -> (rb <= cb), tiles enumerated column by column: tile = cb*(cb+1)/2 + rb; diag_only: the col_blocks diagonal tiles only
  // (the Proposal's sweep decides everything off the diagonal from the boxes themselves: nms_sweep_iou_kernel)
  int cb = diag_only ? tile : (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
  if (!diag_only) {
    while ((cb + 1) * (cb + 2) / 2 <= tile) ++cb;
    while (cb * (cb + 1) / 2 > tile) --cb;
  }
  const int rb = diag_only ? tile : tile - cb * (cb + 1) / 2;
  const int img = blockIdx.z;
  const float* boxes = boxes_all + (size_t)img * boxes_img_stride;
  const int* order = order_all ? order_all + (size_t)img * n : nullptr;
  uint64_t* mask = mask_all + (size_t)img * n * col_blocks;

  const int row = rb * 64 + lane, col = cb * 64 + lane;
  const bool row_ok = tile_ok && row < n, col_ok = tile_ok && col < n;
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = make_float4(0.f, 0.f, 0.f, 0.f);
  if (row_ok) {
    if (PACKED4) a = reinterpret_cast<const float4*>(boxes)[row];
    else { const float* p = boxes + (size_t)(order ? order[row] : row) * box_dim; a = make_float4(p[0], p[1], p[2], p[3]); }
  }
  if (col_ok) {
    if (PACKED4) c = reinterpret_cast<const float4*>(boxes)[col];
    else { const float* p = boxes + (size_t)(order ? order[col] : col) * box_dim; c = make_float4(p[0], p[1], p[2], p[3]); }
  }
  const float Sa = box_area(a);
  cbox[wv][lane] = c;
  carea[wv][lane] = box_area(c);
  __syncthreads();
  if (!tile_ok) return;
  const int ncol = min(64, n - cb * 64);
  const bool diag = rb == cb;
  uint32_t word[2] = {0, 0};
  bool unsure = !t.fast;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    uint32_t w = 0;
#pragma unroll 8
    for (int jj = 0; jj < 32; ++jj) {
      const int j = half * 32 + jj;
      const bool p = iou_exceeds_fast(a, Sa, cbox[wv][j], carea[wv][j], t, unsure);
      w |= (uint32_t)p << jj;
    }
    word[half] = w;
  }
  if (__builtin_expect(__any(unsure), 0)) {   // wave-uniform and rare: redo the tile with the real division
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      uint32_t w = 0;
      for (int jj = 0; jj < 32; ++jj) {
        const int j = half * 32 + jj;
        w |= (uint32_t)iou_exceeds_div(a, Sa, cbox[wv][j], carea[wv][j], t) << jj;
      }
      word[half] = w;
    }
  }
  uint64_t bits = ((uint64_t)word[1] << 32) | word[0];
  if (ncol < 64) bits &= (1ULL << ncol) - 1ULL;                       // columns past the last box
  if (diag) bits &= lane == 63 ? 0ULL : ~((2ULL << lane) - 1ULL);      // in a diagonal tile only j > i counts
  if (!row_ok) bits = 0;
  if (row_ok) mask[(size_t)row * col_blocks + cb] = bits;
  if (diag) {
    // transpose the finished 64 x 64 bit block: word of column k = the rows that suppress k
    uint32_t tlo = 0, thi = 0;
#pragma unroll 8
    for (int k = 0; k < 64; ++k) {
      const unsigned long long colword = __ballot((bits >> k) & 1ULL);
      if (lane == k) { tlo = (uint32_t)colword; thi = (uint32_t)(colword >> 32); }
    }
    if (row_ok) diagT_all[(size_t)img * n + row] = ((uint64_t)thi << 32) | tlo;
  }
}

// ---- float64 boxes: lib/nms/nms.py:37-74 run on float64 dets (pred_eval's py_nms_wrapper) --------
// Same tiling as nms_mask_kernel; the column boxes are read at wave-uniform addresses (scalar loads).
// numpy keeps `ovr <= thresh`, so a pair suppresses when NOT (ovr <= thresh); the division-free test is
// IouTest's argument with doubles (u = 2^-52), the sliver and non-positive / NaN unions take the division.
struct IouTest64 { double thresh, thresh_eps; int fast; };
static inline IouTest64 make_iou_test64(double thresh) {
  IouTest64 t;
  t.thresh = thresh;
  t.thresh_eps = thresh * 4.440892098500626e-16;   // 2^-51
  t.fast = (thresh > 1e-300 && thresh < 1e300) ? 1 : 0;
  return t;
}

static __global__ __launch_bounds__(256) void nms_mask_f64_kernel(const double* __restrict__ boxes, int box_dim, int n,
                                                                  IouTest64 t, uint64_t* __restrict__ mask,
                                                                  uint64_t* __restrict__ diagT, int col_blocks) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (tile >= nms_tile_count(col_blocks)) return;
  int cb = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
  while ((cb + 1) * (cb + 2) / 2 <= tile) ++cb;
  while (cb * (cb + 1) / 2 > tile) --cb;
  const int rb = tile - cb * (cb + 1) / 2;
  const int row = rb * 64 + lane;
  const bool row_ok = row < n;
  double ax1 = 0, ay1 = 0, ax2 = 0, ay2 = 0;
  if (row_ok) {
    const double* p = boxes + (size_t)row * box_dim;
    ax1 = p[0]; ay1 = p[1]; ax2 = p[2]; ay2 = p[3];
  }
  const double Sa = (ax2 - ax1 + 1) * (ay2 - ay1 + 1);
  const int ncol = min(64, n - cb * 64);
  const bool diag = rb == cb;
  uint64_t bits = 0, tbits = 0;
  for (int j = 0; j < ncol; ++j) {
    const double* q = boxes + (size_t)(cb * 64 + j) * box_dim;     // wave-uniform
    const double bx1 = q[0], by1 = q[1], bx2 = q[2], by2 = q[3];
    const double Sb = (bx2 - bx1 + 1) * (by2 - by1 + 1);
    const double w = fmax(0.0, fmin(ax2, bx2) - fmax(ax1, bx1) + 1), h = fmax(0.0, fmin(ay2, by2) - fmax(ay1, by1) + 1);
    const double inter = w * h;
    const double uni = Sa + Sb - inter;
    const double d = fma(-t.thresh, uni, inter);
    const bool pos = d > 0.0, clear = d >= t.thresh_eps * uni;
    bool p = pos && clear;
    const bool unsure = !t.fast || !(uni > 0.0) || (pos && !clear);
    if (__builtin_expect(__any(unsure), 0)) {
      if (unsure) p = !(inter / uni <= t.thresh);
    }
    p = p && row_ok && (!diag || j > lane);
    bits |= (uint64_t)p << j;
    if (diag) {
      const unsigned long long colword = __ballot(p);
      if (lane == j) tbits = colword;
    }
  }
  if (row_ok) mask[(size_t)row * col_blocks + cb] = bits;
  if (diag && row_ok) diagT[row] = tbits;
}

constexpr int kSweepMaxBlocks = 512;   // n <= 32768
constexpr int kSweepMaxOut = 1024;     // Proposal output rows
constexpr int kSweepLazyMax = 4096;
constexpr int kSweepPairMax = 320;     // survivors the two-blocks-per-trip sweep gathers with five loads per lane    // survivors whose positions fit the LDS list of the lazy sweep

__device__ __forceinline__ uint64_t wave_or64(uint64_t v) {
  uint32_t lo = (uint32_t)v, hi = (uint32_t)(v >> 32);
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) { lo |= __shfl_xor(lo, d, 64); hi |= __shfl_xor(hi, d, 64); }
  return ((uint64_t)hi << 32) | lo;
}

// OR over the wave, returned as a wave-uniform value, on the DPP cross-lane paths: quad swaps, half-row and row
// mirrors (every lane of a 16-lane row then holds the row's OR), row_bcast15 into rows 1 and 3, row_bcast31 into
// rows 2 and 3, read lane 63.  About 25 VALU instructions and no LDS traffic (the ds_bpermute butterfly of
// wave_or64 is 12 dependent LDS-crossbar round trips).
__device__ __forceinline__ uint32_t wave_or32_dpp(uint32_t x) {
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0xB1, 0xF, 0xF, false);    // quad_perm [1,0,3,2]
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x4E, 0xF, 0xF, false);    // quad_perm [2,3,0,1]
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x141, 0xF, 0xF, false);   // row_half_mirror
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x140, 0xF, 0xF, false);   // row_mirror
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);   // row_bcast15 -> rows 1, 3
  x |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);   // row_bcast31 -> rows 2, 3
  return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ __forceinline__ uint64_t wave_or64_uniform(uint64_t v) {
  return ((uint64_t)wave_or32_dpp((uint32_t)(v >> 32)) << 32) | wave_or32_dpp((uint32_t)v);
}

// grid (images); block 64 (one wave).  keep: (images, n) positions in the sorted list, or NULL; num_keep:
// (images) or NULL.  Proposal epilogue when rois != NULL: sorted_box (images, n) float4, sorted_key (images, n)
// order keys (for `scores`), rois (images*post_n, 5), scores (images*post_n) or NULL; max_keep = post_n.
//
// Which candidates of block b are already suppressed is found LAZILY when max_keep <= kSweepLazyMax: the
// survivors so far sit in an LDS list, each lane gathers one word of a survivor's mask row and the lanes OR
// them into the next block's removed-set word (nothing is ever loaded for blocks the sweep does not reach —
// Proposal stops after 300 survivors, usually within a few blocks — and the loads for block b+1 are in
// flight while block b is resolved).
// Larger keep budgets use the eager form: every survivor's whole row is ORed into a removed-set in LDS.
static __global__ __launch_bounds__(64) void nms_sweep_kernel(const uint64_t* __restrict__ mask_all,
                                                              const uint64_t* __restrict__ diagT_all, int n, int col_blocks,
                                                              int max_keep, int* __restrict__ keep_all,
                                                              int* __restrict__ num_keep_all,
                                                              const float4* __restrict__ sorted_box_all,
                                                              const uint32_t* __restrict__ sorted_key_all,
                                                              float* __restrict__ rois, float* __restrict__ scores) {
  __shared__ uint64_t remv[kSweepMaxBlocks];
  __shared__ int kept_pos[kSweepLazyMax];
#ifdef LSFA_SWEEP_PRIO      // lab (tools/lab/build_variant.sh): the one wave of a sweep is a latency chain; under the pipeline's contention it takes 67 us for 12 us of work
  __builtin_amdgcn_s_setprio(LSFA_SWEEP_PRIO);
#endif
  const int img = blockIdx.x;
  const uint64_t* mask = mask_all + (size_t)img * n * col_blocks;
  const uint64_t* diagT = diagT_all + (size_t)img * n;
  int* keep = keep_all ? keep_all + (size_t)img * n : nullptr;
  const int lane = threadIdx.x;
  const bool lazy = max_keep <= kSweepLazyMax;
  if (!lazy) {
    for (int i = lane; i < col_blocks; i += 64) remv[i] = 0;
  }
  __syncthreads();  // single-wave workgroup: orders the LDS accesses of different lanes
  int num = 0;
  if (lazy && max_keep <= kSweepPairMax) {
    // Two blocks per trip to memory (Proposal: 300 survivors).  For a pair (b, b+1) everything that does not depend on
    // the pair's own survivors was put together one iteration earlier: cur0 = removed(b), and cur1 = removed(b+1)
    // WITHOUT block b's contribution, which is added after block b is resolved from `intra` = word b+1 of block b's
    // rows (state-independent, loaded an iteration ahead).  While the pair is resolved, the loads for the NEXT pair are
    // in flight: word b+2 / b+3 of the rows that survived before b (gathered through kept_pos), of block b's and
    // block b+1's own rows (all 64 each, the survivors' words selected afterwards), the next pair's diagonal words and
    // its `intra`.  On heavy-overlap inputs (3-5 survivors per block, 70-90 blocks to reach 300) a block's resolution is
    // far shorter than a trip to L2/MALL, so halving the trips nearly halves the sweep.
    uint64_t colw0 = lane < n ? diagT[lane] : 0ULL, rowd0 = lane < n ? mask[(size_t)lane * col_blocks] : 0ULL;
    uint64_t colw1 = 0, rowd1 = 0, intra = 0;
    if (col_blocks > 1) {
      const int r1 = min(64 + lane, n - 1);
      const uint64_t cw = diagT[r1], rd = mask[(size_t)r1 * col_blocks + 1], it = mask[(size_t)min(lane, n - 1) * col_blocks + 1];
      colw1 = 64 + lane < n ? cw : 0ULL;
      rowd1 = 64 + lane < n ? rd : 0ULL;
      intra = lane < n ? it : 0ULL;
    }
    uint64_t cur0 = 0, cur1 = 0;
    for (int b = 0; b < col_blocks && num < max_keep; b += 2) {
      const bool more = b + 2 < col_blocks, more3 = b + 3 < col_blocks;
      // ---- requests for the next pair (t2 = b + 2, t3 = b + 3); unconditional loads on clamped indices ----
      const int t2 = more ? b + 2 : b, t3 = more3 ? b + 3 : t2;
      const uint64_t* m2 = mask + t2;
      const uint64_t* m3 = mask + t3;
      const int k0 = lane < num ? kept_pos[lane] : 0, k1 = lane + 64 < num ? kept_pos[lane + 64] : 0;
      const int k2 = lane + 128 < num ? kept_pos[lane + 128] : 0, k3 = lane + 192 < num ? kept_pos[lane + 192] : 0;
      const int k4 = lane + 256 < num ? kept_pos[lane + 256] : 0;
      const int ra = min(b * 64 + lane, n - 1), rb = min(b * 64 + 64 + lane, n - 1);
      const int rc = min(t2 * 64 + lane, n - 1), rd_ = min(t3 * 64 + lane, n - 1);
      const uint64_t g20 = m2[(size_t)k0 * col_blocks], g21 = m2[(size_t)k1 * col_blocks], g22 = m2[(size_t)k2 * col_blocks];
      const uint64_t g23 = m2[(size_t)k3 * col_blocks], g24 = m2[(size_t)k4 * col_blocks];
      const uint64_t g30 = m3[(size_t)k0 * col_blocks], g31 = m3[(size_t)k1 * col_blocks], g32 = m3[(size_t)k2 * col_blocks];
      const uint64_t g33 = m3[(size_t)k3 * col_blocks], g34 = m3[(size_t)k4 * col_blocks];
      const uint64_t sa2 = m2[(size_t)ra * col_blocks], sa3 = m3[(size_t)ra * col_blocks];      // block b's rows
      // block b+1's rows.  When there is no next pair (t2 / t3 clamped to b) word b of those rows lies LEFT of the diagonal,
      // which nms_mask_kernel never writes: the value would be discarded below (cur0 / cur1 are forced to 0), but read a
      // word that exists instead (their own diagonal word, or the last row's) so that nothing uninitialised is ever loaded
      const int tb2 = min(max(t2, b + 1), col_blocks - 1), tb3 = min(max(t3, b + 1), col_blocks - 1);
      const uint64_t sb2 = mask[(size_t)rb * col_blocks + tb2], sb3 = mask[(size_t)rb * col_blocks + tb3];
      const uint64_t n_colw0 = diagT[rc], n_rowd0 = m2[(size_t)rc * col_blocks], n_intra = m3[(size_t)rc * col_blocks];
      const uint64_t n_colw1 = diagT[rd_], n_rowd1 = m3[(size_t)rd_ * col_blocks];
      const bool h0 = lane < num, h1 = lane + 64 < num, h2 = lane + 128 < num, h3 = lane + 192 < num, h4 = lane + 256 < num;
      // ---- block b ----
      const int nb0 = min(64, n - b * 64);
      const bool alive0 = lane < nb0 && !((cur0 >> lane) & 1ULL);
      const uint64_t kept0 = resolve_block(__ballot(alive0), alive0, colw0, rowd0, max_keep - num);
      const bool mine0 = (kept0 >> lane) & 1ULL;
      if (mine0) {
        const int pos = num + __popcll(kept0 & ((1ULL << lane) - 1ULL));
        if (keep) keep[pos] = b * 64 + lane;
        kept_pos[pos] = b * 64 + lane;
      }
      num += __popcll(kept0);
      // ---- block b + 1 ----
      uint64_t kept1 = 0;
      bool mine1 = false;
      if (b + 1 < col_blocks && num < max_keep) {
        const uint64_t c1 = cur1 | wave_or64_uniform(mine0 ? intra : 0ULL);
        const int nb1 = min(64, n - (b + 1) * 64);
        const bool alive1 = lane < nb1 && !((c1 >> lane) & 1ULL);
        kept1 = resolve_block(__ballot(alive1), alive1, colw1, rowd1, max_keep - num);
        mine1 = (kept1 >> lane) & 1ULL;
        if (mine1) {
          const int pos = num + __popcll(kept1 & ((1ULL << lane) - 1ULL));
          if (keep) keep[pos] = (b + 1) * 64 + lane;
          kept_pos[pos] = (b + 1) * 64 + lane;
        }
        num += __popcll(kept1);
      }
      // ---- the next pair's state ----
      const bool va = b * 64 + lane < n, vb = b * 64 + 64 + lane < n;
      const uint64_t x2 = (h0 ? g20 : 0ULL) | (h1 ? g21 : 0ULL) | (h2 ? g22 : 0ULL) | (h3 ? g23 : 0ULL) | (h4 ? g24 : 0ULL) |
                          ((mine0 && va) ? sa2 : 0ULL) | ((mine1 && vb) ? sb2 : 0ULL);
      const uint64_t x3 = (h0 ? g30 : 0ULL) | (h1 ? g31 : 0ULL) | (h2 ? g32 : 0ULL) | (h3 ? g33 : 0ULL) | (h4 ? g34 : 0ULL) |
                          ((mine0 && va) ? sa3 : 0ULL) | ((mine1 && vb) ? sb3 : 0ULL);
      cur0 = more ? wave_or64_uniform(x2) : 0ULL;
      cur1 = more3 ? wave_or64_uniform(x3) : 0ULL;
      const bool vc = more && t2 * 64 + lane < n, vd = more3 && t3 * 64 + lane < n;
      colw0 = vc ? n_colw0 : 0ULL; rowd0 = vc ? n_rowd0 : 0ULL;
      intra = (vc && more3) ? n_intra : 0ULL;
      colw1 = vd ? n_colw1 : 0ULL; rowd1 = vd ? n_rowd1 : 0ULL;
      __syncthreads();
    }
  } else if (lazy) {
    // Software-pipelined: everything block b+1 needs from memory is requested before block b is resolved.
    //   removed(b+1) = OR of word b+1 of the rows kept in blocks < b   (gathered through kept_pos: known now)
    //                | OR of word b+1 of the rows block b keeps         (all 64 rows of block b are loaded
    //                                                                    speculatively, the kept lanes contribute)
    // so one block costs a ballot fixpoint and one cross-lane OR, not a dependent trip to memory.
    uint64_t colw = lane < n ? diagT[lane] : 0ULL;
    uint64_t rowd = lane < n ? mask[(size_t)lane * col_blocks] : 0ULL;
    // vmcnt(0) here, so that inside the loop the compiler knows colw / rowd have landed and their uses do not
    // wait for the loads the same iteration has just issued (the counter retires in order)
    __builtin_amdgcn_s_waitcnt(0x0F70);
    uint64_t cur = 0;     // removed-set word of the block being resolved (wave-uniform)
    for (int b = 0; b < col_blocks && num < max_keep; ++b) {
      const int base = b * 64;
      const int nb = min(64, n - base);
      const bool more = b + 1 < col_blocks;
      uint64_t col_n = 0, rowd_n = 0, spec = 0, v0 = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0, vrest = 0;
      if (more) {
        const int nxt = base + 64 + lane;
        if (nxt < n) { col_n = diagT[nxt]; rowd_n = mask[(size_t)nxt * col_blocks + b + 1]; }
        if (lane < nb) spec = mask[(size_t)(base + lane) * col_blocks + b + 1];
        const uint64_t* mcol = mask + b + 1;
        if (lane < num) v0 = mcol[(size_t)kept_pos[lane] * col_blocks];
        if (lane + 64 < num) v1 = mcol[(size_t)kept_pos[lane + 64] * col_blocks];
        if (lane + 128 < num) v2 = mcol[(size_t)kept_pos[lane + 128] * col_blocks];
        if (lane + 192 < num) v3 = mcol[(size_t)kept_pos[lane + 192] * col_blocks];
        if (lane + 256 < num) v4 = mcol[(size_t)kept_pos[lane + 256] * col_blocks];
        for (int k = lane + 320; k < num; k += 64) vrest |= mcol[(size_t)kept_pos[k] * col_blocks];
      }
      const bool alive = lane < nb && !((cur >> lane) & 1ULL);
      const uint64_t kept = resolve_block(__ballot(alive), alive, colw, rowd, max_keep - num);
      const bool mine = (kept >> lane) & 1ULL;
      if (mine) {
        const int pos = num + __popcll(kept & ((1ULL << lane) - 1ULL));
        if (keep) keep[pos] = base + lane;
        kept_pos[pos] = base + lane;
      }
      num += __popcll(kept);
      if (more) cur = wave_or64_uniform(v0 | v1 | v2 | v3 | v4 | vrest | (mine ? spec : 0ULL));
      colw = col_n;
      rowd = rowd_n;
      __syncthreads();
    }
  } else {
  uint64_t col_next = lane < n ? diagT[lane] : 0ULL;
  for (int b = 0; b < col_blocks && num < max_keep; ++b) {
    const int base = b * 64;
    const int nb = min(64, n - base);
    const uint64_t colw = col_next;
    const int nxt = base + 64 + lane;
    col_next = nxt < n ? diagT[nxt] : 0ULL;     // next block's transposed diagonal: independent of the sweep state
    const uint64_t cur = remv[b];
    const bool alive = lane < nb && !((cur >> lane) & 1ULL);
    const uint64_t rowd = lane < nb ? mask[(size_t)(base + lane) * col_blocks + b] : 0ULL;
    const uint64_t kept = resolve_block(__ballot(alive), alive, colw, rowd, max_keep - num);
    if (kept == 0) continue;
    if ((kept >> lane) & 1ULL) {
      const int pos = num + __popcll(kept & ((1ULL << lane) - 1ULL));
      if (keep) keep[pos] = base + lane;
    }
    num += __popcll(kept);
    __syncthreads();
    if (num >= max_keep || b + 1 >= col_blocks) break;
    // eager: OR the survivors' rows into remv for the blocks to the right of b, 8 rows per batch of loads
    uint64_t rem = kept;
    while (rem) {
      int ks[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (rem) { ks[u] = __builtin_ctzll(rem); rem &= rem - 1; } else ks[u] = -1;
      }
      for (int w0 = b + 1; w0 < col_blocks; w0 += 64) {
        const int w = w0 + lane;
        uint64_t v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
          v[u] = (ks[u] >= 0 && w < col_blocks) ? mask[(size_t)(base + ks[u]) * col_blocks + w] : 0ULL;
        uint64_t acc = 0;
#pragma unroll
        for (int u = 0; u < 8; ++u) acc |= v[u];
        if (w < col_blocks) remv[w] |= acc;
      }
    }
    __syncthreads();
  }
  }
  if (num_keep_all && lane == 0) num_keep_all[img] = num;
  if (rois) {      // Proposal: max_keep = post_n <= kSweepMaxOut <= kSweepLazyMax, so kept_pos holds the survivors
    __syncthreads();
    const float4* sbox = sorted_box_all + (size_t)img * n;
    const uint32_t* skey = sorted_key_all + (size_t)img * n;
    for (int index = lane; index < max_keep; index += 64) {
      const int p = kept_pos[index < num ? index : index % num];
      const float4 bx = sbox[p];
      float* o = rois + ((size_t)img * max_keep + index) * 5;
      o[0] = (float)img;
      o[1] = bx.x; o[2] = bx.y; o[3] = bx.z; o[4] = bx.w;
      if (scores) {
        const uint32_t asc = ~skey[p];
        const uint32_t u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
        scores[(size_t)img * max_keep + index] = __uint_as_float(u);
      }
    }
  }
}

// ---- r3 experiment (LSFA_PROPOSAL_PLAN_CHIP_WIDE_BOX_SWEEP): the Proposal's sweep without the suppression mask -------------------
// MEASURED SLOWER than mask + sweep (160 us against 18 + 48 on the benchmark's RPN): a wave64 VALU instruction occupies its SIMD for
// four cycles and a CU has four SIMDs, so the 64 x (survivors so far) IoU tests of a block - ~12 K wave-tests of ~25 instructions
// over the call - are 150 us of ONE CU's issue slots; the mask kernel spreads the same tests over the chip.  Kept as an opt-in plan
// because it is bit-exact and states the trade; frames/s did not move either way (the pipeline's throughput is set by the chip-filling
// kernels; a one-CU kernel's latency overlaps the other streams' work).  The reasoning that led here:
// nms_sweep_kernel above pays one trip to L2 per pair of 64-box blocks for the mask words of the survivors so far: ~0.55 us a
// block, 70-90 blocks on the benchmark's untrained RPN (every anchor regresses to a large box, 3-5 survivors per block) = 48 us,
// behind an 18 us mask kernel that fills 4465 tiles of which the sweep reads a sliver.  The Proposal keeps at most post_n <= 1024
// boxes, so the survivors' boxes fit in LDS and "is candidate j suppressed by an earlier survivor" can be decided from the boxes
// themselves: one 16-wave workgroup; for block b every wave tests the block's 64 candidates (lane = candidate) against its share
// of the survivor list (survivor k goes to wave k % 16; its box is a broadcast LDS read), the 16 partial words are ORed, wave 0
// resolves the block with resolve_block (the block's own 64 x 64 IoU bits come from nms_mask_kernel run on the DIAGONAL tiles
// only: 94 tiles instead of 4465) and appends the survivors' boxes.  Nothing the loop loads from global memory depends on the
// sweep's state (the next block's boxes and diagonal words are requested a block ahead); a block costs two workgroup barriers and
// ~num/16 IoU tests per lane.  Decisions are the mask kernel's, bit for bit: the same iou_exceeds_fast / iou_exceeds_div, whose
// value does not depend on which of the two boxes is called the row.
constexpr int kSweepIouWaves = 16;

static __global__ __launch_bounds__(64 * kSweepIouWaves) void nms_sweep_iou_kernel(
    const float4* __restrict__ sorted_box_all, const uint32_t* __restrict__ sorted_key_all, const uint64_t* __restrict__ mask_all,
    const uint64_t* __restrict__ diagT_all, int n, int col_blocks, int max_keep, IouTest t, float* __restrict__ rois,
    float* __restrict__ scores) {
  __shared__ float4 kbox[kSweepMaxOut];
  __shared__ float karea[kSweepMaxOut];
  __shared__ int kept_pos[kSweepMaxOut];
  __shared__ uint64_t part[kSweepIouWaves];
  __shared__ int s_num;
  const int img = blockIdx.x;
  const float4* sbox = sorted_box_all + (size_t)img * n;
  const uint64_t* mask = mask_all + (size_t)img * n * col_blocks;
  const uint64_t* diagT = diagT_all + (size_t)img * n;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x == 0) s_num = 0;
  // block 0's operands
  float4 c_next = lane < n ? sbox[lane] : make_float4(0.f, 0.f, 0.f, 0.f);
  uint64_t colw_next = 0, rowd_next = 0;
  if (wave == 0 && lane < n) { colw_next = diagT[lane]; rowd_next = mask[(size_t)lane * col_blocks]; }
  __syncthreads();
  int num = 0;
  for (int b = 0; b < col_blocks && num < max_keep; ++b) {
    const float4 c = c_next;
    const uint64_t colw = colw_next, rowd = rowd_next;
    const int cand = b * 64 + lane;
    {   // requests for block b + 1: independent of what this block decides
      const int nc = cand + 64;
      c_next = nc < n ? sbox[nc] : make_float4(0.f, 0.f, 0.f, 0.f);
      if (wave == 0) {
        colw_next = nc < n ? diagT[nc] : 0ULL;
        rowd_next = nc < n ? mask[(size_t)nc * col_blocks + b + 1] : 0ULL;
      }
    }
    const float Sc = box_area(c);
    bool sup = false, unsure = !t.fast;
    for (int k = wave; k < num; k += kSweepIouWaves) sup = sup || iou_exceeds_fast(kbox[k], karea[k], c, Sc, t, unsure);
    if (__builtin_expect(__any(unsure), 0)) {     // rare: redo this wave's share with the real division
      sup = false;
      for (int k = wave; k < num; k += kSweepIouWaves) sup = sup || iou_exceeds_div(kbox[k], karea[k], c, Sc, t);
    }
    const uint64_t w = __ballot(sup);
    if (lane == 0) part[wave] = w;
    __syncthreads();
    if (wave == 0) {
      uint64_t cur = 0;
#pragma unroll
      for (int i = 0; i < kSweepIouWaves; ++i) cur |= part[i];
      const int nb = min(64, n - b * 64);
      const bool alive = lane < nb && !((cur >> lane) & 1ULL);
      const uint64_t kept = resolve_block(__ballot(alive), alive, colw, rowd, max_keep - num);
      if ((kept >> lane) & 1ULL) {
        const int pos = num + __popcll(kept & ((1ULL << lane) - 1ULL));
        kbox[pos] = c;
        karea[pos] = Sc;
        kept_pos[pos] = cand;
      }
      if (lane == 0) s_num = num + __popcll(kept);
    }
    __syncthreads();
    num = s_num;
  }
  // output: the first max_keep survivors, cyclic pad (PrepareOutput, multi_proposal.cu:363-388)
  for (int index = threadIdx.x; index < max_keep; index += 64 * kSweepIouWaves) {
    const int p = kept_pos[index < num ? index : index % num];
    const float4 bx = sbox[p];
    float* o = rois + ((size_t)img * max_keep + index) * 5;
    o[0] = (float)img;
    o[1] = bx.x; o[2] = bx.y; o[3] = bx.z; o[4] = bx.w;
    if (scores) {
      const uint32_t asc = ~sorted_key_all[(size_t)img * n + p];
      const uint32_t u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
      scores[(size_t)img * max_keep + index] = __uint_as_float(u);
    }
  }
}

}  // namespace lsfa
