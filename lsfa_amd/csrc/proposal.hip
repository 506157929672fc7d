// RPN proposal generation, fully device resident (no malloc/free, no D2H, no host sweep; the
// reference does 4 cudaMalloc/cudaFree, a thrust sort, a 4.5 MB D2H and a serial host loop per call,
// multi_proposal.cu:454-557).  Two launch plans, same results bit for bit:
//
// CHIP-WIDE plan (count <= 32768 anchors, pre_nms_top_n <= 8192: the LSFA shapes), 5 short kernels:
//   proposal_decode_kernel   anchors + decode + clip + filter (ProposalGrid/BBoxPred/FilterBox,
//                            multi_proposal.cu:47-216) -> float4 boxes + 32-bit order keys, and per
//                            workgroup a 4096-bin histogram of the keys (128 bins per octave of score)
//   proposal_compact_kernel  every workgroup sums the partial histograms, finds the bin the
//                            pre_nms_top_n-th best key falls in, and writes ITS keys at or above that bin
//                            into the candidate list GROUPED BY BIN (group offsets from the histograms;
//                            no global atomics, no counters to zero).  M candidates, M - pre_n surplus.
//   proposal_rank_kernel     exact position of a candidate in thrust::stable_sort_by_key(greater) order
//                            (:517-521) = start of its bin's group + candidates of its own bin ordered
//                            before it by (key, anchor index) — a handful of compares — then it writes
//                            its box and key at that position (rank < pre_n: the exact top pre_n, in order)
//   nms_mask_kernel + nms_sweep_kernel (nms_kernels.h)  64x64 IoU tiles of the upper triangle over the
//                            chip, then one wave sweeps, stops at post_n survivors and writes the
//                            output with the cyclic pad (PrepareOutput, :363-388)
// SINGLE-WORKGROUP plan (anything larger): proposal_select_nms_kernel, one 16-wave workgroup per
//   image with everything in its CU's LDS: radix select, stable LSD radix (or bitonic) sort, greedy
//   NMS with IoUs on the fly against the survivor list.
// Arithmetic = oracle orc_proposal_decode, operation for operation.
#include <math.h>
#include <string.h>

#include "nms_kernels.h"

using namespace lsfa;

namespace {

constexpr int kMaxAnchors = 32;
constexpr int kTopkThreads = 1024;

struct Anchors { float v[kMaxAnchors * 4]; };

// multi_proposal-inl.h:256-295 (host code in the reference too)
void generate_anchors(int feature_stride, const float* ratios, int nr, const float* scales, int ns, Anchors* out) {
  const float base[4] = {0.0f, 0.0f, (float)(feature_stride - 1.0), (float)(feature_stride - 1.0)};
  int n = 0;
  for (int j = 0; j < nr; ++j)
    for (int k = 0; k < ns; ++k) {
      const float scale = scales[k], ratio = ratios[j];
      const float w = base[2] - base[1] + 1.0f;
      const float h = base[3] - base[1] + 1.0f;
      const float x_ctr = (float)(base[0] + 0.5 * (w - 1.0f));
      const float y_ctr = (float)(base[1] + 0.5 * (h - 1.0f));
      const float size = w * h;
      const float size_ratios = floorf(size / ratio);
      const float new_w = floorf(sqrtf(size_ratios) + 0.5f) * scale;
      const float new_h = floorf((new_w / scale * ratio) + 0.5f) * scale;
      out->v[n * 4 + 0] = x_ctr - 0.5f * (new_w - 1.0f);
      out->v[n * 4 + 1] = y_ctr - 0.5f * (new_h - 1.0f);
      out->v[n * 4 + 2] = x_ctr + 0.5f * (new_w - 1.0f);
      out->v[n * 4 + 3] = y_ctr + 0.5f * (new_h - 1.0f);
      ++n;
    }
}

// ascending-orderable image of a float, complemented: smaller key == larger score
__device__ __forceinline__ uint32_t desc_key(float score) {
  score = score + 0.0f;  // -0 -> +0, so that equal scores tie like thrust::greater sees them
  const uint32_t u = __float_as_uint(score);
  const uint32_t asc = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
  return ~asc;
}
__device__ __forceinline__ float key_score(uint32_t key) {
  const uint32_t asc = ~key;
  const uint32_t u = (asc & 0x80000000u) ? (asc & 0x7fffffffu) : ~asc;
  return __uint_as_float(u);
}

constexpr int kDecodeThreads = 1024;
constexpr int kBins = 4096;

// 12-bit monotone image of an order key: 128 bins per octave of score over [2^-32, 1) (7 mantissa
// bits); everything below (negative scores, the -1 marks) lands in bin 0, everything above in 4095.
// Monotone in the key, so "key order" never contradicts "bin order"; ties inside a bin are resolved
// exactly by the rank kernel.
__device__ __forceinline__ int key_bin(uint32_t key) {
  const uint32_t asc = ~key;
  if (!(asc & 0x80000000u)) return 0;                     // negative float
  const int e7 = (int)((asc & 0x7fffffffu) >> 16);        // exponent + 7 mantissa bits of a positive float
  const int b = e7 - ((127 - 32) << 7);
  return b < 0 ? 0 : (b > kBins - 1 ? kBins - 1 : b);
}

// thread t -> (a, h, w) with w fastest so the five NCHW planes are read coalesced;
// output position follows the reference's enumeration index = (h*W + w)*A + a.
// hist_part (images, gridDim.x, kBins) or NULL: histogram of key_bin over THIS workgroup's anchors,
// every bin written (the workspace is not assumed to be zeroed).
__global__ __launch_bounds__(kDecodeThreads) void proposal_decode_kernel(
    const float* __restrict__ cls_prob, const float* __restrict__ bbox_pred, const float* __restrict__ im_info,
    int A, int H, int W, int feature_stride, int rpn_min_size, Anchors anchors, float4* __restrict__ boxes_all,
    uint32_t* __restrict__ keys_all, uint32_t* __restrict__ hist_part) {
  __shared__ uint32_t hist[kBins];
  const int b = blockIdx.y;
  const int HW = H * W;
  const int t = blockIdx.x * kDecodeThreads + threadIdx.x;
  if (hist_part) {
    for (int i = threadIdx.x; i < kBins; i += kDecodeThreads) hist[i] = 0;
    __syncthreads();
  }
  if (t < A * HW) {
    const int a = t / HW, p = t - a * HW;
    const int h = p / W, w = p - h * W;
    const float im_height = im_info[b * 3], im_width = im_info[b * 3 + 1];
    const int real_height = (int)(im_height / feature_stride);
    const int real_width = (int)(im_width / feature_stride);
    const float min_size = (float)rpn_min_size * im_info[b * 3 + 2];
    const float x1 = anchors.v[a * 4 + 0] + (float)(w * feature_stride);
    const float y1 = anchors.v[a * 4 + 1] + (float)(h * feature_stride);
    const float x2 = anchors.v[a * 4 + 2] + (float)(w * feature_stride);
    const float y2 = anchors.v[a * 4 + 3] + (float)(h * feature_stride);
    float score = cls_prob[((size_t)b * (2 * A) + a + A) * HW + p];
    const float width = x2 - x1 + 1.0f;
    const float height = y2 - y1 + 1.0f;
    const float ctr_x = fmaf(0.5f, width - 1.0f, x1);
    const float ctr_y = fmaf(0.5f, height - 1.0f, y1);
    const size_t ba = (size_t)b * A + a;
    const float dx = bbox_pred[(ba * 4 + 0) * HW + p];
    const float dy = bbox_pred[(ba * 4 + 1) * HW + p];
    const float dw = bbox_pred[(ba * 4 + 2) * HW + p];
    const float dh = bbox_pred[(ba * 4 + 3) * HW + p];
    const float pred_ctr_x = fmaf(dx, width, ctr_x);
    const float pred_ctr_y = fmaf(dy, height, ctr_y);
    const float pred_w = expf_cr(dw) * width;
    const float pred_h = expf_cr(dh) * height;
    float pred_x1 = fmaf(-0.5f, pred_w - 1.0f, pred_ctr_x);
    float pred_y1 = fmaf(-0.5f, pred_h - 1.0f, pred_ctr_y);
    float pred_x2 = fmaf(0.5f, pred_w - 1.0f, pred_ctr_x);
    float pred_y2 = fmaf(0.5f, pred_h - 1.0f, pred_ctr_y);
    pred_x1 = fmaxf(fminf(pred_x1, im_width - 1.0f), 0.0f);
    pred_y1 = fmaxf(fminf(pred_y1, im_height - 1.0f), 0.0f);
    pred_x2 = fmaxf(fminf(pred_x2, im_width - 1.0f), 0.0f);
    pred_y2 = fmaxf(fminf(pred_y2, im_height - 1.0f), 0.0f);
    if (h >= real_height || w >= real_width) score = -1.0f;
    const float iw = pred_x2 - pred_x1 + 1.0f;
    const float ih = pred_y2 - pred_y1 + 1.0f;
    if (iw < min_size || ih < min_size) {
      pred_x1 -= min_size / 2; pred_y1 -= min_size / 2;
      pred_x2 += min_size / 2; pred_y2 += min_size / 2;
      score = -1.0f;
    }
    const size_t index = (size_t)b * A * HW + (size_t)p * A + a;
    boxes_all[index] = make_float4(pred_x1, pred_y1, pred_x2, pred_y2);
    const uint32_t key = desc_key(score);
    keys_all[index] = key;
    if (hist_part) atomicAdd(&hist[key_bin(key)], 1u);
  }
  if (hist_part) {
    __syncthreads();
    uint32_t* dst = hist_part + ((size_t)b * gridDim.x + blockIdx.x) * kBins;
    for (int i = threadIdx.x; i < kBins; i += kDecodeThreads) dst[i] = hist[i];
  }
}

// block-wide exclusive scan of one int per thread (1024 threads = 16 waves)
__device__ __forceinline__ int block_exclusive_scan(int v, int* wave_sums /*LDS, 16*/, int* total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const int o = __shfl_up(incl, d, 64);
    if (lane >= d) incl += o;
  }
  if (lane == 63) wave_sums[wid] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int i = 0; i < kTopkThreads / 64; ++i) {
    const int s = wave_sums[i];
    if (i < wid) base += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// ---- chip-wide plan: compact (grouped by bin) -> rank within the bin + scatter ----------------------
constexpr int kRankThreads = 256;
constexpr int kFastMaxCount = 32768;     // anchors per image the chip-wide plan accepts

// info (images, 4) int32: [0] = M, the number of candidates.  bin_start / bin_count (images, kBins): where
// the candidates of a bin sit in the candidate list (bins in DESCENDING order: bin_start[b] = candidates in
// bins above b) and how many they are — written by workgroup 0, read by the rank kernel.
// grid (G = decode's gridDim.x, images); block kDecodeThreads.  Every workgroup redoes the small histogram
// reduction (G x 16 KB from L2) instead of a separate one-workgroup launch.
__global__ __launch_bounds__(kDecodeThreads) void proposal_compact_kernel(
    const uint32_t* __restrict__ keys_all, const uint32_t* __restrict__ hist_part, int A, int HW, int pre_n,
    uint64_t* __restrict__ cand_all, int* __restrict__ info, int* __restrict__ bin_start, int* __restrict__ bin_count) {
  __shared__ int wave_sums[kDecodeThreads / 64];
  __shared__ int sh[4];
  __shared__ int slot[kBins];            // next free position of each bin's group, for THIS workgroup's candidates
  const int img = blockIdx.y, g = blockIdx.x, G = gridDim.x;
  const int tid = threadIdx.x;
  const int N = A * HW;
  const uint32_t* parts = hist_part + (size_t)img * G * kBins;
  // this thread's anchor (same chunking as the decode kernel's): its key is needed last, so load it first
  const int t = g * kDecodeThreads + tid;
  uint32_t key = 0, index = 0;
  if (t < N) {
    const int a = t / HW, p = t - a * HW;
    index = (uint32_t)(p * A + a);
    key = keys_all[(size_t)img * N + index];
  }
  // histograms: this thread owns bins 4*tid .. 4*tid+3; totals over all workgroups, and the part held by the
  // workgroups before this one
  uint32_t c[4] = {0, 0, 0, 0}, before_me[4] = {0, 0, 0, 0};
  for (int q0 = 0; q0 < G; q0 += 8) {      // eight workgroups' partial histograms in flight per step (G = 22: a chain of 22 trips otherwise)
    uint4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u < G ? q0 + u : G - 1;
      v[u] = *reinterpret_cast<const uint4*>(parts + (size_t)q * kBins + 4 * tid);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int q = q0 + u;
      if (q < G) {
        c[0] += v[u].x; c[1] += v[u].y; c[2] += v[u].z; c[3] += v[u].w;
        if (q < g) { before_me[0] += v[u].x; before_me[1] += v[u].y; before_me[2] += v[u].z; before_me[3] += v[u].w; }
      }
    }
  }
  const int mine = (int)(c[0] + c[1] + c[2] + c[3]);
  int total;
  const int below = block_exclusive_scan(mine, wave_sums, &total);     // keys in bins < 4*tid
  // walk this thread's bins downwards: `above` = keys in bins above the current one.  The bin where the count
  // of keys at or above it first reaches pre_n is the threshold bin (exactly one thread finds it; total >= pre_n)
  int above = total - below - mine;
#pragma unroll
  for (int k = 3; k >= 0; --k) {
    const int bin = 4 * tid + k;
    slot[bin] = above + (int)before_me[k];
    if (g == 0) { bin_start[img * kBins + bin] = above; bin_count[img * kBins + bin] = (int)c[k]; }
    if (above < pre_n && above + (int)c[k] >= pre_n) { sh[0] = bin; sh[1] = above + (int)c[k]; }
    above += (int)c[k];
  }
  __syncthreads();
  const int bin_t = sh[0], M = sh[1];
  if (t < N) {
    const int bin = key_bin(key);
    if (bin >= bin_t) {
      const int pos = atomicAdd(&slot[bin], 1);       // order inside a bin's group is arbitrary; the rank kernel sorts it out
      cand_all[(size_t)img * N + pos] = ((uint64_t)key << 32) | index;
    }
  }
  if (g == 0 && tid == 0) info[img * 4] = M;
}

// grid (ceil(N / 256), images); block 256.  Candidate i sits in its bin's group; bins are disjoint key ranges,
// so its position in the stable descending sort (== thrust::stable_sort_by_key(greater), :517-521) is
// bin_start + the number of candidates OF ITS OWN BIN ordered before it ((key, index) pairs are distinct).
// Groups hold a handful of candidates unless the scores collapse into one bin, when this degrades to M^2.
// Candidates ranked below pre_n drop out; the others write their box and key at their sorted position.
__global__ __launch_bounds__(kRankThreads) void proposal_rank_kernel(
    const uint64_t* __restrict__ cand_all, const int* __restrict__ info, const int* __restrict__ bin_start,
    const int* __restrict__ bin_count, int N, int pre_n, const float4* __restrict__ boxes_all,
    float4* __restrict__ sorted_box, uint32_t* __restrict__ sorted_key) {
  const int img = blockIdx.y;
  const int M = info[img * 4];
  const int i = blockIdx.x * kRankThreads + threadIdx.x;
  if (i >= M) return;
  const uint64_t* cand = cand_all + (size_t)img * N;
  const uint64_t e = cand[i];
  const int bin = key_bin((uint32_t)(e >> 32));
  const int s0 = bin_start[img * kBins + bin], n = bin_count[img * kBins + bin];
  int before = 0;
  {
    const uint64_t* b = cand + s0;
    int j = 0;
    for (; j + 8 <= n; j += 8) {       // eight loads in flight: the walk over a bin's members is a latency chain otherwise
      uint64_t v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = b[j + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) before += v[u] < e;
    }
    for (; j < n; ++j) before += b[j] < e;
  }
  const int rank = s0 + before;
  if (rank >= pre_n) return;
  sorted_box[(size_t)img * pre_n + rank] = boxes_all[(size_t)img * N + (uint32_t)e];
  sorted_key[(size_t)img * pre_n + rank] = (uint32_t)(e >> 32);
}

// ---- bitonic sort of Kpad 64-bit keys, 8 per thread in registers ---------------------------
// element e = PER*tid + r (PER = 8, or 16 above 8192 keys).  Strides j < PER stay in registers,
// PER <= j < 64*PER are lane exchanges inside a wave (shuffles), only j >= 64*PER go through
// LDS with a workgroup barrier.
__device__ __forceinline__ void cswap(uint64_t& a, uint64_t& b, bool up) {
  const bool sw = (a > b) == up;
  const uint64_t lo = sw ? b : a, hi = sw ? a : b;
  a = lo; b = hi;
}

template <int kPerThread>
__device__ __forceinline__ void bitonic_sort_regs(uint64_t (&v)[kPerThread], uint64_t* buf, int Kpad) {
  const int tid = threadIdx.x;
  const int nthr = Kpad / kPerThread;          // threads that hold data (Kpad >= 8)
  const bool active = tid < nthr;
  for (int k = 2; k <= Kpad; k <<= 1) {
    for (int j = k >> 1; j >= kPerThread; j >>= 1) {
      if (j >= kPerThread * 64) {
        // cross-wave: through LDS
        __syncthreads();
        if (active) {
#pragma unroll
          for (int r = 0; r < kPerThread; ++r) buf[tid * kPerThread + r] = v[r];
        }
        __syncthreads();
        if (active) {
#pragma unroll
          for (int r = 0; r < kPerThread; ++r) {
            const int e = tid * kPerThread + r;
            const uint64_t p = buf[e ^ j];
            const bool keep_min = ((e & k) == 0) == ((e & j) == 0);
            v[r] = ((v[r] > p) == keep_min) ? p : v[r];
          }
        }
      } else {
        const int lane_mask = j / kPerThread;   // < 64
#pragma unroll
        for (int r = 0; r < kPerThread; ++r) {
          const int e = tid * kPerThread + r;
          const uint64_t p = __shfl_xor(v[r], lane_mask, 64);
          const bool keep_min = ((e & k) == 0) == ((e & j) == 0);
          v[r] = ((v[r] > p) == keep_min) ? p : v[r];
        }
      }
    }
    // strides below kPerThread: in registers.  The stride must be a compile-time constant here
    // (a runtime register index would push v[] into scratch memory), hence the unrolled ladder.
#pragma unroll
    for (int jj = kPerThread / 2; jj > 0; jj >>= 1) {
      if (jj < k) {
#pragma unroll
        for (int r = 0; r < kPerThread; ++r) {
          if ((r & jj) == 0) {
            const int e = tid * kPerThread + r;
            cswap(v[r], v[r | jj], (e & k) == 0);
          }
        }
      }
    }
  }
}

// LDS carve of proposal_select_nms_kernel (offsets in bytes, all multiples of 16):
//   [0, Kpad*8)            sortbuf  — radix-select histograms (16 replicas x 256) overlay it first
//   [Kpad*8, +256)         misc
//   [Kpad*8+256, ...)      lkeys (N*4, KEYS_LDS) — overlaid after the sort by the NMS state:
//                          kept_box float4[1024] | kept_idx int[1024] | cand float4[64] | colsupp u64[64]
//                          | kept_area float[1024] | alive_box float4[1024] | alive_idx int[1024]
constexpr int kHistReplicas = 16;
constexpr size_t kNmsCoreBytes = 1024 * 16 + 1024 * 4 + 64 * 16 + 64 * 8 + 1024 * 4;   // survivors + step scratch
constexpr size_t kNmsStateBytes = kNmsCoreBytes + 1024 * 16 + 1024 * 4;                 // + alive candidates of a chunk

// devIoU(a, b) > thresh (multi_proposal.cu:252-260, :295): IouTest / iou_exceeds_fast / iou_exceeds_div of nms_kernels.h
// one pair (used where the loop is short); wave-uniform branch so the division is really skipped
__device__ __forceinline__ bool iou_exceeds(const float4& a, float Sa, const float4& b, float Sb, const IouTest& t) {
  bool degenerate = !t.fast;
  bool r = iou_exceeds_fast(a, Sa, b, Sb, t, degenerate);
  if (__builtin_expect(__any(degenerate), 0)) {
    if (degenerate) r = iou_exceeds_div(a, Sa, b, Sb, t);
  }
  return r;
}
// candidate b against survivors [k0, k1) stepping by `stride`: branch-free main loop (LDS reads batch and
// pipeline), and a division-based second pass only if some lane met a degenerate union
__device__ __forceinline__ bool suppressed_by(const float4* kept_box, const float* kept_area, int k0, int k1, int stride,
                                              const float4& b, float Sb, const IouTest& t) {
  bool hit = false, degenerate = !t.fast;
#pragma unroll 4
  for (int k = k0; k < k1; k += stride) hit |= iou_exceeds_fast(kept_box[k], kept_area[k], b, Sb, t, degenerate);
  if (__builtin_expect(__any(degenerate), 0)) {
    if (degenerate) {
      hit = false;
      for (int k = k0; k < k1; k += stride) hit |= iou_exceeds_div(kept_box[k], kept_area[k], b, Sb, t);
    }
  }
  return hit;
}

template <bool KEYS_LDS, int kPerThread, bool RADIX, bool ALIVE_BOX>
__global__ __launch_bounds__(kTopkThreads) void proposal_select_nms_kernel(
    const uint32_t* __restrict__ keys_all, const float4* __restrict__ boxes_all, int N, int K, int Kpad,
    int sort_bytes, IouTest iou, int post_n, float* __restrict__ rois, float* __restrict__ scores,
    unsigned long long* __restrict__ stamps /* diagnostic: phase timestamps of workgroup 0, or NULL */) {
#define LSFA_STAMP(i) do { if (stamps && threadIdx.x == 0 && blockIdx.x == 0) stamps[i] = __builtin_readcyclecounter(); } while (0)
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LSFA_STAMP(0);
  uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem);
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  int* misc = reinterpret_cast<int*>(smem + (size_t)sort_bytes);
  unsigned char* region = smem + (size_t)sort_bytes + 256;
  uint32_t* lkeys = reinterpret_cast<uint32_t*>(region);
  const int img = blockIdx.x;
  const uint32_t* gkeys = keys_all + (size_t)img * N;
  const float4* boxes = boxes_all + (size_t)img * N;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wid = tid >> 6;
  if (KEYS_LDS) {
    for (int i = tid; i < N; i += kTopkThreads) lkeys[i] = gkeys[i];
  }
  const uint32_t* keys = KEYS_LDS ? lkeys : gkeys;

  __syncthreads();
  LSFA_STAMP(1);
  // ---- radix select: T = K-th smallest key; r_eq = how many keys == T to take ------------
  // replica = lane & 15 spreads same-bucket increments of one wave instruction over 16 words
  uint32_t prefix = 0;
  int remaining = K;
  for (int shift = 24; shift >= 0; shift -= 8) {
    for (int i = tid; i < kHistReplicas * 256; i += kTopkThreads) hist[i] = 0;
    __syncthreads();
    uint32_t* myhist = hist + (lane & (kHistReplicas - 1)) * 256;
    for (int i = tid; i < N; i += kTopkThreads) {
      const uint32_t k = keys[i];
      const bool match = (shift == 24) || (((k ^ prefix) >> (shift + 8)) == 0);
      if (match) atomicAdd(&myhist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 256) {
      uint32_t c = 0;
#pragma unroll
      for (int r = 0; r < kHistReplicas; ++r) c += hist[r * 256 + tid];
      // inclusive scan over the 256 buckets: 4 waves, wave scan + carry through misc[8..11]
      int incl = (int)c;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
      }
      if (lane == 63) misc[8 + wid] = incl;
      // thread `tid` is the only reader of column `tid` of the replicas, so it may reuse them:
      hist[tid] = (uint32_t)incl;          // replica 0 now holds the per-wave inclusive scan
      hist[256 + tid] = c;                 // replica 1 holds the bucket counts
    }
    __syncthreads();
    if (tid < 256) {
      int carry = 0;
      for (int w = 0; w < wid; ++w) carry += misc[8 + w];
      const int incl = (int)hist[tid] + carry;
      const int c = (int)hist[256 + tid];
      // the unique bucket where the running count crosses `remaining`
      if (incl >= remaining && incl - c < remaining) {
        misc[0] = tid;
        misc[1] = remaining - (incl - c);
      }
    }
    __syncthreads();
    prefix |= (uint32_t)misc[0] << shift;
    remaining = misc[1];
    __syncthreads();
  }
  const uint32_t T = prefix;
  const int r_eq = remaining;
  const int n_less = K - r_eq;

  LSFA_STAMP(2);
  // ---- stable compaction (index order): keys < T, and the first r_eq keys == T -------------
  const int chunk = (N + kTopkThreads - 1) / kTopkThreads;
  const int i0 = min(tid * chunk, N), i1 = min(i0 + chunk, N);
  int my_eq = 0, my_less = 0;
  for (int i = i0; i < i1; ++i) { const uint32_t k = keys[i]; my_eq += (k == T); my_less += (k < T); }
  int total_eq, total_sel;
  const int eq_rank0 = block_exclusive_scan(my_eq, misc + 8, &total_eq);
  const int my_sel = my_less + min(max(r_eq - eq_rank0, 0), my_eq);
  int pos = block_exclusive_scan(my_sel, misc + 8, &total_sel);
  {
    int er = eq_rank0;
    for (int i = i0; i < i1; ++i) {
      const uint32_t k = keys[i];
      bool sel = k < T;
      if (k == T) { sel = er < r_eq; ++er; }
      if (sel) sortbuf[pos++] = ((uint64_t)k << 32) | (uint32_t)i;
    }
  }
  (void)n_less;
  __syncthreads();

  LSFA_STAMP(3);
  // ---- sort by key (ascending == score desc); equal keys keep index order -------------------
  if (RADIX) {
    // LSD radix sort, 4-bit digits, stable: thread t owns the contiguous run [c0, c1) of the K
    // candidates; per pass: digit counts packed in one u64 -> cnt16[d][t] -> one flat exclusive
    // scan (digit-major) -> scatter.  Passes whose digit is identical for all keys are skipped.
    uint64_t* bufA = sortbuf;
    uint64_t* bufB = reinterpret_cast<uint64_t*>(region);
    uint16_t* cnt16 = reinterpret_cast<uint16_t*>(region + (((size_t)K * 8 + 15) & ~(size_t)15));
    const int ck = (K + kTopkThreads - 1) / kTopkThreads;   // <= 8
    const int c0 = min(tid * ck, K), c1 = min(c0 + ck, K);
    uint32_t vo = 0, va = ~0u;
    for (int i = c0; i < c1; ++i) { const uint32_t hw = (uint32_t)(bufA[i] >> 32); vo |= hw; va &= hw; }
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { vo |= __shfl_xor(vo, d, 64); va &= __shfl_xor(va, d, 64); }
    if (lane == 0) { misc[32 + wid] = (int)vo; misc[48 + wid] = (int)va; }
    __syncthreads();
    vo = 0; va = ~0u;
    for (int w = 0; w < kTopkThreads / 64; ++w) { vo |= (uint32_t)misc[32 + w]; va &= (uint32_t)misc[48 + w]; }
    const uint32_t differ = vo ^ va;
    uint64_t* src = bufA;
    uint64_t* dst = bufB;
    for (int pass = 0; pass < 8; ++pass) {
      if (((differ >> (4 * pass)) & 15u) == 0) continue;
      const int shift = 32 + 4 * pass;
      uint64_t h = 0;
      for (int i = c0; i < c1; ++i) h += 1ULL << (4 * (int)((src[i] >> shift) & 15u));
#pragma unroll
      for (int d = 0; d < 16; ++d) cnt16[d * kTopkThreads + tid] = (uint16_t)((h >> (4 * d)) & 15u);
      __syncthreads();
      {
        uint4* p4 = reinterpret_cast<uint4*>(cnt16 + 16 * tid);
        uint4 q0 = p4[0], q1 = p4[1];
        uint32_t w[8] = {q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, q1.z, q1.w};
        int run = 0;
        uint32_t ex[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const uint32_t c = (w[e >> 1] >> (16 * (e & 1))) & 0xffffu;
          ex[e] = (uint32_t)run;
          run += (int)c;
        }
        int tot;
        const int base = block_exclusive_scan(run, misc + 8, &tot);
#pragma unroll
        for (int e = 0; e < 8; ++e) w[e] = ((ex[2 * e] + base) & 0xffffu) | ((ex[2 * e + 1] + base) << 16);
        p4[0] = make_uint4(w[0], w[1], w[2], w[3]);
        p4[1] = make_uint4(w[4], w[5], w[6], w[7]);
      }
      __syncthreads();
      for (int i = c0; i < c1; ++i) {
        const uint64_t e = src[i];
        uint16_t* slot = cnt16 + (int)((e >> shift) & 15u) * kTopkThreads + tid;
        const int o = *slot;
        *slot = (uint16_t)(o + 1);
        dst[o] = e;
      }
      __syncthreads();
      uint64_t* tsw = src; src = dst; dst = tsw;
    }
    if (src != bufA) {
      for (int i = c0; i < c1; ++i) bufA[i] = src[i];
    }
    __syncthreads();
  } else {
    for (int i = K + tid; i < Kpad; i += kTopkThreads) sortbuf[i] = ~0ULL;
    __syncthreads();
    uint64_t v[kPerThread];
    const bool holds = tid < Kpad / kPerThread;
#pragma unroll
    for (int r = 0; r < kPerThread; ++r) v[r] = holds ? sortbuf[tid * kPerThread + r] : ~0ULL;
    bitonic_sort_regs<kPerThread>(v, sortbuf, Kpad);
    __syncthreads();
    if (holds) {
#pragma unroll
      for (int r = 0; r < kPerThread; ++r) sortbuf[tid * kPerThread + r] = v[r];
    }
    __syncthreads();
  }

  LSFA_STAMP(4);
  // ---- greedy NMS over the sorted candidates, IoU on the fly against the survivors --------
  // (same decisions as nms_kernel + the host sweep, multi_proposal.cu:262-357).  Candidates are
  // taken 1024 at a time: every thread tests ONE candidate against all survivors so far (no
  // barriers, full occupancy — in the heavy-overlap regime this discards ~95 % of them), the
  // still-alive ones are compacted in order, and only those go through the 64-wide step that
  // tests them against the survivors added since (waves split that short list), builds the
  // in-block relation with ballots and resolves it as a fixpoint.
  float4* kept_box = reinterpret_cast<float4*>(region);
  int* kept_idx = reinterpret_cast<int*>(region + 1024 * 16);
  float4* cand_lds = reinterpret_cast<float4*>(region + 1024 * 16 + 1024 * 4);
  uint64_t* colsupp = reinterpret_cast<uint64_t*>(region + 1024 * 16 + 1024 * 4 + 64 * 16);
  float* kept_area = reinterpret_cast<float*>(region + 1024 * 16 + 1024 * 4 + 64 * 16 + 64 * 8);
  float4* alive_box = reinterpret_cast<float4*>(region + kNmsCoreBytes);            // 1024
  int* alive_idx = reinterpret_cast<int*>(region + kNmsCoreBytes + (ALIVE_BOX ? 1024 * 16 : 0));   // 1024
  unsigned int* supp = reinterpret_cast<unsigned int*>(misc + 4);  // 2 words
  if (tid == 0) { misc[3] = 0; supp[0] = 0; supp[1] = 0; }
  __syncthreads();
  int num = 0;
  int dbg_blocks = 0;
  unsigned long long dbg_a = 0, dbg_b = 0, dbg_c = 0, dbg_t = 0;
  for (int chunk = 0; chunk < K && num < post_n; chunk += kTopkThreads) {
    if (stamps) dbg_t = __builtin_readcyclecounter();
    // -- filter: one candidate per thread against the survivors so far
    const int ci = chunk + tid;
    const bool valid = ci < K;
    const float4 mybox = valid ? boxes[(uint32_t)sortbuf[ci]] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float myarea = (mybox.z - mybox.x + 1) * (mybox.w - mybox.y + 1);
    bool dead = !valid;
    dead = dead || suppressed_by(kept_box, kept_area, 0, num, 1, mybox, myarea, iou);
    // -- ordered compaction of the alive candidates
    int total_alive;
    const int apos = block_exclusive_scan(dead ? 0 : 1, misc + 8, &total_alive);
    if (!dead) { if (ALIVE_BOX) alive_box[apos] = mybox; alive_idx[apos] = ci; }
    __syncthreads();
    if (stamps) { const unsigned long long t1 = __builtin_readcyclecounter(); dbg_a += t1 - dbg_t; dbg_t = t1; }
    const int m0 = num;   // survivors the filter already accounted for
    for (int base = 0; base < total_alive && num < post_n; base += 64) {
      ++dbg_blocks;
      if (stamps) dbg_t = __builtin_readcyclecounter();
      const int nb = min(64, total_alive - base);
      const float4 cb = lane < nb ? (ALIVE_BOX ? alive_box[base + lane] : boxes[(uint32_t)sortbuf[alive_idx[base + lane]]])
                                  : make_float4(0.f, 0.f, 0.f, 0.f);
      const float carea = (cb.z - cb.x + 1) * (cb.w - cb.y + 1);
      if (wid == 0) cand_lds[lane] = cb;
      // survivors added since the filter (earlier steps of this chunk), split over the waves
      const bool s = suppressed_by(kept_box, kept_area, m0 + wid, num, kTopkThreads / 64, cb, carea, iou);
      const unsigned long long bal = __ballot(s && lane < nb);
      if (lane == 0 && bal) {
        atomicOr(&supp[0], (unsigned int)bal);
        atomicOr(&supp[1], (unsigned int)(bal >> 32));
      }
      __syncthreads();
      // in-block relation, only among the candidates still alive: wave q takes the q-th alive one as
      // column j (or four fixed columns when many are alive) and ballots the rows that suppress it
      {
        const uint64_t sprev_u = ((uint64_t)supp[1] << 32) | supp[0];
        const uint64_t valid_m = nb == 64 ? ~0ULL : ((1ULL << nb) - 1ULL);
        uint64_t am = valid_m & ~sprev_u;                       // wave-uniform
        const bool row_alive = (am >> lane) & 1ULL;
        if (__popcll(am) <= kTopkThreads / 64) {
          for (int skip = 0; skip < wid && am; ++skip) am &= am - 1;
          if (am) {
            const int j = __builtin_ctzll(am);
            const float4 jb = cand_lds[j];
            const float jarea = (jb.z - jb.x + 1) * (jb.w - jb.y + 1);
            const bool pred = row_alive && (lane < j) && iou_exceeds(cb, carea, jb, jarea, iou);
            const unsigned long long col = __ballot(pred);
            if (lane == 0) colsupp[j] = col;
          }
        } else {
#pragma unroll
          for (int jj = 0; jj < 4; ++jj) {
            const int j = wid * 4 + jj;
            const float4 jb = cand_lds[j];
            const float jarea = (jb.z - jb.x + 1) * (jb.w - jb.y + 1);
            const bool pred = row_alive && (lane < j) && (j < nb) && iou_exceeds(cb, carea, jb, jarea, iou);
            const unsigned long long col = __ballot(pred);
            if (lane == 0) colsupp[j] = col;
          }
        }
      }
      __syncthreads();
      if (stamps) { const unsigned long long t1 = __builtin_readcyclecounter(); dbg_b += t1 - dbg_t; dbg_t = t1; }
      if (wid == 0) {
        // In-block greedy as a fixpoint: G(k) = alive(k) && no earlier j in G suppresses k.  Starting
        // from G = alive, iteration t fixes the first t decisions, so it converges to the serial
        // sweep's answer in (dependency-chain depth) rounds — a handful — instead of 64 scalar steps.
        const uint64_t col = colsupp[lane];
        const uint64_t sprev = ((uint64_t)supp[1] << 32) | supp[0];
        const bool alive = lane < nb && !((sprev >> lane) & 1ULL);
        uint64_t G = __ballot(alive);
        for (int it = 0; it < 64; ++it) {
          const uint64_t G2 = __ballot(alive && (col & G) == 0);
          if (G2 == G) break;
          G = G2;
        }
        // keep only as many as the output still needs (later survivors never affect earlier ones)
        const int budget = post_n - num;
        uint64_t kept = G;
        if (__popcll(G) > budget) {
          const bool mine = (G >> lane) & 1ULL;
          const int rank = __popcll(G & ((1ULL << lane) - 1ULL));
          kept = __ballot(mine && rank < budget);
        }
        if ((kept >> lane) & 1ULL) {
          const int pos = num + __popcll(kept & ((1ULL << lane) - 1ULL));
          kept_box[pos] = cb;
          kept_area[pos] = carea;
          kept_idx[pos] = alive_idx[base + lane];
        }
        if (lane == 0) { misc[3] = num + __popcll(kept); supp[0] = 0; supp[1] = 0; }
      }
      __syncthreads();
      if (stamps) { const unsigned long long t1 = __builtin_readcyclecounter(); dbg_c += t1 - dbg_t; dbg_t = t1; }
      num = misc[3];
    }
  }

  LSFA_STAMP(5);
  if (stamps && tid == 0 && blockIdx.x == 0) { stamps[7] = (unsigned long long)num; stamps[8] = (unsigned long long)dbg_blocks; stamps[9] = dbg_a; stamps[10] = dbg_b; stamps[11] = dbg_c; }
  // ---- PrepareOutput (multi_proposal.cu:363-388): first post_n survivors, cyclic pad ------
  for (int index = tid; index < post_n; index += kTopkThreads) {
    const int kpos = index < num ? index : index % num;
    const float4 bx = kept_box[kpos];
    float* o = rois + ((size_t)img * post_n + index) * 5;
    o[0] = (float)img;
    o[1] = bx.x; o[2] = bx.y; o[3] = bx.z; o[4] = bx.w;
    if (scores) scores[(size_t)img * post_n + index] = key_score((uint32_t)(sortbuf[kept_idx[kpos]] >> 32));
  }
  LSFA_STAMP(6);
#undef LSFA_STAMP
}

struct WsLayout {
  size_t boxes, keys, hist, info, bins, cand, sbox, skey, mask, diagT, total;
  int groups;      // decode workgroups per image
  bool chip_wide, chip_wide_possible;
};
int clamp_pre_n(int rpn_pre_nms_top_n, int count) {
  int pre_n = rpn_pre_nms_top_n > 0 ? rpn_pre_nms_top_n : count;  // multi_proposal.cu:435-436
  return pre_n < count ? pre_n : count;
}
WsLayout ws_layout(int B, int count, int pre_n, int post_n) {
  WsLayout l;
  size_t o = 0;
  l.groups = ceil_div(count, kDecodeThreads);
  l.chip_wide = count <= kFastMaxCount && pre_n <= 8192 && post_n <= kSweepMaxOut;
  l.chip_wide_possible = l.chip_wide;
  l.boxes = o; o += align_up((size_t)B * count * sizeof(float4), 256);
  l.keys = o; o += align_up((size_t)B * count * sizeof(uint32_t), 256);
  l.hist = l.info = l.bins = l.cand = l.sbox = l.skey = l.mask = l.diagT = o;
  if (l.chip_wide) {
    const size_t col_blocks = (size_t)ceil_div(pre_n, 64);
    l.hist = o; o += align_up((size_t)B * l.groups * kBins * sizeof(uint32_t), 256);
    l.info = o; o += align_up((size_t)B * 4 * sizeof(int), 256);
    l.bins = o; o += align_up((size_t)B * 2 * kBins * sizeof(int), 256);
    l.cand = o; o += align_up((size_t)B * count * sizeof(uint64_t), 256);
    l.sbox = o; o += align_up((size_t)B * pre_n * sizeof(float4), 256);
    l.skey = o; o += align_up((size_t)B * pre_n * sizeof(uint32_t), 256);
    l.mask = o; o += align_up((size_t)B * pre_n * col_blocks * sizeof(uint64_t), 256);
    l.diagT = o; o += align_up((size_t)B * pre_n * sizeof(uint64_t), 256);
  }
  l.total = o;
  return l;
}

}  // namespace

static std::atomic<int> g_plan{LSFA_PROPOSAL_PLAN_AUTO};
extern "C" int lsfa_proposal_set_plan(int plan) {
  LSFA_REQUIRE((plan >= LSFA_PROPOSAL_PLAN_AUTO && plan <= LSFA_PROPOSAL_PLAN_CHIP_WIDE_BOX_SWEEP) || plan == LSFA_PROPOSAL_PLAN_LAB_NO_NMS,
               "lsfa_proposal_set_plan: unknown plan %d", plan);
  g_plan.store(plan);
  return LSFA_OK;
}

extern "C" size_t lsfa_proposal_workspace_bytes(int B, int A, int H, int W, int pre_nms_top_n) {
  if (B <= 0 || A <= 0 || H <= 0 || W <= 0) return 0;
  const int count = A * H * W;
  const int pre_n = clamp_pre_n(pre_nms_top_n, count);
  // post_nms_top_n is not an argument here: size for the chip-wide plan whenever the other limits allow it
  return ws_layout(B, count, pre_n, 1).total;
}

extern "C" int lsfa_proposal(const float* cls_prob, const float* bbox_pred, const float* im_info, int B, int A,
                             int H, int W, int feature_stride, const float* scales_host, int n_scales,
                             const float* ratios_host, int n_ratios, int rpn_pre_nms_top_n,
                             int rpn_post_nms_top_n, float threshold, int rpn_min_size, float* rois, float* scores,
                             void* ws, size_t ws_bytes, void* stream) {
  LSFA_REQUIRE(cls_prob && bbox_pred && im_info && rois && ws, "lsfa_proposal: NULL argument");
  LSFA_REQUIRE(B > 0 && A > 0 && H > 0 && W > 0 && feature_stride > 0, "lsfa_proposal: bad shape");
  // multi_proposal.cu:446: CHECK_EQ(num_anchors, ratios.size() * scales.size())
  LSFA_REQUIRE(scales_host && ratios_host && A == n_scales * n_ratios,
               "lsfa_proposal: num_anchors %d != ratios %d * scales %d", A, n_ratios, n_scales);
  LSFA_REQUIRE(rpn_post_nms_top_n > 0, "lsfa_proposal: rpn_post_nms_top_n must be positive");
  if (A > kMaxAnchors) { set_error("lsfa_proposal: more than %d anchors per cell", kMaxAnchors); return LSFA_ENOTSUP; }
  const int count = A * H * W;
  const int pre_n = clamp_pre_n(rpn_pre_nms_top_n, count);
  const int post_n = rpn_post_nms_top_n < pre_n ? rpn_post_nms_top_n : pre_n;  // :437
  if (pre_n > 16384 || post_n > 1024 || B > 65535) {
    set_error("lsfa_proposal: pre_nms_top_n %d (max 16384) / post_nms_top_n %d (max %d) unsupported", pre_n, post_n,
              1024);
    return LSFA_ENOTSUP;
  }
  WsLayout l = ws_layout(B, count, pre_n, post_n);     // sized for the chip-wide plan whenever the shape allows it
  const int plan = g_plan.load();
  if (plan == LSFA_PROPOSAL_PLAN_SINGLE_WORKGROUP) l.chip_wide = false;
  if (ws_bytes < l.total) { set_error("lsfa_proposal: workspace %zu < %zu bytes", ws_bytes, l.total); return LSFA_EWORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  unsigned char* base = (unsigned char*)ws;
  float4* boxes = (float4*)(base + l.boxes);
  uint32_t* keys = (uint32_t*)(base + l.keys);

  Anchors anchors;
  generate_anchors(feature_stride, ratios_host, n_ratios, scales_host, n_scales, &anchors);

  ProfScope prof(LSFA_OP_PROPOSAL, s);
  uint32_t* hist = l.chip_wide ? (uint32_t*)(base + l.hist) : nullptr;
  hipLaunchKernelGGL(proposal_decode_kernel, dim3(l.groups, B), dim3(kDecodeThreads), 0, s, cls_prob, bbox_pred,
                     im_info, A, H, W, feature_stride, rpn_min_size, anchors, boxes, keys, hist);
  const IouTest iou = make_iou_test(threshold);
  if (l.chip_wide) {
    int* info = (int*)(base + l.info);
    uint64_t* cand = (uint64_t*)(base + l.cand);
    int* bin_start = (int*)(base + l.bins);
    int* bin_count = bin_start + (size_t)B * kBins;
    float4* sbox = (float4*)(base + l.sbox);
    uint32_t* skey = (uint32_t*)(base + l.skey);
    uint64_t* mask = (uint64_t*)(base + l.mask);
    uint64_t* diagT = (uint64_t*)(base + l.diagT);
    const int col_blocks = ceil_div(pre_n, 64);
    const int gi = ceil_div(count, kRankThreads);
    hipLaunchKernelGGL(proposal_compact_kernel, dim3(l.groups, B), dim3(kDecodeThreads), 0, s, (const uint32_t*)keys,
                       (const uint32_t*)hist, A, H * W, pre_n, cand, info, bin_start, bin_count);
    hipLaunchKernelGGL(proposal_rank_kernel, dim3(gi, B), dim3(kRankThreads), 0, s, (const uint64_t*)cand, (const int*)info,
                       (const int*)bin_start, (const int*)bin_count, count, pre_n, (const float4*)boxes, sbox, skey);
    if (plan == LSFA_PROPOSAL_PLAN_LAB_NO_NMS) {      // ablation: what the suppression stage costs the pipeline (no rois are written)
      LSFA_LAUNCH_CHECK("lsfa_proposal");
      return LSFA_OK;
    }
    // r6: AUTO takes the mask-free sweep for launches of several images.  The full mask costs the WHOLE CHIP 100 us per nine images (4465 tiles
    // each, of which the sweep reads a sliver); the box sweep costs one workgroup per image 160 us.  Alone that is slower - r3's measurement, and
    // why one image keeps the mask (18 + 12 us against 160: the frame-by-frame pipeline loses 6 % on the box sweep) - but in the batched
    // pipeline a kernel's cost is the share of the chip it takes from the other streams: +2.2 ... +2.8 % frames/s (fp32), +4.2 % (bf16, four
    // clips) with the box sweep (profiles/r6/proposal_plan_ab*.txt).  Same results bit for bit (tests/test_hip_ops.py::test_proposal_launch_plans_agree).
    const bool box_sweep = plan == LSFA_PROPOSAL_PLAN_CHIP_WIDE_BOX_SWEEP || (plan == LSFA_PROPOSAL_PLAN_AUTO && B >= 2);
    if (!box_sweep) {
      hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(ceil_div(nms_tile_count(col_blocks), 4), 1, B), dim3(256), 0, s,
                         (const float*)sbox, (long)pre_n * 4, 4, (const int*)nullptr, pre_n, iou, mask, diagT, col_blocks, 0);
      hipLaunchKernelGGL(nms_sweep_kernel, dim3(B), dim3(64), 0, s, (const uint64_t*)mask, (const uint64_t*)diagT, pre_n,
                         col_blocks, post_n, (int*)nullptr, (int*)nullptr, (const float4*)sbox, (const uint32_t*)skey,
                         rois, scores);
    } else {
      // diagonal tiles only (each block's own 64 x 64 bits); everything across blocks is decided by the sweep from the boxes
      hipLaunchKernelGGL(nms_mask_kernel<true>, dim3(ceil_div(col_blocks, 4), 1, B), dim3(256), 0, s,
                         (const float*)sbox, (long)pre_n * 4, 4, (const int*)nullptr, pre_n, iou, mask, diagT, col_blocks, 1);
      hipLaunchKernelGGL(nms_sweep_iou_kernel, dim3(B), dim3(64 * kSweepIouWaves), 0, s, (const float4*)sbox,
                         (const uint32_t*)skey, (const uint64_t*)mask, (const uint64_t*)diagT, pre_n, col_blocks, post_n, iou,
                         rois, scores);
    }
    LSFA_LAUNCH_CHECK("lsfa_proposal");
    return LSFA_OK;
  }

  int Kpad = 16;
  while (Kpad < pre_n) Kpad <<= 1;
  constexpr size_t kLdsMax = 160 * 1024;
  static PerDeviceOnce lds_attr;
  lds_attr.run([] {
    (void)hipFuncSetAttribute((const void*)proposal_select_nms_kernel<true, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    (void)hipFuncSetAttribute((const void*)proposal_select_nms_kernel<false, 8, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    (void)hipFuncSetAttribute((const void*)proposal_select_nms_kernel<false, 8, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    (void)hipFuncSetAttribute((const void*)proposal_select_nms_kernel<false, 16, false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
  });
  // radix path: sort buffer A = K*8 (>= the 16 KB of select histograms), region = buffer B + counters
  const size_t hist_bytes = (size_t)kHistReplicas * 256 * 4;
  const size_t sortA = align_up((size_t)pre_n * 8 > hist_bytes ? (size_t)pre_n * 8 : hist_bytes, 16);
  const size_t radix_region = align_up((size_t)pre_n * 8, 16) + 16 * kTopkThreads * 2;
  const size_t region_min = radix_region > kNmsStateBytes ? radix_region : kNmsStateBytes;
  const size_t lds_radix_keys = sortA + 256 + align_up(((size_t)count * 4 > region_min ? (size_t)count * 4 : region_min), 16);
  const size_t lds_radix = sortA + 256 + align_up(region_min, 16);
  const size_t lds_bitonic = (size_t)Kpad * 8 + 256 + kNmsStateBytes;
#define LSFA_LAUNCH_SELECT(KL, PER, RAD, AB, LDS, SORTB)                                                     \
  hipLaunchKernelGGL((proposal_select_nms_kernel<KL, PER, RAD, AB>), dim3(B), dim3(kTopkThreads), (LDS), s,   \
                     (const uint32_t*)keys, (const float4*)boxes, count, pre_n, Kpad, (int)(SORTB), iou, post_n, \
                     rois, scores, (unsigned long long*)nullptr)
  // 8193..16384 candidates: 128 KB sort buffer, so the alive list keeps indices only (boxes re-read from L2)
  const size_t lds_bitonic16 = (size_t)Kpad * 8 + 256 + kNmsCoreBytes + 1024 * 4;
  if (pre_n <= 8192 && lds_radix_keys <= kLdsMax) LSFA_LAUNCH_SELECT(true, 8, true, true, lds_radix_keys, sortA);
  else if (pre_n <= 8192 && lds_radix <= kLdsMax) LSFA_LAUNCH_SELECT(false, 8, true, true, lds_radix, sortA);
  else if (Kpad <= 8192) LSFA_LAUNCH_SELECT(false, 8, false, true, lds_bitonic, (size_t)Kpad * 8);
  else LSFA_LAUNCH_SELECT(false, 16, false, false, lds_bitonic16, (size_t)Kpad * 8);
#undef LSFA_LAUNCH_SELECT
  LSFA_LAUNCH_CHECK("lsfa_proposal");
  return LSFA_OK;
}
