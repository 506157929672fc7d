// RPN proposal generation, fully device resident (no malloc/free, no D2H, no host sweep):
//   proposal_decode_kernel  anchors + decode + clip + filter (ProposalGrid/BBoxPred/FilterBox,
//                           multi_proposal.cu:47-216) -> float4 boxes + 32-bit order keys
//   proposal_select_nms_kernel   ONE workgroup (16 waves) per image, everything in its CU's LDS:
//       radix-select the pre_nms_top_n best (score desc, anchor index asc ==
//       thrust::stable_sort_by_key(greater), :517-521) with the 21,546 keys held in LDS (86 KB),
//       bitonic-sort just those 6,000 (8 keys per thread in registers, lane shuffles below stride
//       512, LDS only above), then greedy NMS with IoUs computed on the fly against the survivors
//       (no 4.5 MB mask, no second kernel, stops at post_nms_top_n survivors) and PrepareOutput
//       (:363-388).
// Selecting before sorting cuts the sort from 21,546 to 6,000 keys; the NMS only ever evaluates
// (candidates visited) x (survivors so far) IoUs instead of the full 6000^2/2 mask.
// Arithmetic = oracle orc_proposal_decode, operation for operation.
#include <math.h>

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

// thread t -> (a, h, w) with w fastest so the five NCHW planes are read coalesced;
// output position follows the reference's enumeration index = (h*W + w)*A + a.
__global__ __launch_bounds__(256) void proposal_decode_kernel(
    const float* __restrict__ cls_prob, const float* __restrict__ bbox_pred, const float* __restrict__ im_info,
    int A, int H, int W, int feature_stride, int rpn_min_size, Anchors anchors, float4* __restrict__ boxes_all,
    uint32_t* __restrict__ keys_all) {
  const int b = blockIdx.y;
  const int HW = H * W;
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= A * HW) return;
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
  keys_all[index] = desc_key(score);
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
    for (int j = k >> 1; j > 0; j >>= 1) {
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
            const bool up = (e & k) == 0;
            const bool lower = (e & j) == 0;
            const uint64_t mn = v[r] < p ? v[r] : p, mx = v[r] < p ? p : v[r];
            v[r] = (lower == up) ? mn : mx;
          }
        }
      } else if (j >= kPerThread) {
        const int lane_mask = j / kPerThread;   // < 64
#pragma unroll
        for (int r = 0; r < kPerThread; ++r) {
          const int e = tid * kPerThread + r;
          const uint64_t p = __shfl_xor(v[r], lane_mask, 64);
          const bool up = (e & k) == 0;
          const bool lower = (e & j) == 0;
          const uint64_t mn = v[r] < p ? v[r] : p, mx = v[r] < p ? p : v[r];
          v[r] = (lower == up) ? mn : mx;
        }
      } else {
        // in registers: pairs (r, r^j) with r & j == 0; direction from the element index
#pragma unroll
        for (int r = 0; r < kPerThread; ++r) {
          if ((r & j) == 0) {
            const int e = tid * kPerThread + r;
            cswap(v[r], v[r | j], (e & k) == 0);
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
constexpr int kHistReplicas = 16;
constexpr size_t kNmsStateBytes = 1024 * 16 + 1024 * 4 + 64 * 16 + 64 * 8;

template <bool KEYS_LDS, int kPerThread>
__global__ __launch_bounds__(kTopkThreads) void proposal_select_nms_kernel(
    const uint32_t* __restrict__ keys_all, const float4* __restrict__ boxes_all, int N, int K, int Kpad,
    float thresh, int post_n, float* __restrict__ rois, float* __restrict__ scores) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem);
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);
  int* misc = reinterpret_cast<int*>(smem + (size_t)Kpad * 8);
  unsigned char* region = smem + (size_t)Kpad * 8 + 256;
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

  // ---- compaction: keys < T in any order; keys == T in index order, first r_eq -----------
  if (tid == 0) misc[2] = 0;
  for (int i = K + tid; i < Kpad; i += kTopkThreads) sortbuf[i] = ~0ULL;
  const int chunk = (N + kTopkThreads - 1) / kTopkThreads;
  const int i0 = min(tid * chunk, N), i1 = min(i0 + chunk, N);
  int my_eq = 0;
  for (int i = i0; i < i1; ++i) my_eq += (keys[i] == T);
  int total_eq;
  int eq_rank = block_exclusive_scan(my_eq, misc + 8, &total_eq);  // barriers inside order misc[2] = 0
  for (int i = i0; i < i1; ++i) {
    const uint32_t k = keys[i];
    if (k < T) {
      const int slot = atomicAdd(&misc[2], 1);
      sortbuf[slot] = ((uint64_t)k << 32) | (uint32_t)i;
    } else if (k == T) {
      if (eq_rank < r_eq) sortbuf[n_less + eq_rank] = ((uint64_t)k << 32) | (uint32_t)i;
      ++eq_rank;
    }
  }
  __syncthreads();

  // ---- sort (ascending composite key == score desc, anchor index asc) ---------------------
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

  // ---- greedy NMS over the sorted candidates, IoU on the fly against the survivors --------
  // (same decisions as nms_kernel + the host sweep, multi_proposal.cu:262-357).  Per block of 64
  // candidates: wave q tests candidate `lane` against survivors q, q+16, ...; then the 64x64
  // in-block relation is built column-wise by ballots and resolved by a 64-step scalar loop.
  float4* kept_box = reinterpret_cast<float4*>(region);
  int* kept_idx = reinterpret_cast<int*>(region + 1024 * 16);
  float4* cand_lds = reinterpret_cast<float4*>(region + 1024 * 16 + 1024 * 4);
  uint64_t* colsupp = reinterpret_cast<uint64_t*>(region + 1024 * 16 + 1024 * 4 + 64 * 16);
  unsigned int* supp = reinterpret_cast<unsigned int*>(misc + 4);  // 2 words
  if (tid == 0) { misc[3] = 0; supp[0] = 0; supp[1] = 0; }
  const int nblocks = (K + 63) / 64;
  float4 next_cand = make_float4(0.f, 0.f, 0.f, 0.f);
  if (wid == 0 && lane < K) next_cand = boxes[(uint32_t)sortbuf[lane]];
  __syncthreads();
  int num = 0;
  for (int b = 0; b < nblocks && num < post_n; ++b) {
    const int base = b * 64;
    const int nb = min(64, K - base);
    if (wid == 0) {
      cand_lds[lane] = next_cand;
      const int nxt = base + 64 + lane;
      if (nxt < K) next_cand = boxes[(uint32_t)sortbuf[nxt]];   // prefetch the next block's boxes
    }
    __syncthreads();
    const float4 cb = cand_lds[lane];
    const float cbox[4] = {cb.x, cb.y, cb.z, cb.w};
    bool s = false;
    for (int k = wid; k < num; k += kTopkThreads / 64) {
      const float4 kb = kept_box[k];
      const float kbox[4] = {kb.x, kb.y, kb.z, kb.w};
      s = s || (dev_iou(kbox, cbox) > thresh);
    }
    const unsigned long long bal = __ballot(s && lane < nb);
    if (lane == 0 && bal) {
      atomicOr(&supp[0], (unsigned int)bal);
      atomicOr(&supp[1], (unsigned int)(bal >> 32));
    }
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int j = wid * 4 + jj;
      const float4 jb = cand_lds[j];
      const float jbox[4] = {jb.x, jb.y, jb.z, jb.w};
      const bool pred = (lane < j) && (j < nb) && (dev_iou(cbox, jbox) > thresh);
      const unsigned long long col = __ballot(pred);
      if (lane == 0) colsupp[j] = col;
    }
    __syncthreads();
    if (wid == 0) {
      const uint64_t col = colsupp[lane];
      const uint32_t col_lo = (uint32_t)col, col_hi = (uint32_t)(col >> 32);
      const uint32_t s_lo = __builtin_amdgcn_readfirstlane(supp[0]), s_hi = __builtin_amdgcn_readfirstlane(supp[1]);
      uint32_t kept_lo = 0, kept_hi = 0;
      int budget = post_n - num;
      for (int k = 0; k < nb && budget > 0; ++k) {
        const uint32_t c_lo = __builtin_amdgcn_readlane(col_lo, k), c_hi = __builtin_amdgcn_readlane(col_hi, k);
        const bool sup_prev = k < 32 ? ((s_lo >> k) & 1u) : ((s_hi >> (k - 32)) & 1u);
        const bool removed = sup_prev || ((c_lo & kept_lo) | (c_hi & kept_hi)) != 0;
        if (!removed) {
          if (k < 32) kept_lo |= 1u << k; else kept_hi |= 1u << (k - 32);
          --budget;
        }
      }
      const uint64_t kept = ((uint64_t)kept_hi << 32) | kept_lo;
      if ((kept >> lane) & 1ULL) {
        const int pos = num + __popcll(kept & ((1ULL << lane) - 1ULL));
        kept_box[pos] = cb;
        kept_idx[pos] = base + lane;
      }
      if (lane == 0) { misc[3] = num + __popcll(kept); supp[0] = 0; supp[1] = 0; }
    }
    __syncthreads();
    num = misc[3];
  }

  // ---- PrepareOutput (multi_proposal.cu:363-388): first post_n survivors, cyclic pad ------
  for (int index = tid; index < post_n; index += kTopkThreads) {
    const int kpos = index < num ? index : index % num;
    const float4 bx = kept_box[kpos];
    float* o = rois + ((size_t)img * post_n + index) * 5;
    o[0] = (float)img;
    o[1] = bx.x; o[2] = bx.y; o[3] = bx.z; o[4] = bx.w;
    if (scores) scores[(size_t)img * post_n + index] = key_score((uint32_t)(sortbuf[kept_idx[kpos]] >> 32));
  }
}

struct WsLayout {
  size_t boxes, keys, total;
};
WsLayout ws_layout(int B, int count, int pre_n) {
  (void)pre_n;
  WsLayout l;
  size_t o = 0;
  l.boxes = o; o += align_up((size_t)B * count * sizeof(float4), 256);
  l.keys = o; o += align_up((size_t)B * count * sizeof(uint32_t), 256);
  l.total = o;
  return l;
}
int clamp_pre_n(int rpn_pre_nms_top_n, int count) {
  int pre_n = rpn_pre_nms_top_n > 0 ? rpn_pre_nms_top_n : count;  // multi_proposal.cu:435-436
  return pre_n < count ? pre_n : count;
}

}  // namespace

extern "C" size_t lsfa_proposal_workspace_bytes(int B, int A, int H, int W, int pre_nms_top_n) {
  if (B <= 0 || A <= 0 || H <= 0 || W <= 0) return 0;
  const int count = A * H * W;
  return ws_layout(B, count, clamp_pre_n(pre_nms_top_n, count)).total;
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
  if (pre_n > 16384 || post_n > kSweepKeepLds || B > 65535) {
    set_error("lsfa_proposal: pre_nms_top_n %d (max 16384) / post_nms_top_n %d (max %d) unsupported", pre_n, post_n,
              kSweepKeepLds);
    return LSFA_ENOTSUP;
  }
  const WsLayout l = ws_layout(B, count, pre_n);
  if (ws_bytes < l.total) { set_error("lsfa_proposal: workspace %zu < %zu bytes", ws_bytes, l.total); return LSFA_EWORKSPACE; }
  hipStream_t s = (hipStream_t)stream;
  unsigned char* base = (unsigned char*)ws;
  float4* boxes = (float4*)(base + l.boxes);
  uint32_t* keys = (uint32_t*)(base + l.keys);

  Anchors anchors;
  generate_anchors(feature_stride, ratios_host, n_ratios, scales_host, n_scales, &anchors);

  ProfScope prof(LSFA_OP_PROPOSAL, s);
  hipLaunchKernelGGL(proposal_decode_kernel, dim3(ceil_div(count, 256), B), dim3(256), 0, s, cls_prob, bbox_pred,
                     im_info, A, H, W, feature_stride, rpn_min_size, anchors, boxes, keys);

  int Kpad = 16;
  while (Kpad < pre_n) Kpad <<= 1;
  const size_t lds_base = (size_t)Kpad * 8 + 256;
  size_t lds_keys = lds_base + ((size_t)count * 4 > kNmsStateBytes ? (size_t)count * 4 : kNmsStateBytes);
  lds_keys = align_up(lds_keys, 16);
  const size_t lds_nokeys = lds_base + kNmsStateBytes;
  const size_t kLdsMax = 160 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)proposal_select_nms_kernel<true, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    (void)hipFuncSetAttribute((const void*)proposal_select_nms_kernel<false, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    (void)hipFuncSetAttribute((const void*)proposal_select_nms_kernel<false, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    attr_set = true;
  }
  if (Kpad > 8192) {   // 8193..16384 candidates: 16 keys per thread, keys stay in global memory (L2)
    hipLaunchKernelGGL((proposal_select_nms_kernel<false, 16>), dim3(B), dim3(kTopkThreads), lds_nokeys, s, (const uint32_t*)keys,
                       (const float4*)boxes, count, pre_n, Kpad, threshold, post_n, rois, scores);
  } else if (lds_keys <= kLdsMax) {
    hipLaunchKernelGGL((proposal_select_nms_kernel<true, 8>), dim3(B), dim3(kTopkThreads), lds_keys, s, (const uint32_t*)keys,
                       (const float4*)boxes, count, pre_n, Kpad, threshold, post_n, rois, scores);
  } else {
    hipLaunchKernelGGL((proposal_select_nms_kernel<false, 8>), dim3(B), dim3(kTopkThreads), lds_nokeys, s, (const uint32_t*)keys,
                       (const float4*)boxes, count, pre_n, Kpad, threshold, post_n, rois, scores);
  }
  LSFA_LAUNCH_CHECK("lsfa_proposal");
  return LSFA_OK;
}
