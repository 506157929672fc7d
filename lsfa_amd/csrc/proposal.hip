// RPN proposal generation, fully device resident (no malloc/free, no D2H, no host sweep):
//   proposal_decode_kernel  anchors + decode + clip + filter (ProposalGrid/BBoxPred/FilterBox,
//                           multi_proposal.cu:47-216) -> float4 boxes + 32-bit order keys
//   proposal_topk_kernel    ONE workgroup per image: radix-select the pre_nms_top_n best
//                           (score desc, anchor index asc == thrust::stable_sort_by_key(greater),
//                           :517-521) with the keys held in LDS, then a bitonic sort of just those
//                           in LDS, then the gather of ReorderProposals (:238-250)
//   nms_mask_kernel / nms_sweep_kernel<true>   nms_kernels.h (+ PrepareOutput :363-388)
// Selecting before sorting cuts the sort from 21,546 to 6,000 keys and keeps everything in
// one CU's 160 KB LDS: 86 KB of keys + 64 KB sort buffer.
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

template <bool KEYS_LDS>
__global__ __launch_bounds__(kTopkThreads) void proposal_topk_kernel(
    const uint32_t* __restrict__ keys_all, const float4* __restrict__ boxes_all, int N, int K, int Kpad,
    float4* __restrict__ sorted_boxes_all, float* __restrict__ sorted_scores_all) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  uint64_t* sortbuf = reinterpret_cast<uint64_t*>(smem);                        // Kpad
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem + (size_t)Kpad * 8);        // 256
  int* misc = reinterpret_cast<int*>(smem + (size_t)Kpad * 8 + 1024);           // 32 ints
  uint32_t* lkeys = reinterpret_cast<uint32_t*>(smem + (size_t)Kpad * 8 + 1024 + 128);  // N (KEYS_LDS)
  const int img = blockIdx.x;
  const uint32_t* gkeys = keys_all + (size_t)img * N;
  const float4* boxes = boxes_all + (size_t)img * N;
  const int tid = threadIdx.x;
  if (KEYS_LDS) {
    for (int i = tid; i < N; i += kTopkThreads) lkeys[i] = gkeys[i];
    __syncthreads();
  }
  const uint32_t* keys = KEYS_LDS ? lkeys : gkeys;

  // ---- radix select: T = K-th smallest key; r_eq = how many keys == T to take --------
  uint32_t prefix = 0;
  int remaining = K;
  for (int shift = 24; shift >= 0; shift -= 8) {
    if (tid < 256) hist[tid] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += kTopkThreads) {
      const uint32_t k = keys[i];
      const bool match = (shift == 24) || (((k ^ prefix) >> (shift + 8)) == 0);
      if (match) atomicAdd(&hist[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (tid < 64) {
      const uint32_t c0 = hist[tid * 4], c1 = hist[tid * 4 + 1], c2 = hist[tid * 4 + 2], c3 = hist[tid * 4 + 3];
      const int s = (int)(c0 + c1 + c2 + c3);
      int incl = s;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (tid >= d) incl += o;
      }
      const unsigned long long hit = __ballot(incl >= remaining);
      const int first = __builtin_ctzll(hit);  // total count >= remaining, so some lane hits
      if (tid == first) {
        int before = incl - s;
        int bucket = tid * 4;
        const uint32_t cs[4] = {c0, c1, c2, c3};
        for (int q = 0; q < 4; ++q) {
          if (before + (int)cs[q] >= remaining) { bucket = tid * 4 + q; break; }
          before += (int)cs[q];
        }
        misc[0] = bucket;
        misc[1] = remaining - before;
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

  // ---- compaction: keys < T in any order; keys == T in index order, first r_eq --------
  if (tid == 0) misc[2] = 0;
  for (int i = K + tid; i < Kpad; i += kTopkThreads) sortbuf[i] = ~0ULL;
  const int chunk = (N + kTopkThreads - 1) / kTopkThreads;
  const int i0 = min(tid * chunk, N), i1 = min(i0 + chunk, N);
  int my_eq = 0;
  for (int i = i0; i < i1; ++i) my_eq += (keys[i] == T);
  int total_eq;
  int eq_rank = block_exclusive_scan(my_eq, misc + 8, &total_eq);  // also orders misc[2] = 0
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

  // ---- bitonic sort of the Kpad composite keys (ascending) ------------------------------
  for (int k = 2; k <= Kpad; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int t = tid; t < (Kpad >> 1); t += kTopkThreads) {
        const int i = ((t & ~(j - 1)) << 1) | (t & (j - 1));
        const int l = i | j;
        const uint64_t a = sortbuf[i], b = sortbuf[l];
        const bool up = (i & k) == 0;
        if ((a > b) == up) { sortbuf[i] = b; sortbuf[l] = a; }
      }
      __syncthreads();
    }
  }

  // ---- ReorderProposals: gather the K best boxes in order -------------------------------
  float4* sorted_boxes = sorted_boxes_all + (size_t)img * K;
  float* sorted_scores = sorted_scores_all + (size_t)img * K;
  for (int i = tid; i < K; i += kTopkThreads) {
    const uint64_t e = sortbuf[i];
    sorted_boxes[i] = boxes[(uint32_t)e];
    sorted_scores[i] = key_score((uint32_t)(e >> 32));
  }
}

struct WsLayout {
  size_t boxes, keys, sorted_boxes, sorted_scores, mask, total;
};
WsLayout ws_layout(int B, int count, int pre_n) {
  WsLayout l;
  size_t o = 0;
  l.boxes = o; o += align_up((size_t)B * count * sizeof(float4), 256);
  l.keys = o; o += align_up((size_t)B * count * sizeof(uint32_t), 256);
  l.sorted_boxes = o; o += align_up((size_t)B * pre_n * sizeof(float4), 256);
  l.sorted_scores = o; o += align_up((size_t)B * pre_n * sizeof(float), 256);
  l.mask = o; o += align_up((size_t)B * pre_n * ceil_div(pre_n, 64) * sizeof(uint64_t), 256);
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
  float4* sorted_boxes = (float4*)(base + l.sorted_boxes);
  float* sorted_scores = (float*)(base + l.sorted_scores);
  uint64_t* mask = (uint64_t*)(base + l.mask);

  Anchors anchors;
  generate_anchors(feature_stride, ratios_host, n_ratios, scales_host, n_scales, &anchors);

  ProfScope prof(LSFA_OP_PROPOSAL, s);
  hipLaunchKernelGGL(proposal_decode_kernel, dim3(ceil_div(count, 256), B), dim3(256), 0, s, cls_prob, bbox_pred,
                     im_info, A, H, W, feature_stride, rpn_min_size, anchors, boxes, keys);

  int Kpad = 2;
  while (Kpad < pre_n) Kpad <<= 1;
  const size_t lds_base = (size_t)Kpad * 8 + 1024 + 128;
  const size_t lds_keys = lds_base + (size_t)count * 4;
  const size_t kLdsMax = 160 * 1024;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute((const void*)proposal_topk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    (void)hipFuncSetAttribute((const void*)proposal_topk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsMax);
    attr_set = true;
  }
  if (lds_keys <= kLdsMax) {
    hipLaunchKernelGGL(proposal_topk_kernel<true>, dim3(B), dim3(kTopkThreads), lds_keys, s, (const uint32_t*)keys,
                       (const float4*)boxes, count, pre_n, Kpad, sorted_boxes, sorted_scores);
  } else {
    hipLaunchKernelGGL(proposal_topk_kernel<false>, dim3(B), dim3(kTopkThreads), lds_base, s, (const uint32_t*)keys,
                       (const float4*)boxes, count, pre_n, Kpad, sorted_boxes, sorted_scores);
  }
  const int col_blocks = ceil_div(pre_n, 64);
  hipLaunchKernelGGL(nms_mask_kernel, dim3(col_blocks, col_blocks, B), dim3(64), 0, s, (const float*)sorted_boxes, pre_n,
                     4, threshold, mask, col_blocks);
  ProposalOut po{sorted_boxes, sorted_scores, rois, scores, post_n};
  hipLaunchKernelGGL(nms_sweep_kernel<true>, dim3(B), dim3(64), 0, s, (const uint64_t*)mask, pre_n, col_blocks, post_n,
                     (int*)nullptr, (int*)nullptr, po);
  LSFA_LAUNCH_CHECK("lsfa_proposal");
  return LSFA_OK;
}
