// Device kernels of nms.hip (lsfa_nms_sorted, _nms).
//
// Structure on CDNA4 (replaces nms_kernel + D2H of the 4.5 MB mask + serial host sweep +
// H2D of the keep list, lib/nms/nms_kernel.cu:40-150, multi_proposal.cu:262-357):
//   1. nms_mask_kernel — 64x64 IoU tiles, one 64-lane wave per tile, upper triangle only
//      (the sweep never reads words left of the diagonal); bit k of word (i, cb) is set
//      iff IoU(box_i, box_{64cb+k}) > thresh and 64cb+k > i, as in the reference.
//   2. nms_sweep_kernel — ONE wave walks the 64-box blocks in score order.  Per block it
//      loads the 64 diagonal words (one per lane), resolves the block with a 64-step
//      scalar loop on SGPRs (v_readlane of the diagonal word of each survivor), then ORs
//      the survivors' mask rows into the removed-set kept in LDS, loads batched 8 rows
//      at a time so their latencies overlap.  It stops as soon as `max_keep` survivors exist.
// (Proposal does not use these: it keeps everything in one workgroup's LDS, proposal.hip.)
// devIoU arithmetic = oracle dev_iou (nms_kernel.cu:30-38), -ffp-contract=off.
#pragma once
#include "common.h"

namespace lsfa {

__device__ __forceinline__ float dev_iou(const float* a, const float* b) {
  const float left = fmaxf(a[0], b[0]), right = fminf(a[2], b[2]);
  const float top = fmaxf(a[1], b[1]), bottom = fminf(a[3], b[3]);
  const float width = fmaxf(right - left + 1, 0.f), height = fmaxf(bottom - top + 1, 0.f);
  const float interS = width * height;
  const float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  const float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

// grid (col_blocks, col_blocks, images); block 64.  boxes: (images, n, box_dim).
// mask: (images, n, col_blocks) uint64.
static __global__ __launch_bounds__(64) void nms_mask_kernel(const float* __restrict__ boxes_all, int n, int box_dim,
                                                      float thresh, uint64_t* __restrict__ mask_all,
                                                      int col_blocks) {
  const int row_start = blockIdx.y, col_start = blockIdx.x;
  if (col_start < row_start) return;
  const float* boxes = boxes_all + (size_t)blockIdx.z * n * box_dim;
  uint64_t* mask = mask_all + (size_t)blockIdx.z * n * col_blocks;
  const int row_size = min(n - row_start * 64, 64);
  const int col_size = min(n - col_start * 64, 64);
  __shared__ float block_boxes[64 * 4];
  const int t = threadIdx.x;
  if (t < col_size) {
    const float* src = boxes + (size_t)(64 * col_start + t) * box_dim;
    block_boxes[t * 4 + 0] = src[0];
    block_boxes[t * 4 + 1] = src[1];
    block_boxes[t * 4 + 2] = src[2];
    block_boxes[t * 4 + 3] = src[3];
  }
  __syncthreads();
  if (t < row_size) {
    const int cur = 64 * row_start + t;
    const float* src = boxes + (size_t)cur * box_dim;
    const float cur_box[4] = {src[0], src[1], src[2], src[3]};
    uint64_t bits = 0;
    const int start = (row_start == col_start) ? t + 1 : 0;
    for (int i = start; i < col_size; ++i) {
      if (dev_iou(cur_box, block_boxes + i * 4) > thresh) bits |= 1ULL << i;
    }
    mask[(size_t)cur * col_blocks + col_start] = bits;
  }
}

constexpr int kSweepMaxBlocks = 512;   // n <= 32768

// grid (images); block 64 (one wave).  keep: (images, n) indices into boxes; num_keep: (images).
__global__ __launch_bounds__(64) void nms_sweep_kernel(const uint64_t* __restrict__ mask_all, int n, int col_blocks,
                                                       int max_keep, int* __restrict__ keep_all,
                                                       int* __restrict__ num_keep_all) {
  __shared__ uint64_t remv[kSweepMaxBlocks];
  const int img = blockIdx.x;
  const uint64_t* mask = mask_all + (size_t)img * n * col_blocks;
  int* keep = keep_all + (size_t)img * n;
  const int lane = threadIdx.x;
  for (int i = lane; i < col_blocks; i += 64) remv[i] = 0;
  __syncthreads();  // single-wave workgroup: orders the LDS accesses of different lanes
  int num = 0;
  for (int b = 0; b < col_blocks && num < max_keep; ++b) {
    const int base = b * 64;
    const int nb = min(64, n - base);
    const uint64_t diag = (lane < nb) ? mask[(size_t)(base + lane) * col_blocks + b] : 0ULL;
    const uint64_t cur0 = remv[b];
    uint32_t cur_lo = __builtin_amdgcn_readfirstlane((uint32_t)cur0);
    uint32_t cur_hi = __builtin_amdgcn_readfirstlane((uint32_t)(cur0 >> 32));
    const uint32_t diag_lo = (uint32_t)diag, diag_hi = (uint32_t)(diag >> 32);
    uint32_t kept_lo = 0, kept_hi = 0;
    int budget = max_keep - num;
    for (int k = 0; k < nb && budget > 0; ++k) {
      const bool removed = k < 32 ? ((cur_lo >> k) & 1u) : ((cur_hi >> (k - 32)) & 1u);
      if (!removed) {
        if (k < 32) kept_lo |= 1u << k; else kept_hi |= 1u << (k - 32);
        cur_lo |= __builtin_amdgcn_readlane(diag_lo, k);
        cur_hi |= __builtin_amdgcn_readlane(diag_hi, k);
        --budget;
      }
    }
    const uint64_t kept = ((uint64_t)kept_hi << 32) | kept_lo;
    if ((kept >> lane) & 1ULL) keep[num + __popcll(kept & ((1ULL << lane) - 1ULL))] = base + lane;
    num += __popcll(kept);
    if (num >= max_keep || b + 1 >= col_blocks) break;
    // OR the survivors' rows into remv for the blocks to the right of b, 8 rows per batch of loads
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
  if (lane == 0) num_keep_all[img] = num;
}

}  // namespace lsfa
