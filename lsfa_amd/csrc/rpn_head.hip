// RPN head, second half: the per-anchor two-way softmax and the split into the two NCHW maps MultiProposal takes.
// rpn_cls_score and rpn_bbox_pred (two 1x1 convolutions of channels [0, 512) of the feature map, resnet_v1_101_flownet_rfcn.py:479-488)
// run as ONE convolution of the own family straight on the NCHW map the reference's operators exchange (lsfa_conv_fwd with x_nchw: the
// direct kernel reads the K-major operand as 128-byte runs of the planes), which leaves `logits` (N, H*W, ld) channels-last: column o <
// 2A = score channel o (background a = o, foreground A + a), 2A <= o < 6A = box delta o - 2A.  This kernel turns a 64-pixel x ld tile
// through LDS, applies Reshape (2, A*H, W) -> SoftmaxActivation(channel) -> Reshape (:489-494) to the score pairs and stores
// rpn_cls_prob (N, 2A, H, W) and rpn_bbox_pred (N, 4A, H, W) as 256-byte runs of their planes.
// (r4 history: the 0.13-GFLOP contraction as fp32 FMAs with lane = pixel took 53-147 us for nine frames in four variants - scalar-register
// weights without room to load ahead, every wave re-reading all planes, weights and planes both through LDS - against ~15 on the matrix pipe.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "lsfa_hip.h"

namespace {
constexpr int kPix = 64, kMaxOut = 64;

// grid (ceil(HW / 64), N); block 256
__global__ __launch_bounds__(256) void rpn_softmax_split_kernel(const float* __restrict__ logits, int HW, int ld, int A,
                                                                float* __restrict__ cls_prob, float* __restrict__ bbox) {
  __shared__ float t[kMaxOut][kPix + 1];
  const int n = blockIdx.y, p0 = blockIdx.x * kPix;
  const int nout = 6 * A;
  // in: 256 threads walk the tile's rows, `ld` floats each, coalesced along the channels
  for (int e = threadIdx.x; e < kPix * nout; e += 256) {
    const int px = e / nout, o = e - px * nout;
    if (p0 + px < HW) t[o][px] = logits[((size_t)n * HW + p0 + px) * ld + o];
  }
  __syncthreads();
  // out: thread -> (channel o = wave, wave + 4, ..., pixel lane)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int p = p0 + lane;
  if (p >= HW) return;
  for (int o = wave; o < nout; o += 4) {
    if (o < 2 * A) {
      const int a = o < A ? o : o - A;
      const float bg = t[a][lane], fg = t[A + a][lane];
      const float m = fmaxf(bg, fg);
      const float eb = expf(bg - m), ef = expf(fg - m);
      cls_prob[((size_t)n * 2 * A + o) * HW + p] = (o < A ? eb : ef) / (eb + ef);
    } else {
      bbox[((size_t)n * 4 * A + (o - 2 * A)) * HW + p] = t[o][lane];
    }
  }
}
}  // namespace

extern "C" int lsfa_rpn_softmax_split(const float* logits, int N, int H, int W, int ld, int A, float* cls_prob, float* bbox_pred, void* stream) {
  LSFA_REQUIRE(logits && cls_prob && bbox_pred, "lsfa_rpn_softmax_split: NULL argument");
  LSFA_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0, "lsfa_rpn_softmax_split: bad shape N=%d H=%d W=%d", N, H, W);
  if (A < 1 || 6 * A > kMaxOut || ld < 6 * A) {
    lsfa::set_error("lsfa_rpn_softmax_split: %d anchors per position with rows of %d floats unsupported (6 A <= %d, ld >= 6 A)", A, ld, kMaxOut);
    return LSFA_ENOTSUP;
  }
  const int HW = H * W;
  hipLaunchKernelGGL(rpn_softmax_split_kernel, dim3((unsigned)((HW + kPix - 1) / kPix), (unsigned)N), dim3(256), 0, (hipStream_t)stream, logits, HW, ld, A,
                     cls_prob, bbox_pred);
  LSFA_LAUNCH_CHECK("lsfa_rpn_softmax_split");
  return LSFA_OK;
}
