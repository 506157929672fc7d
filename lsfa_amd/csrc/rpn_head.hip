// RPN head: rpn_cls_score and rpn_bbox_pred (two 1x1 convolutions of channels [0, 512) of the feature map,
// resnet_v1_101_flownet_rfcn.py:479-488) + bias + the two-way softmax over (background, foreground) of every anchor (:489-494:
// Reshape (2, A*H, W) -> SoftmaxActivation(channel) -> Reshape) as ONE launch on the NCHW map the reference's operators exchange.
// 54 outputs x 512 inputs per pixel is 0.13 GFLOP per frame: too small for matrix-pipe tiles and K-major (NCHW) for a row GEMM, so
// it is plain fp32 FMA with lane = pixel (the map's planes are read coalesced) and the weights broadcast from scalar registers.
// A workgroup = 64 pixels x 8 waves.  The 64-pixel runs of the 512 planes go through LDS in eight chunks of 64 planes (each wave
// fetches eight of a chunk's planes, one chunk ahead in registers): every plane is read from memory ONCE per workgroup.  Wave w owns
// outputs [8 w, 8 w + 8) and walks all 512 input channels in ascending order (one fmaf chain per output; the chunk's weights sit in LDS
// too: a broadcast read of 32 bytes per channel), so no partial sums meet anywhere.  The logits cross waves once, through LDS, for the softmax's (a, A + a) pairs;
// stores are coalesced (a plane's 64 consecutive pixels per output channel).
// (Measured on nine frames: a wave per 64 input channels with all 64 outputs - 64 scalar registers of weights per channel, nothing
// loaded ahead - 104 us; every wave reading all planes itself - eight times the traffic - 147 us; planes through LDS but weights by scalar
// loads - both wait on the same counter, out of order - 60 us.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "lsfa_hip.h"

namespace {
constexpr int kRpnCin = 512, kRpnOut = 64, kRpnWaves = 8, kRpnPix = 64, kRpnPerWave = kRpnOut / kRpnWaves, kRpnChunk = 64;

// grid (ceil(HW / 64), N); block 512.  w_t: (512, 64) floats [cin][out] (outputs past 6A are zero columns); bias (64)
__global__ __launch_bounds__(kRpnPix * kRpnWaves) void rpn_head_kernel(const float* __restrict__ feat, long img_stride, int HW,
                                                                       const float* __restrict__ w_t, const float* __restrict__ bias, int A,
                                                                       float* __restrict__ cls_prob, float* __restrict__ bbox) {
  __shared__ float logit[kRpnOut][kRpnPix + 1];
  __shared__ float xs[2][kRpnChunk][kRpnPix];                     // two chunks of 64 planes x 64 pixels
  __shared__ __attribute__((aligned(16))) float ws[2][kRpnChunk * kRpnOut];      // and of their 64 x 64 weights
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = blockIdx.y;
  const int p = blockIdx.x * kRpnPix + lane;
  const bool ok = p < HW;
  const float* x = feat + (size_t)n * img_stride + (size_t)(wave * 8) * HW + (ok ? p : 0);      // this wave's eight planes of a chunk
  const float4* wg = reinterpret_cast<const float4*>(w_t) + threadIdx.x * 2;                    // this thread's eight weights of a chunk
  float acc[kRpnPerWave];
#pragma unroll
  for (int o = 0; o < kRpnPerWave; ++o) acc[o] = 0.f;
  float xn[8];
  float4 wn0 = wg[0], wn1 = wg[1];
#pragma unroll
  for (int j = 0; j < 8; ++j) xn[j] = ok ? x[(size_t)j * HW] : 0.f;
  for (int c = 0; c < kRpnCin / kRpnChunk; ++c) {
    float (*buf)[kRpnPix] = xs[c & 1];
    float* wb = ws[c & 1];
#pragma unroll
    for (int j = 0; j < 8; ++j) buf[wave * 8 + j][lane] = xn[j];
    reinterpret_cast<float4*>(wb)[threadIdx.x * 2] = wn0;
    reinterpret_cast<float4*>(wb)[threadIdx.x * 2 + 1] = wn1;
    __syncthreads();          // also: every wave is done reading the other buffers (they were these two chunks ago)
    if (c + 1 < kRpnCin / kRpnChunk) {
#pragma unroll
      for (int j = 0; j < 8; ++j) xn[j] = ok ? x[(size_t)((c + 1) * kRpnChunk + j) * HW] : 0.f;
      wn0 = wg[(size_t)(c + 1) * (kRpnChunk * kRpnOut / 4)];
      wn1 = wg[(size_t)(c + 1) * (kRpnChunk * kRpnOut / 4) + 1];
    }
    // everything the loop reads comes from LDS (one counter, in order): the pixel's value and the wave's eight weights (a broadcast read)
#pragma unroll 8
    for (int k = 0; k < kRpnChunk; ++k) {
      const float xv = buf[k][lane];
      const float4 w0 = *reinterpret_cast<const float4*>(wb + k * kRpnOut + wave * kRpnPerWave);
      const float4 w1 = *reinterpret_cast<const float4*>(wb + k * kRpnOut + wave * kRpnPerWave + 4);
      acc[0] = fmaf(xv, w0.x, acc[0]); acc[1] = fmaf(xv, w0.y, acc[1]); acc[2] = fmaf(xv, w0.z, acc[2]); acc[3] = fmaf(xv, w0.w, acc[3]);
      acc[4] = fmaf(xv, w1.x, acc[4]); acc[5] = fmaf(xv, w1.y, acc[5]); acc[6] = fmaf(xv, w1.z, acc[6]); acc[7] = fmaf(xv, w1.w, acc[7]);
    }
  }
#pragma unroll
  for (int o = 0; o < kRpnPerWave; ++o) logit[wave * kRpnPerWave + o][lane] = acc[o] + bias[wave * kRpnPerWave + o];
  __syncthreads();
  // stores: thread -> (output o, pixel lane), o = wave, wave + 8, ...; channels [0, 2A) are the scores (background a, foreground A + a)
  if (!ok) return;
  for (int o = wave; o < 6 * A; o += kRpnWaves) {
    const float v = logit[o][lane];
    if (o < 2 * A) {
      const int a = o < A ? o : o - A;
      const float bg = logit[a][lane], fg = logit[A + a][lane];
      const float m = fmaxf(bg, fg);
      const float eb = expf(bg - m), ef = expf(fg - m);
      cls_prob[((size_t)n * 2 * A + o) * HW + p] = (o < A ? eb : ef) / (eb + ef);
    } else {
      bbox[((size_t)n * 4 * A + (o - 2 * A)) * HW + p] = v;
    }
  }
}
}  // namespace

extern "C" int lsfa_rpn_head(const float* feat, int N, int C_total, int H, int W, const float* w_t, const float* bias, int A,
                             float* cls_prob, float* bbox_pred, void* stream) {
  LSFA_REQUIRE(feat && w_t && bias && cls_prob && bbox_pred, "lsfa_rpn_head: NULL argument");
  LSFA_REQUIRE(N > 0 && N <= 65535 && H > 0 && W > 0 && C_total >= kRpnCin, "lsfa_rpn_head: bad shape N=%d C=%d H=%d W=%d (the head reads channels [0, 512))", N,
               C_total, H, W);
  if (A < 1 || 6 * A > kRpnOut) {
    lsfa::set_error("lsfa_rpn_head: %d anchors per position unsupported (6 A <= %d)", A, kRpnOut);
    return LSFA_ENOTSUP;
  }
  const int HW = H * W;
  hipLaunchKernelGGL(rpn_head_kernel, dim3((unsigned)((HW + kRpnPix - 1) / kRpnPix), (unsigned)N), dim3(kRpnPix * kRpnWaves), 0, (hipStream_t)stream, feat,
                     (long)C_total * HW, HW, w_t, bias, A, cls_prob, bbox_pred);
  LSFA_LAUNCH_CHECK("lsfa_rpn_head");
  return LSFA_OK;
}
