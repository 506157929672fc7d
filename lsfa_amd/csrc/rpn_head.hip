// RPN head: rpn_cls_score and rpn_bbox_pred (two 1x1 convolutions of channels [0, 512) of the feature map,
// resnet_v1_101_flownet_rfcn.py:479-488) + bias + the two-way softmax over (background, foreground) of every anchor (:489-494:
// Reshape (2, A*H, W) -> SoftmaxActivation(channel) -> Reshape) as ONE launch on the NCHW map the reference's operators exchange.
// 54 outputs x 512 inputs per pixel is 0.13 GFLOP per frame: too small for matrix-pipe tiles and K-major (NCHW) for a row GEMM, so
// it is plain fp32 FMA with lane = pixel (the map's planes are read coalesced, each once) and the weights broadcast from scalar
// registers: a workgroup = 64 pixels x 8 waves, wave w sums input channels [64 w, 64 w + 64) for all outputs in ascending channel
// order (one fmaf chain per output), the eight partial sums meet in LDS and are added in wave order, then bias, softmax, stores
// (coalesced: a plane's 64 consecutive pixels per output channel).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"
#include "lsfa_hip.h"

namespace {
constexpr int kRpnCin = 512, kRpnOut = 64, kRpnWaves = 8, kRpnPix = 64;

// grid (ceil(HW / 64), N); block 512.  w_t: (512, 64) floats [cin][out] (outputs past 6A are zero columns); bias (64)
__global__ __launch_bounds__(kRpnPix * kRpnWaves) void rpn_head_kernel(const float* __restrict__ feat, long img_stride, int HW,
                                                                       const float* __restrict__ w_t, const float* __restrict__ bias, int A,
                                                                       float* __restrict__ cls_prob, float* __restrict__ bbox) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float (*part)[kRpnOut][kRpnPix] = reinterpret_cast<float (*)[kRpnOut][kRpnPix]>(smem);      // waves 1..7 (wave 0 keeps its sums in registers): 114,688 bytes
  float (*logit)[kRpnPix + 1] = reinterpret_cast<float (*)[kRpnPix + 1]>(smem + (kRpnWaves - 1) * kRpnOut * kRpnPix);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int n = blockIdx.y;
  const int p = blockIdx.x * kRpnPix + lane;
  const bool ok = p < HW;
  const float* x = feat + (size_t)n * img_stride + (size_t)(wave * (kRpnCin / kRpnWaves)) * HW + (ok ? p : 0);
  const float* w = w_t + (size_t)(wave * (kRpnCin / kRpnWaves)) * kRpnOut;
  float acc[kRpnOut];
#pragma unroll
  for (int o = 0; o < kRpnOut; ++o) acc[o] = 0.f;
  for (int k0 = 0; k0 < kRpnCin / kRpnWaves; k0 += 8) {
    float xv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) xv[j] = ok ? x[(size_t)(k0 + j) * HW] : 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* wk = w + (size_t)(k0 + j) * kRpnOut;          // wave-uniform address: scalar loads
#pragma unroll
      for (int o = 0; o < kRpnOut; ++o) acc[o] = fmaf(xv[j], wk[o], acc[o]);
    }
  }
  if (wave > 0) {
#pragma unroll
    for (int o = 0; o < kRpnOut; ++o) part[wave - 1][o][lane] = acc[o];
  }
  __syncthreads();
  if (wave == 0) {
#pragma unroll
    for (int o = 0; o < kRpnOut; ++o) {
      float v = acc[o];
#pragma unroll
      for (int s = 0; s < kRpnWaves - 1; ++s) v = v + part[s][o][lane];
      logit[o][lane] = v + bias[o];
    }
  }
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
  const size_t lds = ((size_t)(kRpnWaves - 1) * kRpnOut * kRpnPix + (size_t)kRpnOut * (kRpnPix + 1)) * sizeof(float);
  static lsfa::PerDeviceOnce lds_attr;
  lds_attr.run([] { (void)hipFuncSetAttribute((const void*)rpn_head_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
  hipLaunchKernelGGL(rpn_head_kernel, dim3((unsigned)((HW + kRpnPix - 1) / kRpnPix), (unsigned)N), dim3(kRpnPix * kRpnWaves), lds, (hipStream_t)stream, feat,
                     (long)C_total * HW, HW, w_t, bias, A, cls_prob, bbox_pred);
  LSFA_LAUNCH_CHECK("lsfa_rpn_head");
  return LSFA_OK;
}
