// Bilinear feature warp (GridGenerator 'warp' + BilinearSampler) with the fused key-path (x scale_map) and
// cur-path (+ rnet_conv0(res_diff) + small-net feature) epilogues.  Kernel: warp_kernels.h.
// See include/lsfa_hip.h for the reference interfaces it replaces.
#include "warp_kernels.h"

namespace {

using namespace lsfa::warp;

template <int VEC>
void launch(dim3 grid, hipStream_t s, bool has_mul, bool has_add, bool has_res,
            const float* feat, int feat_n, const float* flow, int C, int H, int W, const float* mul,
            const float* add, const float* res, int res_c, const float* res_w, const float* res_b,
            float* out, int cpb) {
#define LSFA_WARP_CASE(M, A, R)                                                                        \
  if (has_mul == M && has_add == A && has_res == R) {                                                  \
    hipLaunchKernelGGL((warp_kernel<VEC, M, A, R>), grid, dim3(kThreads), 0, s, feat, feat_n, flow, C, \
                       H, W, mul, add, res, res_c, res_w, res_b, out, cpb);                            \
    return;                                                                                            \
  }
  LSFA_WARP_CASE(false, false, false)
  LSFA_WARP_CASE(true, false, false)
  LSFA_WARP_CASE(false, true, false)
  LSFA_WARP_CASE(false, false, true)
  LSFA_WARP_CASE(true, true, false)
  LSFA_WARP_CASE(true, false, true)
  LSFA_WARP_CASE(false, true, true)
  LSFA_WARP_CASE(true, true, true)
#undef LSFA_WARP_CASE
}

inline bool aligned(const void* p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }



}  // namespace

extern "C" int lsfa_warp_bilinear(const float* feat, int feat_n, const float* flow, int N, int C, int H,
                                  int W, const float* mul, const float* add, const float* res, int res_c,
                                  const float* res_w, const float* res_b, float* out, void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(feat && flow && out, "lsfa_warp_bilinear: feat, flow and out must be non-NULL");
  LSFA_REQUIRE(N > 0 && C > 0 && H > 1 && W > 1, "lsfa_warp_bilinear: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
  LSFA_REQUIRE(feat_n == 1 || feat_n == N, "lsfa_warp_bilinear: feat batch %d must be 1 or N=%d", feat_n, N);
  LSFA_REQUIRE(N <= 65535, "lsfa_warp_bilinear: N=%d exceeds grid.z", N);
  if (res) {
    LSFA_REQUIRE(res_w && res_b, "lsfa_warp_bilinear: res given without res_w/res_b");
    if (res_c < 1 || res_c > kResMax) {
      set_error("lsfa_warp_bilinear: res_c=%d not in [1,%d]", res_c, kResMax);
      return LSFA_ENOTSUP;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  int vec = (HW % 4 == 0) ? 4 : (HW % 2 == 0) ? 2 : 1;
  const size_t al = sizeof(float) * vec;
  if (!(aligned(flow, al) && aligned(mul, al) && aligned(add, al) && aligned(res, al) && aligned(out, al))) vec = 1;
  const int gx = ceil_div(HW, kThreads * vec);
  // enough workgroups to cover 256 CUs several times over, while amortising the tap
  // computation over the channel run
  int cpb = 8;
  while (cpb > 1 && (long)gx * ceil_div(C, cpb) * N < 1024) cpb >>= 1;
  dim3 grid(gx, ceil_div(C, cpb), N);
  ProfScope prof(LSFA_OP_WARP, s);
  if (vec == 4) launch<4>(grid, s, mul != nullptr, add != nullptr, res != nullptr, feat, feat_n, flow, C, H, W, mul, add, res, res_c, res_w, res_b, out, cpb);
  else if (vec == 2) launch<2>(grid, s, mul != nullptr, add != nullptr, res != nullptr, feat, feat_n, flow, C, H, W, mul, add, res, res_c, res_w, res_b, out, cpb);
  else launch<1>(grid, s, mul != nullptr, add != nullptr, res != nullptr, feat, feat_n, flow, C, H, W, mul, add, res, res_c, res_w, res_b, out, cpb);
  LSFA_LAUNCH_CHECK("lsfa_warp_bilinear");
  return LSFA_OK;
}
