// Bilinear feature warp (GridGenerator 'warp' + BilinearSampler) with the fused key-path (x scale_map) and
// cur-path (+ rnet_conv0(res_diff) + small-net feature) epilogues.  Kernel: warp_kernels.h.
// See include/lsfa_hip.h for the reference interfaces it replaces.
#include <atomic>

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

std::atomic<int> g_variant{0};       // lsfa_warp_set_variant: 0 auto, 1 gather kernel only, 2 staged kernel wherever it applies

// The staged kernel (warp_kernels.h, round 3) for one (THREADS, NPAIR, NDMA) instance; returns false when the shape does not fit it.
template <int THREADS, int NPAIR, int NDMA>
bool launch_staged(hipStream_t s, StagedArgs a) {
  constexpr int kStages = 3;
  const int HW = a.H * a.W;
  constexpr int kRegion = THREADS * NDMA * 4;
  a.guard = (2 * a.W + 6 + 3) & ~3;
  // the lanes cover the plane; the DMA pass covers its chunks; the farthest tap (plane start + <= 3 floats of shift + H*W + 2W + 1, taken
  // and discarded for a flow that leaves the map at the bottom right) stays inside the slot
  if (HW > 2 * THREADS * NPAIR || (3 + HW + 3) / 4 > THREADS * NDMA || 3 + HW + 2 * a.W + 2 > kRegion + a.guard) return false;
  const int ops = (a.mul ? 1 : 0) + (a.add ? 1 : 0);
  const size_t lds_bytes = (size_t)kStages * (2 * a.guard + (1 + ops) * kRegion) * 4;
  if (lds_bytes > 160 * 1024) return false;
  static lsfa::PerDeviceOnce attr;
  attr.run([] {
#define LSFA_WS_ATTR(M, A, R) (void)hipFuncSetAttribute((const void*)warp_staged_kernel<THREADS, NPAIR, NDMA, kStages, M, A, R, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LSFA_WS_ATTR(false, false, false) LSFA_WS_ATTR(true, false, false) LSFA_WS_ATTR(false, true, false) LSFA_WS_ATTR(false, false, true)
    LSFA_WS_ATTR(true, true, false) LSFA_WS_ATTR(true, false, true) LSFA_WS_ATTR(false, true, true) LSFA_WS_ATTR(true, true, true)
#define LSFA_WS_ATTR3(M, A) (void)hipFuncSetAttribute((const void*)warp_staged_kernel<THREADS, NPAIR, NDMA, kStages, M, A, true, 3>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    LSFA_WS_ATTR3(false, false) LSFA_WS_ATTR3(true, false) LSFA_WS_ATTR3(false, true) LSFA_WS_ATTR3(true, true)
#undef LSFA_WS_ATTR3
#undef LSFA_WS_ATTR
  });
  const bool m = a.mul != nullptr, ad = a.add != nullptr, r = a.res != nullptr;
  const dim3 grid(a.N * (a.C / a.cg));
#define LSFA_WS_CASE(M, A, R, RC)                                                                                              \
  if (m == M && ad == A && r == R && (RC == 0 || a.res_c == RC)) {                                                             \
    hipLaunchKernelGGL((warp_staged_kernel<THREADS, NPAIR, NDMA, kStages, M, A, R, RC>), grid, dim3(THREADS), lds_bytes, s, a); \
    return true;                                                                                                               \
  }
  // the frame path's residual has 3 channels (rnet_conv0): those instances keep 3 values per pixel and unroll the dot product
  LSFA_WS_CASE(false, false, true, 3) LSFA_WS_CASE(true, false, true, 3) LSFA_WS_CASE(false, true, true, 3) LSFA_WS_CASE(true, true, true, 3)
  LSFA_WS_CASE(false, false, false, 0) LSFA_WS_CASE(true, false, false, 0) LSFA_WS_CASE(false, true, false, 0) LSFA_WS_CASE(false, false, true, 0)
  LSFA_WS_CASE(true, true, false, 0) LSFA_WS_CASE(true, false, true, 0) LSFA_WS_CASE(false, true, true, 0) LSFA_WS_CASE(true, true, true, 0)
#undef LSFA_WS_CASE
  return false;
}



}  // namespace

extern "C" int lsfa_warp_set_variant(int variant) {
  LSFA_REQUIRE(variant >= 0 && variant <= 2, "lsfa_warp_set_variant: unknown variant %d", variant);
  g_variant.store(variant);
  return LSFA_OK;
}

extern "C" int lsfa_warp_bilinear(const float* feat, int feat_n, const float* flow, int N, int C, int H,
                                  int W, const float* mul, const float* add, const float* res, int res_c,
                                  const float* res_w, const float* res_b, float* out, void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(feat && flow && out, "lsfa_warp_bilinear: feat, flow and out must be non-NULL");
  LSFA_REQUIRE(N > 0 && C > 0 && H > 1 && W > 1, "lsfa_warp_bilinear: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
  LSFA_REQUIRE(feat_n >= 1 && N % feat_n == 0, "lsfa_warp_bilinear: feat batch %d must divide N=%d", feat_n, N);
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
  ProfScope prof(LSFA_OP_WARP, s);
  // round 3: planes staged in LDS by DMA (warp_staged_kernel) wherever the shape allows it: whole planes of 1,024 .. 4,096 even
  // pixels, 16-byte aligned maps whose images end on a 16-byte boundary, channel runs that divide C.  Same bits as warp_kernel.
  const int variant = g_variant.load();
  if (variant != 1 && HW % 2 == 0 && (HW >= 1024 || variant == 2) && ((size_t)C * HW) % 4 == 0 && aligned(feat, 16) && aligned(mul, 16) &&
      aligned(add, 16) && aligned(flow, 8) && aligned(res, 8) && aligned(out, 8)) {
    StagedArgs a = {feat, feat_n, flow, N, C, H, W, mul, add, res, res_c, res_w, res_b, out, 1, 0};
    const long planes = (long)N * C;
    // many planes: 4-wave workgroups of 8 channels, two or three to a CU (5.1-5.35 TB/s at 32 maps); few (one map = 1,024 planes):
    // 10-wave workgroups of 4 channels, one per CU and a short prologue (8.4 us against 9.1)
    bool done = false;
    auto run_len = [&](int want) { int g = want; while (g > 1 && C % g) g >>= 1; return g; };
    if (planes >= 4096) { a.cg = run_len(8); done = launch_staged<256, 5, 3>(s, a); }
    if (!done) { a.cg = run_len(planes >= 4096 ? 8 : 4); done = launch_staged<640, 2, 1>(s, a); }
    if (!done) { a.cg = run_len(planes >= 4096 ? 8 : 4); done = launch_staged<512, 4, 2>(s, a); }
    if (done) {
      LSFA_LAUNCH_CHECK("lsfa_warp_bilinear");
      return LSFA_OK;
    }
  }
  if (variant == 2) { set_error("lsfa_warp_bilinear: the staged kernel does not take this shape / alignment"); return LSFA_ENOTSUP; }
  int vec = (HW % 4 == 0) ? 4 : (HW % 2 == 0) ? 2 : 1;
  const size_t al = sizeof(float) * vec;
  if (!(aligned(flow, al) && aligned(mul, al) && aligned(add, al) && aligned(res, al) && aligned(out, al))) vec = 1;
  const int gx = ceil_div(HW, kThreads * vec);
  // enough workgroups to cover 256 CUs several times over, while amortising the tap
  // computation over the channel run
  int cpb = 8;
  while (cpb > 1 && (long)gx * ceil_div(C, cpb) * N < 1024) cpb >>= 1;
  dim3 grid(gx, ceil_div(C, cpb), N);
  if (vec == 4) launch<4>(grid, s, mul != nullptr, add != nullptr, res != nullptr, feat, feat_n, flow, C, H, W, mul, add, res, res_c, res_w, res_b, out, cpb);
  else if (vec == 2) launch<2>(grid, s, mul != nullptr, add != nullptr, res != nullptr, feat, feat_n, flow, C, H, W, mul, add, res, res_c, res_w, res_b, out, cpb);
  else launch<1>(grid, s, mul != nullptr, add != nullptr, res != nullptr, feat, feat_n, flow, C, H, W, mul, add, res, res_c, res_w, res_b, out, cpb);
  LSFA_LAUNCH_CHECK("lsfa_warp_bilinear");
  return LSFA_OK;
}
