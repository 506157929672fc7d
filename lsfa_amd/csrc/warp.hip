// Bilinear feature warp (GridGenerator 'warp' + BilinearSampler) with the fused
// key-path (x scale_map) and cur-path (+ rnet_conv0(res_diff) + small-net feature)
// epilogues.  See include/lsfa_hip.h for the reference interfaces it replaces.
//
// Mapping on CDNA4: a thread owns VEC horizontally adjacent output pixels and a run
// of channels.  The per-pixel tap offsets, validity and weights are computed once
// and reused across the channel run (the reference materialises a (1,2,H,W) grid and
// recomputes the taps per channel).  Lanes of a wave cover 64*VEC consecutive pixels
// of one channel plane, so flow / mul / add / out move as 4*VEC-byte coalesced vectors
// and the taps are near-contiguous gathers served by L1/L2: in the interior fast path
// the left/right taps of a row are ONE 4-byte-aligned 8-byte load.  Planes of an NCHW
// tensor start at multiples of H*W floats, hence VEC = 4, 2 or 1 by H*W mod 4.
//
// Arithmetic is the oracle's, operation for operation (orc_warp_bilinear): built with
// -ffp-contract=off so nothing fuses.
#include "warp_kernels.h"

// second pixel of a lane takes its taps from the first pixel's load / the next lane's (warp_kernels.h);
// decided by measurement, see DESIGN.md "Kernels: warp"
#ifndef LSFA_WARP_SHARE
#define LSFA_WARP_SHARE false
#endif

namespace {
inline bool aligned(const void* p, size_t a) { return p == nullptr || ((uintptr_t)p % a) == 0; }
}  // namespace

extern "C" int lsfa_warp_bilinear(const float* feat, int feat_n, const float* flow, int N, int C, int H,
                                  int W, const float* mul, const float* add, const float* res, int res_c,
                                  const float* res_w, const float* res_b, float* out, void* stream) {
  using namespace lsfa;
  using warp::kResMax;
  LSFA_REQUIRE(feat && flow && out, "lsfa_warp_bilinear: feat, flow and out must be non-NULL");
  LSFA_REQUIRE(N > 0 && C > 0 && H > 1 && W > 1, "lsfa_warp_bilinear: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
  LSFA_REQUIRE(feat_n == 1 || feat_n == N, "lsfa_warp_bilinear: feat batch %d must be 1 or N=%d", feat_n, N);
  LSFA_REQUIRE((long)N * C * ((H * W + 63) / 64) < (1L << 31), "lsfa_warp_bilinear: N=%d C=%d H=%d W=%d has too many work items", N, C, H, W);
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
  const warp::Args a = {feat, feat_n, flow, N, C, H, W, mul, add, res, res_c, res_w, res_b, out};
  // channel run per wave: the largest of 8 / 4 / 2 / 1 that divides C (the kernel has no per-channel guards);
  // small problems halve it so that the chip still gets a few waves per SIMD
  int cpr = (C % 8 == 0) ? 8 : (C % 4 == 0) ? 4 : 1;
  if (cpr == 8 && (long)N * (C / 8) * warp::pixel_tiles(HW, vec) < 2048) cpr = 4;
  if (cpr == 4 && (long)N * (C / 4) * warp::pixel_tiles(HW, vec) < 1024) cpr = 1;
  ProfScope prof(LSFA_OP_WARP, s);
#define LSFA_WARP_RUN(V, SH)                                                   \
  switch (cpr) {                                                               \
    case 8: warp::launch<V, 8, SH>(s, a); break;                               \
    case 4: warp::launch<V, 4, SH>(s, a); break;                               \
    default: warp::launch<V, 1, SH>(s, a); break;                              \
  }
  if (vec == 4) { LSFA_WARP_RUN(4, false) }
  else if (vec == 2) { LSFA_WARP_RUN(2, LSFA_WARP_SHARE) }
  else { LSFA_WARP_RUN(1, false) }
#undef LSFA_WARP_RUN
  LSFA_LAUNCH_CHECK("lsfa_warp_bilinear");
  return LSFA_OK;
}
