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

// ---- r6: the non-key path's warp on CHANNELS-LAST maps --------------------------------------------------------------------------------------
// On a non-key frame the warped feature is read by two 1x1 convolutions only (the RPN head on channels [0, 512), the R-FCN score maps on
// [512, 1024): resnet_v1_101_flownet_rfcn.py:479-499) - GEMMs over the channel axis, i.e. consumers of channels-last rows.  With the NCHW
// operator layout every pass carried a transposing copy of half the map in front of the R-FCN convolution (lsfa_nchw_to_nhwc: 32 us per
// nine-frame segment, 2.2 % of frames/s, profiles/r6/tail_ablation.txt).  Here the key feature is turned channels-last ONCE per pass (one map
// instead of one per frame), the small net's fuse convolution writes its natural layout, and the warp reads and writes (pixel, channel) rows:
// thread t owns the channel quad 4t .. 4t+3 (C = 1024: one quad per thread of a 256-thread workgroup), a workgroup walks a run of pixels,
// every access is a 4 KB row (256 threads x float4).  The arithmetic is lsfa_warp_bilinear's general path, operation for operation (taps
// outside the map contribute 0 * w; then + rnet_conv0(res_diff) + add): the two layouts give the same bits (tests/test_hip_ops.py).
// amax_out (or NULL): 256 zeroed slots that receive max|out| - the next convolution's scale, as the convolutions' own epilogues leave it.
namespace {
__device__ __attribute__((aligned(16))) float4 g_warp_zero4 = {0.f, 0.f, 0.f, 0.f};      // (not const: a constant-address-space pointer in the select below turns the loads into flat ones)

template <bool HAS_ADD, bool HAS_RES>
__global__ __launch_bounds__(256) void warp_cl_kernel(const float* __restrict__ feat, int feat_n, const float* __restrict__ flow, int N, int C,
                                                      int H, int W, const float* __restrict__ add, const float* __restrict__ res, int res_c,
                                                      const float* __restrict__ res_w, const float* __restrict__ res_b,
                                                      float* __restrict__ out, unsigned* __restrict__ amax_out, int amax_c0, int pix_per_wg) {
  using namespace lsfa::warp;
  const int HW = H * W, C4 = C >> 2;
  const int P = N * HW;                                   // (< 2^31: checked by the host)
  const int p_begin = blockIdx.x * pix_per_wg;
  const int p_end = min(p_begin + pix_per_wg, P);
  const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
  float mx = 0.f;
  for (int q = threadIdx.x; q < C4; q += 256) {
    float rw[4][kResMax], rb[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      rb[j] = HAS_RES ? res_b[4 * q + j] : 0.f;
#pragma unroll
      for (int k = 0; k < kResMax; ++k) rw[j][k] = (HAS_RES && k < res_c) ? res_w[(size_t)(4 * q + j) * res_c + k] : 0.f;
    }
    const bool counts = 4 * q >= amax_c0;            // the maximum is taken over channels [amax_c0, C): the ones the scaled consumer reads
    int n = p_begin / HW;
    int r = p_begin - n * HW;
    int y = r / W, x = r - y * W;
    for (int p = p_begin; p < p_end; ++p) {
      const float fx = flow[((size_t)n * 2 + 0) * HW + r], fy = flow[((size_t)n * 2 + 1) * HW + r];
      const float gx = ((float)x + fx) / half_w - 1.0f;
      const float gy = ((float)y + fy) / half_h - 1.0f;
      const float x_real = (gx + 1.0f) * (float)(W - 1) / 2.0f;
      const float y_real = (gy + 1.0f) * (float)(H - 1) / 2.0f;
      const float fx0 = floorf(x_real), fy0 = floorf(y_real);
      const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)W);       // (clamped before the conversion: outside the map either way)
      const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)H);
      const float wx0 = 1.0f - (x_real - fx0), wy0 = 1.0f - (y_real - fy0);
      const float wx1 = 1.0f - wx0, wy1 = 1.0f - wy0;
      const bool vx0 = (x0 >= 0 && x0 <= W - 1), vx1 = (x0 + 1 >= 0 && x0 + 1 <= W - 1);
      const bool vy0 = (y0 >= 0 && y0 <= H - 1), vy1 = (y0 + 1 >= 0 && y0 + 1 <= H - 1);
      const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const float4* fb = reinterpret_cast<const float4*>(feat + (size_t)(n % feat_n) * HW * C) + q;
      const long off = (long)y0 * W + x0;
      // (every thread of the workgroup works on the same pixel: the four branches are uniform)
      // a tap outside the map reads a block of zeros (a select between two GLOBAL addresses: `cond ? load : 0` made hipcc spill a zero to
      // scratch and load through a flat pointer)
      const float4 tl = *((vx0 && vy0) ? fb + (size_t)off * C4 : &g_warp_zero4);
      const float4 tr = *((vx1 && vy0) ? fb + (size_t)(off + 1) * C4 : &g_warp_zero4);
      const float4 bl = *((vx0 && vy1) ? fb + (size_t)(off + W) * C4 : &g_warp_zero4);
      const float4 br = *((vx1 && vy1) ? fb + (size_t)(off + W + 1) * C4 : &g_warp_zero4);
      float4 a4 = zero4;
      if (HAS_ADD) a4 = reinterpret_cast<const float4*>(add)[(size_t)p * C4 + q];
      // (named scalars, not arrays: indexed arrays ended up in scratch memory here)
      const float* rp = res + (size_t)n * res_c * HW + r;
      const float rv0 = HAS_RES ? rp[0] : 0.f, rv1 = (HAS_RES && res_c > 1) ? rp[HW] : 0.f;
      const float rv2 = (HAS_RES && res_c > 2) ? rp[2 * (size_t)HW] : 0.f, rv3 = (HAS_RES && res_c > 3) ? rp[3 * (size_t)HW] : 0.f;
      // one component: the oracle's expression, then the residual's 1x1 convolution and the small net's feature
#define LSFA_WARP_CL_ONE(J_, TL_, TR_, BL_, BR_, A_, O_)                                                                     \
      float O_ = TL_ * wy0 * wx0 + TR_ * wy0 * wx1 + BL_ * wy1 * wx0 + BR_ * wy1 * wx1;                                      \
      if (HAS_RES) {                                                                                                         \
        float qv = rw[J_][0] * rv0;                                                                                          \
        if (res_c > 1) qv = qv + rw[J_][1] * rv1;                                                                            \
        if (res_c > 2) qv = qv + rw[J_][2] * rv2;                                                                            \
        if (res_c > 3) qv = qv + rw[J_][3] * rv3;                                                                            \
        qv = qv + rb[J_];                                                                                                    \
        O_ = O_ + qv;                                                                                                        \
      }                                                                                                                      \
      if (HAS_ADD) O_ = O_ + A_;                                                                                             \
      if (counts) mx = fmaxf(mx, fabsf(O_));
      LSFA_WARP_CL_ONE(0, tl.x, tr.x, bl.x, br.x, a4.x, o0)
      LSFA_WARP_CL_ONE(1, tl.y, tr.y, bl.y, br.y, a4.y, o1)
      LSFA_WARP_CL_ONE(2, tl.z, tr.z, bl.z, br.z, a4.z, o2)
      LSFA_WARP_CL_ONE(3, tl.w, tr.w, bl.w, br.w, a4.w, o3)
#undef LSFA_WARP_CL_ONE
      reinterpret_cast<float4*>(out)[(size_t)p * C4 + q] = make_float4(o0, o1, o2, o3);
      if (++x == W) { x = 0; if (++y == H) { y = 0; ++n; } }
      if (++r == HW) r = 0;
    }
  }
  if (amax_out) {
    uint32_t m = __float_as_uint(mx);       // non-negative, or NaN bits (fmaxf drops a NaN: a non-finite output shows in the consumer's own status)
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(amax_out + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255), m);
  }
}
}  // namespace

extern "C" int lsfa_warp_bilinear_cl(const float* feat_cl, int feat_n, const float* flow, int N, int C, int H, int W, const float* add_cl,
                                     const float* res, int res_c, const float* res_w, const float* res_b, float* out_cl, unsigned* amax_out,
                                     int amax_c0, void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(feat_cl && flow && out_cl, "lsfa_warp_bilinear_cl: feat, flow and out must be non-NULL");
  LSFA_REQUIRE(N > 0 && C > 0 && C % 4 == 0 && H > 1 && W > 1, "lsfa_warp_bilinear_cl: bad shape N=%d C=%d (a multiple of 4) H=%d W=%d", N, C, H, W);
  LSFA_REQUIRE(feat_n >= 1 && N % feat_n == 0, "lsfa_warp_bilinear_cl: feat batch %d must divide N=%d", feat_n, N);
  LSFA_REQUIRE(amax_c0 >= 0 && amax_c0 < C && amax_c0 % 4 == 0, "lsfa_warp_bilinear_cl: amax_c0=%d must be a multiple of 4 in [0, C)", amax_c0);
  LSFA_REQUIRE(aligned(feat_cl, 16) && aligned(add_cl, 16) && aligned(out_cl, 16), "lsfa_warp_bilinear_cl: maps must be 16-byte aligned");
  if (res) {
    LSFA_REQUIRE(res_w && res_b, "lsfa_warp_bilinear_cl: res given without res_w/res_b");
    if (res_c < 1 || res_c > warp::kResMax) {
      set_error("lsfa_warp_bilinear_cl: res_c=%d not in [1,%d]", res_c, warp::kResMax);
      return LSFA_ENOTSUP;
    }
  }
  hipStream_t s = (hipStream_t)stream;
  const long P = (long)N * H * W;
  LSFA_REQUIRE(P * C < (1L << 31), "lsfa_warp_bilinear_cl: map of 2^31 elements or more");
  long per = P / 1536;                     // ~6 workgroups per CU; a run of pixels amortises the quad's residual weights
  if (per < 1) per = 1;
  if (per > 16) per = 16;
  const dim3 grid((unsigned)((P + per - 1) / per));
  ProfScope prof(LSFA_OP_WARP, s);
#define LSFA_WARP_CL(A_, R_) hipLaunchKernelGGL((warp_cl_kernel<A_, R_>), grid, dim3(256), 0, s, feat_cl, feat_n, flow, N, C, H, W, add_cl, res, res_c, \
                                                res_w, res_b, out_cl, amax_out, amax_c0, (int)per)
  if (add_cl && res) LSFA_WARP_CL(true, true);
  else if (add_cl) LSFA_WARP_CL(true, false);
  else if (res) LSFA_WARP_CL(false, true);
  else LSFA_WARP_CL(false, false);
#undef LSFA_WARP_CL
  LSFA_LAUNCH_CHECK("lsfa_warp_bilinear_cl");
  return LSFA_OK;
}

