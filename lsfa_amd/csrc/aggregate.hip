// Long-term aggregation combines: Nq_net tail (2-way softmax + weighted sum) and the
// Fgfa cosine-similarity variant.  Reference interfaces: include/lsfa_hip.h.
//
// Mapping: like the warp kernel, a thread owns VEC adjacent pixels and a run of
// channels; the per-pixel softmax weights (two correctly rounded exps on the fp64
// unit) are computed once per thread and reused over the run.  The cosine variant
// first reduces the two E-channel embeddings per pixel: lanes walk pixels (coalesced
// along H*W), each lane accumulating its own pixel over E sequentially, which is the
// oracle's summation order, so results are bit-identical and no cross-lane reduction
// is needed at all.
#include "common.h"

namespace {

constexpr int kThreads = 256;

template <int VEC> struct VecT;
template <> struct VecT<1> { using type = float; };
template <> struct VecT<2> { using type = float2; };
template <> struct VecT<4> { using type = float4; };

template <int VEC>
__device__ __forceinline__ void load_vec(const float* p, float (&v)[VEC]) {
  using T = typename VecT<VEC>::type;
  T t = *reinterpret_cast<const T*>(p);
  const float* f = reinterpret_cast<const float*>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) v[i] = f[i];
}
template <int VEC>
__device__ __forceinline__ void store_vec(float* p, const float (&v)[VEC]) {
  using T = typename VecT<VEC>::type;
  T t;
  float* f = reinterpret_cast<float*>(&t);
#pragma unroll
  for (int i = 0; i < VEC; ++i) f[i] = v[i];
  *reinterpret_cast<T*>(p) = t;
}

__device__ __forceinline__ void softmax2(float l0, float l1, float& w0, float& w1) {
  const float m = fmaxf(l0, l1);
  const float e0 = expf_cr(l0 - m), e1 = expf_cr(l1 - m);
  const float s = e0 + e1;
  w0 = e0 / s;
  w1 = e1 / s;
}

// out = w0*a + w1*b with (w0,w1) = softmax of the (2, HW) logits; shared by both variants.
// CPB (channels per thread, compile time): all 2*CPB operand loads are issued BEFORE the softmax —
// the two correctly rounded exps are a ~1 us dependent chain on the fp64 unit, and with the loads
// behind it every wave paid logits latency + exp chain + operand latency in series.
template <int VEC, int CPB>
__global__ __launch_bounds__(kThreads) void combine_kernel(const float* __restrict__ a,
                                                           const float* __restrict__ b,
                                                           const float* __restrict__ logits, int C, int HW,
                                                           float* __restrict__ out) {
  const int c0 = blockIdx.y * CPB;
  const int p0 = (blockIdx.x * kThreads + threadIdx.x) * VEC;
  if (p0 >= HW) return;
  float l0[VEC], l1[VEC], w0[VEC], w1[VEC];
  load_vec<VEC>(logits + p0, l0);
  load_vec<VEC>(logits + HW + p0, l1);
  float va[CPB][VEC], vb[CPB][VEC];
#pragma unroll
  for (int k = 0; k < CPB; ++k)
    if (c0 + k < C) {
      const size_t o = (size_t)(c0 + k) * HW + p0;
      load_vec<VEC>(a + o, va[k]);
      load_vec<VEC>(b + o, vb[k]);
    }
#pragma unroll
  for (int i = 0; i < VEC; ++i) softmax2(l0[i], l1[i], w0[i], w1[i]);
#pragma unroll
  for (int k = 0; k < CPB; ++k)
    if (c0 + k < C) {
      float v[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) v[i] = w0[i] * va[k][i] + w1[i] * vb[k][i];
      store_vec<VEC>(out + (size_t)(c0 + k) * HW + p0, v);
    }
}

// cosine logits -> (2, HW) logits buffer laid out like Nq's, so the combine is shared.
__global__ __launch_bounds__(kThreads) void cosine_logits_kernel(const float* __restrict__ emb_warp,
                                                                 const float* __restrict__ emb_cur, int E,
                                                                 int HW, float* __restrict__ logits) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= HW) return;
  float sw = 0.f, sc = 0.f;
  for (int e = 0; e < E; ++e) { const float v = emb_warp[(size_t)e * HW + p]; sw += v * v; }
  for (int e = 0; e < E; ++e) { const float v = emb_cur[(size_t)e * HW + p]; sc += v * v; }
  const float nw = sqrtf(sw + 1e-10f), nc = sqrtf(sc + 1e-10f);
  float l0 = 0.f, l1 = 0.f;
  for (int e = 0; e < E; ++e) {
    const float vw = emb_warp[(size_t)e * HW + p] / nw, vc = emb_cur[(size_t)e * HW + p] / nc;
    l0 += vw * vc;
    l1 += vc * vc;
  }
  logits[p] = l0;
  logits[HW + p] = l1;
}

// last two channels of the cosine variant (their planes held the logits scratch)
__global__ __launch_bounds__(kThreads) void cosine_tail_kernel(const float* __restrict__ a,
                                                               const float* __restrict__ b, int C, int HW,
                                                               float* out) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= HW) return;
  const size_t o0 = (size_t)(C - 2) * HW + p, o1 = (size_t)(C - 1) * HW + p;
  float w0, w1;
  softmax2(out[o0], out[o1], w0, w1);
  const float r0 = w0 * a[o0] + w1 * b[o0];
  const float r1 = w0 * a[o1] + w1 * b[o1];
  out[o0] = r0;
  out[o1] = r1;
}

inline bool aligned(const void* p, size_t a) { return ((uintptr_t)p % a) == 0; }

int launch_combine(const float* a, const float* b, const float* logits, int C, int HW, float* out, hipStream_t s) {
  using namespace lsfa;
  int vec = (HW % 4 == 0) ? 4 : (HW % 2 == 0) ? 2 : 1;
  const size_t al = sizeof(float) * vec;
  if (!(aligned(a, al) && aligned(b, al) && aligned(logits, al) && aligned(out, al))) vec = 1;
  const int gx = ceil_div(HW, kThreads * vec);
  // 8 channels per thread amortise the softmax; fall back to 4 when that leaves too few workgroups
  const bool c8 = (long)gx * ceil_div(C, 8) >= 1024;
  const int cpb = c8 ? 8 : 4;
  dim3 grid(gx, ceil_div(C, cpb));
#define LSFA_COMBINE(V)                                                                                          \
  if (c8) hipLaunchKernelGGL((combine_kernel<V, 8>), grid, dim3(kThreads), 0, s, a, b, logits, C, HW, out);      \
  else hipLaunchKernelGGL((combine_kernel<V, 4>), grid, dim3(kThreads), 0, s, a, b, logits, C, HW, out);
  if (vec == 4) { LSFA_COMBINE(4) }
  else if (vec == 2) { LSFA_COMBINE(2) }
  else { LSFA_COMBINE(1) }
#undef LSFA_COMBINE
  return 0;
}

}  // namespace

extern "C" int lsfa_aggregate_softmax2(const float* a, const float* b, const float* logits, int C, int H,
                                       int W, float* out, void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(a && b && logits && out, "lsfa_aggregate_softmax2: NULL argument");
  LSFA_REQUIRE(C > 0 && H > 0 && W > 0, "lsfa_aggregate_softmax2: bad shape C=%d H=%d W=%d", C, H, W);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(LSFA_OP_AGG, s);
  launch_combine(a, b, logits, C, H * W, out, s);
  LSFA_LAUNCH_CHECK("lsfa_aggregate_softmax2");
  return LSFA_OK;
}

// The cosine variant needs a (2,H,W) scratch for the logits and the ABI has no workspace
// argument, so the last two channel planes of `out` serve as the scratch: pass 1 writes
// the logits there, pass 2 combines channels [0, C-2) reading them, pass 3 finishes the
// last two channels with each thread reading its own pixel's two logits before it
// overwrites them.  `out` must not alias a or b.
extern "C" int lsfa_aggregate_cosine(const float* a, const float* b, const float* emb_warp,
                                     const float* emb_cur, int C, int E, int H, int W, float* out,
                                     void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(a && b && emb_warp && emb_cur && out, "lsfa_aggregate_cosine: NULL argument");
  LSFA_REQUIRE(C > 2 && E > 0 && H > 0 && W > 0, "lsfa_aggregate_cosine: bad shape C=%d E=%d H=%d W=%d", C, E, H, W);
  LSFA_REQUIRE(out != a && out != b, "lsfa_aggregate_cosine: out must not alias a or b");
  hipStream_t s = (hipStream_t)stream;
  const int HW = H * W;
  ProfScope prof(LSFA_OP_AGG, s);
  float* scratch = out + (size_t)(C - 2) * HW;
  hipLaunchKernelGGL(cosine_logits_kernel, dim3(ceil_div(HW, kThreads)), dim3(kThreads), 0, s, emb_warp, emb_cur, E, HW, scratch);
  launch_combine(a, b, scratch, C - 2, HW, out, s);
  hipLaunchKernelGGL(cosine_tail_kernel, dim3(ceil_div(HW, kThreads)), dim3(kThreads), 0, s, a, b, C, HW, out);
  LSFA_LAUNCH_CHECK("lsfa_aggregate_cosine");
  return LSFA_OK;
}
