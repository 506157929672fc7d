// Long-term aggregation combines: Nq_net tail (2-way softmax + weighted sum) and the
// Fgfa cosine-similarity variant.  Reference interfaces: include/lsfa_hip.h.
//
// Mapping: like the warp kernel, a thread owns VEC adjacent pixels and a run of
// channels; the per-pixel softmax weights (two correctly rounded exps on the fp64
// unit) are computed once per thread and reused over the run.  The cosine variant
// first reduces the two E-channel embeddings per pixel with the E axis across the
// lanes of a wave and a fixed __shfl_xor butterfly (cosine_logits_kernel below).
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
// Batched form (gridDim.z = N maps per launch, BASELINE configs[4]): a, b, out are (N, C, HW); the logits are
// (2, N, HW) — what the Nq convolutions produce for Concat(warp x N, cur x N) on the batch axis.
template <int VEC, int CPB>
__global__ __launch_bounds__(kThreads) void combine_kernel(const float* __restrict__ a,
                                                           const float* __restrict__ b,
                                                           const float* __restrict__ logits, long lrow, int C, int HW,
                                                           float* __restrict__ out) {
  const int c0 = blockIdx.y * CPB;
  const int p0 = (blockIdx.x * kThreads + threadIdx.x) * VEC;
  if (p0 >= HW) return;
  const int n = blockIdx.z, N = gridDim.z;
  a += (size_t)n * C * HW; b += (size_t)n * C * HW; out += (size_t)n * C * HW;
  float l0[VEC], l1[VEC], w0[VEC], w1[VEC];
  load_vec<VEC>(logits + (size_t)n * lrow + p0, l0);            // lrow: floats between logit rows (HW when they are dense)
  load_vec<VEC>(logits + (size_t)(N + n) * lrow + p0, l1);
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
//
// Per pixel: L2 norms of the two E-channel embeddings, then <emb_warp/|.|, emb_cur/|.|> and
// <emb_cur/|.|, emb_cur/|.|> — four reductions over E (2048 for LSFA).  The sum order is the fixed tree
// of orc_aggregate_cosine: 64 partial sums (partial j = channels j, j+64, ... in order), then a
// butterfly over j.  Mapping: a workgroup owns kCosPx consecutive pixels; each 64-channel chunk is read
// from HBM/L2 coalesced along the pixels, staged in LDS, and re-read transposed so that LANE j of a
// wave holds channel class j of the wave's pixels: the chunk loop is the in-order accumulation of
// partial j, and the butterfly is six __shfl_xor steps — the cross-lane reduction north_star asks
// for — with bit-identical results on both sides.
constexpr int kCosPx = 16;                    // pixels per workgroup (64-byte segments per channel row; 150 workgroups at 38x63)
constexpr int kCosWaves = kThreads / 64;      // 4
constexpr int kCosPxPerWave = kCosPx / kCosWaves;

__device__ __forceinline__ float wave_tree_total(float v) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) v = v + __shfl_xor(v, d, 64);
  return v;
}

__global__ __launch_bounds__(kThreads) void cosine_logits_kernel(const float* __restrict__ emb_warp,
                                                                 const float* __restrict__ emb_cur, int E,
                                                                 int HW, float* __restrict__ logits) {
  __shared__ float tw[64][kCosPx + 1], tc[64][kCosPx + 1];
  constexpr int kPerThread = 64 * kCosPx / kThreads;      // staged values per thread, tensor and chunk (8)
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int p0 = blockIdx.x * kCosPx;
  const int chunks = (E + 63) / 64;
  // staging role: value q of this thread is (row r = (q*256 + tid) / 32, pixel x = (q*256 + tid) % 32) of a chunk
  int srow[kPerThread], scol[kPerThread];
  size_t soff[kPerThread];
#pragma unroll
  for (int q = 0; q < kPerThread; ++q) {
    const int idx = q * kThreads + tid;
    srow[q] = idx / kCosPx;
    scol[q] = idx - srow[q] * kCosPx;
    soff[q] = (size_t)srow[q] * HW + min(p0 + scol[q], HW - 1);
  }
  float nw[kCosPxPerWave], nc[kCosPxPerWave];
  float l0[kCosPxPerWave], l1[kCosPxPerWave];
#pragma unroll 1
  for (int pass = 0; pass < 2; ++pass) {
    float acc_a[kCosPxPerWave], acc_b[kCosPxPerWave];
#pragma unroll
    for (int i = 0; i < kCosPxPerWave; ++i) { acc_a[i] = 0.f; acc_b[i] = 0.f; }
    // chunk ch+1 is in flight (registers) while chunk ch is reduced out of LDS: with one 4-wave workgroup per
    // CU there is nothing else to hide the load latency behind
    float vw[kPerThread], vc[kPerThread];
#pragma unroll
    for (int q = 0; q < kPerThread; ++q) {
      const bool ok = srow[q] < E;
      vw[q] = ok ? emb_warp[soff[q]] : 0.f;
      vc[q] = ok ? emb_cur[soff[q]] : 0.f;
    }
    for (int ch = 0; ch < chunks; ++ch) {
#pragma unroll
      for (int q = 0; q < kPerThread; ++q) { tw[srow[q]][scol[q]] = vw[q]; tc[srow[q]][scol[q]] = vc[q]; }
      __syncthreads();
      if (ch + 1 < chunks) {
        const size_t base = (size_t)(ch + 1) * 64 * HW;
#pragma unroll
        for (int q = 0; q < kPerThread; ++q) {
          const bool ok = (ch + 1) * 64 + srow[q] < E;
          vw[q] = ok ? emb_warp[base + soff[q]] : 0.f;
          vc[q] = ok ? emb_cur[base + soff[q]] : 0.f;
        }
      }
#pragma unroll
      for (int i = 0; i < kCosPxPerWave; ++i) {
        const int x = wid * kCosPxPerWave + i;
        const float a = tw[lane][x], c = tc[lane][x];
        if (pass == 0) {
          acc_a[i] += a * a;
          acc_b[i] += c * c;
        } else {
          const float uw = a / nw[i], uc = c / nc[i];
          acc_a[i] += uw * uc;
          acc_b[i] += uc * uc;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < kCosPxPerWave; ++i) {
      const float ta = wave_tree_total(acc_a[i]), tb = wave_tree_total(acc_b[i]);
      if (pass == 0) { nw[i] = sqrtf(ta + 1e-10f); nc[i] = sqrtf(tb + 1e-10f); }
      else { l0[i] = ta; l1[i] = tb; }
    }
  }
  if (lane < kCosPxPerWave) {
    float v0 = 0.f, v1 = 0.f;
#pragma unroll
    for (int i = 0; i < kCosPxPerWave; ++i) if (lane == i) { v0 = l0[i]; v1 = l1[i]; }
    const int p = p0 + wid * kCosPxPerWave + lane;
    if (p < HW) { logits[p] = v0; logits[HW + p] = v1; }
  }
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

int launch_combine(const float* a, const float* b, const float* logits, long lrow, int N, int C, int HW, float* out, hipStream_t s) {
  using namespace lsfa;
  int vec = (HW % 4 == 0) ? 4 : (HW % 2 == 0) ? 2 : 1;
  const size_t al = sizeof(float) * vec;
  if (!(aligned(a, al) && aligned(b, al) && aligned(logits, al) && aligned(out, al)) || lrow % vec != 0) vec = 1;
  const int gx = ceil_div(HW, kThreads * vec);
  // 8 channels per thread amortise the softmax; fall back to 4 when that leaves too few workgroups
  const bool c8 = (long)gx * ceil_div(C, 8) * N >= 1024;
  const int cpb = c8 ? 8 : 4;
  dim3 grid(gx, ceil_div(C, cpb), N);
#define LSFA_COMBINE(V)                                                                                          \
  if (c8) hipLaunchKernelGGL((combine_kernel<V, 8>), grid, dim3(kThreads), 0, s, a, b, logits, lrow, C, HW, out);      \
  else hipLaunchKernelGGL((combine_kernel<V, 4>), grid, dim3(kThreads), 0, s, a, b, logits, lrow, C, HW, out);
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
  launch_combine(a, b, logits, (long)H * W, 1, C, H * W, out, s);
  LSFA_LAUNCH_CHECK("lsfa_aggregate_softmax2");
  return LSFA_OK;
}

extern "C" int lsfa_aggregate_softmax2_batched(const float* a, const float* b, const float* logits, int N, int C, int H,
                                               int W, float* out, void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(a && b && logits && out, "lsfa_aggregate_softmax2_batched: NULL argument");
  LSFA_REQUIRE(N > 0 && N <= 65535 && C > 0 && H > 0 && W > 0, "lsfa_aggregate_softmax2_batched: bad shape N=%d C=%d H=%d W=%d", N, C, H, W);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(LSFA_OP_AGG, s);
  launch_combine(a, b, logits, (long)H * W, N, C, H * W, out, s);
  LSFA_LAUNCH_CHECK("lsfa_aggregate_softmax2_batched");
  return LSFA_OK;
}

extern "C" int lsfa_aggregate_softmax2_rows(const float* a, const float* b, const float* logits, long logit_row_stride, int N, int C,
                                            int H, int W, float* out, void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(a && b && logits && out, "lsfa_aggregate_softmax2_rows: NULL argument");
  LSFA_REQUIRE(N > 0 && N <= 65535 && C > 0 && H > 0 && W > 0 && logit_row_stride >= (long)H * W,
               "lsfa_aggregate_softmax2_rows: bad shape N=%d C=%d H=%d W=%d row stride %ld", N, C, H, W, logit_row_stride);
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(LSFA_OP_AGG, s);
  launch_combine(a, b, logits, logit_row_stride, N, C, H * W, out, s);
  LSFA_LAUNCH_CHECK("lsfa_aggregate_softmax2_rows");
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
  hipLaunchKernelGGL(cosine_logits_kernel, dim3(ceil_div(HW, kCosPx)), dim3(kThreads), 0, s, emb_warp, emb_cur, E, HW, scratch);
  launch_combine(a, b, scratch, (long)HW, 1, C - 2, HW, out, s);
  hipLaunchKernelGGL(cosine_tail_kernel, dim3(ceil_div(HW, kThreads)), dim3(kThreads), 0, s, a, b, C, HW, out);
  LSFA_LAUNCH_CHECK("lsfa_aggregate_cosine");
  return LSFA_OK;
}
