// Lab copy of conv.hip's conv_igemm_kernel<64> with parts switched off, to see what bounds it
// (tools/lab/conv_lab.py).  MODE 0 = the product loop; 1 = no global loads inside the loop (the chunk staged
// before the loop is re-used); 2 = also no LDS writes / barriers inside the loop; 3 = MFMAs only (fragments read
// from LDS once).  Modes > 0 compute garbage on purpose: only their time matters.
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace {
constexpr int kBM = 64, kBN = 64, kThreads = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));
struct ConvArgs {
  const float* x; const float* w; float* part;
  int N, H, W, Cin, Cout, kh, kw, stride, pad, dil, Ho, Wo, taps_per_slice;
};

template <int BK, int MODE>
__global__ __launch_bounds__(kThreads) void conv_lab_kernel(ConvArgs a) {
  constexpr int kLdk = BK + 4;
  constexpr int NV = BK / 16;
  __shared__ __attribute__((aligned(16))) float As[2][kBM * kLdk];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBN * kLdk];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int P = a.N * a.Ho * a.Wo;
  const int m0 = blockIdx.x * kBM, n0 = blockIdx.y * kBN;
  const int taps = a.kh * a.kw;
  const int tap0 = blockIdx.z * a.taps_per_slice, tap1 = min(tap0 + a.taps_per_slice, taps);
  const int chunks_per_tap = a.Cin / BK;
  const int nchunks = (tap1 - tap0) * chunks_per_tap;
  const int srow = tid >> 2, scol = (tid & 3) * (BK / 4);
  const int pix = m0 + srow;
  const bool pix_ok = pix < P;
  int py = 0, px = 0, pn = 0;
  if (pix_ok) { pn = pix / (a.Ho * a.Wo); const int r = pix - pn * a.Ho * a.Wo; py = r / a.Wo; px = r - py * a.Wo; }
  const float* wrow = a.w + ((size_t)(n0 + srow) * taps) * a.Cin + scol;
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  ra2 = ra3 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
  float a_keep = 0.f;
#define FETCH(chunk_)                                                                                                  \
  {                                                                                                                    \
    const int t_ = (chunk_) / chunks_per_tap;                                                                          \
    const int tap = tap0 + t_;                                                                                         \
    const int ci0 = ((chunk_) - t_ * chunks_per_tap) * BK;                                                             \
    const int ty = tap / a.kw, tx = tap - ty * a.kw;                                                                   \
    const int iy = py * a.stride - a.pad + ty * a.dil, ix = px * a.stride - a.pad + tx * a.dil;                        \
    const bool ok = pix_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;                                              \
    const float4* src = reinterpret_cast<const float4*>(a.x + (((size_t)pn * a.H + (ok ? iy : 0)) * a.W + (ok ? ix : 0)) * a.Cin + ci0 + scol); \
    const float4* wsrc = reinterpret_cast<const float4*>(wrow + (size_t)tap * a.Cin + ci0);                            \
    ra0 = src[0]; ra1 = src[1]; rb0 = wsrc[0]; rb1 = wsrc[1];                                                          \
    if (NV > 2) { ra2 = src[2]; ra3 = src[3]; rb2 = wsrc[2]; rb3 = wsrc[3]; }                                          \
    a_keep = ok ? 1.0f : 0.0f;                                                                                         \
  }
#define STASH(buf_)                                                                                                    \
  {                                                                                                                    \
    float4* da = reinterpret_cast<float4*>(&As[buf_][srow * kLdk + scol]);                                             \
    float4* db = reinterpret_cast<float4*>(&Bs[buf_][srow * kLdk + scol]);                                             \
    da[0] = make_float4(ra0.x * a_keep, ra0.y * a_keep, ra0.z * a_keep, ra0.w * a_keep);                               \
    da[1] = make_float4(ra1.x * a_keep, ra1.y * a_keep, ra1.z * a_keep, ra1.w * a_keep);                               \
    db[0] = rb0; db[1] = rb1;                                                                                          \
    if (NV > 2) {                                                                                                      \
      da[2] = make_float4(ra2.x * a_keep, ra2.y * a_keep, ra2.z * a_keep, ra2.w * a_keep);                             \
      da[3] = make_float4(ra3.x * a_keep, ra3.y * a_keep, ra3.z * a_keep, ra3.w * a_keep);                             \
      db[2] = rb2; db[3] = rb3;                                                                                        \
    }                                                                                                                  \
  }
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  FETCH(0)
  STASH(0)
  if (MODE >= 2) STASH(1)
  __syncthreads();
  const int arow = (wr * 32 + (lane & 31)) * kLdk + 4 * (lane >> 5);
  const int brow = (wc * 32 + (lane & 31)) * kLdk + 4 * (lane >> 5);
  float4 av0 = *reinterpret_cast<const float4*>(&As[0][arow]), bv0 = *reinterpret_cast<const float4*>(&Bs[0][brow]);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    if (MODE == 0 && chunk + 1 < nchunks) FETCH(chunk + 1)
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      float4 av = av0, bv = bv0;
      if (MODE < 3) {
        av = *reinterpret_cast<const float4*>(&As[buf][arow + 8 * c]);
        bv = *reinterpret_cast<const float4*>(&Bs[buf][brow + 8 * c]);
      }
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (MODE < 2 && chunk + 1 < nchunks) {
      STASH(buf ^ 1)
      __syncthreads();
    }
  }
#undef FETCH
#undef STASH
  const int ch = n0 + wc * 32 + (lane & 31);
  float* part = a.part + (size_t)blockIdx.z * P * a.Cout;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int p = m0 + wr * 32 + row;
    if (p < P) part[(size_t)p * a.Cout + ch] = acc[r];
  }
}
}  // namespace

// x (N,H,W,Cin), w (Cout, 9, Cin), part (slices, P, Cout); 3x3, stride 1, pad = dil
extern "C" int conv_lab_run(int mode, const float* x, const float* w, float* part, int N, int H, int W, int Cin, int Cout,
                            int dil, int slices, void* stream) {
  ConvArgs a = {x, w, part, N, H, W, Cin, Cout, 3, 3, 1, dil, dil, H, W, 9 / slices};
  const int P = N * H * W;
  const dim3 grid((P + kBM - 1) / kBM, Cout / kBN, slices);
  hipStream_t s = (hipStream_t)stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL((conv_lab_kernel<64, 0>), grid, dim3(kThreads), 0, s, a); break;
    case 1: hipLaunchKernelGGL((conv_lab_kernel<64, 1>), grid, dim3(kThreads), 0, s, a); break;
    case 2: hipLaunchKernelGGL((conv_lab_kernel<64, 2>), grid, dim3(kThreads), 0, s, a); break;
    case 3: hipLaunchKernelGGL((conv_lab_kernel<64, 3>), grid, dim3(kThreads), 0, s, a); break;
    default: return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 2;
}
