// The parts of FlowNet-S (dff_rfcn/symbols/resnet_v1_101_flownet_rfcn.py:150-207) that are not MFMA-sized contractions,
// on channels-last maps, so that together with lsfa_conv_split_view_fwd the whole network runs without a library call:
//   flow_conv1 (7x7 / 2, 6 -> 64 channels)   the stem kernel of stem.hip run once per image of the pair (3 channels each), the
//                                            second pass adding onto the first and applying bias + LeakyReLU: stem.hip
//   head_conv3x3_kernel    Convolution1..5 (:178, :183, :188, :193, :203): 3x3, pad 1, Cin up to 1026 -> 2 channels.  One
//                          workgroup per output pixel, threads over the input channels (coalesced reads), nine taps, a fixed
//                          reduction tree: deterministic, a few microseconds (the maps are 5x8 ... 38x63).
//   upflow_kernel          upsample_flow6to5 ... 3to2 (:180 ...): Deconvolution(kernel 4, stride 2, 2 -> 2 channels) + Crop(offset 1)
//                          written into the two flow channels of the next concatenated map.
//   avgpool2_cl_kernel     Pooling(2x2 / 2, avg, pooling_convention='full') on a channels-last map (:201, and the frame pair).
// Arithmetic: plain fp32 fmaf chains in a stated order; these are "dense" stages compared by tolerance like every convolution.
#include "common.h"

using namespace lsfa;

namespace {

constexpr int kHeadMaxCout = 4;

// grid (P); block 256 = 4 waves, ONE output pixel per workgroup: the (tap, channel) products of a pixel are dealt to the 256
// threads (thread t takes channels t, t + 256, ... of every tap), so that a 9 x 1026-long dot product is 9 x 4 loads deep per
// thread instead of 9 x 16, all of them independent (unrolled): the first form of this kernel walked taps x channels with
// one wave and sat at 60 us per call on load latency.  x (N, H, W, lda) channels-last, Cin channels used.
// w (Cout, 3, 3, Cin); out: NCHW (N, Cout, H, W) when out_nchw, else channels [c0, c0 + Cout) of an (N, H, W, ldy) map.
// Sum order (fixed): per thread channels ascending, taps in (ky, kx) order inside a channel; then the lanes of a wave pairwise (xor 32 ... 1), then
// the four waves in order.
__global__ __launch_bounds__(256) void head_conv3x3_kernel(const float* __restrict__ x, int lda, int N, int H, int W, int Cin,
                                                           const float* __restrict__ w, const float* __restrict__ bias, int Cout,
                                                           float mul, float* __restrict__ y, int out_nchw, int ldy, int c0) {
  __shared__ float wave_sum[4][kHeadMaxCout];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int p = blockIdx.x;
  const int n = p / (H * W), r = p - n * H * W, oy = r / W, ox = r - oy * W;
  float acc[kHeadMaxCout];
#pragma unroll
  for (int o = 0; o < kHeadMaxCout; ++o) acc[o] = 0.f;
  // taps outside the image contribute zero through a 0 / 1 factor on a clamped (always valid) address instead of a branch:
  // no load then waits behind a branch, and the unrolled loops keep all of a thread's loads in flight
  for (int c = tid; c < Cin; c += 256) {
    float xv[9];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int iy = oy - 1 + tap / 3, ix = ox - 1 + tap % 3;
      const bool ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
      const int cy = min(max(iy, 0), H - 1), cx = min(max(ix, 0), W - 1);
      xv[tap] = x[((size_t)(n * H + cy) * W + cx) * lda + c] * (ok ? 1.f : 0.f);
    }
#pragma unroll
    for (int o = 0; o < kHeadMaxCout; ++o) {
      if (o >= Cout) break;
      float wv[9];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) wv[tap] = w[((size_t)o * 9 + tap) * Cin + c];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) acc[o] = fmaf(xv[tap], wv[tap], acc[o]);
    }
  }
#pragma unroll
  for (int o = 0; o < kHeadMaxCout; ++o) {
    float s = acc[o];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s = s + __shfl_xor(s, d, 64);
    if (lane == 0) wave_sum[wv][o] = s;
  }
  __syncthreads();
  if (tid < Cout) {
    const float s = ((wave_sum[0][tid] + wave_sum[1][tid]) + wave_sum[2][tid]) + wave_sum[3][tid];
    const float v = (s + (bias ? bias[tid] : 0.f)) * mul;
    if (out_nchw) y[((size_t)n * Cout + tid) * H * W + r] = v;
    else y[(size_t)p * ldy + c0 + tid] = v;
  }
}

// One thread per output element of the cropped map: out[n, oy, ox, c0 + co], (oy, ox) in Hc x Wc.
// Deconvolution: full[y, x, co] = bias[co] + sum over (ci, ky, kx) with y = 2*iy + ky, x = 2*ix + kx of in[iy, ix, ci] * w[ci, co, ky, kx];
// Crop(offset 1): out[oy, ox] = full[oy + 1, ox + 1].  Sum order: ci, ky, kx ascending.
__global__ __launch_bounds__(256) void upflow_kernel(const float* __restrict__ in, int N, int Hi, int Wi, int C, const float* __restrict__ w,
                                                     const float* __restrict__ bias, int Hc, int Wc, float* __restrict__ out, int ldy,
                                                     int c0, unsigned* __restrict__ amax_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float v = 0.f;
  if (i < N * Hc * Wc * C) {
  const int co = i % C;
  int r = i / C;
  const int ox = r % Wc; r /= Wc;
  const int oy = r % Hc;
  const int n = r / Hc;
  const int fy = oy + 1, fx = ox + 1;
  float s = 0.f;
  for (int ci = 0; ci < C; ++ci)
    for (int ky = 0; ky < 4; ++ky) {
      const int ty = fy - ky;
      if (ty < 0 || (ty & 1) || (ty >> 1) >= Hi) continue;
      for (int kx = 0; kx < 4; ++kx) {
        const int tx = fx - kx;
        if (tx < 0 || (tx & 1) || (tx >> 1) >= Wi) continue;
        s = fmaf(in[((size_t)(n * Hi + (ty >> 1)) * Wi + (tx >> 1)) * C + ci], w[((ci * C + co) * 4 + ky) * 4 + kx], s);
      }
    }
  v = s + (bias ? bias[co] : 0.f);
  out[((size_t)(n * Hc + oy) * Wc + ox) * ldy + c0 + co] = v;
  }
  if (amax_out) {      // max|v| of the wave into one of the 256 slots the map's other producers (convolutions) also write (lsfa_conv_fwd)
    uint32_t m = __float_as_uint(v) & 0x7FFFFFFFu;
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(amax_out + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255), m);
  }
}

// channels-last 2x2 / 2 average, windows clipped to the map ('full' convention); a float4 of channels per thread
__global__ __launch_bounds__(256) void avgpool2_cl_kernel(const float4* __restrict__ x, int N, int H, int W, int C4, int Ho, int Wo,
                                                          float4* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)N * Ho * Wo * C4) return;
  const int c = (int)(i % C4);
  long r = i / C4;
  const int ox = (int)(r % Wo); r /= Wo;
  const int oy = (int)(r % Ho);
  const int n = (int)(r / Ho);
  const int y0 = 2 * oy, x0 = 2 * ox, y1 = min(y0 + 2, H), x1 = min(x0 + 2, W);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      const float4 v = x[(((size_t)n * H + yy) * W + xx) * C4 + c];
      s.x = s.x + v.x; s.y = s.y + v.y; s.z = s.z + v.z; s.w = s.w + v.w;
    }
  const float d = (float)((y1 - y0) * (x1 - x0));
  y[i] = make_float4(s.x / d, s.y / d, s.z / d, s.w / d);
}

// (N, Ctot, HW) planes -> (N, HW, C) rows for channels [c0, c0 + C): 64 x 64 tiles through LDS (padded rows: conflict-free both ways),
// coalesced 256-byte reads along the pixels and writes along the channels
__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const float* __restrict__ x, int Ctot, int HW, int c0, int C, float* __restrict__ y,
                                                           unsigned* __restrict__ amax_out) {
  __shared__ float tile[64][65];
  const int n = blockIdx.z, p0 = blockIdx.x * 64, cb = blockIdx.y * 64;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const float* xs = x + ((size_t)n * Ctot + c0) * HW;
  float top = 0.f;
  for (int r = ty; r < 64; r += 4) {
    const int c = cb + r, p = p0 + tx;
    const float v = (c < C && p < HW) ? xs[(size_t)c * HW + p] : 0.f;
    tile[r][tx] = v;
    top = fmaxf(top, fabsf(v));
  }
  if (amax_out) {      // the slots lsfa_conv_fwd reads as amax_in (a wave's maximum per atomic)
    uint32_t m = __float_as_uint(top);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (tx == 0) atomicMax(amax_out + ((((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + ty) & 255), m);
  }
  __syncthreads();
  float* ys = y + (size_t)n * HW * C;
  for (int r = ty; r < 64; r += 4) {
    const int p = p0 + r, c = cb + tx;
    if (p < HW && c < C) ys[(size_t)p * C + c] = tile[tx][r];
  }
}

}  // namespace

extern "C" int lsfa_nchw_to_nhwc(const float* x, int N, int Ctot, int HW, int c0, int C, float* y, unsigned* amax_out, void* stream) {
  LSFA_REQUIRE(x && y, "lsfa_nchw_to_nhwc: NULL argument");
  LSFA_REQUIRE(N > 0 && Ctot > 0 && HW > 0 && c0 >= 0 && C > 0 && c0 + C <= Ctot, "lsfa_nchw_to_nhwc: bad shape");
  ProfScope prof(LSFA_OP_FLOWNET, (hipStream_t)stream);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((HW + 63) / 64), (unsigned)((C + 63) / 64), (unsigned)N), dim3(256), 0,
                     (hipStream_t)stream, x, Ctot, HW, c0, C, y, amax_out);
  LSFA_LAUNCH_CHECK("lsfa_nchw_to_nhwc");
  return LSFA_OK;
}

extern "C" int lsfa_head_conv3x3(const float* x, int lda, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                                 float mul, float* y, int out_nchw, int ldy, int c0, void* stream) {
  LSFA_REQUIRE(x && w && y, "lsfa_head_conv3x3: NULL argument");
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && Cin > 0 && lda >= Cin && Cout > 0 && Cout <= kHeadMaxCout,
               "lsfa_head_conv3x3: bad shape (Cout must be 1..%d)", kHeadMaxCout);
  LSFA_REQUIRE(out_nchw || (ldy >= c0 + Cout && c0 >= 0), "lsfa_head_conv3x3: channels [%d, %d) do not fit ldy %d", c0, c0 + Cout, ldy);
  const int P = N * H * W;
  ProfScope prof(LSFA_OP_FLOWNET, (hipStream_t)stream);
  hipLaunchKernelGGL(head_conv3x3_kernel, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, x, lda, N, H, W, Cin, w, bias,
                     Cout, mul, y, out_nchw, ldy, c0);
  LSFA_LAUNCH_CHECK("lsfa_head_conv3x3");
  return LSFA_OK;
}

extern "C" int lsfa_upsample_flow(const float* in, int N, int Hi, int Wi, int C, const float* w, const float* bias, int Hc, int Wc,
                                  float* out, int ldy, int c0, unsigned* amax_out, void* stream) {
  LSFA_REQUIRE(in && w && out, "lsfa_upsample_flow: NULL argument");
  LSFA_REQUIRE(N > 0 && Hi > 0 && Wi > 0 && C > 0 && C <= 8 && Hc > 0 && Wc > 0 && Hc <= 2 * Hi + 1 && Wc <= 2 * Wi + 1 && ldy >= c0 + C && c0 >= 0,
               "lsfa_upsample_flow: bad shape");
  const int total = N * Hc * Wc * C;
  ProfScope prof(LSFA_OP_FLOWNET, (hipStream_t)stream);
  hipLaunchKernelGGL(upflow_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, in, N, Hi, Wi, C, w, bias, Hc, Wc,
                     out, ldy, c0, amax_out);
  LSFA_LAUNCH_CHECK("lsfa_upsample_flow");
  return LSFA_OK;
}

extern "C" int lsfa_avgpool2_nhwc(const float* x, int N, int H, int W, int C, float* y, void* stream) {
  LSFA_REQUIRE(x && y, "lsfa_avgpool2_nhwc: NULL argument");
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "lsfa_avgpool2_nhwc: bad shape (C must be a multiple of 4)");
  LSFA_REQUIRE(!((uintptr_t)x & 15) && !((uintptr_t)y & 15), "lsfa_avgpool2_nhwc: x / y must be 16-byte aligned");
  const int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
  const long total = (long)N * Ho * Wo * (C / 4);
  ProfScope prof(LSFA_OP_FLOWNET, (hipStream_t)stream);
  hipLaunchKernelGGL(avgpool2_cl_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, N, H, W,
                     C / 4, Ho, Wo, (float4*)y);
  LSFA_LAUNCH_CHECK("lsfa_avgpool2_nhwc");
  return LSFA_OK;
}
