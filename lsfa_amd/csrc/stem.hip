// The stem of the ResNets (dff_rfcn/symbols/resnet.py:151-163: bn_data -> conv0 7x7/2 + bn0 + relu0 -> pool0 3x3/2 max)
// and the average pooling that shrinks the frame in front of the small net (resnet_v1_101_flownet_rfcn.py:216, `resize_data`,
// 4x4/4 avg).  Through the libraries this is six launches per frame (avg pool, bn_data pass, layout copy, convolution,
// bias + ReLU pass, max pool: ~70 us for the small net's 150x250 input, ~215 us for the backbone's 600x1000); here three:
//   avgpool_kernel        NCHW k x k / k average
//   stem_conv_kernel      bn_data (per-channel affine, applied while the input patch is staged: padding stays 0) + 7x7 stride-2
//                         convolution with bn0 folded + bias + ReLU, 3 -> 64 channels, NCHW in, channels-last out
//   maxpool3x3s2_kernel   channels-last 3x3 / 2, pad 1
// The convolution is 147 MACs per output and channel: vector fp32 FMAs.  A workgroup computes 4 rows x 8 columns of conv
// outputs x 64 channels; lane = output channel, wave = row.  The 147 weights of a channel stay in the thread's registers;
// the input patch (3 x 13 x 24 floats) sits in LDS: a thread reads its row of 24 inputs (the same address in all lanes:
// broadcast ds_read_b128) and does 56 FMAs per (ci, ky).
// Summation order: ci, ky, kx ascending, one fmaf chain per output (the oracle's conv is a library stage: compared by
// tolerance, like every convolution).
#include "common.h"

using namespace lsfa;

namespace {

constexpr int kStemCout = 64, kStemCin = 3, kK = 7;
constexpr int kTileRows = 4, kTileCols = 8;
constexpr int kInRows = 2 * (kTileRows - 1) + kK;       // 13
constexpr int kInCols = 2 * (kTileCols - 1) + kK;       // 21
constexpr int kInPitch = 24;

__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, int NC, int H, int W, int k, int Ho, int Wo,
                                                      float* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)NC * Ho * Wo) return;
  const int ox = (int)(i % Wo);
  const long r = i / Wo;
  const int oy = (int)(r % Ho);
  const long nc = r / Ho;
  const int y0 = oy * k, x0 = ox * k, y1 = min(y0 + k, H), x1 = min(x0 + k, W);
  const float* p = x + nc * (long)H * W;
  float s = 0.f;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) s = s + p[(long)yy * W + xx];
  y[i] = s / (float)((y1 - y0) * (x1 - x0));       // ceil-mode windows are clipped to the image (no padding)
}

// grid (ceil(ceil(Wo / 8) / tiles_per_wg), ceil(Ho / 4), N); block 256.  w_l: (3, 7, 7, 64) floats = [ci][ky][kx][co].
// A thread keeps the 147 weights of its output channel in registers (loaded once, coalesced across the wave) and walks
// `tiles_per_wg` tiles of 4 rows x 8 columns along x; only the input patch goes through LDS (all lanes of a wave read the
// same row: broadcast ds_read_b128).  The empty asm statements keep the compiler from pairing the FMAs into
// v_pk_fma_f32 (the operands are not register-adjacent: it cost 1.4 v_mov per FMA).
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ x, int H, int W, const float* __restrict__ in_scale,
                                                        const float* __restrict__ in_shift, const float* __restrict__ w_l,
                                                        const float* __restrict__ bias, int Ho, int Wo, int tiles_per_wg,
                                                        const float* accum, int act, float* y, unsigned* __restrict__ amax_out) {
  __shared__ __attribute__((aligned(16))) float in_s[kStemCin][kInRows][kInPitch];
  const int tid = threadIdx.x, co = tid & 63, row = tid >> 6;
  const int n = blockIdx.z, oy0 = blockIdx.y * kTileRows;
  const int iy0 = 2 * oy0 - 3;
  const float* xin = x + (size_t)n * kStemCin * H * W;
  float wreg[kStemCin * kK * kK];
#pragma unroll
  for (int k = 0; k < kStemCin * kK * kK; ++k) wreg[k] = w_l[k * kStemCout + co];
  const float b = bias ? bias[co] : 0.f;
  const float sc0 = in_scale ? in_scale[0] : 1.f, sc1 = in_scale ? in_scale[1] : 1.f, sc2 = in_scale ? in_scale[2] : 1.f;
  const float sh0 = in_shift ? in_shift[0] : 0.f, sh1 = in_shift ? in_shift[1] : 0.f, sh2 = in_shift ? in_shift[2] : 0.f;
  const int oy = oy0 + row;
  float top = 0.f;                        // max |y| this thread stored
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int ox0 = (blockIdx.x * tiles_per_wg + t) * kTileCols;
    if (ox0 >= Wo) break;
    const int ix0 = 2 * ox0 - 3;
    __syncthreads();                      // the previous tile's readers are done with in_s
    for (int e = tid; e < kStemCin * kInRows * kInPitch; e += 256) {
      const int c = e % kInPitch, r = (e / kInPitch) % kInRows, ci = e / (kInPitch * kInRows);
      const int iy = iy0 + r, ix = ix0 + c;
      float v = 0.f;
      if (c < kInCols && iy >= 0 && iy < H && ix >= 0 && ix < W) {
        v = xin[((size_t)ci * H + iy) * W + ix];
        if (in_scale) v = v * (ci == 0 ? sc0 : (ci == 1 ? sc1 : sc2)) + (ci == 0 ? sh0 : (ci == 1 ? sh1 : sh2));   // bn_data; the zero padding is applied to ITS output
      }
      in_s[ci][r][c] = v;
    }
    __syncthreads();
    float acc[kTileCols];
#pragma unroll
    for (int p = 0; p < kTileCols; ++p) acc[p] = 0.f;
#pragma unroll
    for (int ci = 0; ci < kStemCin; ++ci) {
#pragma unroll
      for (int ky = 0; ky < kK; ++ky) {
        const float4* ir = reinterpret_cast<const float4*>(&in_s[ci][2 * row + ky][0]);
        const float4 i0 = ir[0], i1 = ir[1], i2 = ir[2], i3 = ir[3], i4 = ir[4], i5 = ir[5];
        const float in[24] = {i0.x, i0.y, i0.z, i0.w, i1.x, i1.y, i1.z, i1.w, i2.x, i2.y, i2.z, i2.w,
                              i3.x, i3.y, i3.z, i3.w, i4.x, i4.y, i4.z, i4.w, i5.x, i5.y, i5.z, i5.w};
#pragma unroll
        for (int kx = 0; kx < kK; ++kx) {
          const float wv = wreg[(ci * kK + ky) * kK + kx];
#pragma unroll
          for (int p = 0; p < kTileCols; ++p) {
            acc[p] = fmaf(in[2 * p + kx], wv, acc[p]);
            asm volatile("" : "+v"(acc[p]));
          }
        }
      }
    }
    if (oy < Ho) {
      float* out = y + (((size_t)n * Ho + oy) * Wo) * kStemCout + co;
#pragma unroll
      for (int p = 0; p < kTileCols; ++p)
        if (ox0 + p < Wo) {
          float v = acc[p] + b;
          if (accum) v = v + accum[(((size_t)n * Ho + oy) * Wo + ox0 + p) * kStemCout + co];
          v = act == 1 ? fmaxf(v, 0.f) : (act == 2 ? (v > 0.f ? v : v * 0.1f) : v);
          out[(size_t)(ox0 + p) * kStemCout] = v;
          top = fmaxf(top, fabsf(v));
        }
    }
  }
  if (amax_out) {      // the slots lsfa_conv_fwd reads as amax_in
    uint32_t m = __float_as_uint(top);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (co == 0) atomicMax(amax_out + ((((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + row) & 255), m);
  }
}

// channels-last 3x3 / 2 max pooling, pad 1 (the padding never wins: windows are clipped); a thread = 4 channels of one output.
// y2 (optional): max(y * scale2[c] + shift2[c], 0), the first unit's bn1 + relu1 of the pooled map, as a second output.
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float4* __restrict__ x, int N, int H, int W, int C4, int Ho, int Wo,
                                                           float4* __restrict__ y, float4* __restrict__ y2,
                                                           const float4* __restrict__ scale2, const float4* __restrict__ shift2,
                                                           unsigned* __restrict__ amax_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float top = 0.f;        // max |.| of what this thread hands to the next convolution (y2 when present, else y)
  if (i < (long)N * Ho * Wo * C4) {
  const int c = (int)(i % C4);
  long r = i / C4;
  const int ox = (int)(r % Wo); r /= Wo;
  const int oy = (int)(r % Ho);
  const int n = (int)(r / Ho);
  const int y0 = max(2 * oy - 1, 0), y1 = min(2 * oy + 2, H), x0 = max(2 * ox - 1, 0), x1 = min(2 * ox + 2, W);
  float4 m = x[(((size_t)n * H + y0) * W + x0) * C4 + c];
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      const float4 v = x[(((size_t)n * H + yy) * W + xx) * C4 + c];
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  y[i] = m;
  if (y2) {
    const float4 sc = scale2[c], sh = shift2[c];
    m = make_float4(fmaxf(m.x * sc.x + sh.x, 0.f), fmaxf(m.y * sc.y + sh.y, 0.f), fmaxf(m.z * sc.z + sh.z, 0.f),
                    fmaxf(m.w * sc.w + sh.w, 0.f));
    y2[i] = m;
  }
  top = fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fmaxf(fabsf(m.z), fabsf(m.w)));
  }
  if (amax_out) {      // the slots lsfa_conv_fwd reads as amax_in (bit patterns of non-negative floats order like the floats)
    uint32_t b = __float_as_uint(top);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) b = max(b, (uint32_t)__shfl_xor((int)b, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(amax_out + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255), b);
  }
}

}  // namespace

extern "C" int lsfa_avgpool_nchw(const float* x, int N, int C, int H, int W, int k, float* y, void* stream) {
  LSFA_REQUIRE(x && y, "lsfa_avgpool_nchw: NULL argument");
  LSFA_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && k > 0, "lsfa_avgpool_nchw: bad shape");
  const int Ho = (H + k - 1) / k, Wo = (W + k - 1) / k;
  const long total = (long)N * C * Ho * Wo;
  ProfScope prof(LSFA_OP_STEM, (hipStream_t)stream);
  hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, N * C, H, W, k, Ho, Wo, y);
  LSFA_LAUNCH_CHECK("lsfa_avgpool_nchw");
  return LSFA_OK;
}

extern "C" int lsfa_stem_conv7x7s2(const float* x, int N, int H, int W, const float* in_scale, const float* in_shift,
                                   const float* w_l, const float* bias, float* y, void* stream) {
  return lsfa_stem_conv7x7s2_ex(x, N, H, W, in_scale, in_shift, w_l, bias, nullptr, 1, y, nullptr, stream);
}

extern "C" int lsfa_stem_conv7x7s2_ex(const float* x, int N, int H, int W, const float* in_scale, const float* in_shift,
                                      const float* w_l, const float* bias, const float* accum, int act, float* y, unsigned* amax_out,
                                      void* stream) {
  LSFA_REQUIRE(x && w_l && y, "lsfa_stem_conv7x7s2: NULL argument");
  LSFA_REQUIRE(act >= 0 && act <= 2, "lsfa_stem_conv7x7s2_ex: act must be 0, 1 or 2");
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0, "lsfa_stem_conv7x7s2: bad shape");
  LSFA_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lsfa_stem_conv7x7s2: in_scale and in_shift go together");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  ProfScope prof(LSFA_OP_STEM, (hipStream_t)stream);
  // enough workgroups to fill the chip (>= ~768), each walking as many tiles along x as that allows (the 147 weight
  // registers are loaded once per workgroup)
  const int xt = (Wo + kTileCols - 1) / kTileCols, yt = (Ho + kTileRows - 1) / kTileRows;
  int tpw = (int)(((long)xt * yt * N) / 768);
  if (tpw < 1) tpw = 1;
  if (tpw > xt) tpw = xt;
  hipLaunchKernelGGL(stem_conv_kernel, dim3((xt + tpw - 1) / tpw, yt, N), dim3(256), 0, (hipStream_t)stream, x, H, W, in_scale, in_shift,
                     w_l, bias, Ho, Wo, tpw, accum, act, y, amax_out);
  LSFA_LAUNCH_CHECK("lsfa_stem_conv7x7s2");
  return LSFA_OK;
}

extern "C" int lsfa_maxpool3x3s2_nhwc(const float* x, int N, int H, int W, int C, float* y, float* y2, const float* scale2,
                                      const float* shift2, unsigned* amax_out, void* stream) {
  LSFA_REQUIRE(x && y, "lsfa_maxpool3x3s2_nhwc: NULL argument");
  LSFA_REQUIRE(!y2 || (scale2 && shift2 && y2 != y), "lsfa_maxpool3x3s2_nhwc: y2 needs scale2 / shift2 and must not alias y");
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "lsfa_maxpool3x3s2_nhwc: bad shape (C must be a multiple of 4)");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / 4);
  ProfScope prof(LSFA_OP_STEM, (hipStream_t)stream);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, N, H, W,
                     C / 4, Ho, Wo, (float4*)y, (float4*)y2, (const float4*)scale2, (const float4*)shift2, amax_out);
  LSFA_LAUNCH_CHECK("lsfa_maxpool3x3s2_nhwc");
  return LSFA_OK;
}
