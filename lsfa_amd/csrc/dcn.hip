// Deformable-convolution bilinear im2col and the fused inference BatchNorm+ReLU pass.
// Reference interfaces: include/lsfa_hip.h.  Arithmetic = oracle orc_deform_im2col /
// orc_scale_shift_relu.
#include "common.h"

using namespace lsfa;

namespace {

constexpr int kThreads = 256;

__device__ __forceinline__ float dcn_bilinear(const float* __restrict__ plane, int data_width, int height, int width,
                                              float h, float w) {
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high, w_high;
  if (h_low >= height - 1) { h_high = h_low = height - 1; h = (float)h_low; } else { h_high = h_low + 1; }
  if (w_low >= width - 1) { w_high = w_low = width - 1; w = (float)w_low; } else { w_high = w_low + 1; }
  const float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  const float v1 = plane[h_low * data_width + w_low], v2 = plane[h_low * data_width + w_high];
  const float v3 = plane[h_high * data_width + w_low], v4 = plane[h_high * data_width + w_high];
  const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}

// thread -> (n, c, tap, ho, wo) with wo fastest: offsets and col are accessed coalesced,
// the four data taps are near-contiguous gathers.  A thread handles one tap so that the
// (kh*kw) x (Ho*Wo) plane of col rows is written as full lines.
__global__ __launch_bounds__(kThreads) void deform_im2col_kernel(const float* __restrict__ data,
                                                                 const float* __restrict__ offset, int C, int H, int W,
                                                                 int kh, int kw, int pad, int stride, int dilate,
                                                                 int dg, int Ho, int Wo, float* __restrict__ col,
                                                                 size_t total) {
  const int HoWo = Ho * Wo, KK = kh * kw, cpg = C / dg;
  for (size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x; idx < total; idx += (size_t)gridDim.x * kThreads) {
    const int sp = (int)(idx % HoWo);
    const int tap = (int)((idx / HoWo) % KK);
    const int c = (int)((idx / HoWo / KK) % C);
    const int n = (int)(idx / HoWo / KK / C);
    const int ho = sp / Wo, wo = sp - ho * Wo;
    const int i = tap / kw, j = tap - i * kw;
    const int g = c / cpg;
    const float* off = offset + ((size_t)n * dg + g) * 2 * KK * HoWo;
    const float oh = off[(size_t)(2 * tap) * HoWo + sp];
    const float ow = off[(size_t)(2 * tap + 1) * HoWo + sp];
    const float h_im = (float)(ho * stride - pad + i * dilate) + oh;
    const float w_im = (float)(wo * stride - pad + j * dilate) + ow;
    float val = 0.f;
    if (h_im >= 0 && w_im >= 0 && h_im < H && w_im < W)
      val = dcn_bilinear(data + ((size_t)n * C + c) * H * W, W, H, W, h_im, w_im);
    col[idx] = val;
  }
}

// Channels-last form: data (N, H, W, C), offset (N, Ho, Wo, 2*KK*dg), col (N, Ho*Wo, KK, C).
// A thread produces four consecutive channels of one (pixel, tap): the 128 channels of a deformable
// group share the tap's offset, so a wave reads each of the four corners as 1 KB of consecutive
// channels and writes 1 KB of col — every access is a full line.  The corner selection and the
// weights are those of dcn_bilinear (same expressions, same order => same bits per channel).
// one float4 of channels of one (pixel, tap): the bilinear sample of DeformableConvolution's im2col (zeros outside the image)
__device__ __forceinline__ float4 deform_sample4(const float* __restrict__ data, const float* __restrict__ offset, int C, int H, int W, int kw,
                                                 int pad, int stride, int dilate, int cpg, int KK, int off_ld, size_t pix, int n, int ho, int wo,
                                                 int tap, int c) {
  const int i = tap / kw, j = tap - i * kw;
  const int g = c / cpg;
  const float* off = offset + pix * (size_t)off_ld + (size_t)g * 2 * KK + 2 * tap;
  const float oh = off[0], ow = off[1];
  float h = (float)(ho * stride - pad + i * dilate) + oh;
  float w = (float)(wo * stride - pad + j * dilate) + ow;
  float4 val = make_float4(0.f, 0.f, 0.f, 0.f);
  if (h >= 0 && w >= 0 && h < H && w < W) {
    int h_low = (int)floorf(h), w_low = (int)floorf(w);
    int h_high, w_high;
    if (h_low >= H - 1) { h_high = h_low = H - 1; h = (float)h_low; } else { h_high = h_low + 1; }
    if (w_low >= W - 1) { w_high = w_low = W - 1; w = (float)w_low; } else { w_high = w_low + 1; }
    const float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
    const float* base = data + (size_t)n * H * W * C + c;
    const float4 v1 = *reinterpret_cast<const float4*>(base + ((size_t)h_low * W + w_low) * C);
    const float4 v2 = *reinterpret_cast<const float4*>(base + ((size_t)h_low * W + w_high) * C);
    const float4 v3 = *reinterpret_cast<const float4*>(base + ((size_t)h_high * W + w_low) * C);
    const float4 v4 = *reinterpret_cast<const float4*>(base + ((size_t)h_high * W + w_high) * C);
    const float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
    val.x = w1 * v1.x + w2 * v2.x + w3 * v3.x + w4 * v4.x;
    val.y = w1 * v1.y + w2 * v2.y + w3 * v3.y + w4 * v4.y;
    val.z = w1 * v1.z + w2 * v2.z + w3 * v3.z + w4 * v4.z;
    val.w = w1 * v1.w + w2 * v2.w + w3 * v3.w + w4 * v4.w;
  }
  return val;
}

__global__ __launch_bounds__(kThreads) void deform_im2col_cl_kernel(const float* __restrict__ data,
                                                                    const float* __restrict__ offset, int C, int H,
                                                                    int W, int kh, int kw, int pad, int stride,
                                                                    int dilate, int dg, int Ho, int Wo, int off_ld,
                                                                    float* __restrict__ col, size_t total4) {
  const int KK = kh * kw, C4 = C / 4, cpg = C / dg, HoWo = Ho * Wo;
  for (size_t idx = (size_t)blockIdx.x * kThreads + threadIdx.x; idx < total4; idx += (size_t)gridDim.x * kThreads) {
    const int c = (int)(idx % C4) * 4;
    const int tap = (int)((idx / C4) % KK);
    const size_t pix = idx / C4 / KK;                 // n * HoWo + sp
    const int sp = (int)(pix % HoWo);
    const int n = (int)(pix / HoWo);
    const int ho = sp / Wo, wo = sp - ho * Wo;
    reinterpret_cast<float4*>(col)[idx] = deform_sample4(data, offset, C, H, W, kw, pad, stride, dilate, cpg, KK, off_ld, pix, n, ho, wo, tap, c);
  }
}

// n / d for 0 <= n < 2^24, 0 < d, inv = 1.0f / d from the host: the float product is off by at most one, the remainder fixes it
__device__ __forceinline__ int fdiv24(int n, int d, float inv) {
  int q = (int)((float)n * inv);
  const int r = n - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// The same with the index arithmetic the network's shapes allow (r4): C / 4 a power of two <= 256 and fewer than 2^24 (pixel, tap) units.
// The general kernel spends five 64-bit integer divisions (~100 instructions each) per float4 it writes - more than the sampling
// itself; here a thread's channel group is a mask of its id, its unit a shift, and the three remaining divisions run through float
// reciprocals.  One (pixel, tap) unit per C / 4 threads, 256 / (C / 4) units per workgroup.  Same values.
__global__ __launch_bounds__(kThreads) void deform_im2col_cl_pow2_kernel(const float* __restrict__ data, const float* __restrict__ offset, int C,
                                                                         int H, int W, int kh, int kw, int pad, int stride, int dilate, int dg,
                                                                         int Ho, int Wo, int off_ld, float* __restrict__ col, int units,
                                                                         int c4_shift, float inv_kk, float inv_howo, float inv_wo) {
  const int KK = kh * kw, cpg = C / dg, HoWo = Ho * Wo;
  const int c = (int)(threadIdx.x & ((1u << c4_shift) - 1u)) * 4;
  const int per_wg = kThreads >> c4_shift;
  for (int u = blockIdx.x * per_wg + (int)(threadIdx.x >> c4_shift); u < units; u += gridDim.x * per_wg) {
    const int pix = fdiv24(u, KK, inv_kk), tap = u - pix * KK;
    const int n = fdiv24(pix, HoWo, inv_howo), sp = pix - n * HoWo;
    const int ho = fdiv24(sp, Wo, inv_wo), wo = sp - ho * Wo;
    reinterpret_cast<float4*>(col)[((size_t)u << c4_shift) + (c >> 2)] =
        deform_sample4(data, offset, C, H, W, kw, pad, stride, dilate, cpg, KK, off_ld, (size_t)pix, n, ho, wo, tap, c);
  }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void scale_shift_relu_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, int C, int HW,
                                                                    int relu, float slope, float* __restrict__ y,
                                                                    size_t nvec) {
  for (size_t v = (size_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += (size_t)gridDim.x * kThreads) {
    const size_t e = v * VEC;
    const int c = (int)((e / HW) % C);
    const float sc = scale[c], sh = shift[c];
    float in[VEC], out[VEC];
    if (VEC == 4) { const float4 t = *reinterpret_cast<const float4*>(x + e); in[0] = t.x; in[1] = t.y; in[2] = t.z; in[3] = t.w; }
    else if (VEC == 2) { const float2 t = *reinterpret_cast<const float2*>(x + e); in[0] = t.x; in[1] = t.y; }
    else in[0] = x[e];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float r = in[i] * sc + sh;
      out[i] = relu == 2 ? (r > 0.f ? r : r * slope) : ((relu && r < 0.f) ? 0.f : r);
    }
    if (VEC == 4) *reinterpret_cast<float4*>(y + e) = make_float4(out[0], out[1], out[2], out[3]);
    else if (VEC == 2) *reinterpret_cast<float2*>(y + e) = make_float2(out[0], out[1]);
    else y[e] = out[0];
  }
}

// channels-last rows: x[r][c], channel fastest.  One float4 of channels per thread; C % 4 == 0.
__global__ __launch_bounds__(kThreads) void scale_shift_relu_cl_kernel(const float4* __restrict__ x,
                                                                       const float4* __restrict__ scale,
                                                                       const float4* __restrict__ shift, int C4, int relu,
                                                                       float4* __restrict__ y, size_t nvec) {
  for (size_t v = (size_t)blockIdx.x * kThreads + threadIdx.x; v < nvec; v += (size_t)gridDim.x * kThreads) {
    const int c4 = (int)(v % C4);
    const float4 sc = scale[c4], sh = shift[c4], t = x[v];
    float4 r;
    r.x = t.x * sc.x + sh.x; r.y = t.y * sc.y + sh.y; r.z = t.z * sc.z + sh.z; r.w = t.w * sc.w + sh.w;
    if (relu) {
      r.x = r.x < 0.f ? 0.f : r.x; r.y = r.y < 0.f ? 0.f : r.y; r.z = r.z < 0.f ? 0.f : r.z; r.w = r.w < 0.f ? 0.f : r.w;
    }
    y[v] = r;
  }
}

}  // namespace

extern "C" int lsfa_scale_shift_relu_cl(const float* x, const float* scale, const float* shift, long long rows, int C,
                                        int relu, float* y, void* stream) {
  LSFA_REQUIRE(x && scale && shift && y, "lsfa_scale_shift_relu_cl: NULL argument");
  LSFA_REQUIRE(rows > 0 && C > 0 && C % 4 == 0, "lsfa_scale_shift_relu_cl: rows=%lld C=%d (C must be a positive multiple of 4)",
               rows, C);
  LSFA_REQUIRE(!((uintptr_t)x % 16) && !((uintptr_t)y % 16) && !((uintptr_t)scale % 16) && !((uintptr_t)shift % 16),
               "lsfa_scale_shift_relu_cl: pointers must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const size_t nvec = (size_t)rows * (C / 4);
  size_t nb = (nvec + kThreads - 1) / kThreads;
  if (nb > 4096) nb = 4096;
  ProfScope prof(LSFA_OP_BNRELU, s);
  hipLaunchKernelGGL(scale_shift_relu_cl_kernel, dim3((unsigned)nb), dim3(kThreads), 0, s, (const float4*)x,
                     (const float4*)scale, (const float4*)shift, C / 4, relu, (float4*)y, nvec);
  LSFA_LAUNCH_CHECK("lsfa_scale_shift_relu_cl");
  return LSFA_OK;
}

extern "C" int lsfa_deform_im2col(const float* data, const float* offset, int N, int C, int H, int W, int kh, int kw,
                                  int pad, int stride, int dilate, int deform_groups, int Ho, int Wo, float* col,
                                  void* stream) {
  LSFA_REQUIRE(data && offset && col, "lsfa_deform_im2col: NULL argument");
  LSFA_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && dilate > 0 && Ho > 0 && Wo > 0,
               "lsfa_deform_im2col: bad shape");
  LSFA_REQUIRE(deform_groups > 0 && C % deform_groups == 0, "lsfa_deform_im2col: C=%d not divisible by deformable groups %d",
               C, deform_groups);
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * C * kh * kw * Ho * Wo;
  size_t nb = (total + kThreads - 1) / kThreads;
  if (nb > 65536) nb = 65536;
  ProfScope prof(LSFA_OP_DCN_IM2COL, s);
  hipLaunchKernelGGL(deform_im2col_kernel, dim3((unsigned)nb), dim3(kThreads), 0, s, data, offset, C, H, W, kh, kw, pad,
                     stride, dilate, deform_groups, Ho, Wo, col, total);
  LSFA_LAUNCH_CHECK("lsfa_deform_im2col");
  return LSFA_OK;
}

extern "C" int lsfa_deform_im2col_cl(const float* data, const float* offset, int N, int C, int H, int W, int kh, int kw,
                                     int pad, int stride, int dilate, int deform_groups, int Ho, int Wo, float* col,
                                     void* stream) {
  return lsfa_deform_im2col_cl_ld(data, offset, 2 * kh * kw * deform_groups, N, C, H, W, kh, kw, pad, stride, dilate, deform_groups,
                                  Ho, Wo, col, stream);
}

extern "C" int lsfa_deform_im2col_cl_ld(const float* data, const float* offset, int offset_ld, int N, int C, int H, int W, int kh,
                                        int kw, int pad, int stride, int dilate, int deform_groups, int Ho, int Wo, float* col,
                                        void* stream) {
  LSFA_REQUIRE(data && offset && col, "lsfa_deform_im2col_cl: NULL argument");
  LSFA_REQUIRE(offset_ld >= 2 * kh * kw * deform_groups, "lsfa_deform_im2col_cl_ld: offset_ld %d < %d offset channels", offset_ld,
               2 * kh * kw * deform_groups);
  LSFA_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && dilate > 0 && Ho > 0 && Wo > 0,
               "lsfa_deform_im2col_cl: bad shape");
  LSFA_REQUIRE(deform_groups > 0 && C % deform_groups == 0 && (C / deform_groups) % 4 == 0,
               "lsfa_deform_im2col_cl: C=%d, deformable groups %d: channels per group must be a multiple of 4", C,
               deform_groups);
  LSFA_REQUIRE(!((uintptr_t)data % 16) && !((uintptr_t)col % 16), "lsfa_deform_im2col_cl: data/col must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  const size_t total4 = (size_t)N * Ho * Wo * kh * kw * (C / 4);
  size_t nb = (total4 + kThreads - 1) / kThreads;
  if (nb > 65536) nb = 65536;
  ProfScope prof(LSFA_OP_DCN_IM2COL, s);
  const int C4 = C / 4;
  const size_t units = (size_t)N * Ho * Wo * kh * kw;
  if ((C4 & (C4 - 1)) == 0 && C4 <= kThreads && units < (1u << 24)) {
    int shift = 0;
    while ((1 << shift) < C4) ++shift;
    const int per_wg = kThreads >> shift;
    size_t nbu = (units + per_wg - 1) / per_wg;
    if (nbu > 65536) nbu = 65536;
    hipLaunchKernelGGL(deform_im2col_cl_pow2_kernel, dim3((unsigned)nbu), dim3(kThreads), 0, s, data, offset, C, H, W, kh, kw, pad, stride, dilate,
                       deform_groups, Ho, Wo, offset_ld, col, (int)units, shift, 1.0f / (float)(kh * kw), 1.0f / (float)(Ho * Wo), 1.0f / (float)Wo);
    LSFA_LAUNCH_CHECK("lsfa_deform_im2col_cl");
    return LSFA_OK;
  }
  hipLaunchKernelGGL(deform_im2col_cl_kernel, dim3((unsigned)nb), dim3(kThreads), 0, s, data, offset, C, H, W, kh, kw,
                     pad, stride, dilate, deform_groups, Ho, Wo, offset_ld, col, total4);
  LSFA_LAUNCH_CHECK("lsfa_deform_im2col_cl");
  return LSFA_OK;
}

static int scale_shift_act(const float* x, const float* scale, const float* shift, int N, int C, int HW, int relu,
                           float slope, float* y, void* stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t total = (size_t)N * C * HW;
  int vec = (HW % 4 == 0) ? 4 : (HW % 2 == 0) ? 2 : 1;
  if (((uintptr_t)x % (4 * vec)) || ((uintptr_t)y % (4 * vec))) vec = 1;
  const size_t nvec = total / vec;
  size_t nb = (nvec + kThreads - 1) / kThreads;
  if (nb > 4096) nb = 4096;
  ProfScope prof(LSFA_OP_BNRELU, s);
  if (vec == 4) hipLaunchKernelGGL(scale_shift_relu_kernel<4>, dim3((unsigned)nb), dim3(kThreads), 0, s, x, scale, shift, C, HW, relu, slope, y, nvec);
  else if (vec == 2) hipLaunchKernelGGL(scale_shift_relu_kernel<2>, dim3((unsigned)nb), dim3(kThreads), 0, s, x, scale, shift, C, HW, relu, slope, y, nvec);
  else hipLaunchKernelGGL(scale_shift_relu_kernel<1>, dim3((unsigned)nb), dim3(kThreads), 0, s, x, scale, shift, C, HW, relu, slope, y, nvec);
  LSFA_LAUNCH_CHECK("lsfa_scale_shift_relu");
  return LSFA_OK;
}

extern "C" int lsfa_scale_shift_relu(const float* x, const float* scale, const float* shift, int N, int C, int HW,
                                     int relu, float* y, void* stream) {
  LSFA_REQUIRE(x && scale && shift && y, "lsfa_scale_shift_relu: NULL argument");
  LSFA_REQUIRE(N > 0 && C > 0 && HW > 0, "lsfa_scale_shift_relu: bad shape");
  return scale_shift_act(x, scale, shift, N, C, HW, relu ? 1 : 0, 0.f, y, stream);
}

extern "C" int lsfa_scale_shift_leaky(const float* x, const float* scale, const float* shift, int N, int C, int HW,
                                      float slope, float* y, void* stream) {
  LSFA_REQUIRE(x && scale && shift && y, "lsfa_scale_shift_leaky: NULL argument");
  LSFA_REQUIRE(N > 0 && C > 0 && HW > 0, "lsfa_scale_shift_leaky: bad shape");
  return scale_shift_act(x, scale, shift, N, C, HW, 2, slope, y, stream);
}
