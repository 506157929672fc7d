// Compressed-domain motion vectors on the device: accumulation of per-frame macroblock vectors back
// to the key frame, the accumulated motion-vector field and the residual image
// (external/data_loader_py2/coviar_data_loader.c:71-177, create_and_load_mv_residual, accumulate = 1).
//
// The reference walks the decoder's block list serially and lets later blocks overwrite earlier ones.
// Every write reads the PREVIOUS frame's map (accu_src_old), so a frame is one gather once it is known,
// per pixel, WHICH block wrote it last:
//   mv_owner_kernel    one thread per (block, pixel of the block): atomicMax of the block index into an
//                      owner map (-1 = untouched) — "last writer wins" without the serial walk;
//   mv_gather_kernel   one thread per pixel: new[p] = owner >= 0 ? old[p - dst + src] : old[p].  The
//                      output is complete, so old/new ping-pong and the reference's per-frame memcpy
//                      (:123-125) disappears.
// Pure int32 index work, HBM/L2-bound, bit-exact with orc_coviar_* by construction.  Maps are (H, W, 2)
// row-major (the reference's are x-major; only its index arithmetic differs).
#include "common.h"

using namespace lsfa;

namespace {

constexpr int kThreads = 256;

__global__ __launch_bounds__(kThreads) void mv_fill_kernel(int* __restrict__ p, int n, int v) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i < n) p[i] = v;
}

__global__ __launch_bounds__(kThreads) void mv_identity_kernel(int* __restrict__ accu, int width, int height) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  if (i >= width * height) return;
  const int y = i / width, x = i - y * width;
  reinterpret_cast<int2*>(accu)[i] = make_int2(x, y);
}

// mvs (n, 7) int32 = {source, w, h, src_x, src_y, dst_x, dst_y}; grid (n, ceil(max_area / 256))
__global__ __launch_bounds__(kThreads) void mv_owner_kernel(const int* __restrict__ mvs, int n, int width, int height,
                                                            int* __restrict__ owner) {
  const int i = blockIdx.x;
  const int* mv = mvs + (size_t)i * 7;
  const int w = mv[1], h = mv[2], src_x = mv[3], src_y = mv[4], dst_x = mv[5], dst_y = mv[6];
  if (dst_x - src_x == 0 && dst_y - src_y == 0) return;
  // the reference's loops: x_start in [(-1 * w) / 2, w / 2), C division truncating toward zero
  const int x_lo = (-1 * w) / 2, x_hi = w / 2, y_lo = (-1 * h) / 2, y_hi = h / 2;
  const int bw = x_hi - x_lo, bh = y_hi - y_lo;
  if (bw <= 0 || bh <= 0) return;
  const int t = blockIdx.y * kThreads + threadIdx.x;
  if (t >= bw * bh) return;
  const int xs = x_lo + t % bw, ys = y_lo + t / bw;
  const int pdx = dst_x + xs, pdy = dst_y + ys, psx = src_x + xs, psy = src_y + ys;
  if (pdy >= 0 && pdy < height && pdx >= 0 && pdx < width && psy >= 0 && psy < height && psx >= 0 && psx < width)
    atomicMax(&owner[pdy * width + pdx], i);
}

__global__ __launch_bounds__(kThreads) void mv_gather_kernel(const int* __restrict__ mvs, const int* __restrict__ owner,
                                                             const int* __restrict__ accu_old, int* __restrict__ accu_new,
                                                             int width, int height) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= width * height) return;
  const int o = owner[p];
  int src = p;
  if (o >= 0) {
    const int* mv = mvs + (size_t)o * 7;
    const int y = p / width, x = p - y * width;
    src = (y - mv[6] + mv[4]) * width + (x - mv[5] + mv[3]);
  }
  reinterpret_cast<int2*>(accu_new)[p] = reinterpret_cast<const int2*>(accu_old)[src];
}

__global__ __launch_bounds__(kThreads) void mv_field_kernel(const int* __restrict__ accu, int width, int height,
                                                            int* __restrict__ mv) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= width * height) return;
  const int y = p / width, x = p - y * width;
  const int2 a = reinterpret_cast<const int2*>(accu)[p];
  reinterpret_cast<int2*>(mv)[p] = make_int2(x - a.x, y - a.y);
}

__global__ __launch_bounds__(kThreads) void mv_residual_kernel(const unsigned char* __restrict__ cur,
                                                               const unsigned char* __restrict__ ref,
                                                               const int* __restrict__ accu, int width, int height,
                                                               int* __restrict__ res) {
  const int p = blockIdx.x * kThreads + threadIdx.x;
  if (p >= width * height) return;
  const int2 a = reinterpret_cast<const int2*>(accu)[p];
  const size_t s = ((size_t)a.y * width + a.x) * 3, o = (size_t)p * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) res[o + c] = (int)cur[o + c] - (int)ref[s + c];
}

// ---- r5: transform_mv_res (lib/utils/image.py:202-228) on the device ---------------------------------------------------------------------
// The reference resizes the (H, W, 2) motion vectors and the (H, W, 3) residual by im_scale (cv2.resize, INTER_LINEAR, float32 images: OpenCV
// 3.2's HResizeLinear / VResizeLinear<float, float, float>), copies them into zero maps padded to the stride (np.zeros: float64), rewrites the
// residual's channels IN PLACE in float64 - channel 2 is computed from the already rewritten channel 0 (:218-219) -, resizes both by 1 / stride
// (CV_64F: double work type, the same float coefficients), scales the motion vectors by im_scale / stride and hands float64 arrays to the
// executor, which rounds them to float32.  An output element depends on 2 x 2 padded positions (1 / 16: rows 16 Y + 7 and 16 Y + 8, weights
// 0.5) and each of those on 2 x 2 source pixels: one thread per output element computes exactly that chain - the first resize's four values
// in float32 (mul, mul, add: the two passes' roundings), everything behind them in float64, one rounding to float32 at the end - instead of
// materialising three full-resolution maps.  oracle/np_ref.py::transform_mv_res is the same arithmetic statement by statement.
struct ResizeTap { int i0, i1; float a; };

// resize.cpp: `fx = (float)((dx + 0.5) * scale_x - 0.5); sx = cvFloor(fx); fx -= sx;` with scale = 1. / f; taps clamped into the image
__device__ __forceinline__ ResizeTap resize_tap(int d, int src_n, double inv_f) {
  const float f = (float)(((double)d + 0.5) * inv_f - 0.5);
  int s0 = (int)floorf(f);
  float a = f - (float)s0;
  if (s0 < 0) { s0 = 0; a = 0.f; }
  if (s0 >= src_n - 1) { s0 = src_n - 1; a = 0.f; }
  ResizeTap t;
  t.i0 = s0;
  t.i1 = min(s0 + 1, src_n - 1);
  t.a = a;
  return t;
}

template <typename T>
__device__ __forceinline__ float first_resize(const T* __restrict__ src, int H, int W, int C, int c, int y, int x, double inv_scale, float sign = 1.f) {
  const ResizeTap tx = resize_tap(x, W, inv_scale), ty = resize_tap(y, H, inv_scale);
  const float bx = 1.f - tx.a, by = 1.f - ty.a;
  // (sign: the reference negates the decoder's motion vectors before the transform, image.py:54 - exact, applied to the source values)
  const float s00 = (float)src[((size_t)ty.i0 * W + tx.i0) * C + c] * sign, s01 = (float)src[((size_t)ty.i0 * W + tx.i1) * C + c] * sign;
  const float s10 = (float)src[((size_t)ty.i1 * W + tx.i0) * C + c] * sign, s11 = (float)src[((size_t)ty.i1 * W + tx.i1) * C + c] * sign;
  const float h0 = s00 * bx + s01 * tx.a;          // the horizontal pass of the two rows, rounded to float like the work buffer
  const float h1 = s10 * bx + s11 * tx.a;
  return h0 * by + h1 * ty.a;
}

struct MvResArgs {
  int H, W;             // the decoded frame
  int h1, w1;           // cvRound(H im_scale), cvRound(W im_scale): the first resize's output
  int ph, pw;           // padded to the stride
  int oh, ow;           // cvRound(ph / stride), cvRound(pw / stride): the network's feature grid
  double inv_scale;     // 1. / im_scale
  double inv_rcnn;      // 1. / (1. / stride)
  double mv_mul;        // im_scale * (1. / stride)
  double m0, m1, m2, pixel_scale;       // pixel_means in B, G, R order
  float mv_sign;        // -1: the motion vectors are negated first (`motion_vector = - motion_vector`, image.py:54)
};

// value of padded map `which` (0: motion vectors, channel c; 1: the residual AFTER the in-place loop, channel c) at (y, x), float64
template <typename T>
__device__ __forceinline__ double padded_value(const T* __restrict__ mv, const T* __restrict__ res, const MvResArgs& a, int which, int c, int y, int x) {
  const bool inside = y < a.h1 && x < a.w1;
  if (which == 0) return inside ? (double)first_resize(mv, a.H, a.W, 2, c, y, x, a.inv_scale, a.mv_sign) : 0.0;
  // i = 0: p0 = (p2 - mean[2]) scale;  i = 1: p1 = (p1 - mean[1]) scale;  i = 2: p2 = (p0 - mean[0]) scale with the NEW p0
  if (c == 1) return ((inside ? (double)first_resize(res, a.H, a.W, 3, 1, y, x, a.inv_scale) : 0.0) - a.m1) * a.pixel_scale;
  const double p0 = ((inside ? (double)first_resize(res, a.H, a.W, 3, 2, y, x, a.inv_scale) : 0.0) - a.m2) * a.pixel_scale;
  return c == 0 ? p0 : (p0 - a.m0) * a.pixel_scale;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void transform_mv_res_kernel(const T* __restrict__ mv, const T* __restrict__ res, MvResArgs a,
                                                                    float* __restrict__ out_mv, float* __restrict__ out_res) {
  const int i = blockIdx.x * kThreads + threadIdx.x;
  const int plane = a.oh * a.ow;
  if (i >= 5 * plane) return;
  const int ch = i / plane, r = i - ch * plane, Y = r / a.ow, X = r - Y * a.ow;
  const int which = ch < 2 ? 0 : 1, c = ch < 2 ? ch : ch - 2;
  const ResizeTap tx = resize_tap(X, a.pw, a.inv_rcnn), ty = resize_tap(Y, a.ph, a.inv_rcnn);
  const double ax = (double)tx.a, bx = (double)(1.f - tx.a), ay = (double)ty.a, by = (double)(1.f - ty.a);      // `1.f - fx` in float, used in double
  const double h0 = padded_value(mv, res, a, which, c, ty.i0, tx.i0) * bx + padded_value(mv, res, a, which, c, ty.i0, tx.i1) * ax;
  const double h1 = padded_value(mv, res, a, which, c, ty.i1, tx.i0) * bx + padded_value(mv, res, a, which, c, ty.i1, tx.i1) * ax;
  double v = h0 * by + h1 * ay;
  if (which == 0) { v *= a.mv_mul; out_mv[c * plane + r] = (float)v; }
  else out_res[c * plane + r] = (float)v;
}

// ---- r5: resize + transform (lib/utils/image.py:266-308) of a decoded frame in one launch -----------------------------------------------
// `resize`: cv2.resize by im_scale on the FLOAT image (get_image converts the decoder's frame with .astype(np.float32) first, :52: OpenCV's
// float path, as above), zero padding to the image stride; `transform`: channel i = (im[..., 2 - i] - pixel_means[2 - i]) * pixel_scale: the
// subtraction in float32 when stride == 0 (a float32 image minus a Python float: config.py:172-182 leaves a list) and in float64 when the frame was
// padded (the padded copy is np.zeros(...): float64, image.py:288-293), the product in float64 (np.zeros), rounded to float32 at the executor.  Padding pixels are transformed like any other ((0 - mean) * scale), as in the
// reference.  A thread per output pixel, the three channels together.
template <typename T>
__global__ __launch_bounds__(kThreads) void resize_transform_kernel(const T* __restrict__ im, int N, int H, int W, int h1, int w1, int ph, int pw,
                                                                    double inv_scale, double m0, double m1, double m2, double pixel_scale,
                                                                    int sub_f64, float* __restrict__ out) {
  const long i = (long)blockIdx.x * kThreads + threadIdx.x;
  const long plane = (long)ph * pw;
  if (i >= N * plane) return;
  const int n = (int)(i / plane);
  const long r = i - n * plane;
  const int y = (int)(r / pw), x = (int)(r - (long)y * pw);
  const T* src = im + (size_t)n * H * W * 3;
  float b = 0.f, g = 0.f, rr = 0.f;
  if (y < h1 && x < w1) {
    b = first_resize(src, H, W, 3, 0, y, x, inv_scale);
    g = first_resize(src, H, W, 3, 1, y, x, inv_scale);
    rr = first_resize(src, H, W, 3, 2, y, x, inv_scale);
  }
  float* o = out + (size_t)n * 3 * plane + r;
  if (sub_f64) {
    // stride > 0: `resize` copied the float32 frame into np.zeros(...) - a float64 image (image.py:288-293) - so `transform` subtracts in float64
    // and the executor rounds once
    o[0] = (float)(((double)rr - m2) * pixel_scale);
    o[plane] = (float)(((double)g - m1) * pixel_scale);
    o[2 * plane] = (float)(((double)b - m0) * pixel_scale);
    return;
  }
  // stride == 0: a float32 image minus a Python float is a float32 subtraction (the mean rounded to float32 first); the product with pixel_scale is float64
  const float tr = rr - (float)m2, tg = g - (float)m1, tb = b - (float)m0;
  o[0] = (float)((double)tr * pixel_scale);
  o[plane] = (float)((double)tg * pixel_scale);
  o[2 * plane] = (float)((double)tb * pixel_scale);
}

// r6: the LAST frame of a video reaches `resize` as the uint8 image cv2.imread returned (lib/utils/image.py:45), and OpenCV interpolates uint8
// images in fixed point: shorts `saturate_cast<short>(c * 2048)` of the two float coefficients of an axis, an int32 horizontal pass
// `D = S[x0] a0 + S[x1] a1`, and `uchar((((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2)` vertically (OpenCV 3.2 imgwarp.cpp,
// VResizeLinear<uchar, int, short, FixedPtCast<int, uchar, 22>>; restated as oracle/np_ref.py::cv2_resize_linear_u8 - parity unpinned: no
// OpenCV in the image).  `transform` then sees a uint8 (or, padded, float64) image: float64 subtraction.  One thread per output pixel.
__global__ __launch_bounds__(kThreads) void resize_u8_fixed_transform_kernel(const unsigned char* __restrict__ im, int N, int H, int W, int h1, int w1,
                                                                             int ph, int pw, double inv_scale, double m0, double m1, double m2,
                                                                             double pixel_scale, float* __restrict__ out) {
  const long i = (long)blockIdx.x * kThreads + threadIdx.x;
  const long plane = (long)ph * pw;
  if (i >= N * plane) return;
  const int n = (int)(i / plane);
  const long r = i - n * plane;
  const int y = (int)(r / pw), x = (int)(r - (long)y * pw);
  const unsigned char* src = im + (size_t)n * H * W * 3;
  int v[3] = {0, 0, 0};
  if (y < h1 && x < w1) {
    const ResizeTap tx = resize_tap(x, W, inv_scale), ty = resize_tap(y, H, inv_scale);
    // cvRound of the float coefficient times 2048 (round to nearest even: rintf in the default mode)
    const int a0 = (int)rintf((1.f - tx.a) * 2048.f), a1 = (int)rintf(tx.a * 2048.f);
    const int b0 = (int)rintf((1.f - ty.a) * 2048.f), b1 = (int)rintf(ty.a * 2048.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int s00 = src[((size_t)ty.i0 * W + tx.i0) * 3 + c], s01 = src[((size_t)ty.i0 * W + tx.i1) * 3 + c];
      const int s10 = src[((size_t)ty.i1 * W + tx.i0) * 3 + c], s11 = src[((size_t)ty.i1 * W + tx.i1) * 3 + c];
      const int h0 = s00 * a0 + s01 * a1, h1_ = s10 * a0 + s11 * a1;
      v[c] = ((((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1_ >> 4)) >> 16) + 2) >> 2) & 0xFF;
    }
  }
  float* o = out + (size_t)n * 3 * plane + r;
  o[0] = (float)(((double)v[2] - m2) * pixel_scale);
  o[plane] = (float)(((double)v[1] - m1) * pixel_scale);
  o[2 * plane] = (float)(((double)v[0] - m0) * pixel_scale);
}

}  // namespace

extern "C" int lsfa_image_resize_transform(const void* im_hwc_bgr, int is_u8, int N, int H, int W, double im_scale, int h1, int w1, int stride,
                                           const double* pixel_means_bgr_host, double pixel_scale, float* data_nchw, int out_h, int out_w,
                                           void* stream) {
  LSFA_REQUIRE(im_hwc_bgr && pixel_means_bgr_host && data_nchw, "lsfa_image_resize_transform: NULL argument");
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && h1 > 0 && w1 > 0 && stride >= 0 && im_scale > 0.0, "lsfa_image_resize_transform: bad shape");
  const int ph = stride > 0 ? (h1 + stride - 1) / stride * stride : h1, pw = stride > 0 ? (w1 + stride - 1) / stride * stride : w1;
  if (ph != out_h || pw != out_w) {
    set_error("lsfa_image_resize_transform: output is %d x %d, the resized %d x %d frame padded to %d gives %d x %d", out_h, out_w, h1, w1, stride, ph, pw);
    return LSFA_EINVAL;
  }
  const long total = (long)N * ph * pw;
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(LSFA_OP_STEM, s);
  const dim3 grid((unsigned)((total + kThreads - 1) / kThreads));
  LSFA_REQUIRE(is_u8 >= 0 && is_u8 <= 2, "lsfa_image_resize_transform: is_u8 is 0 (float32), 1 (uint8, interpolated in float) or 2 (uint8, OpenCV's fixed-point path)");
  if (is_u8 == 2)
    hipLaunchKernelGGL(resize_u8_fixed_transform_kernel, grid, dim3(kThreads), 0, s, (const unsigned char*)im_hwc_bgr, N, H, W, h1, w1, ph, pw,
                       1.0 / im_scale, pixel_means_bgr_host[0], pixel_means_bgr_host[1], pixel_means_bgr_host[2], pixel_scale, data_nchw);
  else if (is_u8)
    hipLaunchKernelGGL(resize_transform_kernel<unsigned char>, grid, dim3(kThreads), 0, s, (const unsigned char*)im_hwc_bgr, N, H, W, h1, w1, ph, pw,
                       1.0 / im_scale, pixel_means_bgr_host[0], pixel_means_bgr_host[1], pixel_means_bgr_host[2], pixel_scale, stride > 0 ? 1 : 0, data_nchw);
  else
    hipLaunchKernelGGL(resize_transform_kernel<float>, grid, dim3(kThreads), 0, s, (const float*)im_hwc_bgr, N, H, W, h1, w1, ph, pw, 1.0 / im_scale,
                       pixel_means_bgr_host[0], pixel_means_bgr_host[1], pixel_means_bgr_host[2], pixel_scale, stride > 0 ? 1 : 0, data_nchw);
  LSFA_LAUNCH_CHECK("lsfa_image_resize_transform");
  return LSFA_OK;
}

extern "C" int lsfa_transform_mv_res(const void* motion_vector, const void* res_diff, int flags, int H, int W, double im_scale, int h1, int w1,
                                     int rcnn_stride, const double* pixel_means_bgr_host, double pixel_scale, float* out_mv, float* out_res,
                                     int out_h, int out_w, void* stream) {
  LSFA_REQUIRE(motion_vector && res_diff && pixel_means_bgr_host && out_mv && out_res, "lsfa_transform_mv_res: NULL argument");
  LSFA_REQUIRE(H > 0 && W > 0 && h1 > 0 && w1 > 0 && rcnn_stride > 0 && im_scale > 0.0 && (flags & ~3) == 0, "lsfa_transform_mv_res: bad shape or flags");
  MvResArgs a;
  a.H = H; a.W = W; a.h1 = h1; a.w1 = w1;
  a.ph = (h1 + rcnn_stride - 1) / rcnn_stride * rcnn_stride;
  a.pw = (w1 + rcnn_stride - 1) / rcnn_stride * rcnn_stride;
  const double rcnn_scale = 1.0 / (double)rcnn_stride;
  a.oh = (int)nearbyint((double)a.ph * rcnn_scale);        // cvRound (ties to even); a multiple of the stride divides exactly
  a.ow = (int)nearbyint((double)a.pw * rcnn_scale);
  if (a.oh != out_h || a.ow != out_w) {
    set_error("lsfa_transform_mv_res: outputs are %d x %d, the padded %d x %d map gives %d x %d", out_h, out_w, a.ph, a.pw, a.oh, a.ow);
    return LSFA_EINVAL;
  }
  a.inv_scale = 1.0 / im_scale;
  a.inv_rcnn = 1.0 / rcnn_scale;
  a.mv_mul = im_scale * rcnn_scale;
  a.m0 = pixel_means_bgr_host[0]; a.m1 = pixel_means_bgr_host[1]; a.m2 = pixel_means_bgr_host[2];
  a.pixel_scale = pixel_scale;
  a.mv_sign = (flags & 2) ? -1.f : 1.f;
  const bool is_int32 = (flags & 1) != 0;
  const int total = 5 * a.oh * a.ow;
  hipStream_t s = (hipStream_t)stream;
  if (is_int32)
    hipLaunchKernelGGL(transform_mv_res_kernel<int>, dim3(ceil_div(total, kThreads)), dim3(kThreads), 0, s, (const int*)motion_vector, (const int*)res_diff, a,
                       out_mv, out_res);
  else
    hipLaunchKernelGGL(transform_mv_res_kernel<float>, dim3(ceil_div(total, kThreads)), dim3(kThreads), 0, s, (const float*)motion_vector,
                       (const float*)res_diff, a, out_mv, out_res);
  LSFA_LAUNCH_CHECK("lsfa_transform_mv_res");
  return LSFA_OK;
}

extern "C" size_t lsfa_mv_workspace_bytes(int width, int height) {
  if (width <= 0 || height <= 0) return 0;
  return align_up((size_t)width * height * sizeof(int), 256);
}

extern "C" int lsfa_mv_identity(int* accu, int width, int height, void* stream) {
  LSFA_REQUIRE(accu && width > 0 && height > 0, "lsfa_mv_identity: bad argument");
  hipLaunchKernelGGL(mv_identity_kernel, dim3(ceil_div(width * height, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     accu, width, height);
  LSFA_LAUNCH_CHECK("lsfa_mv_identity");
  return LSFA_OK;
}

extern "C" int lsfa_mv_accumulate(const int* mvs, int n_mvs, int max_block_area, const int* accu_old, int* accu_new,
                                  int width, int height, void* ws, size_t ws_bytes, void* stream) {
  LSFA_REQUIRE(accu_old && accu_new && accu_old != accu_new, "lsfa_mv_accumulate: accu_old/accu_new must be two buffers");
  LSFA_REQUIRE(width > 0 && height > 0 && n_mvs >= 0 && max_block_area >= 0, "lsfa_mv_accumulate: bad shape");
  LSFA_REQUIRE(n_mvs == 0 || mvs, "lsfa_mv_accumulate: mvs is NULL");
  LSFA_REQUIRE((long)width * height < (1L << 30), "lsfa_mv_accumulate: frame too large");
  if (ws_bytes < lsfa_mv_workspace_bytes(width, height) || !ws) {
    set_error("lsfa_mv_accumulate: workspace %zu < %zu bytes", ws_bytes, lsfa_mv_workspace_bytes(width, height));
    return LSFA_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  int* owner = (int*)ws;
  const int px = width * height;
  hipLaunchKernelGGL(mv_fill_kernel, dim3(ceil_div(px, kThreads)), dim3(kThreads), 0, s, owner, px, -1);
  if (n_mvs > 0 && max_block_area > 0) {
    hipLaunchKernelGGL(mv_owner_kernel, dim3(n_mvs, ceil_div(max_block_area, kThreads)), dim3(kThreads), 0, s, mvs, n_mvs,
                       width, height, owner);
  }
  hipLaunchKernelGGL(mv_gather_kernel, dim3(ceil_div(px, kThreads)), dim3(kThreads), 0, s, mvs, (const int*)owner, accu_old,
                     accu_new, width, height);
  LSFA_LAUNCH_CHECK("lsfa_mv_accumulate");
  return LSFA_OK;
}

extern "C" int lsfa_mv_field(const int* accu, int width, int height, int* mv, void* stream) {
  LSFA_REQUIRE(accu && mv && width > 0 && height > 0, "lsfa_mv_field: bad argument");
  hipLaunchKernelGGL(mv_field_kernel, dim3(ceil_div(width * height, kThreads)), dim3(kThreads), 0, (hipStream_t)stream, accu,
                     width, height, mv);
  LSFA_LAUNCH_CHECK("lsfa_mv_field");
  return LSFA_OK;
}

extern "C" int lsfa_mv_residual(const unsigned char* bgr_cur, const unsigned char* bgr_ref, const int* accu, int width,
                                int height, int* res, void* stream) {
  LSFA_REQUIRE(bgr_cur && bgr_ref && accu && res && width > 0 && height > 0, "lsfa_mv_residual: bad argument");
  hipLaunchKernelGGL(mv_residual_kernel, dim3(ceil_div(width * height, kThreads)), dim3(kThreads), 0, (hipStream_t)stream,
                     bgr_cur, bgr_ref, accu, width, height, res);
  LSFA_LAUNCH_CHECK("lsfa_mv_residual");
  return LSFA_OK;
}
