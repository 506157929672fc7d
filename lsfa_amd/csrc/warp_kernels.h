// Device kernel of warp.hip: bilinear feature warp (GridGenerator 'warp' + BilinearSampler) with the
// fused key-path (x scale_map) and cur-path (+ rnet_conv0(res_diff) + small-net feature) epilogues.
//
// Mapping on CDNA4: a thread owns VEC horizontally adjacent output pixels and a run of channels
// (gridDim.x tiles the flattened H*W plane, gridDim.y the channel runs, gridDim.z the images).  The
// per-pixel tap offsets, validity and weights are computed once and reused across the channel run (the
// reference materialises a (1,2,H,W) grid and recomputes the taps per channel).  Lanes of a wave cover
// 64*VEC consecutive pixels of one channel plane, so flow / mul / add / out move as 4*VEC-byte
// coalesced vectors and the taps are near-contiguous gathers served by L1/L2: in the interior fast
// path (wave-uniform test) the left/right taps of a row are ONE 4-byte-aligned 8-byte load.  Planes of
// an NCHW tensor start at multiples of H*W floats, hence VEC = 4, 2 or 1 by H*W mod 4.
//
// This is the round-1 kernel.  Round 2 measured fourteen restructurings against it in one process
// (tools/lab/warp_lab.py; profiles/r2/warp_lab.txt): wave-granular balanced work items, one batch of 8
// channels instead of two of 4, operand loads hoisted above the flow-dependent part, 4 / 16 channels
// per wave, 64- and 1024-thread workgroups, 4-byte taps, taps shared between the two pixels of a lane
// through lane shuffles.  None beat it by more than 5 % at one map (10.7-11.3 us for 29 MB) and all
// lost 3-15 % at 32 maps per launch (296 us for 942 MB = 3.2 TB/s): the kernel sits on a plateau that
// mapping and batching do not move.
//
// Arithmetic is the oracle's, operation for operation (orc_warp_bilinear): built with
// -ffp-contract=off so nothing fuses.
#pragma once
#include "common.h"

namespace lsfa {
namespace warp {

constexpr int kThreads = 256;
constexpr int kResMax = 4;

typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));

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

template <int VEC, bool HAS_MUL, bool HAS_ADD, bool HAS_RES>
__global__ __launch_bounds__(kThreads) void warp_kernel(
    const float* __restrict__ feat, int feat_n, const float* __restrict__ flow, int C, int H, int W,
    const float* __restrict__ mul, const float* __restrict__ add, const float* __restrict__ res,
    int res_c, const float* __restrict__ res_w, const float* __restrict__ res_b,
    float* __restrict__ out, int ch_per_block) {
  const int HW = H * W;
  const int n = blockIdx.z;
  const int c0 = blockIdx.y * ch_per_block;
  const int p0 = (blockIdx.x * kThreads + threadIdx.x) * VEC;
  if (p0 >= HW) return;

  float fx[VEC], fy[VEC];
  load_vec<VEC>(flow + ((size_t)n * 2 + 0) * HW + p0, fx);
  load_vec<VEC>(flow + ((size_t)n * 2 + 1) * HW + p0, fy);

  const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
  int off[VEC];
  bool v00[VEC], v01[VEC], v10[VEC], v11[VEC];
  float wx0[VEC], wx1[VEC], wy0[VEC], wy1[VEC];
#pragma unroll
  for (int i = 0; i < VEC; ++i) {
    const int p = p0 + i;
    const int y = p / W, x = p - y * W;
    const float gx = ((float)x + fx[i]) / half_w - 1.0f;
    const float gy = ((float)y + fy[i]) / half_h - 1.0f;
    const float x_real = (gx + 1.0f) * (float)(W - 1) / 2.0f;
    const float y_real = (gy + 1.0f) * (float)(H - 1) / 2.0f;
    const float fx0 = floorf(x_real), fy0 = floorf(y_real);
    // clamp before the int conversion so wild flows cannot overflow; clamped values
    // are outside the map either way
    const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)W);
    const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)H);
    wx0[i] = 1.0f - (x_real - fx0);
    wy0[i] = 1.0f - (y_real - fy0);
    wx1[i] = 1.0f - wx0[i];
    wy1[i] = 1.0f - wy0[i];
    const bool vx0 = (x0 >= 0 && x0 <= W - 1), vx1 = (x0 + 1 >= 0 && x0 + 1 <= W - 1);
    const bool vy0 = (y0 >= 0 && y0 <= H - 1), vy1 = (y0 + 1 >= 0 && y0 + 1 <= H - 1);
    v00[i] = vx0 && vy0; v01[i] = vx1 && vy0; v10[i] = vx0 && vy1; v11[i] = vx1 && vy1;
    off[i] = y0 * W + x0;
  }

  float rv[kResMax][VEC];
  if (HAS_RES) {
#pragma unroll
    for (int k = 0; k < kResMax; ++k)
      if (k < res_c) load_vec<VEC>(res + ((size_t)n * res_c + k) * HW + p0, rv[k]);
  }

  const float* fbase = feat + (feat_n == 1 ? (size_t)0 : (size_t)n * C * HW);
  const int c1 = min(c0 + ch_per_block, C);
  bool interior = true;
#pragma unroll
  for (int i = 0; i < VEC; ++i) interior = interior && v00[i] && v01[i] && v10[i] && v11[i];
  // wave-uniform split: a wave whose pixels all sample inside the map (nearly all of them) runs
  // the loop without any validity logic; the general loop handles map borders and escaping flows
  if (__all(interior)) {
#pragma unroll 4
    for (int c = c0; c < c1; ++c) {
      const float* plane = fbase + (size_t)c * HW;
      const size_t o = ((size_t)n * C + c) * HW + p0;
      float m[VEC], a[VEC], v[VEC];
      if (HAS_MUL) load_vec<VEC>(mul + o, m);
      if (HAS_ADD) load_vec<VEC>(add + o, a);
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float2u t = *reinterpret_cast<const float2u*>(plane + off[i]);
        const float2u b = *reinterpret_cast<const float2u*>(plane + off[i] + W);
        float r = t.x * wy0[i] * wx0[i] + t.y * wy0[i] * wx1[i] + b.x * wy1[i] * wx0[i] + b.y * wy1[i] * wx1[i];
        if (HAS_MUL) r = r * m[i];
        if (HAS_RES) {
          float q = res_w[(size_t)c * res_c] * rv[0][i];
#pragma unroll
          for (int k = 1; k < kResMax; ++k)
            if (k < res_c) q = q + res_w[(size_t)c * res_c + k] * rv[k][i];
          q = q + res_b[c];
          r = r + q;
        }
        if (HAS_ADD) r = r + a[i];
        v[i] = r;
      }
      store_vec<VEC>(out + o, v);
    }
    return;
  }
#pragma unroll 2
  for (int c = c0; c < c1; ++c) {
    const float* plane = fbase + (size_t)c * HW;
    const size_t o = ((size_t)n * C + c) * HW + p0;
    float m[VEC], a[VEC], v[VEC];
    if (HAS_MUL) load_vec<VEC>(mul + o, m);
    if (HAS_ADD) load_vec<VEC>(add + o, a);
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float tl = v00[i] ? plane[off[i]] : 0.f;
      const float tr = v01[i] ? plane[off[i] + 1] : 0.f;
      const float bl = v10[i] ? plane[off[i] + W] : 0.f;
      const float br = v11[i] ? plane[off[i] + W + 1] : 0.f;
      float r = tl * wy0[i] * wx0[i] + tr * wy0[i] * wx1[i] + bl * wy1[i] * wx0[i] + br * wy1[i] * wx1[i];
      if (HAS_MUL) r = r * m[i];
      if (HAS_RES) {
        float q = res_w[(size_t)c * res_c] * rv[0][i];
#pragma unroll
        for (int k = 1; k < kResMax; ++k)
          if (k < res_c) q = q + res_w[(size_t)c * res_c + k] * rv[k][i];
        q = q + res_b[c];
        r = r + q;
      }
      if (HAS_ADD) r = r + a[i];
      v[i] = r;
    }
    store_vec<VEC>(out + o, v);
  }
}


}  // namespace warp
}  // namespace lsfa
