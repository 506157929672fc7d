// A round-2 rewrite of the warp kernel that LOST to the round-1 kernel (tools/lab/warp_lab.py: 25 us against 11 us
// at one 38x63x1024 map, equal at 32 maps) and is kept only so the measurement can be repeated.  Not part of the library.
//
// Mapping on CDNA4 — a WAVE owns 64*VEC consecutive pixels of the flattened H*W plane and a run of CPR
// channels; work items (image, channel run, pixel tile) are enumerated linearly over waves, so every
// workgroup is full whatever H*W is (the r1 kernel tiled workgroups over pixels: at 38x63 its last
// x-block was 68 % populated and its 640 workgroups left the 256 CUs with 2 or 3 each).
//   * the per-pixel tap offsets, validity and the four weights are computed once per lane and reused
//     over the channel run (the reference materialises a (1,2,H,W) grid and recomputes per channel);
//   * the kernel is a latency chain, not a bandwidth stream, at LSFA's size (29 MB, resident in the
//     Infinity Cache): flow -> addresses -> taps -> store.  So the loads that do not depend on the flow
//     (scale map / small-net feature) are issued first, and the taps of ALL CPR channels are issued as
//     one batch before any of them is used: one memory round trip per wave instead of CPR/4;
//   * in the interior fast path (wave-uniform test) the left/right taps of a row are ONE 8-byte load
//     at a 4-byte-aligned address; with SHARE (VEC = 2) the second pixel takes its taps from the first
//     pixel's load and from the next lane's (the usual case: neighbouring pixels sample neighbouring
//     cells) and issues its own loads only on the lanes where the flow breaks that pattern.
// Planes of an NCHW tensor start at multiples of H*W floats, hence VEC = 4, 2 or 1 by H*W mod 4.
//
// Arithmetic is the oracle's, operation for operation (orc_warp_bilinear): built with
// -ffp-contract=off so nothing fuses.
#pragma once
#include "common.h"

namespace warp_r2 {

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

struct Args {
  const float* feat; int feat_n; const float* flow; int N, C, H, W;
  const float* mul; const float* add; const float* res; int res_c; const float* res_w; const float* res_b;
  float* out;
};

// work items per image-independent axis
__host__ __device__ static inline int pixel_tiles(int HW, int vec) { return (HW + 64 * vec - 1) / (64 * vec); }

// RES_C: channels of the residual input folded in by the 1x1 convolution epilogue — 0 = none, 3 = LSFA's
// (compile time: the weight loads are then straight-line scalar loads), -1 = A.res_c at run time (1..kResMax).
template <int VEC, int CPR, bool HAS_MUL, bool HAS_ADD, int RES_C, bool SHARE, int MINW>
__global__ __launch_bounds__(kThreads, (MINW > 0 ? MINW : 2)) void warp_kernel(Args A) {
  constexpr bool HAS_RES = RES_C != 0;
  const int res_c = RES_C > 0 ? RES_C : A.res_c;
  static_assert(!SHARE || VEC == 2, "tap sharing is written for two pixels per lane");
  const int H = A.H, W = A.W, C = A.C;
  const int HW = H * W;
  const int lane = threadIdx.x & 63;
  const int tiles = pixel_tiles(HW, VEC);
  const int runs = C / CPR;                       // the launcher picks a CPR that divides C: no per-channel guards, the
                                                  // loads of a run stay one straight-line batch
  // the wave index is uniform across the wave; readfirstlane tells the compiler so (everything derived from it —
  // channel run, image, the `c0 + k < C` guards — then lives in SGPRs and branches are scalar, not exec-masked)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const unsigned item = blockIdx.x * (kThreads / 64) + wave;
  if (item >= (unsigned)A.N * runs * tiles) return;
  const int tile = (int)(item % (unsigned)tiles);
  const int run = (int)((item / (unsigned)tiles) % (unsigned)runs);
  const int n = (int)(item / ((unsigned)tiles * runs));
  const int c0 = run * CPR;
  const int p_raw = (tile * 64 + lane) * VEC;
  const bool active = p_raw < HW;                 // HW % VEC == 0: a lane is wholly inside or outside
  const int p0 = active ? p_raw : HW - VEC;       // parked lanes redo the last pixels and store nothing

  // ---- loads that do not depend on the flow go first -----------------------------------------
  float fx[VEC], fy[VEC];
  load_vec<VEC>(A.flow + ((size_t)n * 2 + 0) * HW + p0, fx);
  load_vec<VEC>(A.flow + ((size_t)n * 2 + 1) * HW + p0, fy);
  const size_t o0 = ((size_t)n * C + c0) * HW + p0;
  float m[CPR][VEC], ad[CPR][VEC];
#pragma unroll
  for (int k = 0; k < CPR; ++k) {
    if (HAS_MUL) load_vec<VEC>(A.mul + o0 + (size_t)k * HW, m[k]);
    if (HAS_ADD) load_vec<VEC>(A.add + o0 + (size_t)k * HW, ad[k]);
  }
  float rv[kResMax][VEC];
  float rw[CPR][kResMax], rb[CPR];       // rnet_conv0 weights of this run: wave-uniform, read before any store
  if (HAS_RES) {
#pragma unroll
    for (int k = 0; k < kResMax; ++k)
      if (k < res_c) load_vec<VEC>(A.res + ((size_t)n * res_c + k) * HW + p0, rv[k]);
#pragma unroll
    for (int k = 0; k < CPR; ++k) {
#pragma unroll
      for (int j = 0; j < kResMax; ++j)
        if (j < res_c) rw[k][j] = A.res_w[(size_t)(c0 + k) * res_c + j];
      rb[k] = A.res_b[c0 + k];
    }
  }

  // ---- taps: offsets, validity, weights (GridGenerator + BilinearSampler arithmetic) -----------
  const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
  int off[VEC];
  bool v00[VEC], v01[VEC], v10[VEC], v11[VEC];
  float wx0[VEC], wx1[VEC], wy0[VEC], wy1[VEC];
  bool interior = true;
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
    interior = interior && v00[i] && v01[i] && v10[i] && v11[i];
  }
  const float* fbase = A.feat + (A.feat_n == 1 ? (size_t)0 : (size_t)n * C * HW) + (size_t)c0 * HW;
  float* obase = A.out + o0;

  auto epilogue = [&](int k, int i, float r) -> float {
    if (HAS_MUL) r = r * m[k][i];
    if (HAS_RES) {
      float q = rw[k][0] * rv[0][i];
#pragma unroll
      for (int j = 1; j < kResMax; ++j)
        if (j < res_c) q = q + rw[k][j] * rv[j][i];
      q = q + rb[k];
      r = r + q;
    }
    if (HAS_ADD) r = r + ad[k][i];
    return r;
  };

  // wave-uniform split: a wave whose pixels all sample inside the map (nearly all of them) issues every
  // tap of its channel run as one batch of 8-byte loads, without any validity logic
  if (__all(interior)) {
    float2u t[CPR][VEC], b[CPR][VEC];
    if (SHARE) {
      // second pixel: its left taps are the first pixel's right taps when it samples the next cell, and
      // its right taps are the next lane's left taps when that lane starts two cells further
      const int next_off = __shfl_down(off[0], 1, 64);
      const bool chained = off[1] == off[0] + 1 && next_off == off[0] + 2 && lane < 63;
#pragma unroll
      for (int k = 0; k < CPR; ++k) {
        const float* plane = fbase + (size_t)k * HW;
        t[k][0] = *reinterpret_cast<const float2u*>(plane + off[0]);
        b[k][0] = *reinterpret_cast<const float2u*>(plane + off[0] + W);
      }
      if (!chained) {
#pragma unroll
        for (int k = 0; k < CPR; ++k) {
          const float* plane = fbase + (size_t)k * HW;
          t[k][1] = *reinterpret_cast<const float2u*>(plane + off[1]);
          b[k][1] = *reinterpret_cast<const float2u*>(plane + off[1] + W);
        }
      }
#pragma unroll
      for (int k = 0; k < CPR; ++k) {
        const float tn = __shfl_down(t[k][0].x, 1, 64), bn = __shfl_down(b[k][0].x, 1, 64);
        if (chained) {
          t[k][1].x = t[k][0].y; t[k][1].y = tn;
          b[k][1].x = b[k][0].y; b[k][1].y = bn;
        }
      }
    } else {
#pragma unroll
      for (int k = 0; k < CPR; ++k) {
        const float* plane = fbase + (size_t)k * HW;
#pragma unroll
        for (int i = 0; i < VEC; ++i) {
          t[k][i] = *reinterpret_cast<const float2u*>(plane + off[i]);
          b[k][i] = *reinterpret_cast<const float2u*>(plane + off[i] + W);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < CPR; ++k) {
      float v[VEC];
#pragma unroll
      for (int i = 0; i < VEC; ++i) {
        const float r = t[k][i].x * wy0[i] * wx0[i] + t[k][i].y * wy0[i] * wx1[i] + b[k][i].x * wy1[i] * wx0[i] +
                        b[k][i].y * wy1[i] * wx1[i];
        v[i] = epilogue(k, i, r);
      }
      if (active) store_vec<VEC>(obase + (size_t)k * HW, v);
    }
    return;
  }
  // general path: map borders and flows that leave the map; each tap individually zero when outside
#pragma unroll 2
  for (int k = 0; k < CPR; ++k) {
    const float* plane = fbase + (size_t)k * HW;
    float v[VEC];
#pragma unroll
    for (int i = 0; i < VEC; ++i) {
      const float tl = v00[i] ? plane[off[i]] : 0.f;
      const float tr = v01[i] ? plane[off[i] + 1] : 0.f;
      const float bl = v10[i] ? plane[off[i] + W] : 0.f;
      const float br = v11[i] ? plane[off[i] + W + 1] : 0.f;
      const float r = tl * wy0[i] * wx0[i] + tr * wy0[i] * wx1[i] + bl * wy1[i] * wx0[i] + br * wy1[i] * wx1[i];
      v[i] = epilogue(k, i, r);
    }
    if (active) store_vec<VEC>(obase + (size_t)k * HW, v);
  }
}

// a.C % CPR == 0 (the kernel has no per-channel guard)
template <int VEC, int CPR, bool SHARE, int MINW>
static void launch(hipStream_t s, const Args& a) {
  const long items = (long)a.N * (a.C / CPR) * pixel_tiles(a.H * a.W, VEC);   // < 2^31, checked by the caller
  const dim3 grid((unsigned)((items + kThreads / 64 - 1) / (kThreads / 64)));
  const bool has_mul = a.mul != nullptr, has_add = a.add != nullptr;
  const int res_kind = a.res == nullptr ? 0 : (a.res_c == 3 ? 3 : -1);
#define LSFA_WARP_CASE(M, AD, R)                                                                             \
  if (has_mul == M && has_add == AD && res_kind == R) {                                                      \
    hipLaunchKernelGGL((warp_kernel<VEC, CPR, M, AD, R, SHARE, MINW>), grid, dim3(kThreads), 0, s, a);       \
    return;                                                                                                  \
  }
  LSFA_WARP_CASE(true, false, 0)
  LSFA_WARP_CASE(false, true, 3)
#undef LSFA_WARP_CASE
}

}  // namespace warp_r2
