// Generic warp kernel for A/B experiments (tools/lab/warp_lab.py): the arithmetic of lsfa_warp_bilinear with the
// work mapping, the batch of channels whose loads are issued together, the position of the operand loads and the
// tap load width as template parameters.  VEC = 2 (H*W even), interior fast path + general path as in the product.
#pragma once
#include "common.h"

namespace warp_lab {

typedef float float2u __attribute__((ext_vector_type(2), aligned(4)));

struct Args {
  const float* feat; int feat_n; const float* flow; int N, C, H, W;
  const float* mul; const float* add; const float* res; const float* res_w; const float* res_b; float* out;
};

// FLAT: wave-granular work items (image, channel run, 128-pixel tile) vs workgroup tiles (blockIdx.x pixels,
// blockIdx.y channel runs, blockIdx.z image).  RUN: channels per wave / thread.  U: channels whose loads are issued
// as one batch.  HOIST: operand (mul / add) loads of the whole RUN issued before the flow is known.
// DWORD: taps as four 4-byte loads instead of two 4-byte-aligned 8-byte loads.  THREADS: workgroup size.
template <int RUN, int U, bool FLAT, bool HOIST, bool DWORD, int THREADS, bool HAS_MUL, bool HAS_ADD, bool HAS_RES>
__global__ __launch_bounds__(THREADS) void kernel(Args A) {
  constexpr int VEC = 2;
  const int H = A.H, W = A.W, C = A.C, HW = H * W;
  const int lane = threadIdx.x & 63;
  int n, c0, p_raw;
  if (FLAT) {
    const int tiles = (HW + 127) / 128, runs = C / RUN;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const unsigned item = blockIdx.x * (THREADS / 64) + wave;
    if (item >= (unsigned)A.N * runs * tiles) return;
    const int tile = (int)(item % (unsigned)tiles);
    c0 = (int)((item / (unsigned)tiles) % (unsigned)runs) * RUN;
    n = (int)(item / ((unsigned)tiles * runs));
    p_raw = (tile * 64 + lane) * VEC;
  } else {
    n = blockIdx.z;
    c0 = blockIdx.y * RUN;
    p_raw = (blockIdx.x * THREADS + threadIdx.x) * VEC;
  }
  const bool active = p_raw < HW;
  const int p0 = active ? p_raw : HW - VEC;
  const float2 fx = *reinterpret_cast<const float2*>(A.flow + ((size_t)n * 2 + 0) * HW + p0);
  const float2 fy = *reinterpret_cast<const float2*>(A.flow + ((size_t)n * 2 + 1) * HW + p0);
  const size_t o0 = ((size_t)n * C + c0) * HW + p0;
  float2 hm[HOIST ? RUN : 1], ha[HOIST ? RUN : 1];
  if (HOIST) {
#pragma unroll
    for (int k = 0; k < RUN; ++k) {
      if (HAS_MUL) hm[k] = *reinterpret_cast<const float2*>(A.mul + o0 + (size_t)k * HW);
      if (HAS_ADD) ha[k] = *reinterpret_cast<const float2*>(A.add + o0 + (size_t)k * HW);
    }
  }
  float2 rv[3];
  if (HAS_RES) {
#pragma unroll
    for (int j = 0; j < 3; ++j) rv[j] = *reinterpret_cast<const float2*>(A.res + ((size_t)n * 3 + j) * HW + p0);
  }
  const float fxs[2] = {fx.x, fx.y}, fys[2] = {fy.x, fy.y};
  const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
  int off[2];
  bool v00[2], v01[2], v10[2], v11[2];
  float wx0[2], wx1[2], wy0[2], wy1[2];
  bool interior = true;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = p0 + i;
    const int y = p / W, x = p - y * W;
    const float gx = ((float)x + fxs[i]) / half_w - 1.0f;
    const float gy = ((float)y + fys[i]) / half_h - 1.0f;
    const float x_real = (gx + 1.0f) * (float)(W - 1) / 2.0f;
    const float y_real = (gy + 1.0f) * (float)(H - 1) / 2.0f;
    const float fx0 = floorf(x_real), fy0 = floorf(y_real);
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
  const float rvs[3][2] = {{rv[0].x, rv[0].y}, {rv[1].x, rv[1].y}, {rv[2].x, rv[2].y}};
  if (__all(interior)) {
#pragma unroll 1
    for (int kb = 0; kb < RUN; kb += U) {
      float tl[U][2], tr[U][2], bl[U][2], br[U][2];
      float2 m[U], ad[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const float* plane = fbase + (size_t)(kb + u) * HW;
        if (!HOIST) {
          if (HAS_MUL) m[u] = *reinterpret_cast<const float2*>(A.mul + o0 + (size_t)(kb + u) * HW);
          if (HAS_ADD) ad[u] = *reinterpret_cast<const float2*>(A.add + o0 + (size_t)(kb + u) * HW);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (DWORD) {
            tl[u][i] = plane[off[i]]; tr[u][i] = plane[off[i] + 1];
            bl[u][i] = plane[off[i] + W]; br[u][i] = plane[off[i] + W + 1];
          } else {
            const float2u t = *reinterpret_cast<const float2u*>(plane + off[i]);
            const float2u b = *reinterpret_cast<const float2u*>(plane + off[i] + W);
            tl[u][i] = t.x; tr[u][i] = t.y; bl[u][i] = b.x; br[u][i] = b.y;
          }
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        float v[2];
        const float ms[2] = {HOIST ? hm[(kb + u) % (HOIST ? RUN : 1)].x : m[u].x, HOIST ? hm[(kb + u) % (HOIST ? RUN : 1)].y : m[u].y};
        const float as[2] = {HOIST ? ha[(kb + u) % (HOIST ? RUN : 1)].x : ad[u].x, HOIST ? ha[(kb + u) % (HOIST ? RUN : 1)].y : ad[u].y};
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          float r = tl[u][i] * wy0[i] * wx0[i] + tr[u][i] * wy0[i] * wx1[i] + bl[u][i] * wy1[i] * wx0[i] + br[u][i] * wy1[i] * wx1[i];
          if (HAS_MUL) r = r * ms[i];
          if (HAS_RES) {
            const int c = c0 + kb + u;
            float q = A.res_w[(size_t)c * 3] * rvs[0][i];
            q = q + A.res_w[(size_t)c * 3 + 1] * rvs[1][i];
            q = q + A.res_w[(size_t)c * 3 + 2] * rvs[2][i];
            q = q + A.res_b[c];
            r = r + q;
          }
          if (HAS_ADD) r = r + as[i];
          v[i] = r;
        }
        if (active) *reinterpret_cast<float2*>(obase + (size_t)(kb + u) * HW) = make_float2(v[0], v[1]);
      }
    }
    return;
  }
  for (int k = 0; k < RUN; ++k) {
    const float* plane = fbase + (size_t)k * HW;
    float v[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float a = v00[i] ? plane[off[i]] : 0.f, b = v01[i] ? plane[off[i] + 1] : 0.f;
      const float c = v10[i] ? plane[off[i] + W] : 0.f, d = v11[i] ? plane[off[i] + W + 1] : 0.f;
      float r = a * wy0[i] * wx0[i] + b * wy0[i] * wx1[i] + c * wy1[i] * wx0[i] + d * wy1[i] * wx1[i];
      if (HAS_MUL) r = r * A.mul[o0 + (size_t)k * HW + i];
      if (HAS_RES) {
        const int cc = c0 + k;
        float q = A.res_w[(size_t)cc * 3] * rvs[0][i];
        q = q + A.res_w[(size_t)cc * 3 + 1] * rvs[1][i];
        q = q + A.res_w[(size_t)cc * 3 + 2] * rvs[2][i];
        q = q + A.res_b[cc];
        r = r + q;
      }
      if (HAS_ADD) r = r + A.add[o0 + (size_t)k * HW + i];
      v[i] = r;
    }
    if (active) *reinterpret_cast<float2*>(obase + (size_t)k * HW) = make_float2(v[0], v[1]);
  }
}

template <int RUN, int U, bool FLAT, bool HOIST, bool DWORD, int THREADS>
static int launch(hipStream_t s, const Args& a) {
  const int HW = a.H * a.W;
  dim3 grid;
  if (FLAT) {
    const long items = (long)a.N * (a.C / RUN) * ((HW + 127) / 128);
    grid = dim3((unsigned)((items + THREADS / 64 - 1) / (THREADS / 64)));
  } else {
    grid = dim3((HW + THREADS * 2 - 1) / (THREADS * 2), a.C / RUN, a.N);
  }
  if (a.mul && !a.add && !a.res)
    hipLaunchKernelGGL((kernel<RUN, U, FLAT, HOIST, DWORD, THREADS, true, false, false>), grid, dim3(THREADS), 0, s, a);
  else if (!a.mul && a.add && a.res)
    hipLaunchKernelGGL((kernel<RUN, U, FLAT, HOIST, DWORD, THREADS, false, true, true>), grid, dim3(THREADS), 0, s, a);
  else
    return -2;
  return 0;
}

}  // namespace warp_lab
