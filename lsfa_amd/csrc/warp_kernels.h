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

  const float* fbase = feat + (size_t)(n % feat_n) * C * HW;       // map n samples feature n mod feat_n (1: one feature for all; N: its own)
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


// ---- round 3: the source plane staged in LDS --------------------------------------------------------------------------------
// warp_kernel above and the sixteen restructurings round 2 measured against it all sit at 3.0-3.2 TB/s at 32 maps per launch: per
// output value they issue two 8-byte gathers next to one coalesced operand load and one store, and the gathers, not HBM, set the
// pace.  This form takes them off the vector-memory path.  A workgroup owns (image n, a run of `cg` channels), computes the taps
// of ALL H*W pixels once (a lane keeps NPAIR pixel pairs: LDS index, four weights, four validity bits, the residual's values), then
// walks the channels: the plane of channel c - and the same plane of every streamed operand (mul, add) - is copied global -> LDS by
// the DMA path (`global_load_lds_dwordx4`: whole 16-byte chunks of the aligned range that covers the plane, no register stage) into
// a ring of STAGES slots; the four taps of a pixel are two ds_read2_b32, the operands one each; the store is 8-byte coalesced.
// HBM sees full-line streams only.  942 MB (32 maps, x scale map) in 176 us = 5.35 TB/s against 296 us = 3.2 TB/s; one map (29 MB,
// cache-resident) 8.4 us against 11.2 (tools/lab/warp_lab.py, profiles/r3/warp_lab_lds.txt).
// Arithmetic = warp_kernel's general path, operation for operation (a tap outside the map contributes 0 * w): bit-identical.
//
// No load of the loop returns into registers, so every wait in it is written by hand:
//   iteration k = [s_waitcnt vmcnt(K): the own DMAs of plane c have landed] [s_barrier: everybody's have] [taps and operands from
//   LDS, arithmetic, store] [issue the DMAs of plane c + STAGES - 1 into the slot plane c - 1 used: every wave is past the barrier,
//   hence past its reads of it].  K = (planes issued after c) x (DMA instructions per plane): loads return in order among loads,
//   so "at most K outstanding" means plane c has landed whatever the stores issued in between have done (their count only makes
//   the wait stricter); the prefetch is issued LAST so that the youngest entries of the queue are the ones allowed to stay out.
//   Every lane executes every store (a lane past the plane recomputed the last pair: the same bytes twice) so that all waves count
//   the same instructions.
//   LDS reads are inline assembly: hipcc cannot prove that a data-dependent tap address stays clear of the slots being filled and
//   drains the DMAs (`s_waitcnt vmcnt(0)`) before every ds_read it can see.  They land asynchronously: lds_wait(), then pin() every
//   value before its first use.
template <int K> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_float;

__device__ __forceinline__ f32x2 lds_read2(uint32_t byte_addr) {      // the floats at byte_addr and byte_addr + 4 (4-byte aligned)
  f32x2 v;
  asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(byte_addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pin(f32x2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

struct StagedArgs {
  const float* feat; int feat_n; const float* flow; int N, C, H, W;
  const float* mul; const float* add; const float* res; int res_c; const float* res_w; const float* res_b; float* out;
  int cg;       // channels per workgroup (divides C)
  int guard;    // floats in front of and behind a slot's plane regions: >= 2W + 6, a multiple of 4
};

// grid (N * C / cg); block THREADS; dynamic LDS STAGES * (2 * guard + (1 + HAS_MUL + HAS_ADD) * THREADS * NDMA * 4) floats.
// Needs: H*W even, feat / mul / add 16-byte aligned, (C*H*W) % 4 == 0, H*W <= 2 * THREADS * NPAIR,
//        3 + H*W + W + 2 <= THREADS * NDMA * 4 + guard (warp.hip checks all of it and falls back to warp_kernel)
// RC: the residual's channel count when it is known at compile time (3: rnet_conv0 of the frame path), 0 = A.res_c at run time
template <int THREADS, int NPAIR, int NDMA, int STAGES, bool HAS_MUL, bool HAS_ADD, bool HAS_RES, int RC = 0>
__global__ __launch_bounds__(THREADS) void warp_staged_kernel(StagedArgs A) {
  static_assert(STAGES == 3 || STAGES == 4, "the counted waits below cover one or two planes in flight");
  constexpr int kRes = RC > 0 ? RC : kResMax;              // residual values a lane keeps per pixel
  const int res_c = RC > 0 ? RC : A.res_c;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int kOps = (HAS_MUL ? 1 : 0) + (HAS_ADD ? 1 : 0);
  constexpr int kRegion = THREADS * NDMA * 4;            // floats one DMA pass of the workgroup covers
  constexpr int kPerPlane = (1 + kOps) * NDMA;           // DMA instructions a wave issues per channel
  const int guard = A.guard, cg = A.cg;
  const int stage_floats = 2 * guard + (1 + kOps) * kRegion;      // [guard][feat plane][operand plane(s)][guard]
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_float*)lds;
  const int H = A.H, W = A.W, C = A.C, HW = H * W;
  const int runs = C / cg;
  const int n = blockIdx.x / runs, c0 = (blockIdx.x - n * runs) * cg;
  const int tid = threadIdx.x;

  // ---- taps of this thread's pixel pairs (warp_kernel's arithmetic) ----------------------------------------------------------
  int idx[NPAIR][2];            // float index of the top-left tap relative to the plane's first float (negative: in the guard)
  unsigned vb[NPAIR];           // validity bits: pixel j -> bits 4j .. 4j+3 = v00 v01 v10 v11
  float wx0[NPAIR][2], wx1[NPAIR][2], wy0[NPAIR][2], wy1[NPAIR][2];
  float rv[HAS_RES ? NPAIR : 1][kRes][2];
  const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
#pragma unroll
  for (int i = 0; i < NPAIR; ++i) {
    const int praw = (tid + THREADS * i) * 2;
    const int p0 = praw < HW ? praw : HW - 2;
    const float2 fx = *reinterpret_cast<const float2*>(A.flow + ((size_t)n * 2 + 0) * HW + p0);
    const float2 fy = *reinterpret_cast<const float2*>(A.flow + ((size_t)n * 2 + 1) * HW + p0);
    const float fxs[2] = {fx.x, fx.y}, fys[2] = {fy.x, fy.y};
    if (HAS_RES) {
#pragma unroll
      for (int k = 0; k < kRes; ++k) {
        float2 r = make_float2(0.f, 0.f);
        if (k < res_c) r = *reinterpret_cast<const float2*>(A.res + ((size_t)n * res_c + k) * HW + p0);
        rv[i][k][0] = r.x; rv[i][k][1] = r.y;
      }
    }
    unsigned bits = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = p0 + j;
      const int y = p / W, x = p - y * W;
      const float gx = ((float)x + fxs[j]) / half_w - 1.0f;
      const float gy = ((float)y + fys[j]) / half_h - 1.0f;
      const float x_real = (gx + 1.0f) * (float)(W - 1) / 2.0f;
      const float y_real = (gy + 1.0f) * (float)(H - 1) / 2.0f;
      const float fx0 = floorf(x_real), fy0 = floorf(y_real);
      const int x0 = (int)fminf(fmaxf(fx0, -2.0f), (float)W);
      const int y0 = (int)fminf(fmaxf(fy0, -2.0f), (float)H);
      wx0[i][j] = 1.0f - (x_real - fx0);
      wy0[i][j] = 1.0f - (y_real - fy0);
      wx1[i][j] = 1.0f - wx0[i][j];
      wy1[i][j] = 1.0f - wy0[i][j];
      const bool vx0 = (x0 >= 0 && x0 <= W - 1), vx1 = (x0 + 1 >= 0 && x0 + 1 <= W - 1);
      const bool vy0 = (y0 >= 0 && y0 <= H - 1), vy1 = (y0 + 1 >= 0 && y0 + 1 <= H - 1);
      bits |= ((unsigned)(vx0 && vy0) | ((unsigned)(vx1 && vy0) << 1) | ((unsigned)(vx0 && vy1) << 2) | ((unsigned)(vx1 && vy1) << 3)) << (4 * j);
      idx[i][j] = y0 * W + x0;     // in [-2W - 2, HW + W]: the four reads stay inside [plane - guard, plane + HW + W + 2)
    }
    vb[i] = bits;
  }
  if (HAS_RES) {       // the residual's loads have returned before the first DMA is issued: only DMAs and stores are counted below
#pragma unroll
    for (int i = 0; i < NPAIR; ++i)
#pragma unroll
      for (int k = 0; k < kRes; ++k) { pin(rv[i][k][0]); pin(rv[i][k][1]); }
  }

  // ---- the ring ------------------------------------------------------------------------------------------------------------------
  const float* fbase = A.feat + (size_t)(n % A.feat_n) * C * HW;       // map n samples feature n mod feat_n
  const size_t nbase = (size_t)n * C * HW;
  // one plane by DMA: lane -> LDS chunk is fixed by the hardware (base + 16 * lane); lanes past the plane re-read its last chunk
  // into the slack behind it
  auto copy_plane = [&](const float* plane, float* dst) {
    const uintptr_t a = reinterpret_cast<uintptr_t>(plane);
    const float* src0 = reinterpret_cast<const float*>(a & ~(uintptr_t)15);
    const int nchunks = (int)(((a & 15) >> 2) + HW + 3) >> 2;
#pragma unroll
    for (int k = 0; k < NDMA; ++k) {
      const int chunk = k * THREADS + tid;
      const int cc = chunk < nchunks ? chunk : nchunks - 1;
      __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(src0 + (size_t)cc * 4),
                                       reinterpret_cast<uint4*>(dst) + k * THREADS + (tid & ~63), 16, 0, 0);
    }
  };
  auto issue = [&](int c, int slot) {
    float* st = lds + (size_t)slot * stage_floats + guard;
    copy_plane(fbase + (size_t)c * HW, st);
    if (HAS_MUL) copy_plane(A.mul + nbase + (size_t)c * HW, st + kRegion);
    if (HAS_ADD) copy_plane(A.add + nbase + (size_t)c * HW, st + (HAS_MUL ? 2 : 1) * kRegion);
  };

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < cg) issue(c0 + s, s);
  int slot = 0;
  for (int k = 0; k < cg; ++k) {
    const int c = c0 + k;
    const int later = min(STAGES - 2, cg - 1 - k);       // planes issued after c
    if (later >= 2) wait_vm<2 * kPerPlane>();
    else if (later == 1) wait_vm<kPerPlane>();
    else wait_vm<0>();
    __builtin_amdgcn_s_barrier();
    const size_t obase = nbase + (size_t)c * HW;
    const uint32_t st_bytes = lds_base + (uint32_t)(slot * stage_floats + guard) * 4u;
    const uint32_t f0 = st_bytes + (uint32_t)(reinterpret_cast<uintptr_t>(fbase + (size_t)c * HW) & 15);      // the plane's first float
    const uint32_t m0 = HAS_MUL ? st_bytes + (uint32_t)kRegion * 4u + (uint32_t)(reinterpret_cast<uintptr_t>(A.mul + obase) & 15) : 0u;
    const uint32_t a0 = HAS_ADD ? st_bytes + (uint32_t)((HAS_MUL ? 2 : 1) * kRegion) * 4u + (uint32_t)(reinterpret_cast<uintptr_t>(A.add + obase) & 15) : 0u;
    f32x2 top[NPAIR][2], bot[NPAIR][2], mv[HAS_MUL ? NPAIR : 1], av[HAS_ADD ? NPAIR : 1];
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      const int praw = (tid + THREADS * i) * 2;
      const int pl = praw < HW ? praw : HW - 2;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const uint32_t t = f0 + (uint32_t)(idx[i][j] * 4);
        top[i][j] = lds_read2(t);
        bot[i][j] = lds_read2(t + (uint32_t)W * 4u);
      }
      if (HAS_MUL) mv[i] = lds_read2(m0 + (uint32_t)pl * 4u);
      if (HAS_ADD) av[i] = lds_read2(a0 + (uint32_t)pl * 4u);
    }
    float rw[kRes], rb = 0.f;
    if (HAS_RES) {
#pragma unroll
      for (int q = 0; q < kRes; ++q) rw[q] = q < res_c ? A.res_w[(size_t)c * res_c + q] : 0.f;
      rb = A.res_b[c];
    }
    lds_wait();
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      pin(top[i][0]); pin(top[i][1]); pin(bot[i][0]); pin(bot[i][1]);
      if (HAS_MUL) pin(mv[i]);
      if (HAS_ADD) pin(av[i]);
    }
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) {
      const int praw = (tid + THREADS * i) * 2;
      float v[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const unsigned b = vb[i] >> (4 * j);
        const float tl = (b & 1u) ? top[i][j].x : 0.f;
        const float tr = (b & 2u) ? top[i][j].y : 0.f;
        const float bl = (b & 4u) ? bot[i][j].x : 0.f;
        const float br = (b & 8u) ? bot[i][j].y : 0.f;
        float r = tl * wy0[i][j] * wx0[i][j] + tr * wy0[i][j] * wx1[i][j] + bl * wy1[i][j] * wx0[i][j] + br * wy1[i][j] * wx1[i][j];
        if (HAS_MUL) r = r * (j == 0 ? mv[i].x : mv[i].y);
        if (HAS_RES) {
          float q = rw[0] * rv[i][0][j];
#pragma unroll
          for (int kk = 1; kk < kRes; ++kk)
            if (kk < res_c) q = q + rw[kk] * rv[i][kk][j];
          q = q + rb;
          r = r + q;
        }
        if (HAS_ADD) r = r + (j == 0 ? av[i].x : av[i].y);
        v[j] = r;
      }
      const int pl = praw < HW ? praw : HW - 2;
      *reinterpret_cast<float2*>(A.out + obase + pl) = make_float2(v[0], v[1]);
    }
    if (k + STAGES - 1 < cg) issue(c + STAGES - 1, (slot + STAGES - 1) % STAGES);
    slot = slot + 1 == STAGES ? 0 : slot + 1;
  }
}


}  // namespace warp
}  // namespace lsfa
