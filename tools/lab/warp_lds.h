// Round-3 experiment: the warp with the source plane staged in LDS ("tap table" form).
//
// Every gather kernel of warp_lab.py sits at 3.0-3.2 TB/s at 32 maps per launch whatever its mapping: per output value they issue
// two 8-byte gathers next to one coalesced operand load and one store.  This form takes the gathers off the vector-memory path:
// a workgroup owns (image n, a run of `cg` channels), computes the taps of ALL H*W pixels once (lane -> pixel pairs, kept in
// registers: LDS index, four weights, four validity bits), then walks the channels: the plane of channel c is copied global ->
// LDS by the DMA path (`global_load_lds_dwordx4`, whole 16-byte chunks of the aligned range that covers the plane, no register
// stage) into a ring of STAGES slots, STAGES - 1 planes ahead; the four taps of a pixel are LDS reads (two ds_read2_b32); the
// store stays 8-byte coalesced.  HBM sees nothing but full-line streams.
// Arithmetic = lsfa_warp_bilinear's general path, operation for operation (a tap outside the map contributes 0 * w).
//
// The streamed operand (mul or add) takes the same road (its plane is a second region of the slot), so the loop issues no
// register-returning load at all and every wait in it is written by hand.
//
// Ring discipline: iteration c = [wait for the own DMAs of plane c] [barrier] [taps from LDS, arithmetic, store] [issue the DMAs
// of plane c + STAGES - 1 into the slot plane c - 1 used: every wave is past its reads of it].  The wait is `s_waitcnt vmcnt(K)`
// with K = (STAGES - 2) * (DMA instructions per plane): loads return in order among loads, so "at most K outstanding" means
// plane c has landed whatever the stores issued in between have done (their count only makes the wait stricter).
#pragma once
#include "common.h"
#include "warp_variants.h"

namespace warp_lds {

template <int K> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory"); }

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) float lds_float;

// LDS reads as inline assembly: hipcc cannot prove that a data-dependent tap address stays clear of the slots the DMAs in flight
// are filling, and drains them (`s_waitcnt vmcnt(0)`) before every ds_read it can see.  The reads land asynchronously: lds_wait()
// then pin() on every value before its first use.
__device__ __forceinline__ f32x2 lds_read2(uint32_t byte_addr) {      // the floats at byte_addr and byte_addr + 4 (4-byte aligned)
  f32x2 v;
  asm volatile("ds_read2_b32 %0, %1 offset1:1" : "=v"(v) : "v"(byte_addr) : "memory");
  return v;
}
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void pin(f32x2& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float& v) { asm volatile("" : "+v"(v)); }

template <int THREADS, int NPAIR, int NDMA, int STAGES, bool HAS_MUL, bool HAS_ADD, bool HAS_RES>
__global__ __launch_bounds__(THREADS) void kernel(warp_lab::Args A, int cg, int guard) {
  static_assert(HAS_MUL != HAS_ADD, "one streamed operand (the two modes the frame path uses)");
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int kRegion = THREADS * NDMA * 4;            // floats one DMA pass of the workgroup covers
  const int stage_floats = 2 * guard + 2 * kRegion;      // [guard][feat plane region][guard][operand plane region]
  const uint32_t lds_base = (uint32_t)(uintptr_t)(lds_float*)lds;
  const int H = A.H, W = A.W, C = A.C, HW = H * W;
  const int runs = C / cg;
  const int n = blockIdx.x / runs, c0 = (blockIdx.x - n * runs) * cg;
  const int tid = threadIdx.x;

  // ---- taps of this thread's pixel pairs -------------------------------------------------------------------------
  int idx[NPAIR][2];            // float index of the top-left tap relative to the plane's first float (may be negative: guard)
  unsigned vb[NPAIR];           // validity bits: pixel j -> bits 4j .. 4j+3 = v00 v01 v10 v11
  float wx0[NPAIR][2], wx1[NPAIR][2], wy0[NPAIR][2], wy1[NPAIR][2];
  float rv[HAS_RES ? NPAIR : 1][3][2];
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
      for (int k = 0; k < 3; ++k) {
        const float2 r = *reinterpret_cast<const float2*>(A.res + ((size_t)n * 3 + k) * HW + p0);
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
      idx[i][j] = y0 * W + x0;     // in [-2W - 2, HW + W]: the reads stay inside [plane - guard, plane + HW + guard)
    }
    vb[i] = bits;
  }

  if (HAS_RES) {       // the residual's loads have returned before the first DMA is issued: nothing but DMAs and stores is counted below
#pragma unroll
    for (int i = 0; i < NPAIR; ++i)
#pragma unroll
      for (int k = 0; k < 3; ++k) { pin(rv[i][k][0]); pin(rv[i][k][1]); }
  }

  // ---- the ring ------------------------------------------------------------------------------------------------------
  const float* fbase = A.feat + (A.feat_n == 1 ? (size_t)0 : (size_t)n * C * HW);
  const float* obase_p = (HAS_MUL ? A.mul : A.add) + (size_t)n * C * HW;
  // one plane by DMA: whole 16-byte chunks of the aligned range that covers it; lane -> LDS chunk is fixed by the hardware
  // (base + 16 * lane), lanes past the plane re-read its last chunk into the slack behind it
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
    float* st = lds + (size_t)slot * stage_floats;
    copy_plane(fbase + (size_t)c * HW, st + guard);
    copy_plane(obase_p + (size_t)c * HW, st + 2 * guard + kRegion);
  };
  constexpr int kPerPlane = 2 * NDMA;                    // DMA instructions a wave issues per channel

#pragma unroll
  for (int s = 0; s < STAGES - 1; ++s)
    if (s < cg) issue(c0 + s, s);
  int slot = 0;
  for (int k = 0; k < cg; ++k) {
    const int c = c0 + k;
    // plane c has landed once at most the DMAs issued after it are outstanding
    const int later = min(STAGES - 2, cg - 1 - k);
    if (STAGES >= 3 && later >= 1) {
      if (STAGES == 3 || later == 1) wait_vm<kPerPlane>(); else wait_vm<2 * kPerPlane>();
    } else {
      wait_vm<0>();
    }
    __builtin_amdgcn_s_barrier();
    const uintptr_t fa = reinterpret_cast<uintptr_t>(fbase + (size_t)c * HW), oa = reinterpret_cast<uintptr_t>(obase_p + (size_t)c * HW);
    const uint32_t st_bytes = lds_base + (uint32_t)(slot * stage_floats) * 4u;
    const uint32_t f0 = st_bytes + (uint32_t)guard * 4u + (uint32_t)(fa & 15);                        // the plane's first float
    const uint32_t o0 = st_bytes + (uint32_t)(2 * guard + kRegion) * 4u + (uint32_t)(oa & 15);
    f32x2 top[NPAIR][2], bot[NPAIR][2], opv[NPAIR];
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
      opv[i] = lds_read2(o0 + (uint32_t)pl * 4u);
    }
    float rw[3], rb = 0.f;
    if (HAS_RES) { rw[0] = A.res_w[(size_t)c * 3]; rw[1] = A.res_w[(size_t)c * 3 + 1]; rw[2] = A.res_w[(size_t)c * 3 + 2]; rb = A.res_b[c]; }
    const size_t obase = ((size_t)n * C + c) * HW;
    lds_wait();
#pragma unroll
    for (int i = 0; i < NPAIR; ++i) { pin(top[i][0]); pin(top[i][1]); pin(bot[i][0]); pin(bot[i][1]); pin(opv[i]); }
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
        const float op = j == 0 ? opv[i].x : opv[i].y;
        if (HAS_MUL) r = r * op;
        if (HAS_RES) {
          float q = rw[0] * rv[i][0][j];
          q = q + rw[1] * rv[i][1][j];
          q = q + rw[2] * rv[i][2][j];
          q = q + rb;
          r = r + q;
        }
        if (HAS_ADD) r = r + op;
        v[j] = r;
      }
      // every lane stores (a lane past the plane recomputed the last pair: the same bytes twice), so that a wave issues the
      // same number of vector-memory instructions whatever its position: the counted waits above rely on it
      const int pl = praw < HW ? praw : HW - 2;
      *reinterpret_cast<float2*>(A.out + obase + pl) = make_float2(v[0], v[1]);
    }
    // the next plane's DMA last: the youngest instructions in the queue are then prefetches, which the counted wait at the top
    // of the next iteration leaves in flight (behind stores it would have to wait for the stores' acknowledgements first).
    // Its slot held plane c - 1: every wave is past the barrier above, hence past its reads of it
    if (k + STAGES - 1 < cg) issue(c + STAGES - 1, (slot + STAGES - 1) % STAGES);
    slot = slot + 1 == STAGES ? 0 : slot + 1;
  }
}

// rc != 0: the shape does not fit this instance
template <int THREADS, int NPAIR, int NDMA, int STAGES>
int launch(hipStream_t s, const warp_lab::Args& A, int cg) {
  const int HW = A.H * A.W;
  if (HW % 2 || A.C % cg || (reinterpret_cast<uintptr_t>(A.feat) & 15) || ((size_t)A.C * HW) % 4) return -11;
  if (THREADS * NPAIR * 2 < HW) return -12;
  if ((reinterpret_cast<uintptr_t>(A.mul ? A.mul : A.add) & 15)) return -11;
  const int guard = (2 * A.W + 2 + 4 + 3) & ~3;       // taps up to 2W + 2 floats before the plane; + the plane's <= 3 floats of shift
  // [guard][<= 3 floats of shift + the plane, as whole chunks, + W + 2 floats of taps below it][guard][operand plane]
  const int chunks = (3 + HW + 3) / 4;
  if (chunks > THREADS * NDMA || 3 + HW + A.W + 2 > THREADS * NDMA * 4 + guard) return -13;
  const int stage_floats = 2 * guard + 2 * THREADS * NDMA * 4;
  const size_t lds_bytes = (size_t)STAGES * stage_floats * 4;
  if (lds_bytes > 160 * 1024) return -14;
  const bool m = A.mul, a = A.add, r = A.res;
  dim3 grid(A.N * (A.C / cg));
#define LSFA_WL_CASE(M, AD, R)                                                                                              \
  if (m == M && a == AD && r == R) {                                                                                        \
    auto kfn = kernel<THREADS, NPAIR, NDMA, STAGES, M, AD, R>;                                                              \
    static bool attr_set = false;                                                                                           \
    if (!attr_set) { (void)hipFuncSetAttribute((const void*)kfn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); attr_set = true; } \
    hipLaunchKernelGGL(kfn, grid, dim3(THREADS), lds_bytes, s, A, cg, guard);                                               \
    return 0;                                                                                                               \
  }
  LSFA_WL_CASE(true, false, false)
  LSFA_WL_CASE(false, true, true)
#undef LSFA_WL_CASE
  return -15;
}

// the floor for this byte mix: out = feat * mul (or feat + add) streamed with 16-byte accesses, no warp at all
template <bool HAS_MUL>
__global__ __launch_bounds__(256) void stream_floor_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ o, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
    const float4 x = a[i], y = b[i];
    o[i] = HAS_MUL ? make_float4(x.x * y.x, x.y * y.y, x.z * y.z, x.w * y.w) : make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
}

}  // namespace warp_lds
