// r4: ONE general convolution / GEMM kernel for the split-operand family, templated on
//   PC  pieces per fp32 operand: 3 = three bf16 pieces, six products (conv_split_kernel.h's arithmetic);
//                                2 = two fp16 pieces + a power-of-two scale per map, three products;
//                                1 = one bf16 piece (round to nearest even), one product: the bf16 mode of BASELINE configs[2]
//   NT  32-column accumulator tiles per wave (2: 128 x 64 workgroup tiles, 4: 128 x 128)
//   ST  stages of the LDS ring (2 ... 4): ST - 1 chunks of K in flight per workgroup
// It replaces conv_split_kernel (NT 2, ST 2), conv_split_deep_kernel (NT 2, ST 4) and conv_split_wide_kernel (NT 4, ST 2)
// of rounds 2-3, which differed only in these three numbers, and adds what the small GEMMs of this network need:
//   * a ring of any depth for every tile shape.  Most launches here are SHORT (a res4 conv1 is 32 chunks of K cut six ways:
//     five or six chunks per workgroup, one workgroup per CU) and were latency-bound with one chunk in flight: each chunk's
//     28-32 KB took ~1 us to arrive from L2 / the Infinity Cache while its MFMAs take 0.15-0.35 us.  With ST - 1 chunks in
//     flight the arrival latency is paid once per workgroup, not once per chunk.
//   * the maximum of |output| written by the EPILOGUE (`amax_out`: kAmaxSlots slots, atomicMax on the bit pattern of |v|,
//     zeroed once per frame by the caller) so that the fp16 form's scale of the NEXT layer needs no pass of its own, and
//   * a status word: a non-finite output (an under-estimated scale makes fp16(x s) overflow to inf) raises bit 0 of
//     `status`, which the host reads with lsfa_status_check() - an overflow is an error, not a silently wrong feature.
// Loop protocol (per chunk c, its data in stage c % ST): wait until at most min(ST - 2, chunks left) chunks' worth of this
// wave's LDS-DMAs are outstanding (DMAs retire in issue order: counted `s_waitcnt vmcnt`), `lgkmcnt(0)`, raw `s_barrier` (every
// wave's share of chunk c has landed and everybody has finished reading chunk c - 1's stage), issue chunk c + ST - 1 into that
// stage, compute chunk c.  One barrier per chunk, all LDS addresses compile-time offsets of ONE __shared__ array (the loop is
// unrolled over the stages), no register-returning vector load inside the loop (the scale is read before the first DMA by an
// inline-asm load with its own wait), so hipcc emits no `vmcnt(0)` of its own.
// TWO-LEVEL ACCUMULATION: an MFMA accumulator is one sequential fp32 chain (feat_conv_3x3: 192 chunks x 6 matrix instructions =
// 1152 rounding steps per K slice), where a CPU library's blocked loops run ~100 short chains and add them at the end.  Measured
// against the float64 graph that made the GPU path 2.2x as far off as the fp32 oracle (tests/test_parity_fullres_gpu.py, r4).
// Every kFlush chunks the accumulators are therefore added into a second set and cleared: chains of <= kFlush x 2 x PC adds, then
// <= ceil(chunks / kFlush) adds of the block sums - the error of a blocked summation, for 16 x NT vector adds per kFlush (= 12) chunks.
#pragma once
#include "conv_split_kernel.h"

namespace lsfa {
namespace convsplit {

constexpr int kFlush = 12;        // a multiple of every ring depth: the flush points do not depend on ST

template <int NT, int PC, int ST> struct Ring {
  static constexpr int kColTile = 128 * PC;                 // uint4 of one 32-column tile of one chunk: 2 steps x PC pieces x 64 lanes
  static constexpr int kStageBn = NT * kColTile;
  static constexpr int kStageN = kStageA + kStageBn;        // uint4 per stage: 16 KB of A + NT x PC x 2 KB of B
  static constexpr int kDmaB = (NT * kColTile) / (4 * 64);  // B DMA instructions per wave and chunk: NT * PC / 2
  static constexpr int kDma = 4 + kDmaB;                    // all DMA instructions per wave and chunk
  static constexpr int kLdsBytes = ST * kStageN * 16;
  static constexpr int kWgPerCu = (2 * kLdsBytes <= 160 * 1024) ? 2 : 1;
  static_assert((NT * kColTile) % 256 == 0, "NT * PC must be even");
  static_assert(kLdsBytes <= 160 * 1024, "ring does not fit the CU's LDS");
  static_assert((ST - 2) * kDma <= 63, "vmcnt is a 6-bit counter");
};

// r4 (Args::in_scale): the 16 channels a lane cuts from chunk `gch` are 32 gch + 16 (lane >> 5) + 0..15 (r0..r3, four each).  T holds
// in_scale * s and, kAffineMaxCin floats further, in_shift * s (s = the fp16 form's power-of-two scale, 1 otherwise):
// max(v (sc s) + sh s, 0) = s max(v sc + sh, 0) exactly, the value the previous layer's epilogue would have stored, scaled - the cut
// then runs with scale 1.
__device__ __forceinline__ float4 affine_relu4(const float4& v, const float4& sc, const float4& sh) {
  return make_float4(fmaxf(v.x * sc.x + sh.x, 0.f), fmaxf(v.y * sc.y + sh.y, 0.f), fmaxf(v.z * sc.z + sh.z, 0.f), fmaxf(v.w * sc.w + sh.w, 0.f));
}
__device__ __forceinline__ void affine_chunk(const float* T, int gch, int lane, float4& c0, float4& c1, float4& c2, float4& c3) {
  const float4* sc = reinterpret_cast<const float4*>(T + gch * kChunk + 16 * (lane >> 5));
  const float4* sh = reinterpret_cast<const float4*>(T + kAffineMaxCin + gch * kChunk + 16 * (lane >> 5));
  c0 = affine_relu4(c0, sc[0], sh[0]);
  c1 = affine_relu4(c1, sc[1], sh[1]);
  c2 = affine_relu4(c2, sc[2], sh[2]);
  c3 = affine_relu4(c3, sc[3], sh[3]);
}

template <int NT, int PC, int ST, int S>
__device__ __forceinline__ void ring_issue_a(uint4 (*R)[(Ring<NT, PC, ST>::kStageN)], const float* __restrict__ x, const Geom& g, const Walk& wk) {
  const int dy = wk.ty * g.dil, dx = wk.tx * g.dil;
  const int doff = (dy * g.W + dx) * g.lda + wk.kc * kChunk;
  uint4* a_dst = &R[S][g.wave * 256];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const bool ok = (unsigned)(g.iy0[i] + dy) < (unsigned)g.H && (unsigned)(g.ix0[i] + dx) < (unsigned)g.W;
    const float* src = ok ? x + (g.off0[i] + doff) : g_zero_block;
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(src), a_dst + i * 64, 16, 0, 0);
  }
}

template <int NT, int PC, int ST, int S, int I0, int I1>
__device__ __forceinline__ void ring_issue_b(uint4 (*R)[(Ring<NT, PC, ST>::kStageN)], const uint4* __restrict__ wblock, const Geom& g, int gch) {
  typedef Ring<NT, PC, ST> RG;
  const uint4* wsrc = wblock + (size_t)gch * g.wstride + g.wave * (RG::kDmaB * 64) + g.lane;
  uint4* b_dst = &R[S][kStageA + g.wave * (RG::kDmaB * 64)];
#pragma unroll
  for (int i = I0; i < I1; ++i) __builtin_amdgcn_global_load_lds(wsrc + i * 64, b_dst + i * 64, 16, 0, 0);
}

// chunk c of n, its data in stage S = c % ST
template <int NT, int PC, int ST, int S, bool AF>
__device__ __forceinline__ void ring_step(uint4 (*R)[(Ring<NT, PC, ST>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                          const Geom& g, Walk& wk, int c, int n, f32x16 (&acc)[NT], float a_scale, const float* T) {
  typedef Ring<NT, PC, ST> RG;
  constexpr int kDmaB = RG::kDmaB;
  constexpr int SN = (S + ST - 1) % ST;                 // the stage chunk c - 1 lived in: free once everybody is past the barrier
  // how many chunks beyond c this wave has already issued (they may stay in flight)
  const int ahead = min(n - 1 - c, ST - 2);
  if (ST >= 4 && ahead >= 2) wait_vmcnt<(ST >= 4 ? 2 : 0) * RG::kDma>();
  else if (ST >= 3 && ahead == 1) wait_vmcnt<(ST >= 3 ? 1 : 0) * RG::kDma>();
  else wait_vmcnt<0>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const bool more = c + ST - 1 < n;
  const int gch_next = wk.gch;
  if (more) ring_issue_a<NT, PC, ST, SN>(R, x, g, wk);
  const uint4* A = &R[S][g.wave * 256];
  const uint4* B = &R[S][kStageA + g.lane];
  const uint4 r0 = A[g.frag[0]], r1 = A[g.frag[1]], r2 = A[g.frag[2]], r3 = A[g.frag[3]];
  float4 c0 = as_f4(r0), c1 = as_f4(r1), c2 = as_f4(r2), c3 = as_f4(r3);
  if (AF) affine_chunk(T, g.chunk0 + c, g.lane, c0, c1, c2, c3);
  const PiecesN s0 = cut8<PC>(c0, c1, AF ? 1.f : a_scale), s1 = cut8<PC>(c2, c3, AF ? 1.f : a_scale);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    // fragment (col tile t, step s, piece p) at ((t*2 + s)*PC + p)*64 + lane
    acc[t] = mma_pc<PC>(s0, B + ((t * 2 + 0) * PC) * 64, acc[t]);
    acc[t] = mma_pc<PC>(s1, B + ((t * 2 + 1) * PC) * 64, acc[t]);
    if (more) {      // this wave's share of the incoming chunk's weights, a slice behind each column tile's MFMAs
      constexpr int kPer = (kDmaB + NT - 1) / NT;
      constexpr int kE0 = (kPer < kDmaB) ? kPer : kDmaB;
      constexpr int kE1 = (2 * kPer < kDmaB) ? 2 * kPer : kDmaB;
      constexpr int kE2 = (3 * kPer < kDmaB) ? 3 * kPer : kDmaB;
      if (t == 0) ring_issue_b<NT, PC, ST, SN, 0, kE0>(R, wblock, g, gch_next);
      if (NT > 1 && t == 1) ring_issue_b<NT, PC, ST, SN, kE0, (NT > 2 ? kE1 : kDmaB)>(R, wblock, g, gch_next);
      if (NT > 2 && t == 2) ring_issue_b<NT, PC, ST, SN, kE1, (NT > 3 ? kE2 : kDmaB)>(R, wblock, g, gch_next);
      if (NT > 3 && t == 3) ring_issue_b<NT, PC, ST, SN, kE2, kDmaB>(R, wblock, g, gch_next);
    }
  }
  if (more) wk.next(g.kw, g.chunks_per_tap);
}

// ---- split roles (SP): waves 4-7 of a 512-thread workgroup issue the copies, waves 0-3 cut and multiply ------------------------------
// Measured (rocprofv3 kernel trace, r4): a chunk costs a mixed-role wave ~1.3 us whatever the ring depth - its 6-8 LDS-DMA
// instructions (~100 cycles of issue each), ~100 VALU instructions of cutting, 16 LDS reads and 12-24 MFMAs are ONE in-order
// instruction stream, and the matrix pipe idles while the wave sits in a DMA issue.  With the copies issued by a partner wave on
// the same SIMD the consumer's stream is LDS reads + cut + MFMAs only and the two streams overlap.  Same LDS image, same barrier
// per chunk (all eight waves), same arithmetic: bit-identical results to the mixed-role form.
template <int NT, int PC, int ST, int S>
__device__ __forceinline__ void ring_load_step(uint4 (*R)[(Ring<NT, PC, ST>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                               const Geom& g, Walk& wk, int c, int n) {
  typedef Ring<NT, PC, ST> RG;
  constexpr int SN = (S + ST - 1) % ST;
  const int ahead = min(n - 1 - c, ST - 2);
  if (ST >= 4 && ahead >= 2) wait_vmcnt<(ST >= 4 ? 2 : 0) * RG::kDma>();
  else if (ST >= 3 && ahead == 1) wait_vmcnt<(ST >= 3 ? 1 : 0) * RG::kDma>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (c + ST - 1 < n) {
    ring_issue_a<NT, PC, ST, SN>(R, x, g, wk);
    ring_issue_b<NT, PC, ST, SN, 0, RG::kDmaB>(R, wblock, g, wk.gch);
    wk.next(g.kw, g.chunks_per_tap);
  }
}

template <int NT, int PC, int ST, int S, bool AF>
__device__ __forceinline__ void ring_consume_step(uint4 (*R)[(Ring<NT, PC, ST>::kStageN)], const Geom& g, f32x16 (&acc)[NT], float a_scale, int c,
                                                  const float* T) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const uint4* A = &R[S][g.wave * 256];
  const uint4* B = &R[S][kStageA + g.lane];
  const uint4 r0 = A[g.frag[0]], r1 = A[g.frag[1]], r2 = A[g.frag[2]], r3 = A[g.frag[3]];
  float4 c0 = as_f4(r0), c1 = as_f4(r1), c2 = as_f4(r2), c3 = as_f4(r3);
  if (AF) affine_chunk(T, g.chunk0 + c, g.lane, c0, c1, c2, c3);
  const PiecesN s0 = cut8<PC>(c0, c1, AF ? 1.f : a_scale), s1 = cut8<PC>(c2, c3, AF ? 1.f : a_scale);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    acc[t] = mma_pc<PC>(s0, B + ((t * 2 + 0) * PC) * 64, acc[t]);
    acc[t] = mma_pc<PC>(s1, B + ((t * 2 + 1) * PC) * 64, acc[t]);
  }
}

template <int NT, int PC, int ST, int S>
__device__ __forceinline__ void ring_prologue(uint4 (*R)[(Ring<NT, PC, ST>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                              const Geom& g, Walk& wk, int n) {
  if (S < n) {
    ring_issue_a<NT, PC, ST, S>(R, x, g, wk);
    ring_issue_b<NT, PC, ST, S, 0, Ring<NT, PC, ST>::kDmaB>(R, wblock, g, wk.gch);
    wk.next(g.kw, g.chunks_per_tap);
  }
}

// grid (8 * ceil(tiles / 8)); block 256 (SP: 512).  tiles = ceil(P / 128) * (Cout / (32 * NT)) * slices (* phases)
// AF: Args::in_scale / in_shift are applied to the input where it is cut (their table, 16 KB, sits in LDS behind the ring)
template <int NT, int PC, int ST, bool SP = false, bool AF = false>
// (two resident workgroups only where their registers allow it: a 128 x 128 tile with loader waves needs ~200 per lane, which two 512-thread
// workgroups per CU - four waves per SIMD, 128 registers - could only have by spilling 100-270 of them)
static __global__ __launch_bounds__((SP ? 2 : 1) * kThreads, (((AF ? 2 * (Ring<NT, PC, ST>::kLdsBytes + 16384) <= 160 * 1024 : Ring<NT, PC, ST>::kWgPerCu == 2) && !(SP && NT == 4)) ? 2 : 1) * (SP ? 2 : 1))
void conv_ring_kernel(Args a, int nx, int ny, int nz) {
  typedef Ring<NT, PC, ST> RG;
  __shared__ __attribute__((aligned(16))) uint4 R[ST][RG::kStageN];
  __shared__ __attribute__((aligned(16))) float T[AF ? 2 * kAffineMaxCin : 4];
  Tile tile = xcd_tile(blockIdx.x, nx, ny, nz, a.tile_order, a.inv_nx, a.inv_ny);
  if (tile.x < 0) return;
  if (a.nphase > 1) { const int slices = nz / a.nphase, phase = tile.z / slices; tile.z -= phase * slices; apply_phase(a, phase, slices); }
  const int tid = threadIdx.x;
  const int P = a.N * a.Ho * a.Wo;
  if (tile.x * kWgPix >= P) return;
  const int taps = a.kh * a.kw;
  Geom g;
  g.lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = SP && wave8 >= 4;                 // wave-uniform role
  g.wave = wave8 & 3;                                    // the consumer wave whose rows / share of B this wave reads or copies
  // fp16 form: the scale that puts max|x| into [2^13, 2^14), and its inverse together with the weights' (both powers of two: exact)
  float a_scale = 1.f, out_scale = 1.f;
  if (PC == 2 && !loader) {
    const int s_exp = 13 - amax_exponent_asm(a.amax, g.lane, a.status);
    a_scale = ldexpf(1.f, s_exp);
    out_scale = ldexpf(1.f, -(s_exp + a.w_exp));
  }
  if (AF && !loader) {      // visible to every consumer after the first chunk's barrier (which an lgkmcnt(0) precedes)
    for (int k = tid; k < a.Cin; k += kThreads) { T[k] = a.in_scale[k] * a_scale; T[kAffineMaxCin + k] = a.in_shift[k] * a_scale; }
  }
  g.H = a.H; g.W = a.W; g.Cin = a.Cin; g.lda = a.lda; g.kw = a.kw; g.stride = a.stride; g.dil = a.dil;
  g.chunks_per_tap = a.Cin / kChunk;
  const int chunk_total = taps * g.chunks_per_tap;
  g.chunk0 = tile.z * a.chunks_per_slice;
  const int nchunks = min(a.chunks_per_slice, chunk_total - g.chunk0);
  const int col_tiles = a.Cout / 32;
  g.wstride = (size_t)col_tiles * RG::kColTile;
  const uint4* wblock = a.wfrag + (size_t)(NT * tile.y) * RG::kColTile;
  const int m0 = tile.x * kWgPix + g.wave * kWavePix;
  // DMA role: instruction i moves pixels 8i .. 8i+7 of the wave's tile, lane -> pixel 8i + (lane >> 3), slot lane & 7
  if (!SP || loader) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int pix = m0 + 8 * i + (g.lane >> 3);
    const int piece = (g.lane & 7) ^ ((4 * i + (g.lane >> 4)) & 7);      // slot -> source piece: the bank swizzle
    g.iy0[i] = g.ix0[i] = -(1 << 24);
    g.off0[i] = 0;
    if (pix < P) {
      const int pn = fdiv(pix, a.Ho * a.Wo, a.inv_howo), r = pix - pn * a.Ho * a.Wo, py = fdiv(r, a.Wo, a.inv_wo), px = r - py * a.Wo;
      g.iy0[i] = py * a.stride - a.pad_h;
      g.ix0[i] = px * a.stride - a.pad_w;
      g.off0[i] = ((pn * a.H + g.iy0[i]) * a.W + g.ix0[i]) * a.lda + 4 * piece;     // may be negative; only used in bounds
    }
  }
  }
  {
    const int r = g.lane & 31, h = g.lane >> 5, sw = (r >> 1) & 7;
#pragma unroll
    for (int j = 0; j < 4; ++j) g.frag[j] = r * 8 + ((4 * h + j) ^ sw);
  }
  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;

  Walk wk;       // the next chunk to fetch
  wk.gch = g.chunk0;
  {
    const int tap = fdiv(g.chunk0, g.chunks_per_tap, a.inv_cpt);
    wk.kc = g.chunk0 - tap * g.chunks_per_tap;
    wk.ty = fdiv(tap, a.kw, a.inv_kw);
    wk.tx = tap - wk.ty * a.kw;
  }
  f32x16 sum[NT];        // block sums (second level)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[t][i] = 0.f;
  if (SP) {
    if (loader) {
      // prologue: chunks 0 .. ST-2 into stages 0 .. ST-2, then one step per chunk: wait for chunk c, meet, refill the stage chunk c - 1 left
      ring_prologue<NT, PC, ST, 0>(R, a.x, wblock, g, wk, nchunks);
      if (ST > 2) ring_prologue<NT, PC, ST, (ST > 2 ? 1 : 0)>(R, a.x, wblock, g, wk, nchunks);
      if (ST > 3) ring_prologue<NT, PC, ST, (ST > 3 ? 2 : 0)>(R, a.x, wblock, g, wk, nchunks);
      for (int c = 0; c < nchunks; c += ST) {
        ring_load_step<NT, PC, ST, 0>(R, a.x, wblock, g, wk, c, nchunks);
        if (c + 1 < nchunks) ring_load_step<NT, PC, ST, 1>(R, a.x, wblock, g, wk, c + 1, nchunks);
        if (ST > 2 && c + 2 < nchunks) ring_load_step<NT, PC, ST, (ST > 2 ? 2 : 0)>(R, a.x, wblock, g, wk, c + 2, nchunks);
        if (ST > 3 && c + 3 < nchunks) ring_load_step<NT, PC, ST, (ST > 3 ? 3 : 0)>(R, a.x, wblock, g, wk, c + 3, nchunks);
      }
      return;                 // the epilogue is the consumers'
    }
    int since = 0;
    for (int c = 0; c < nchunks; c += ST) {
      ring_consume_step<NT, PC, ST, 0, AF>(R, g, acc, a_scale, c, T);
      if (c + 1 < nchunks) ring_consume_step<NT, PC, ST, 1, AF>(R, g, acc, a_scale, c + 1, T);
      if (ST > 2 && c + 2 < nchunks) ring_consume_step<NT, PC, ST, (ST > 2 ? 2 : 0), AF>(R, g, acc, a_scale, c + 2, T);
      if (ST > 3 && c + 3 < nchunks) ring_consume_step<NT, PC, ST, (ST > 3 ? 3 : 0), AF>(R, g, acc, a_scale, c + 3, T);
      since += ST;
      if (since >= kFlush) {
        since = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) { sum[t][i] = sum[t][i] + acc[t][i]; acc[t][i] = 0.f; }
      }
    }
  } else {
    // prologue: chunks 0 .. ST-2 into stages 0 .. ST-2
    ring_prologue<NT, PC, ST, 0>(R, a.x, wblock, g, wk, nchunks);
    if (ST > 2) ring_prologue<NT, PC, ST, (ST > 2 ? 1 : 0)>(R, a.x, wblock, g, wk, nchunks);
    if (ST > 3) ring_prologue<NT, PC, ST, (ST > 3 ? 2 : 0)>(R, a.x, wblock, g, wk, nchunks);
    int since = 0;
    for (int c = 0; c < nchunks; c += ST) {
      ring_step<NT, PC, ST, 0, AF>(R, a.x, wblock, g, wk, c, nchunks, acc, a_scale, T);
      if (c + 1 < nchunks) ring_step<NT, PC, ST, 1, AF>(R, a.x, wblock, g, wk, c + 1, nchunks, acc, a_scale, T);
      if (ST > 2 && c + 2 < nchunks) ring_step<NT, PC, ST, (ST > 2 ? 2 : 0), AF>(R, a.x, wblock, g, wk, c + 2, nchunks, acc, a_scale, T);
      if (ST > 3 && c + 3 < nchunks) ring_step<NT, PC, ST, (ST > 3 ? 3 : 0), AF>(R, a.x, wblock, g, wk, c + 3, nchunks, acc, a_scale, T);
      since += ST;
      if (since >= kFlush) {
        since = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int i = 0; i < 16; ++i) { sum[t][i] = sum[t][i] + acc[t][i]; acc[t][i] = 0.f; }
      }
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = (sum[t][i] + acc[t][i]) * out_scale;       // out_scale = 1 unless PC == 2 (a power of two: exact)

  // C/D layout of 32x32: column = lane & 31 (channel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (pixel)
  const int lane = g.lane;
  // Channels-last outputs leave through LDS: in the accumulator layout a lane holds ONE channel of 16 pixels, i.e. 4-byte accesses
  // (16 per tile and output, plus 16 residual loads: the epilogue of a conv3 - residual, sum, next bn1 / relu1 - was 7 of its 27 us).
  // The ring is free once every wave is past its last chunk: each wave writes its NT 32 x 32 tiles row-major into a private 4 KB x NT
  // region and reads them back as float4 along the channels - lane -> (row 8k + lane / 8, channels 4 (lane % 8) ..): whole 128-byte
  // rows per 8 lanes, 16 bytes per lane, a quarter of the memory instructions.  Same values, same arithmetic per element.
  constexpr bool kRowsFit = RG::kLdsBytes >= 4 * NT * 4096;       // the ring holds the four waves' tiles
  const bool rows_ok = kRowsFit && rows_path_ok(a);
  if (rows_ok) {
    float* part = a.part ? a.part + (size_t)tile.z * P * a.Cout : nullptr;
    RowsIn<NT> in;                                      // residual, bias, second output's affine: in flight across the barrier and the turn through LDS
    tile_rows_in<NT>(a, m0, P, tile.y * (32 * NT), part != nullptr, lane, in);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave is done reading the ring (SP: the loaders have left)
    float* T = reinterpret_cast<float*>(&R[0][0]) + g.wave * (NT * 1024);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) T[t * 1024 + ((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * 32 + (lane & 31)] = acc[t][r];
    const uint32_t m = tile_rows_out<NT>(a, T, m0, P, tile.y * (32 * NT), part, lane, in);
    if (!part) publish_amax(m, a.amax_out, a.status, blockIdx.x * 4 + g.wave);
    return;
  }
  constexpr bool kColsFit = RG::kLdsBytes >= 4 * NT * 32 * kColPitch * 4;       // ... or their padded column-major images
  if (kColsFit && a.y_nchw && !a.part && !a.view) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave is done reading the ring
    float* Tc = reinterpret_cast<float*>(&R[0][0]) + g.wave * (NT * 32 * kColPitch);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) Tc[(t * 32 + (lane & 31)) * kColPitch + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)] = acc[t][r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // the wave reads back what its own lanes wrote
    const uint32_t m = tile_cols_out_nchw<NT>(a, Tc, m0, P, tile.y * (32 * NT), lane);
    publish_amax(m, a.amax_out, a.status, blockIdx.x * 4 + g.wave);
    return;
  }
  int prow[16];
  RowOut ro;
  ro.valid = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    prow[r] = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    if (prow[r] < P) ro.valid |= 1u << r;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) ro.base[r] = ((ro.valid >> r) & 1u) ? out_pixel_base(a, prow[r]) : 0;
  uint32_t m = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) m = max(m, tile_store_max(a, ro, tile.y * (32 * NT) + t * 32 + (lane & 31), acc[t]));
  publish_amax(m, a.amax_out, a.status, blockIdx.x * 4 + g.wave);
}

// weights (Cout, taps, Cin) fp32 -> fragment order, PC pieces.  One thread per (fragment, lane): 8 values.
// out index: ((((g * col_tiles + t) * 2 + s) * PC + piece) * 64 + lane) uint4, g = tap * (Cin/32) + chunk; PC = 2: values w * 2^w_exp
template <int PC>
static __global__ void pack_weights_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int taps, int Cin, int w_exp) {
  const int col_tiles = Cout / 32, chunks = Cin / kChunk;
  const long total = (long)taps * chunks * col_tiles * 2 * 64;
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int lane = (int)(i & 63);
  long r = i >> 6;
  const int s = (int)(r & 1); r >>= 1;
  const int t = (int)(r % col_tiles); r /= col_tiles;
  const int g = (int)r;
  const int tap = g / chunks, kc = g - tap * chunks;
  const int co = t * 32 + (lane & 31);
  const int ci = kc * kChunk + 16 * (lane >> 5) + 8 * s;
  const float* src = w + ((size_t)co * taps + tap) * Cin + ci;
  const float4 v0 = *reinterpret_cast<const float4*>(src), v1 = *reinterpret_cast<const float4*>(src + 4);
  const PiecesN p = cut8<PC>(v0, v1, ldexpf(1.f, w_exp));
  uint4* dst = out + ((((size_t)g * col_tiles + t) * 2 + s) * PC) * 64 + lane;
#pragma unroll
  for (int q = 0; q < PC; ++q) dst[q * 64] = p.p[q];
}

}  // namespace convsplit
}  // namespace lsfa
