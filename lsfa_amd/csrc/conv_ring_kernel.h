// ONE general convolution / GEMM kernel for the split-operand family, templated on
//   PC  pieces per fp32 operand: 3 = three bf16 pieces, six products (conv_split_kernel.h's arithmetic);
//                                2 = two fp16 pieces + a power-of-two scale per map, three products;
//                                1 = one bf16 piece (round to nearest even), one product: the bf16 mode of BASELINE configs[2]
//   NT  32-column accumulator tiles per wave (2: 128 x 64 workgroup tiles, 4: 128 x 128)
//   ST  stages of the LDS ring (2 ... 4): ST - 1 chunks of K in flight per workgroup
//   SP  split roles: waves 4-7 of a 512-thread workgroup issue the copies, waves 0-3 cut and multiply
//   AF  the input's bn + ReLU applied where the operand is cut (Args::in_scale)
//   WV  waves that multiply: 4 (128-pixel tiles) or 8 (r5: 256-pixel tiles, eight mixed-role waves, two per SIMD)
// What the loop is made of (r4): a ring of ST stages filled by LDS-DMA (`global_load_lds_dwordx4`), counted `s_waitcnt vmcnt(N)` (DMAs
// retire in issue order), one raw `s_barrier` per chunk, all LDS addresses compile-time offsets of ONE __shared__ array (the loop is
// unrolled over the stages), no register-returning vector load inside the loop (the scale is read before the first DMA by an inline-asm
// load with its own wait), so hipcc emits no `vmcnt(0)` of its own; the maximum of |output| written by the EPILOGUE (`amax_out`) and a
// status word for non-finite outputs (an under-estimated scale).
// TWO-LEVEL ACCUMULATION: an MFMA accumulator is one sequential fp32 chain (feat_conv_3x3: 192 chunks x 6 matrix instructions =
// 1152 rounding steps per K slice), where a CPU library's blocked loops run ~100 short chains and add them at the end.  Measured
// against the float64 graph that made the GPU path 2.2x as far off as the fp32 oracle (tests/test_parity_fullres_gpu.py, r4).
// Every kFlush chunks the accumulators are therefore added into a second set and cleared: chains of <= kFlush x 2 x PC adds, then
// <= ceil(chunks / kFlush) adds of the block sums - the error of a blocked summation, for 16 x NT vector adds per kFlush (= 12) chunks.
// r5: TWO step forms.  The *pipelined* step (NT = 4: loader / consumer waves, and - one body per step - the mixed-role kernels; WV = 8) cuts chunk v + 1 under the matrix instructions of chunk v, the
// order pinned with sched_group_barrier ("the cut of chunk v + 1 runs UNDER ..." below); every other instantiation keeps r4's step
// ("r4's step, kept for ..." below).  Both compute the same products in the same order per accumulator: results are bit-identical across
// every plan (tests/test_hip_ops.py::test_conv_ring_every_plan_gives_the_same_convolution).
// What bounds the kernel (r5, profiles/r5/ring_ablation.txt, fill_lab.txt): with the copies switched off the six-image feat_conv_3x3 takes
// 1272 us, with the arithmetic switched off 998, with both on 1643 (r4's step: 1758-1892); a CU takes in 32 KB per 128 x 128 x 32 chunk,
// which the LDS-DMA path delivers at 45-75 GB/s per CU for this access pattern (120 from an L2-resident megabyte, 27 from the Infinity
// Cache), against the 768 matrix cycles (0.41 us at the 1.88 GHz the part holds under this load) the chunk is worth: the two sides are
// balanced within 25 %, and neither a deeper ring, tile order, wave priorities nor the pinned step move the sum by more than 5-8 %.
#pragma once
#include "conv_split_kernel.h"

namespace lsfa {
namespace convsplit {

constexpr int kFlush = 12;        // a multiple of every ring depth: the flush points do not depend on ST

// lab (tools/lab/build_variant.sh): cache-policy bits of the ring's LDS-DMA loads (0 = default; 1 = sc0, 2 = nt, 16 = sc1: the latter two bypass the CU's L1)
#ifndef LSFA_RING_A_AUX
#define LSFA_RING_A_AUX 0
#endif
#ifndef LSFA_RING_B_AUX
#define LSFA_RING_B_AUX 0
#endif

#ifdef LSFA_LAB_HALF_A      // lab ablation (tools/lab/build_variant.sh): half of A's copies issued - wrong numbers, the step's timing with 8 KB of A per chunk
constexpr int kRingIssueA = 2;
#else
constexpr int kRingIssueA = 4;
#endif

template <int NT, int PC, int ST, int WV = 4> struct Ring {     // WV: waves that multiply (4: 128-pixel tiles; 8: 256-pixel tiles, mixed roles only)
  static constexpr int kColTile = 128 * PC;                 // uint4 of one 32-column tile of one chunk: 2 steps x PC pieces x 64 lanes
  static constexpr int kStageA = WV * 256;                  // uint4 of A per stage: WV waves x 32 pixels x 8 slots (4 KB per wave)
  static constexpr int kStageBn = NT * kColTile;
  static constexpr int kStageN = kStageA + kStageBn;        // uint4 per stage: 4 WV KB of A + NT x PC x 2 KB of B
  static constexpr int kDmaB = (NT * kColTile) / (WV * 64); // B DMA instructions per wave and chunk: NT * PC / 2 (WV = 8: / 4)
  static constexpr int kDma = kRingIssueA + kDmaB;          // all DMA instructions per wave and chunk (4 of A + B's)
  static constexpr int kLdsBytes = ST * kStageN * 16;
  static constexpr int kWgPerCu = (2 * kLdsBytes <= 160 * 1024) ? 2 : 1;
  static_assert((NT * kColTile) % (WV * 64) == 0, "NT * PC must be a multiple of WV / 2");
  static_assert(kLdsBytes <= 160 * 1024, "ring does not fit the CU's LDS");
  static_assert((ST - 1) * kDma <= 63, "vmcnt is a 6-bit counter");
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

template <int NT, int PC, int ST, int WV, int S>
__device__ __forceinline__ void ring_issue_a(uint4 (*R)[(Ring<NT, PC, ST, WV>::kStageN)], const float* __restrict__ x, const Geom& g, const Walk& wk, bool live) {
  const int dy = wk.ty * g.dil, dx = wk.tx * g.dil;
  const int doff = (dy * g.W + dx) * g.lda + wk.kc * kChunk;
  const int dy_b = live ? dy : (1 << 26);                // not live: every row is out of bounds (a scalar select, no branch)
  uint4* a_dst = &R[S][g.wave * 256];
#pragma unroll
  for (int i = 0; i < kRingIssueA; ++i) {
    const bool ok = (unsigned)(g.iy0[i] + dy_b) < (unsigned)g.H && (unsigned)(g.ix0[i] + dx) < (unsigned)g.W;
    const float* src = ok ? x + (g.off0[i] + doff) : g_zero_block;
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const uint4*>(src), a_dst + i * 64, 16, 0, LSFA_RING_A_AUX);
  }
}

template <int NT, int PC, int ST, int WV, int S, int I0, int I1>
__device__ __forceinline__ void ring_issue_b(uint4 (*R)[(Ring<NT, PC, ST, WV>::kStageN)], const uint4* __restrict__ wblock, const Geom& g, int gch) {
  typedef Ring<NT, PC, ST, WV> RG;
  const uint4* wsrc = wblock + (size_t)gch * g.wstride + g.wave * (RG::kDmaB * 64) + g.lane;
  uint4* b_dst = &R[S][RG::kStageA + g.wave * (RG::kDmaB * 64)];
#pragma unroll
  for (int i = I0; i < I1; ++i) __builtin_amdgcn_global_load_lds(wsrc + i * 64, b_dst + i * 64, 16, 0, LSFA_RING_B_AUX);
}

// ---- r5: the cut of chunk v + 1 runs UNDER the matrix instructions of chunk v ---------------------------------------------------------
// r4's step was [barrier, read A, cut (64-112 vector instructions, ~300-450 cycles), then 8 groups of (2 B reads, wait, 3 MFMAs)]: the
// matrix pipe idled during the cut and at the head of every group (an LDS round trip for 96 cycles of work): 17-35 % duty by PMC.  An MFMA
// holds the SIMD's vector issue for 8 of its 32 cycles (MI355X_MICROARCH.md, cycle constants): the other 24 take ~5 plain vector
// instructions and an LDS read for free.  So the ring's unit is now a *v-chunk*: stage v % ST holds the weights B(v) and the raw
// activations A(v + 1).  Step v multiplies the pieces P(v) it holds in registers by B(v) and, between those MFMAs, reads A(v + 1) and
// cuts it into P(v + 1); the B fragments are requested several MFMAs ahead of their use.  The order of instructions is pinned with
// `sched_group_barrier` (MFMA, LDS read, k vector instructions, repeat).  A(0) arrives as v-chunk -1 (A only, in stage ST - 1) and is cut
// before the loop; the last v-chunk's A half is filled from the block of zeros, so that every v-chunk is the same number of DMAs.  Same cut,
// same MFMAs in the same order per accumulator as r4: bit-identical results.  One barrier more per workgroup (n + 1).
struct Cut { PiecesN s0, s1; };

// the vector instructions of one cut, for the interleave's arithmetic (fp16 pair: 2 mul, cvt_pk, 2 cvt back, 2 sub, cvt_pk; AF: mul, add, max per value)
template <int PC, bool AF> struct CutCost { static constexpr int kValu = (PC == 2 ? 64 : PC == 3 ? 96 : 16) + (AF ? 48 : 0); };

template <int NT, int PC, bool AF, int DMA = 0, int WV = 4> __device__ __forceinline__ void pin_interleave() {
  if (PC == 3 && NT == 4) return;       // (three pieces on 128 x 128 tiles: the pinned order needs more than the 256 registers of two waves per SIMD)
  constexpr int kMfma = NT * 2 * (PC == 3 ? 6 : PC == 2 ? 3 : 1);
  constexpr int kPer = (CutCost<PC, AF>::kValu + (DMA ? 24 : 0) + kMfma - 1) / kMfma + 1;
  // before the first MFMA: the first k-step's B fragments (PC reads), the raw A rows (4 reads; AF: + the table's 8) and the next
  // k-step's fragments; from then on one read per MFMA slot, which keeps a fragment in flight for >= 3 slots (an LDS round trip is
  // 64-128 cycles, a slot 32).  Mixed roles: the wave's DMAs of the v-chunk ST - 1 ahead go one per slot behind the first MFMAs.
  __builtin_amdgcn_sched_group_barrier(0x100, ((PC == 3 || WV == 8) ? PC + 2 : 2 * PC + 4) + (AF ? 8 : 0), 0);      // (WV = 8: the SIMD's other wave covers the round trips; registers are short)
#pragma unroll
  for (int i = 0; i < kMfma; ++i) {
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
    if (i >= 1 && i <= DMA) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
    __builtin_amdgcn_sched_group_barrier(0x002, kPer, 0);
  }
}

// one step's arithmetic: acc += P * B(v) (stage S), and raw A(v + 1) (same stage) -> its pieces.  The loads are written in the order
// they are wanted back.  k-step 0 of every column tile first, then k-step 1 (per accumulator the order of the products is unchanged:
// bit-identical), so that the first k-step's pieces die halfway through the step and the second half of the raw rows is read late: the
// step's peak is ~16 registers lower, which is what lets the pinned order fit 256 registers where two waves share a SIMD.
template <int NT, int PC, int ST, int WV, int S, bool AF, int DMA = 0>
__device__ __forceinline__ Cut ring_mma_cut(const uint4 (*R)[(Ring<NT, PC, ST, WV>::kStageN)], const Geom& g, const Cut& p, f32x16 (&acc)[NT], float a_scale,
                                            int gch_next, const float* T) {
  const uint4* B = &R[S][Ring<NT, PC, ST, WV>::kStageA + g.lane];
  const uint4* A = &R[S][g.wave * 256];
  uint4 bf[NT * 2][PC];      // fragment (col tile t, step s, piece q) at ((t*2 + s)*PC + q)*64 + lane
#pragma unroll
  for (int q = 0; q < PC; ++q) bf[0][q] = B[q * 64];
  const uint4 r0 = A[g.frag[0]], r1 = A[g.frag[1]];
#pragma unroll
  for (int t = 1; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < PC; ++q) bf[t * 2][q] = B[((t * 2) * PC + q) * 64];
  const uint4 r2 = A[g.frag[2]], r3 = A[g.frag[3]];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < PC; ++q) bf[t * 2 + 1][q] = B[((t * 2 + 1) * PC + q) * 64];
  float4 c0 = as_f4(r0), c1 = as_f4(r1), c2 = as_f4(r2), c3 = as_f4(r3);
  if (AF) affine_chunk(T, min(gch_next, g.chunks_per_tap - 1), g.lane, c0, c1, c2, c3);
  Cut n;
  n.s0 = cut8<PC>(c0, c1, AF ? 1.f : a_scale);
  n.s1 = cut8<PC>(c2, c3, AF ? 1.f : a_scale);
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = mma_pc<PC>(p.s0, bf[t * 2][0], bf[t * 2][PC > 1 ? 1 : 0], bf[t * 2][PC > 2 ? 2 : 0], acc[t]);
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = mma_pc<PC>(p.s1, bf[t * 2 + 1][0], bf[t * 2 + 1][PC > 1 ? 1 : 0], bf[t * 2 + 1][PC > 2 ? 2 : 0], acc[t]);
  pin_interleave<NT, PC, AF, DMA, WV>();
  return n;
}

// the first cut (A(0), v-chunk -1)
template <int NT, int PC, int ST, int WV, int S, bool AF>
__device__ __forceinline__ Cut ring_cut(const uint4 (*R)[(Ring<NT, PC, ST, WV>::kStageN)], const Geom& g, float a_scale, int gch, const float* T) {
  const uint4* A = &R[S][g.wave * 256];
  const uint4 r0 = A[g.frag[0]], r1 = A[g.frag[1]], r2 = A[g.frag[2]], r3 = A[g.frag[3]];
  float4 c0 = as_f4(r0), c1 = as_f4(r1), c2 = as_f4(r2), c3 = as_f4(r3);
  if (AF) affine_chunk(T, min(gch, g.chunks_per_tap - 1), g.lane, c0, c1, c2, c3);
  Cut n;
  n.s0 = cut8<PC>(c0, c1, AF ? 1.f : a_scale);
  n.s1 = cut8<PC>(c2, c3, AF ? 1.f : a_scale);
  return n;
}

// (A pipelined step for the four-wave mixed-role kernels - the wave issuing its share of v-chunk v + ST - 1 between its MFMAs, a second body
// for "nothing left to fetch" - was built and measured in r5: 160 (NT = 2) / 230 (NT = 4) registers, one workgroup per CU instead of two,
// slower in the backbone than r4's step on every launch that uses those kernels; they keep r4's step, see below.)
// WV = 8 (256-pixel tiles, eight mixed-role waves, two per SIMD): ONE body per step - a second copy of the step for "nothing left to
// fetch" doubles the accumulators' live ranges across the join (52 registers spilled at the 256 two waves per SIMD leave).  Every
// step issues its DMAs; past the end they fetch the block of zeros / the slice's last weights again into a stage nobody reads (the kernel
// drains them before the epilogue turns the ring into its staging area), so every wait is the same counted `vmcnt`.
template <int NT, int PC, int ST, int WV, int S, bool AF>
__device__ __forceinline__ void ring_step_uniform(uint4 (*R)[(Ring<NT, PC, ST, WV>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                                  const Geom& g, Walk& wk, int v, int n, f32x16 (&acc)[NT], Cut& p, float a_scale, const float* T) {
  typedef Ring<NT, PC, ST, WV> RG;
  constexpr int SN = (S + ST - 1) % ST;
  wait_vmcnt<(ST - 2) * RG::kDma>();
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const Walk cur = wk;
  wk.next(g.kh, g.kw, g.chunks_per_tap, g.k_order);
  ring_issue_a<NT, PC, ST, WV, SN>(R, x, g, cur, v + ST < n);
  ring_issue_b<NT, PC, ST, WV, SN, 0, RG::kDmaB>(R, wblock, g, cur.bprev);      // B(v + ST - 1); past the slice's end: some valid block, into a stage nobody reads
  p = ring_mma_cut<NT, PC, ST, WV, S, AF, RG::kDma>(R, g, p, acc, a_scale, g.chunk0 + v + 1, T);
}

template <int NT, int PC, int ST, int WV, int S>
__device__ __forceinline__ void ring_prologue_uniform(uint4 (*R)[(Ring<NT, PC, ST, WV>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                                      const Geom& g, Walk& wk, int n) {
  ring_issue_a<NT, PC, ST, WV, S>(R, x, g, wk, S + 1 < n);
  ring_issue_b<NT, PC, ST, WV, S, 0, Ring<NT, PC, ST, WV>::kDmaB>(R, wblock, g, wk.bprev);      // B(S): the walk stands at A(S + 1)
  wk.next(g.kh, g.kw, g.chunks_per_tap, g.k_order);
}

// ---- split roles (SP): waves 4-7 of a 512-thread workgroup issue the copies, waves 0-3 cut and multiply ------------------------------
// Measured (rocprofv3 kernel trace, r4): a chunk costs a mixed-role wave ~1.3 us whatever the ring depth - its 6-8 LDS-DMA
// instructions (~100 cycles of issue each), ~100 VALU instructions of cutting, 16 LDS reads and 12-24 MFMAs are ONE in-order
// instruction stream, and the matrix pipe idles while the wave sits in a DMA issue.  With the copies issued by a partner wave on
// the same SIMD the consumer's stream is LDS reads + cut + MFMAs only and the two streams overlap.  Same LDS image, same barrier
// per chunk (all eight waves), same arithmetic: bit-identical results to the mixed-role form.
template <int NT, int PC, int ST, int WV, int S>
__device__ __forceinline__ void ring_load_step(uint4 (*R)[(Ring<NT, PC, ST, WV>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                               const Geom& g, Walk& wk, int v, int n) {
  typedef Ring<NT, PC, ST, WV> RG;
  constexpr int SN = (S + ST - 1) % ST;
  const int ahead = min(n - 1 - v, ST - 2);
  if (ST >= 4 && ahead >= 2) wait_vmcnt<(ST >= 4 ? 2 : 0) * RG::kDma>();
  else if (ST >= 3 && ahead == 1) wait_vmcnt<(ST >= 3 ? 1 : 0) * RG::kDma>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (v + ST - 1 < n) {
    ring_issue_a<NT, PC, ST, WV, SN>(R, x, g, wk, v + ST < n);
    ring_issue_b<NT, PC, ST, WV, SN, 0, RG::kDmaB>(R, wblock, g, wk.bprev);      // B(v + ST - 1): the walk stands at A(v + ST)
    wk.next(g.kh, g.kw, g.chunks_per_tap, g.k_order);
  }
}

template <int NT, int PC, int ST, int WV, int S, bool AF>
__device__ __forceinline__ void ring_consume_step(uint4 (*R)[(Ring<NT, PC, ST, WV>::kStageN)], const Geom& g, f32x16 (&acc)[NT], Cut& p, float a_scale, int v,
                                                  const float* T) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  p = ring_mma_cut<NT, PC, ST, WV, S, AF>(R, g, p, acc, a_scale, g.chunk0 + v + 1, T);
}

// the prologue's v-chunks: -1 (A(0) alone, into stage ST - 1), then 0 .. ST - 2 (B(v) and A(v + 1))
template <int NT, int PC, int ST, int WV, int S>
__device__ __forceinline__ void ring_prologue(uint4 (*R)[(Ring<NT, PC, ST, WV>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                              const Geom& g, Walk& wk, int n) {
  if (S < n) {
    ring_issue_a<NT, PC, ST, WV, S>(R, x, g, wk, S + 1 < n);
    ring_issue_b<NT, PC, ST, WV, S, 0, Ring<NT, PC, ST, WV>::kDmaB>(R, wblock, g, wk.bprev);      // B(S): the walk stands at A(S + 1)
    wk.next(g.kh, g.kw, g.chunks_per_tap, g.k_order);
  }
}

// everything of v-chunk -1 has landed once at most the DMAs of the v-chunks issued behind it are outstanding
template <int NT, int PC, int ST, int WV>
__device__ __forceinline__ void ring_wait_first(int n) {
  typedef Ring<NT, PC, ST, WV> RG;
  const int behind = min(n, ST - 1);
  if (ST >= 4 && behind >= 3) wait_vmcnt<(ST >= 4 ? 3 : 0) * RG::kDma>();
  else if (ST >= 3 && behind == 2) wait_vmcnt<(ST >= 3 ? 2 : 0) * RG::kDma>();
  else if (behind == 1) wait_vmcnt<RG::kDma>();
  else wait_vmcnt<0>();
}

// ---- r4's step, kept for the 128 x 64 tiles (until the end of r5 also the mixed-role 128 x 128 ones: kUniform below) -------------------------------------------------------
// Two workgroups per CU (four waves per SIMD at NT = 2: 128 registers; two 256-thread mixed-role ones at NT = 4) overlap each other
// better than one workgroup's pinned step does on these launches - measured IN the six-image backbone, not in the lab's back-to-back
// repetitions of one layer: stage 2's conv2 84.7 vs 100.4 us, conv3 129.5 vs 150.6; res4 conv3 (mixed roles, two stages) 52.4 vs 58.3
// on 256-pixel tiles - and the pipelined step needs 160 (NT = 2) / 230 (NT = 4) registers.  Here stage c % ST holds A(c) and B(c);
// a step is [wait, barrier, read A, cut, multiply].
// chunk c of n, its data in stage S = c % ST
template <int NT, int PC, int ST, int S, bool AF>
__device__ __forceinline__ void ring_step_r4(uint4 (*R)[(Ring<NT, PC, ST, 4>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                          const Geom& g, Walk& wk, int c, int n, f32x16 (&acc)[NT], float a_scale, const float* T) {
  typedef Ring<NT, PC, ST, 4> RG;
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
  const int gch_next = wk.bidx;
  if (more) ring_issue_a<NT, PC, ST, 4, SN>(R, x, g, wk, true);
  const uint4* A = &R[S][g.wave * 256];
  const uint4* B = &R[S][Ring<NT, PC, ST, 4>::kStageA + g.lane];
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
      if (t == 0) ring_issue_b<NT, PC, ST, 4, SN, 0, kE0>(R, wblock, g, gch_next);
      if (NT > 1 && t == 1) ring_issue_b<NT, PC, ST, 4, SN, kE0, (NT > 2 ? kE1 : kDmaB)>(R, wblock, g, gch_next);
      if (NT > 2 && t == 2) ring_issue_b<NT, PC, ST, 4, SN, kE1, (NT > 3 ? kE2 : kDmaB)>(R, wblock, g, gch_next);
      if (NT > 3 && t == 3) ring_issue_b<NT, PC, ST, 4, SN, kE2, kDmaB>(R, wblock, g, gch_next);
    }
  }
  if (more) wk.next(g.kh, g.kw, g.chunks_per_tap, g.k_order);
}

// ---- split roles (SP): waves 4-7 of a 512-thread workgroup issue the copies, waves 0-3 cut and multiply ------------------------------
// Measured (rocprofv3 kernel trace, r4): a chunk costs a mixed-role wave ~1.3 us whatever the ring depth - its 6-8 LDS-DMA
// instructions (~100 cycles of issue each), ~100 VALU instructions of cutting, 16 LDS reads and 12-24 MFMAs are ONE in-order
// instruction stream, and the matrix pipe idles while the wave sits in a DMA issue.  With the copies issued by a partner wave on
// the same SIMD the consumer's stream is LDS reads + cut + MFMAs only and the two streams overlap.  Same LDS image, same barrier
// per chunk (all eight waves), same arithmetic: bit-identical results to the mixed-role form.
template <int NT, int PC, int ST, int S>
__device__ __forceinline__ void ring_load_step_r4(uint4 (*R)[(Ring<NT, PC, ST, 4>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                               const Geom& g, Walk& wk, int c, int n) {
  typedef Ring<NT, PC, ST, 4> RG;
  constexpr int SN = (S + ST - 1) % ST;
  const int ahead = min(n - 1 - c, ST - 2);
  if (ST >= 4 && ahead >= 2) wait_vmcnt<(ST >= 4 ? 2 : 0) * RG::kDma>();
  else if (ST >= 3 && ahead == 1) wait_vmcnt<(ST >= 3 ? 1 : 0) * RG::kDma>();
  else wait_vmcnt<0>();
  __builtin_amdgcn_s_barrier();
  if (c + ST - 1 < n) {
    ring_issue_a<NT, PC, ST, 4, SN>(R, x, g, wk, true);
    ring_issue_b<NT, PC, ST, 4, SN, 0, RG::kDmaB>(R, wblock, g, wk.bidx);
    wk.next(g.kh, g.kw, g.chunks_per_tap, g.k_order);
  }
}

template <int NT, int PC, int ST, int S, bool AF>
__device__ __forceinline__ void ring_consume_step_r4(uint4 (*R)[(Ring<NT, PC, ST, 4>::kStageN)], const Geom& g, f32x16 (&acc)[NT], float a_scale, int c,
                                                  const float* T) {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  const uint4* A = &R[S][g.wave * 256];
  const uint4* B = &R[S][Ring<NT, PC, ST, 4>::kStageA + g.lane];
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
__device__ __forceinline__ void ring_prologue_r4(uint4 (*R)[(Ring<NT, PC, ST, 4>::kStageN)], const float* __restrict__ x, const uint4* __restrict__ wblock,
                                              const Geom& g, Walk& wk, int n) {
  if (S < n) {
    ring_issue_a<NT, PC, ST, 4, S>(R, x, g, wk, true);
    ring_issue_b<NT, PC, ST, 4, S, 0, Ring<NT, PC, ST, 4>::kDmaB>(R, wblock, g, wk.bidx);
    wk.next(g.kh, g.kw, g.chunks_per_tap, g.k_order);
  }
}

// grid (8 * ceil(tiles / 8)); block 256 (SP: 512).  tiles = ceil(P / 128) * (Cout / (32 * NT)) * slices (* phases)
// AF: Args::in_scale / in_shift are applied to the input where it is cut (their table, 16 KB, sits in LDS behind the ring)
// WV = 8: 256-pixel tiles, eight mixed-role waves (block 512); tiles = ceil(P / 256) * ...
template <int NT, int PC, int ST, bool SP = false, bool AF = false, int WV = 4>
// (two resident workgroups only where their registers allow it: a 128 x 128 tile with loader waves needs ~230 per lane; WV = 8: 512 threads, 256 registers)
static __global__ __launch_bounds__((SP || WV == 8 ? 2 : 1) * kThreads,
    WV == 8 ? 2 : (((AF ? 2 * (Ring<NT, PC, ST, WV>::kLdsBytes + 16384) <= 160 * 1024 : Ring<NT, PC, ST, WV>::kWgPerCu == 2) && !(SP && NT == 4)) ? 2 : 1) * (SP ? 2 : 1))
void conv_ring_kernel(Args a, int nx, int ny, int nz) {
  typedef Ring<NT, PC, ST, WV> RG;
  __shared__ __attribute__((aligned(16))) uint4 R[ST][RG::kStageN];
  __shared__ __attribute__((aligned(16))) float T[AF ? 2 * kAffineMaxCin : 4];
  Tile tile = xcd_tile(blockIdx.x, nx, ny, nz, a.tile_order, a.inv_nx, a.inv_ny);
  if (tile.x < 0) return;
  if (a.nphase > 1) { const int slices = nz / a.nphase, phase = tile.z / slices; tile.z -= phase * slices; apply_phase(a, phase, slices); }
  const int tid = threadIdx.x;
  const int P = a.N * a.Ho * a.Wo;
  static_assert(!(SP && WV != 4), "loader / consumer waves exist for 128-pixel tiles only");
  constexpr int kPix = 32 * WV;                          // pixels of a workgroup's tile
  // the uniform pipelined step (ONE body per step, copies always issued - past the end into a stage nobody reads): the 256-pixel tiles and, since
  // the end of r5, the four-wave mixed-role 128 x 128 kernels: 186-256 registers without a spill, so two workgroups per CU stay (the first attempt
  // at a pipelined mixed-role step had a second body for "nothing left to fetch", needed 512 registers and ONE workgroup per CU, and lost in the
  // backbone).  Measured against r4's step on the same launches (profiles/r5/mixed_pipelined_ab.txt): res4 conv2 forced onto it 100.6 -> 74.5 us,
  // res3 conv2 72 -> 65, the small net's 3x3 at nine frames 416 -> 383, the R-FCN maps 372 -> 334; in situ the nine-frame segment pass 1206 -> 1167 us,
  // the six-image backbone pass 9470 -> 9360.  Two-chunk launches lose 5 % on it: those run on 128 x 64 tiles, which keep r4's step.
  // (the same step on the 128 x 64 tiles - 136 registers, three workgroups per CU - LOSES: res3 conv3 58 -> 63 us, and with the plan's 128 x 64
  //  launches moved from loader / consumer waves onto it the six-image backbone pass goes 9770 -> 9925 us: profiles/r5/mixed_pipelined_nt2_ab.txt)
  constexpr bool kUniform = !SP && (WV == 8 || NT == 4);
  constexpr bool kPipelined = (SP && NT == 4) || kUniform;      // the r5 step: 128 x 128 tiles with loader / consumer waves, 256 x 128 tiles; the rest keep r4's
  if (tile.x * kPix >= P) return;
  const int taps = a.kh * a.kw;
  Geom g;
  g.lane = tid & 63;
  const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
  const bool loader = SP && wave8 >= 4;                 // wave-uniform role
  g.wave = WV == 8 ? wave8 : (wave8 & 3);                // the consumer wave whose rows / share of B this wave reads or copies
  // fp16 form: the scale that puts max|x| into [2^13, 2^14), and its inverse together with the weights' (both powers of two: exact)
  float a_scale = 1.f;
  float out_scale[NT];             // per column tile: 2^-(s_exp + w_exp of this lane's column), 1 unless PC == 2
#pragma unroll
  for (int t = 0; t < NT; ++t) out_scale[t] = 1.f;
  // r6: the scales' loads leave FIRST and are waited for where the scale is needed (LSFA_SCALES_READY below): behind the geometry set-up and,
  // in the mixed-role kernels, behind the prologue's copies - one round trip under cover instead of two in front of everything
  ScaleRegs sr;
  if (PC == 2 && !loader) scale_loads_issue<NT>(a, tile.y * (32 * NT), g.lane, sr);
  // N_: vector-memory operations this wave has issued since (they return in order).  Then the input's activation table (AF), visible to every
  // consumer after the first chunk's barrier (which an lgkmcnt(0) precedes)
#ifdef LSFA_LAB_EARLY_SCALES      // lab A/B (tools/lab/build_variant.sh): r5's order - wait for the scales at once, in front of everything
  constexpr bool kScalesEarly = true;
#else
  constexpr bool kScalesEarly = false;
#endif
#define LSFA_SCALES_READY(N_)                                                                                                   \
  if (!kScalesEarly || (N_) < 0) {                                                                                              \
    if (PC == 2) { scale_loads_wait<((N_) < 0 ? 0 : (N_))>(sr); a_scale = scale_finish<NT>(a, sr, g.lane, out_scale); }         \
    if (AF) {                                                                                                                   \
      for (int k = tid; k < a.Cin; k += 64 * WV) { T[k] = a.in_scale[k] * a_scale; T[kAffineMaxCin + k] = a.in_shift[k] * a_scale; } \
    }                                                                                                                           \
  }
  if (kScalesEarly && !loader) LSFA_SCALES_READY(-1)
  g.H = a.H; g.W = a.W; g.Cin = a.Cin; g.lda = a.lda; g.kh = a.kh; g.kw = a.kw; g.stride = a.stride; g.dil = a.dil;
  g.k_order = taps > 1 ? a.k_order : 0;
  g.chunks_per_tap = a.Cin / kChunk;
  const int chunk_total = taps * g.chunks_per_tap;
  g.chunk0 = tile.z * a.chunks_per_slice;
  const int nchunks = min(a.chunks_per_slice, chunk_total - g.chunk0);
  const int col_tiles = a.Cout / 32;
  g.wstride = (size_t)col_tiles * RG::kColTile;
  const uint4* wblock = a.wfrag + (size_t)(NT * tile.y) * RG::kColTile;
  const int m0 = tile.x * kPix + g.wave * kWavePix;
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

  Walk wk;       // the next chunk whose A is to be fetched
  wk.start(g.chunk0, a.kh, a.kw, g.chunks_per_tap, g.k_order);
  f32x16 sum[NT];        // block sums (second level)
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) sum[t][i] = 0.f;
  if constexpr (!kPipelined) {
    if (SP) {
      if (loader) {
        // prologue: chunks 0 .. ST-2 into stages 0 .. ST-2, then one step per chunk: wait for chunk c, meet, refill the stage chunk c - 1 left
        ring_prologue_r4<NT, PC, ST, 0>(R, a.x, wblock, g, wk, nchunks);
        if (ST > 2) ring_prologue_r4<NT, PC, ST, (ST > 2 ? 1 : 0)>(R, a.x, wblock, g, wk, nchunks);
        if (ST > 3) ring_prologue_r4<NT, PC, ST, (ST > 3 ? 2 : 0)>(R, a.x, wblock, g, wk, nchunks);
        for (int c = 0; c < nchunks; c += ST) {
          ring_load_step_r4<NT, PC, ST, 0>(R, a.x, wblock, g, wk, c, nchunks);
          if (c + 1 < nchunks) ring_load_step_r4<NT, PC, ST, 1>(R, a.x, wblock, g, wk, c + 1, nchunks);
          if (ST > 2 && c + 2 < nchunks) ring_load_step_r4<NT, PC, ST, (ST > 2 ? 2 : 0)>(R, a.x, wblock, g, wk, c + 2, nchunks);
          if (ST > 3 && c + 3 < nchunks) ring_load_step_r4<NT, PC, ST, (ST > 3 ? 3 : 0)>(R, a.x, wblock, g, wk, c + 3, nchunks);
        }
        return;                 // the epilogue is the consumers'
      }
      LSFA_SCALES_READY(0)
      int since = 0;
      for (int c = 0; c < nchunks; c += ST) {
        ring_consume_step_r4<NT, PC, ST, 0, AF>(R, g, acc, a_scale, c, T);
        if (c + 1 < nchunks) ring_consume_step_r4<NT, PC, ST, 1, AF>(R, g, acc, a_scale, c + 1, T);
        if (ST > 2 && c + 2 < nchunks) ring_consume_step_r4<NT, PC, ST, (ST > 2 ? 2 : 0), AF>(R, g, acc, a_scale, c + 2, T);
        if (ST > 3 && c + 3 < nchunks) ring_consume_step_r4<NT, PC, ST, (ST > 3 ? 3 : 0), AF>(R, g, acc, a_scale, c + 3, T);
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
      ring_prologue_r4<NT, PC, ST, 0>(R, a.x, wblock, g, wk, nchunks);
      if (ST > 2) ring_prologue_r4<NT, PC, ST, (ST > 2 ? 1 : 0)>(R, a.x, wblock, g, wk, nchunks);
      if (ST > 3) ring_prologue_r4<NT, PC, ST, (ST > 3 ? 2 : 0)>(R, a.x, wblock, g, wk, nchunks);
      LSFA_SCALES_READY(0)      // (behind the prologue's copies: their count depends on nchunks, so everything is waited for - chunk 0 is needed next anyway)
      int since = 0;
      for (int c = 0; c < nchunks; c += ST) {
        ring_step_r4<NT, PC, ST, 0, AF>(R, a.x, wblock, g, wk, c, nchunks, acc, a_scale, T);
        if (c + 1 < nchunks) ring_step_r4<NT, PC, ST, 1, AF>(R, a.x, wblock, g, wk, c + 1, nchunks, acc, a_scale, T);
        if (ST > 2 && c + 2 < nchunks) ring_step_r4<NT, PC, ST, (ST > 2 ? 2 : 0), AF>(R, a.x, wblock, g, wk, c + 2, nchunks, acc, a_scale, T);
        if (ST > 3 && c + 3 < nchunks) ring_step_r4<NT, PC, ST, (ST > 3 ? 3 : 0), AF>(R, a.x, wblock, g, wk, c + 3, nchunks, acc, a_scale, T);
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
  } else {
  if (SP && loader) {
    // v-chunk -1 (A(0) alone) into stage ST - 1, v-chunks 0 .. ST-2 into stages 0 .. ST-2, then one step per v-chunk: wait for it, meet,
    // refill the stage v-chunk v - 1 left
    ring_issue_a<NT, PC, ST, WV, ST - 1>(R, a.x, g, wk, true);
    wk.next(g.kh, g.kw, g.chunks_per_tap, g.k_order);
    ring_prologue<NT, PC, ST, WV, 0>(R, a.x, wblock, g, wk, nchunks);
    if (ST > 2) ring_prologue<NT, PC, ST, WV, (ST > 2 ? 1 : 0)>(R, a.x, wblock, g, wk, nchunks);
    if (ST > 3) ring_prologue<NT, PC, ST, WV, (ST > 3 ? 2 : 0)>(R, a.x, wblock, g, wk, nchunks);
    ring_wait_first<NT, PC, ST, WV>(nchunks);
    __builtin_amdgcn_s_barrier();
    for (int c = 0; c < nchunks; c += ST) {
      ring_load_step<NT, PC, ST, WV, 0>(R, a.x, wblock, g, wk, c, nchunks);
      if (c + 1 < nchunks) ring_load_step<NT, PC, ST, WV, 1>(R, a.x, wblock, g, wk, c + 1, nchunks);
      if (ST > 2 && c + 2 < nchunks) ring_load_step<NT, PC, ST, WV, (ST > 2 ? 2 : 0)>(R, a.x, wblock, g, wk, c + 2, nchunks);
      if (ST > 3 && c + 3 < nchunks) ring_load_step<NT, PC, ST, WV, (ST > 3 ? 3 : 0)>(R, a.x, wblock, g, wk, c + 3, nchunks);
    }
    return;                 // the epilogue is the consumers'
  }
  if (kUniform) {
    ring_issue_a<NT, PC, ST, WV, ST - 1>(R, a.x, g, wk, true);
    wk.next(g.kh, g.kw, g.chunks_per_tap, g.k_order);
    ring_prologue_uniform<NT, PC, ST, WV, 0>(R, a.x, wblock, g, wk, nchunks);
    if (ST > 2) ring_prologue_uniform<NT, PC, ST, WV, (ST > 2 ? 1 : 0)>(R, a.x, wblock, g, wk, nchunks);
    if (ST > 3) ring_prologue_uniform<NT, PC, ST, WV, (ST > 3 ? 2 : 0)>(R, a.x, wblock, g, wk, nchunks);
    wait_vmcnt<(ST - 1) * RG::kDma>();
    LSFA_SCALES_READY((ST - 1) * RG::kDma)      // (older than every copy: landed with A(0))
  } else {
    LSFA_SCALES_READY(0)                          // a consumer wave issues nothing else
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // AF: this wave's share of the table is written
  __builtin_amdgcn_s_barrier();                            // v-chunk -1 has landed
  Cut p = ring_cut<NT, PC, ST, WV, ST - 1, AF>(R, g, a_scale, g.chunk0, T);
  {
    int since = 0;
    for (int c = 0; c < nchunks; c += ST) {
      if (SP) {
        ring_consume_step<NT, PC, ST, WV, 0, AF>(R, g, acc, p, a_scale, c, T);
        if (c + 1 < nchunks) ring_consume_step<NT, PC, ST, WV, 1, AF>(R, g, acc, p, a_scale, c + 1, T);
        if (ST > 2 && c + 2 < nchunks) ring_consume_step<NT, PC, ST, WV, (ST > 2 ? 2 : 0), AF>(R, g, acc, p, a_scale, c + 2, T);
        if (ST > 3 && c + 3 < nchunks) ring_consume_step<NT, PC, ST, WV, (ST > 3 ? 3 : 0), AF>(R, g, acc, p, a_scale, c + 3, T);
      } else if (kUniform) {
        ring_step_uniform<NT, PC, ST, WV, 0, AF>(R, a.x, wblock, g, wk, c, nchunks, acc, p, a_scale, T);
        if (c + 1 < nchunks) ring_step_uniform<NT, PC, ST, WV, 1, AF>(R, a.x, wblock, g, wk, c + 1, nchunks, acc, p, a_scale, T);
        if (ST > 2 && c + 2 < nchunks) ring_step_uniform<NT, PC, ST, WV, (ST > 2 ? 2 : 0), AF>(R, a.x, wblock, g, wk, c + 2, nchunks, acc, p, a_scale, T);
        if (ST > 3 && c + 3 < nchunks) ring_step_uniform<NT, PC, ST, WV, (ST > 3 ? 3 : 0), AF>(R, a.x, wblock, g, wk, c + 3, nchunks, acc, p, a_scale, T);
      }
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
  }      // kPipelined
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[t][i] = (sum[t][i] + acc[t][i]) * out_scale[t];    // 1 unless PC == 2 (a power of two: exact)

  // C/D layout of 32x32: column = lane & 31 (channel), row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) (pixel)
  const int lane = g.lane;
  // Channels-last outputs leave through LDS: in the accumulator layout a lane holds ONE channel of 16 pixels, i.e. 4-byte accesses
  // (16 per tile and output, plus 16 residual loads: the epilogue of a conv3 - residual, sum, next bn1 / relu1 - was 7 of its 27 us).
  // The ring is free once every wave is past its last chunk: each wave writes its NT 32 x 32 tiles row-major into a private 4 KB x NT
  // region and reads them back as float4 along the channels - lane -> (row 8k + lane / 8, channels 4 (lane % 8) ..): whole 128-byte
  // rows per 8 lanes, 16 bytes per lane, a quarter of the memory instructions.  Same values, same arithmetic per element.
  if (kUniform) wait_vmcnt<0>();      // the copies issued past the end must have landed before the ring becomes the epilogue's staging area
  constexpr bool kRowsFit = RG::kLdsBytes >= WV * NT * 4096;       // the ring holds the waves' tiles
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
    if (!part) publish_amax(m, a.amax_out, a.status, blockIdx.x * WV + g.wave);
    return;
  }
  constexpr bool kColsFit = RG::kLdsBytes >= WV * NT * 32 * kColPitch * 4;       // ... or their padded column-major images
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
    publish_amax(m, a.amax_out, a.status, blockIdx.x * WV + g.wave);
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
  if (a.part) {      // a K slice whose ring is too small for the row path (128 x 128 one-piece tiles at two stages): partial sums, no epilogue
    float* part = a.part + (size_t)tile.z * P * a.Cout;
#pragma unroll
    for (int t = 0; t < NT; ++t) tile_store_part(part, a.Cout, prow, ro.valid, tile.y * (32 * NT) + t * 32 + (lane & 31), acc[t]);
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) ro.base[r] = ((ro.valid >> r) & 1u) ? out_pixel_base(a, prow[r]) : 0;
  uint32_t m = 0;
#pragma unroll
  for (int t = 0; t < NT; ++t) m = max(m, tile_store_max(a, ro, tile.y * (32 * NT) + t * 32 + (lane & 31), acc[t]));
  publish_amax(m, a.amax_out, a.status, blockIdx.x * WV + g.wave);
}

#undef LSFA_SCALES_READY

// weights (Cout, taps, Cin) fp32 -> fragment order, PC pieces.  One thread per (fragment, lane): 8 values.
// out index: ((((g * col_tiles + t) * 2 + s) * PC + piece) * 64 + lane) uint4, g = tap * (Cin/32) + chunk; PC = 2: values w * 2^w_exp
template <int PC>
static __global__ void pack_weights_kernel(const float* __restrict__ w, uint4* __restrict__ out, int Cout, int taps, int Cin, int w_exp,
                                           const int* __restrict__ w_exp_pc = nullptr) {
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
  const PiecesN p = cut8<PC>(v0, v1, ldexpf(1.f, w_exp_pc ? w_exp_pc[co] : w_exp));      // r5: one power of two per OUTPUT channel when given
  uint4* dst = out + ((((size_t)g * col_tiles + t) * 2 + s) * PC) * 64 + lane;
#pragma unroll
  for (int q = 0; q < PC; ++q) dst[q * 64] = p.p[q];
}

}  // namespace convsplit
}  // namespace lsfa
