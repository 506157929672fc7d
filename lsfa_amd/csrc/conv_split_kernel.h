// fp32 convolution / GEMM with every product formed on the 16-bit matrix pipe from PIECES of the fp32 operands (gfx950 has no xf32
// MFMA; this is the construction of 3xTF32 / BF16x9 GEMM emulation with the negligible terms left out).  Three forms, picked per
// weight (lsfa_conv_weights' `pieces`, hip.SplitWeight):
//   3  a = a1 + a2 + a3 EXACTLY in bf16 (8 + 8 + 8 mantissa bits, by truncation), likewise b; a*b ~= a1*b1 + (a1*b2 + a2*b1) +
//      (a1*b3 + a3*b1 + a2*b2): six v_mfma_f32_32x32x16_bf16, the dropped terms < 2^-23 |a*b|.  No scale needed.
//   2  a * 2^s = hi + lo in fp16 (11 + 11 bits; s from the map's maximum, `amax_in`, so that hi stays inside fp16's range; the weights
//      were packed as w * 2^w_exp); a*b ~= hi*hi' + hi*lo' + lo*hi': three v_mfma_f32_32x32x16_f16, the result scaled back by the
//      exact power of two.  The default of the fp32 path since r4.  A non-finite result (an under-estimated amax) raises the status word.
//   1  one bf16 piece, round-to-nearest-even: one instruction per product, the bf16 mode (BASELINE configs[2]).
// Each partial product of two 16-bit values is exact in fp32, so the only differences to an fp32 FMA chain are the dropped terms and
// the summation order; against a float64 convolution forms 3 and 2 are as close as an fp32 FMA chain (tests/test_hip_ops.py).
//
// What is in this file (the 128-pixel ring kernel, the main kernel of the family, lives in conv_ring_kernel.h):
//   Args                    the argument block every kernel of the family takes (filled by conv.hip from lsfa_conv_desc)
//   cut8 / mma_pc           cutting 8 fp32 values into pieces and the 1 / 3 / 6 matrix instructions of one k-step
//   amax plumbing           amax_exponent_asm (the scale from 256 partial maxima), publish_amax (an epilogue's maximum into the slots the
//                           next layer reads), the status word
//   epilogues               tile_store_max (fragment-shaped stores: NCHW, views), tile_rows_out (channels-last rows through LDS as
//                           float4), with bias / residual / activation / second output (the next unit's bn1 + relu1)
//   split_reduce_*          the pass that adds a K-sliced launch's partial sums in slice order and applies the epilogue
//   conv_split_direct_*     small weights on a small map: a wave per 32 x 64 tile, operands straight into registers
// Weights (B) are cut and laid out ONCE at bind time in fragment order (lsfa_conv_weights / pack_weights_kernel):
// [tap][chunk][32-col tile][k-step][piece][lane][8 x 16 bit], so a fragment is 1 KB contiguous.  One chunk = 32 input channels of one tap.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lsfa {
namespace convsplit {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kWavePix = 32, kWgPix = 128, kWgCh = 64, kChunk = 32, kThreads = 256;

struct Args {
  const float* x; const uint4* wfrag; const float* bias; float* y; float* part;
  int N, H, W, Cin, Cout, kh, kw, stride, pad_h, pad_w, dil, Ho, Wo, chunks_per_slice;
  int act;         // 0 none, 1 ReLU, 2 LeakyReLU(0.1) (FlowNet, resnet_v1_101_flownet_rfcn.py:153-176)
  const float* res; float* y2; const float* scale2; const float* shift2;
  int y_nchw;      // y / y2 / res are (N, Cout, Ho, Wo) instead of (N, Ho, Wo, Cout); partial slices stay NHWC
  // r3: operand views, so that producers write into (and consumers read from) channel slices of wider maps without copies
  //   lda   floats between consecutive input pixels (>= Cin; the general kernels only)
  //   ldy   floats between consecutive output pixels (>= Cout): y = channels [c0, c0 + Cout) of an (.., ldy) map, the
  //         pointer already advanced to c0.  view != 0 additionally places output pixel (oy, ox) of image n at
  //         ((n * out_H + oy * out_sy) * out_W + ox * out_sx) * ldy  (the pointer already advanced to the first one):
  //         the four phases of a stride-2 transposed convolution each write every other pixel of the full map.
  int lda, ldy, view, out_H, out_W, out_sy, out_sx;
  // r3: up to four launches in one (the four output parities of a stride-2 transposed convolution): workgroups are dealt
  // over (phase, K slice, channel tile, pixel tile); a phase has its own weights, padding, output grid and first output
  // element; its partial sums start at part + phase * slices * part_stride.  nphase <= 1: a plain launch.
  // nphase == 4: the launch is a Deconvolution(4x4, stride 2) + Crop(1) to out_H x out_W: phase = py * 2 + px has padding
  // (1 - py, 1 - px), output grid ((out_H - py + 1) / 2, (out_W - px + 1) / 2), first output pixel (py, px), and its weights
  // ph_wstride uint4 behind the previous phase's (everything a phase needs follows from (py, px): a table in the argument
  // struct would be indexed dynamically, which makes hipcc keep the struct in scratch)
  int nphase; long part_stride; long ph_wstride;
  // r3, fp16 two-piece form (conv_split_wide_kernel<NT, 2>): `amax` = kAmaxSlots partial maxima of |x| (lsfa_amax_partial), from which
  // every wave derives the power-of-two scale that puts x into fp16's range; the weights were packed as w * 2^w_exp
  const float* amax; int w_exp;
  // r5: per-OUTPUT-channel weight scales: wscale[co] = 2^-w_exp[co] (the weights were packed as w[co] * 2^w_exp[co], lsfa_conv_weights_pc), or
  // NULL = one scale 2^w_exp for the whole tensor.  A BatchNorm folded into a trained network's weights spreads the channels' magnitudes
  // over many octaves; with one scale per tensor the small channels' lo pieces went subnormal and those channels kept 11-15 of their 22
  // bits.  Per channel every output channel has all of them, and the factor comes back in the epilogue per column (a power of two: exact).
  const float* wscale;
  // r4: the epilogue (or the reduce pass of a K-sliced launch) leaves max|output| (of y2 when there is a second output) in
  // amax_out[kAmaxSlots] by atomicMax on the bit patterns (the caller zeroes the slots once per frame), so that the layer that
  // multiplies this output next needs no amax pass; a non-finite output raises bit 0 of *status (lsfa_status_check)
  unsigned* amax_out; unsigned* status;
  // r4: reciprocals of the divisors the kernels' index arithmetic uses (filled by the host; fdiv below): an integer division costs
  // ~35 dependent VALU instructions, and a short launch spent more time in its prologue's divisions than in its matrix instructions
  float inv_wo, inv_howo, inv_nx, inv_ny, inv_cpt, inv_kw;      // 1 / Wo, 1 / (Ho Wo), 1 / nx, 1 / ny, 1 / (Cin / 32), 1 / kw
  int tile_order;   // 0: tiles numbered (slice, channel tile, pixel tile), pixel fastest; 1: (slice, pixel tile, channel tile), channel fastest
  int k_order;      // r6, ring kernel: 0 = a tap's channel chunks before the next tap; 1 = a channel chunk's taps before the next chunk (Walk)
  // r4: the input is channels [0, Cin) of an NCHW map with `lda` channels (K-major for the contraction: element (pixel, channel) at
  // x[(n * lda + channel) * H * W + pixel]); 1x1 / stride 1 / no padding, the direct kernel only (the RPN head on the feature map the
  // reference's operators exchange)
  int x_kmajor;
  // r4: x is a pre-activation map and the convolution multiplies max(x * in_scale[k] + in_shift[k], 0) (k = input channel): the bn1 +
  // relu1 a ResNet unit applies to the previous unit's sum, applied where the operand is cut instead of being written out by the
  // previous conv3 as a second output and read back (a quarter of a unit's map traffic).  1x1, no padding, Cin <= kAffineMaxCin, the
  // ring kernel only.  `amax` is then the maximum of the ACTIVATED map: the previous conv3 publishes it without storing the map
  // (scale2 / shift2 given, y2 == NULL).
  const float* in_scale; const float* in_shift;
};
constexpr int kAffineMaxCin = 2048;

// n / d for 0 <= n < 2^24, 0 < d, with inv = 1.0f / d from the host: the float product is off by at most one, the remainder fixes it
__device__ __forceinline__ int fdiv(int n, int d, float inv) {
  int q = (int)((float)n * inv);
  const int r = n - q * d;
  q += (r >= d) ? 1 : 0;
  q -= (r < 0) ? 1 : 0;
  return q;
}

// the launch as phase `phase` sees it
__device__ __forceinline__ void apply_phase(Args& a, int phase, int slices) {
  if (a.nphase > 1) {
    const int py = phase >> 1, px = phase & 1;
    a.wfrag += (size_t)phase * a.ph_wstride;
    a.y += ((size_t)py * a.out_W + px) * a.ldy;
    a.pad_h = 1 - py; a.pad_w = 1 - px;
    a.Ho = (a.out_H - py + 1) / 2; a.Wo = (a.out_W - px + 1) / 2;
    a.inv_wo = 1.0f / (float)a.Wo; a.inv_howo = 1.0f / (float)(a.Ho * a.Wo);
    if (a.part) a.part += (size_t)phase * slices * a.part_stride;
  }
}

// the leading 8 mantissa bits of v as an fp32 bit pattern (= a bf16 value), and what is left
__device__ __forceinline__ uint32_t lead(float v) { return __float_as_uint(v) & 0xFFFF0000u; }
// two bf16 (upper halves of b's and a's bit patterns) -> one dword, a in the low half
__device__ __forceinline__ uint32_t pack_hi(uint32_t a, uint32_t b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

struct Pieces { uint4 p1, p2, p3; };     // 8 values x 3 pieces, packed bf16

__device__ __forceinline__ void cut3(float v, uint32_t& h, uint32_t& m, uint32_t& l) {
  h = lead(v);
  const float r1 = v - __uint_as_float(h);       // exact
  m = lead(r1);
  const float r2 = r1 - __uint_as_float(m);      // exact, at most 8 significant bits
  l = __float_as_uint(r2);
}

__device__ __forceinline__ Pieces split8(const float4& a, const float4& b) {
  uint32_t h[8], m[8], l[8];
  cut3(a.x, h[0], m[0], l[0]); cut3(a.y, h[1], m[1], l[1]); cut3(a.z, h[2], m[2], l[2]); cut3(a.w, h[3], m[3], l[3]);
  cut3(b.x, h[4], m[4], l[4]); cut3(b.y, h[5], m[5], l[5]); cut3(b.z, h[6], m[6], l[6]); cut3(b.w, h[7], m[7], l[7]);
  Pieces r;
  r.p1 = make_uint4(pack_hi(h[0], h[1]), pack_hi(h[2], h[3]), pack_hi(h[4], h[5]), pack_hi(h[6], h[7]));
  r.p2 = make_uint4(pack_hi(m[0], m[1]), pack_hi(m[2], m[3]), pack_hi(m[4], m[5]), pack_hi(m[6], m[7]));
  r.p3 = make_uint4(pack_hi(l[0], l[1]), pack_hi(l[2], l[3]), pack_hi(l[4], l[5]), pack_hi(l[6], l[7]));
  return r;
}

__device__ __forceinline__ bf16x8 as_bf(const uint4& u) {
  union { uint4 u; bf16x8 v; } c;
  c.u = u;
  return c.v;
}

__device__ __forceinline__ f32x16 mma(const uint4& a, const uint4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(a), as_bf(b), c, 0, 0, 0);
}

// acc += (a1 + a2 + a3) * (b1 + b2 + b3) without the three smallest terms, smallest kept terms first
__device__ __forceinline__ f32x16 mma6(const Pieces& a, const uint4& b1, const uint4& b2, const uint4& b3, f32x16 acc) {
  acc = mma(a.p3, b1, acc);
  acc = mma(a.p1, b3, acc);
  acc = mma(a.p2, b2, acc);
  acc = mma(a.p2, b1, acc);
  acc = mma(a.p1, b2, acc);
  acc = mma(a.p1, b1, acc);
  return acc;
}

// ---- r3 (opt-in): fp16 in TWO pieces, three matrix instructions per k-step ----------------------------------------------------------
// hi = fp16(x s), lo = fp16(x s - hi), s a power of two that puts max|x| into [2^13, 2^14): x s = hi + lo to 2^-24 relative (like fp32
// itself; below 2^-3 of the scaled range lo goes subnormal and the absolute error is 2^-25 of that range), and
// x y = hi hi + hi lo + lo hi + (lo lo <= 2^-24 |x y|, dropped).  Same accumulator, smallest terms first.  Against float64 this is as
// close as the bf16 three-piece form and closer (tools/lab/split_numerics.py), with half the matrix-pipe cycles, which is what the
// large convolutions are bound by.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
constexpr int kAmaxSlots = 256;

struct PiecesH { uint4 hi, lo; };

// two values at a time: gfx950's v_cvt_pk_f16_f32 rounds (to nearest even) and packs both in one instruction; the way back is
// one v_cvt_f32_f16 per half.  8 VALU instructions per pair where the scalar conversions + shifts + ors took 12.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void cut2h_pair(float v0, float v1, float s, uint32_t& h, uint32_t& l) {
  f32x2 xs;
  xs[0] = v0 * s;                                 // exact: s is a power of two
  xs[1] = v1 * s;
  const f16x2 hh = __builtin_convertvector(xs, f16x2);
  f32x2 r;
  r[0] = xs[0] - (float)hh[0];                    // exact (scalar subtractions: no packed fp32 math)
  r[1] = xs[1] - (float)hh[1];
  const f16x2 ll = __builtin_convertvector(r, f16x2);
  h = __builtin_bit_cast(uint32_t, hh);
  l = __builtin_bit_cast(uint32_t, ll);
}

__device__ __forceinline__ PiecesH split8h(const float4& a, const float4& b, float s) {
  PiecesH r;
  cut2h_pair(a.x, a.y, s, r.hi.x, r.lo.x);
  cut2h_pair(a.z, a.w, s, r.hi.y, r.lo.y);
  cut2h_pair(b.x, b.y, s, r.hi.z, r.lo.z);
  cut2h_pair(b.z, b.w, s, r.hi.w, r.lo.w);
  return r;
}

__device__ __forceinline__ f16x8 as_h(const uint4& u) {
  union { uint4 u; f16x8 v; } c;
  c.u = u;
  return c.v;
}

__device__ __forceinline__ f32x16 mma3h(const PiecesH& a, const uint4& bhi, const uint4& blo, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(a.lo), as_h(bhi), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(a.hi), as_h(blo), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(a.hi), as_h(bhi), acc, 0, 0, 0);
  return acc;
}

// ---- epilogue shared by the kernels below -----------------------------------------------------------------------
// Where output pixel p (= (n * Ho + oy) * Wo + ox of THIS launch) starts in y / y2 / res, and the distance between channels.
struct RowOut { int base[16]; unsigned valid; };

__device__ __forceinline__ int out_pixel_base(const Args& a, int p) {
  if (a.y_nchw) { const int hw = a.Ho * a.Wo, pn = fdiv(p, hw, a.inv_howo); return pn * a.Cout * hw + (p - pn * hw); }
  if (a.view) {
    const int hw = a.Ho * a.Wo, pn = fdiv(p, hw, a.inv_howo), r = p - pn * hw, oy = fdiv(r, a.Wo, a.inv_wo), ox = r - oy * a.Wo;
    return ((pn * a.out_H + oy * a.out_sy) * a.out_W + ox * a.out_sx) * a.ldy;
  }
  return p * a.ldy;
}

__device__ __forceinline__ float activate(float v, int act) {
  return act == 1 ? fmaxf(v, 0.f) : (act == 2 ? (v > 0.f ? v : v * 0.1f) : v);
}

// a K slice's partial sums: dense (P, Cout) rows, whatever the output view
__device__ __forceinline__ void tile_store_part(float* part, int Cout, const int (&p)[16], unsigned valid, int ch, const f32x16& acc) {
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if ((valid >> r) & 1u) part[(size_t)p[r] * Cout + ch] = acc[r];
}

// ---- r4: operand pieces generic in PC (pieces per fp32 operand) -----------------------------------------------------------------------
//   3 = three bf16 pieces, six products;  2 = two fp16 pieces + a power-of-two scale per map, three products;
//   1 = one bf16 piece (round to nearest even), one product: the bf16 mode of BASELINE configs[2]
struct PiecesN { uint4 p[3]; };

__device__ __forceinline__ uint32_t bf16_rne_pair(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(uint32_t, v);
}

template <int PC>
__device__ __forceinline__ PiecesN cut8(const float4& a, const float4& b, float s) {
  PiecesN r;
#ifdef LSFA_LAB_NO_CUT      // lab ablation (tools/lab/cut_ab.py): the raw bits as if they were pieces - wrong numbers, the step's timing without the cut
  r.p[0] = make_uint4(__float_as_uint(a.x), __float_as_uint(a.y), __float_as_uint(a.z), __float_as_uint(a.w));
  for (int q = 1; q < PC; ++q) r.p[q] = make_uint4(__float_as_uint(b.x), __float_as_uint(b.y), __float_as_uint(b.z), __float_as_uint(b.w));
  return r;
#endif
  if (PC == 3) {
    const Pieces p = split8(a, b);
    r.p[0] = p.p1; r.p[1] = p.p2; r.p[2] = p.p3;
  } else if (PC == 2) {
    const PiecesH p = split8h(a, b, s);
    r.p[0] = p.hi; r.p[1] = p.lo;
  } else {
    r.p[0] = make_uint4(bf16_rne_pair(a.x, a.y), bf16_rne_pair(a.z, a.w), bf16_rne_pair(b.x, b.y), bf16_rne_pair(b.z, b.w));
  }
  return r;
}

__device__ __forceinline__ float4 as_f4(const uint4& r) {
  return make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
}

// acc += A * B for one k-step; b0, b1, b2 = the (up to) PC fragments of (column tile, step) in piece order
template <int PC>
__device__ __forceinline__ f32x16 mma_pc(const PiecesN& a, const uint4& b0, const uint4& b1, const uint4& b2, f32x16 acc) {
  if (PC == 3) {
    Pieces p; p.p1 = a.p[0]; p.p2 = a.p[1]; p.p3 = a.p[2];
    return mma6(p, b0, b1, b2, acc);
  } else if (PC == 2) {
    PiecesH p; p.hi = a.p[0]; p.lo = a.p[1];
    return mma3h(p, b0, b1, acc);
  }
  return mma(a.p[0], b0, acc);
}
// ... with the fragments 64 uint4 apart in LDS
template <int PC>
__device__ __forceinline__ f32x16 mma_pc(const PiecesN& a, const uint4* B, f32x16 acc) {
  return mma_pc<PC>(a, B[0], B[PC > 1 ? 64 : 0], B[PC > 2 ? 128 : 0], acc);
}

// the accumulator columns' output scales: 2^-(s_exp + w_exp[ch]) for the NT columns ch0 + 32 t + (lane & 31) this lane holds.  The
// per-channel factors are read by inline assembly with their own wait, like the amax slots (no compiler-visible load may be pending
// when a DMA ring starts).
template <int NT>
__device__ __forceinline__ void column_scales(const Args& a, int s_exp, int ch0, int lane, float (&os)[NT]) {
  if (!a.wscale) {
    const float v = ldexpf(1.f, -(s_exp + a.w_exp));
#pragma unroll
    for (int t = 0; t < NT; ++t) os[t] = v;
    return;
  }
  const float* p = a.wscale + ch0 + (lane & 31);
  const float inv = ldexpf(1.f, -s_exp);
  static_assert(NT == 2 || NT == 4, "two or four column tiles per wave");
  if (NT == 2) {
    asm volatile("global_load_dword %0, %2, off\n\t"
                 "global_load_dword %1, %2, off offset:128\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(os[0]), "=&v"(os[1]) : "v"(p) : "memory");
  } else {
    asm volatile("global_load_dword %0, %4, off\n\t"
                 "global_load_dword %1, %4, off offset:128\n\t"
                 "global_load_dword %2, %4, off offset:256\n\t"
                 "global_load_dword %3, %4, off offset:384\n\t"
                 "s_waitcnt vmcnt(0)"
                 : "=&v"(os[0]), "=&v"(os[1]), "=&v"(os[NT > 2 ? 2 : 0]), "=&v"(os[NT > 3 ? 3 : 0]) : "v"(p) : "memory");
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) os[t] = os[t] * inv;
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// r6: the two round trips at the head of every two-piece launch (the 256 amax slots, then the per-channel weight scales) as ONE, issued first
// thing and waited for only when the scale is needed - behind the geometry set-up and, in the mixed-role kernels, behind the prologue's
// copies, whose latency then covers it.  scale_loads_issue: the loads, no wait.  scale_loads_wait<N>: `s_waitcnt vmcnt(N)` with every
// destination register as an in/out operand, so that no use of them can be scheduled above it (N = vector-memory operations the wave has
// issued SINCE: they return in order, so "at most N outstanding" means these have landed).  scale_finish: the arithmetic of
// amax_exponent_asm + column_scales.  Same values, same bits.
struct ScaleRegs { uint32_t m0, m1, m2, m3; float w0, w1, w2, w3; };

template <int NT>
__device__ __forceinline__ void scale_loads_issue(const Args& a, int ch0, int lane, ScaleRegs& r) {
  static_assert(kAmaxSlots == 256, "four slots per lane");
  static_assert(NT == 2 || NT == 4, "two or four column tiles per wave");
  const float* p = a.amax + lane;
  asm volatile("global_load_dword %0, %4, off\n\t"
               "global_load_dword %1, %4, off offset:256\n\t"
               "global_load_dword %2, %4, off offset:512\n\t"
               "global_load_dword %3, %4, off offset:768"
               : "=&v"(r.m0), "=&v"(r.m1), "=&v"(r.m2), "=&v"(r.m3) : "v"(p) : "memory");
  r.w0 = r.w1 = r.w2 = r.w3 = 1.f;
  if (a.wscale) {
    const float* q = a.wscale + ch0 + (lane & 31);
    if (NT == 2) {
      asm volatile("global_load_dword %0, %2, off\n\t"
                   "global_load_dword %1, %2, off offset:128"
                   : "=&v"(r.w0), "=&v"(r.w1) : "v"(q) : "memory");
    } else {
      asm volatile("global_load_dword %0, %4, off\n\t"
                   "global_load_dword %1, %4, off offset:128\n\t"
                   "global_load_dword %2, %4, off offset:256\n\t"
                   "global_load_dword %3, %4, off offset:384"
                   : "=&v"(r.w0), "=&v"(r.w1), "=&v"(r.w2), "=&v"(r.w3) : "v"(q) : "memory");
    }
  }
}

template <int N>
__device__ __forceinline__ void scale_loads_wait(ScaleRegs& r) {
  asm volatile("s_waitcnt vmcnt(%8)"
               : "+v"(r.m0), "+v"(r.m1), "+v"(r.m2), "+v"(r.m3), "+v"(r.w0), "+v"(r.w1), "+v"(r.w2), "+v"(r.w3) : "n"(N) : "memory");
}

// -> a_scale = 2^s_exp (the map's maximum into [2^13, 2^14)), os[t] = 2^-(s_exp + w_exp of this lane's column of tile t)
template <int NT>
__device__ __forceinline__ float scale_finish(const Args& a, const ScaleRegs& r, int lane, float (&os)[NT]) {
  uint32_t m = max(max(r.m0 & 0x7FFFFFFFu, r.m1 & 0x7FFFFFFFu), max(r.m2 & 0x7FFFFFFFu, r.m3 & 0x7FFFFFFFu));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
  const int e = (int)((m >> 23) & 255u);
  if (e == 255 && a.status && lane == 0) atomicOr(a.status, 2u);      // the INPUT map already holds inf / NaN
  const int s_exp = 13 - ((e == 0 || e == 255) ? 0 : __builtin_amdgcn_readfirstlane(e - 127));
  if (!a.wscale) {
    const float v = ldexpf(1.f, -(s_exp + a.w_exp));
#pragma unroll
    for (int t = 0; t < NT; ++t) os[t] = v;
  } else {
    const float inv = ldexpf(1.f, -s_exp);
    const float w[4] = {r.w0, r.w1, r.w2, r.w3};
#pragma unroll
    for (int t = 0; t < NT; ++t) os[t] = w[t] * inv;
  }
  return ldexpf(1.f, s_exp);
}

// the fp16 form's scale from the kAmaxSlots partial maxima (floats or bit patterns of |x|: the same thing for non-negative values),
// read by inline assembly with its own wait so that no compiler-visible vector load is pending when a DMA ring starts.
// -> floor(log2(max)), 0 for an all-zero map; an inf / NaN maximum raises bit 1 of *status
__device__ __forceinline__ int amax_exponent_asm(const float* amax, int lane, unsigned* status) {
  static_assert(kAmaxSlots == 256, "four slots per lane");
  const float* p = amax + lane;
  uint32_t m0, m1, m2, m3;
  asm volatile("global_load_dword %0, %4, off\n\t"
               "global_load_dword %1, %4, off offset:256\n\t"
               "global_load_dword %2, %4, off offset:512\n\t"
               "global_load_dword %3, %4, off offset:768\n\t"
               "s_waitcnt vmcnt(0)"
               : "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
               : "v"(p)
               : "memory");
  uint32_t m = max(max(m0 & 0x7FFFFFFFu, m1 & 0x7FFFFFFFu), max(m2 & 0x7FFFFFFFu, m3 & 0x7FFFFFFFu));
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
  const int e = (int)((m >> 23) & 255u);
  if (e == 255 && status && lane == 0) atomicOr(status, 2u);      // the INPUT map already holds inf / NaN
  return (e == 0 || e == 255) ? 0 : __builtin_amdgcn_readfirstlane(e - 127);
}

// a wave's largest |v| bit pattern -> a slot; non-finite -> status bit 0
__device__ __forceinline__ void publish_amax(uint32_t m, unsigned* amax_out, unsigned* status, int slot) {
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
  if ((threadIdx.x & 63) == 0) {
    if (amax_out) atomicMax(amax_out + (slot & (kAmaxSlots - 1)), m);
    if (m >= 0x7F800000u && status) atomicOr(status, 1u);
  }
}

// |v| as bits, with a non-finite PRE-activation value kept visible (ReLU turns NaN and -inf into 0)
__device__ __forceinline__ uint32_t amax_bits(float pre, float post) {
  const uint32_t b = __float_as_uint(pre) & 0x7FFFFFFFu;
  return b >= 0x7F800000u ? b : (__float_as_uint(post) & 0x7FFFFFFFu);
}

// tile_store that also returns the maximum of what it wrote (of y2 when there is a second output: that is what the next layer multiplies)
__device__ __forceinline__ uint32_t tile_store_max(const Args& a, const RowOut& ro, int ch, const f32x16& acc) {
  const int cs = a.y_nchw ? a.Ho * a.Wo : 1;
  const float bias = a.bias ? a.bias[ch] : 0.f;
  float v[16], pre[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = acc[r] + bias;
  if (a.res) {
    float rv[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) rv[r] = ((ro.valid >> r) & 1u) ? a.res[ro.base[r] + ch * cs] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = v[r] + rv[r];
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) { pre[r] = v[r]; v[r] = activate(v[r], a.act); }
  uint32_t m = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if ((ro.valid >> r) & 1u) { a.y[ro.base[r] + ch * cs] = v[r]; m = max(m, amax_bits(pre[r], a.scale2 ? 0.f : v[r])); }
  if (a.scale2) {        // y2 == NULL: only the maximum of the second output is wanted
    const float sc2 = a.scale2[ch], sh2 = a.shift2[ch];
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((ro.valid >> r) & 1u) {
        const float w = fmaxf(v[r] * sc2 + sh2, 0.f);
        if (a.y2) a.y2[ro.base[r] + ch * cs] = w;
        m = max(m, __float_as_uint(w) & 0x7FFFFFFFu);
      }
  }
  return m;
}

// Channels-last outputs leave through LDS.  In the accumulator layout a lane holds ONE channel of 16 pixels, i.e. 4-byte accesses
// (16 per tile and output, plus 16 residual loads: the epilogue of a conv3 - residual, sum, next bn1 / relu1 - was 7 of its 27 us).
// T holds a wave's NT 32 x 32 tiles row-major ([tile][row][32 channels], written from the accumulators by the caller); here they
// are read back as float4 along the channels - lane -> (row 8k + lane / 8, channels 4 (lane % 8) ..): whole 128-byte rows per
// 8 lanes, 16 bytes per lane, a quarter of the memory instructions.  Same values, same arithmetic per element as tile_store.
// m0 = first pixel of the 32 rows, ch0 = first channel of tile 0; part != NULL: a K slice's partial sums instead of the epilogue.
// -> the wave's contribution to amax_out (bit pattern; 0x7FC00000 if a non-finite value went through this lane)
// Every load of the epilogue is issued before its first store: the residual is usually the output itself (a ResNet unit adds in place),
// so the compiler must assume that a store may change what a later load reads and would run the 4 x NT (load, add, store) steps one
// memory latency after the other - 16 round trips for a 128 x 128 tile, most of a conv3's time (res4 conv3 at six images: 74 us for a
// main loop of 8 chunks).  A lane reads and writes the same addresses, no other lane touches them: reading them all first is safe.
template <int NT> struct RowsIn { float4 rr[4][NT], bb[NT], s2[NT], h2[NT]; int base[4]; bool ok[4]; };

// the loads of tile_rows_out, issued by the caller before it turns its accumulators through LDS (their latency runs meanwhile)
template <int NT>
__device__ __forceinline__ void tile_rows_in(const Args& a, int m0, int P, int ch0, bool part, int lane, RowsIn<NT>& in) {
  const int c4 = (lane & 7) * 4;
  const bool has_y2 = a.scale2 != nullptr, has_res = a.res != nullptr, has_bias = a.bias != nullptr;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int p = m0 + 8 * k + (lane >> 3);
    in.ok[k] = p < P;
    in.base[k] = (in.ok[k] && !part) ? out_pixel_base(a, p) : 0;
  }
  if (part) return;
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int ch = ch0 + t * 32 + c4;
    in.bb[t] = has_bias ? *reinterpret_cast<const float4*>(a.bias + ch) : zero4;
    in.s2[t] = has_y2 ? *reinterpret_cast<const float4*>(a.scale2 + ch) : zero4;
    in.h2[t] = has_y2 ? *reinterpret_cast<const float4*>(a.shift2 + ch) : zero4;
  }
  if (has_res) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int t = 0; t < NT; ++t) in.rr[k][t] = *reinterpret_cast<const float4*>(a.res + in.base[k] + ch0 + t * 32 + c4);      // base 0 for a row past the end: a valid address, not used
  }
}

template <int NT>
__device__ __forceinline__ uint32_t tile_rows_out(const Args& a, const float* T, int m0, int P, int ch0, float* part, int lane, const RowsIn<NT>& in) {
  const int c4 = (lane & 7) * 4;
  // the maximum as a float maximum of |.| (one instruction per value; a NaN drops out of it) and, beside it, the SUM of the
  // pre-activation magnitudes, which is non-finite exactly when one of them is (ReLU would hide a NaN or a -inf): one add per value
  float mx = 0.f, nf = 0.f;
  const int act = a.act;                         // wave-uniform: the branches below are scalar
  const bool has_y2 = a.scale2 != nullptr, store_y2 = a.y2 != nullptr, has_res = a.res != nullptr, has_bias = a.bias != nullptr;
  if (part) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int row = 8 * k + (lane >> 3);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const float4 v = *reinterpret_cast<const float4*>(&T[t * 1024 + row * 32 + c4]);
        if (in.ok[k]) *reinterpret_cast<float4*>(part + (size_t)(m0 + row) * a.Cout + ch0 + t * 32 + c4) = v;
      }
    }
    return 0u;
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int row = 8 * k + (lane >> 3);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int ch = ch0 + t * 32 + c4;
      const float4 v = *reinterpret_cast<const float4*>(&T[t * 1024 + row * 32 + c4]);
      if (!in.ok[k]) continue;
      float4 o = v;
      if (has_bias) { o.x = o.x + in.bb[t].x; o.y = o.y + in.bb[t].y; o.z = o.z + in.bb[t].z; o.w = o.w + in.bb[t].w; }
      if (has_res) { o.x = o.x + in.rr[k][t].x; o.y = o.y + in.rr[k][t].y; o.z = o.z + in.rr[k][t].z; o.w = o.w + in.rr[k][t].w; }
      nf = nf + fabsf(o.x); nf = nf + fabsf(o.y); nf = nf + fabsf(o.z); nf = nf + fabsf(o.w);
      if (act == 1) { o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f); }
      else if (act == 2) { o.x = activate(o.x, 2); o.y = activate(o.y, 2); o.z = activate(o.z, 2); o.w = activate(o.w, 2); }
      *reinterpret_cast<float4*>(a.y + in.base[k] + ch) = o;
      if (has_y2) {
        float4 w;
        w.x = fmaxf(o.x * in.s2[t].x + in.h2[t].x, 0.f); w.y = fmaxf(o.y * in.s2[t].y + in.h2[t].y, 0.f);
        w.z = fmaxf(o.z * in.s2[t].z + in.h2[t].z, 0.f); w.w = fmaxf(o.w * in.s2[t].w + in.h2[t].w, 0.f);
        if (store_y2) *reinterpret_cast<float4*>(a.y2 + in.base[k] + ch) = w;
        mx = fmaxf(fmaxf(mx, fmaxf(w.x, w.y)), fmaxf(w.z, w.w));
      } else {
        mx = fmaxf(fmaxf(mx, fmaxf(fabsf(o.x), fabsf(o.y))), fmaxf(fabsf(o.z), fabsf(o.w)));
      }
    }
  }
  uint32_t m = __float_as_uint(mx);
  if ((__float_as_uint(nf) & 0x7F800000u) == 0x7F800000u) m = 0x7FC00000u;       // a non-finite value went through this lane
  return m;
}

// can a channels-last output take the float4 row path (alignment of the operands it touches)
// NCHW outputs leave through LDS too (r4).  In the accumulator layout a store instruction puts 32 channels of one pixel into 32
// different planes: 4-byte pieces of 32 cache lines, 16 x NT times per wave (fuse_reduce_add writes 88 MB per nine-frame segment that
// way).  Tc holds a wave's NT tiles COLUMN-major with padded columns ([tile][channel][33]: written from the accumulators without bank
// conflicts by the caller); here lane -> (pixel m0 + lane % 32, channel c + lane / 32): an instruction stores two 128-byte runs of two
// planes.  Same values, same arithmetic per element as tile_store_max (bias, residual, activation, second output, maximum, the
// non-finite check on the pre-activation value).
constexpr int kColPitch = 33;
template <int NT>
__device__ __forceinline__ uint32_t tile_cols_out_nchw(const Args& a, const float* Tc, int m0, int P, int ch0, int lane) {
  const int p = m0 + (lane & 31);
  const bool ok = p < P;
  const int hw = a.Ho * a.Wo;
  const int base = ok ? out_pixel_base(a, p) : 0;       // (n * Cout) * hw + the pixel's offset in its plane
  const int act = a.act;
  const bool has_y2 = a.scale2 != nullptr, store_y2 = a.y2 != nullptr, has_res = a.res != nullptr, has_bias = a.bias != nullptr;
  float mx = 0.f, nf = 0.f;
  // per 32-channel tile: every load (bias, residual, the second output's affine: 16 channels per lane) before the first store - the
  // compiler cannot know that the stores leave them alone and would otherwise wait for a load in each of the 16 x NT steps
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    float bb[16], rr[16], s2[16], h2[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int ch = ch0 + t * 32 + 2 * j + (lane >> 5);
      bb[j] = has_bias ? a.bias[ch] : 0.f;
      rr[j] = has_res ? a.res[base + ch * hw] : 0.f;        // base 0 for a pixel past the end: a valid address, not used
      s2[j] = has_y2 ? a.scale2[ch] : 0.f;
      h2[j] = has_y2 ? a.shift2[ch] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int col = 2 * j + (lane >> 5), ch = ch0 + t * 32 + col;
      float o = Tc[(t * 32 + col) * kColPitch + (lane & 31)];
      if (!ok) continue;
      if (has_bias) o = o + bb[j];
      if (has_res) o = o + rr[j];
      nf = nf + fabsf(o);
      o = activate(o, act);
      a.y[base + ch * hw] = o;
      if (has_y2) {
        const float w = fmaxf(o * s2[j] + h2[j], 0.f);
        if (store_y2) a.y2[base + ch * hw] = w;
        mx = fmaxf(mx, w);
      } else {
        mx = fmaxf(mx, fabsf(o));
      }
    }
  }
  uint32_t m = __float_as_uint(mx);
  if ((__float_as_uint(nf) & 0x7F800000u) == 0x7F800000u) m = 0x7FC00000u;       // a non-finite value went through this lane
  return m;
}

__device__ __forceinline__ bool rows_path_ok(const Args& a) {
  return a.part || (!a.y_nchw && (a.ldy & 3) == 0 && ((uintptr_t)a.y & 15) == 0 && (!a.y2 || ((uintptr_t)a.y2 & 15) == 0) &&
                    (!a.res || ((uintptr_t)a.res & 15) == 0));
}

// sum of the K slices in slice order (reproducible), then the same tail as tile_store; a float4 of channels per thread
static __global__ __launch_bounds__(kThreads) void split_reduce_kernel(Args a, long n4, int slices) {
  if (a.nphase > 1) { apply_phase(a, blockIdx.y, slices); n4 = (long)a.N * a.Ho * a.Wo * a.Cout / 4; }
  const long i = (long)blockIdx.x * kThreads + threadIdx.x;
  uint32_t m = 0;
  if (i < n4) {
    const float4* part = reinterpret_cast<const float4*>(a.part);
    float4 s = part[i];
    int z = 1;
    for (; z + 4 <= slices; z += 4) {       // four slices' loads in flight together; the additions stay in slice order
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = part[(size_t)(z + u) * n4 + i];
#pragma unroll
      for (int u = 0; u < 4; ++u) { s.x = s.x + v[u].x; s.y = s.y + v[u].y; s.z = s.z + v[u].z; s.w = s.w + v[u].w; }
    }
    for (; z < slices; ++z) {
      const float4 v = part[(size_t)z * n4 + i];
      s.x = s.x + v.x; s.y = s.y + v.y; s.z = s.z + v.z; s.w = s.w + v.w;
    }
    const int c4 = a.Cout / 4;
    const int p = (int)(i / c4), ch = (int)(i - (long)p * c4) * 4;
    const int base = out_pixel_base(a, p), cs = a.y_nchw ? a.Ho * a.Wo : 1;
    float o1[4] = {s.x, s.y, s.z, s.w}, o2[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v = o1[k] + (a.bias ? a.bias[ch + k] : 0.f);
      if (a.res) v = v + a.res[base + (ch + k) * cs];
      const float pre = v;
      v = activate(v, a.act);
      o1[k] = v;
      o2[k] = a.scale2 ? fmaxf(v * a.scale2[ch + k] + a.shift2[ch + k], 0.f) : 0.f;
      m = max(m, max(amax_bits(pre, a.scale2 ? 0.f : v), __float_as_uint(o2[k]) & 0x7FFFFFFFu));
    }
    if (cs == 1 && ((base + ch) & 3) == 0 && ((uintptr_t)a.y & 15) == 0 && (!a.y2 || ((uintptr_t)a.y2 & 15) == 0)) {
      *reinterpret_cast<float4*>(a.y + base + ch) = make_float4(o1[0], o1[1], o1[2], o1[3]);
      if (a.y2) *reinterpret_cast<float4*>(a.y2 + base + ch) = make_float4(o2[0], o2[1], o2[2], o2[3]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        a.y[base + (ch + k) * cs] = o1[k];
        if (a.y2) a.y2[base + (ch + k) * cs] = o2[k];
      }
    }
  }
  publish_amax(m, a.amax_out, a.status, blockIdx.x * 4 + (threadIdx.x >> 6));
}

// The same for an NCHW output without residual / second output (feat_conv_3x3, fuse_reduce_add): split_reduce_kernel's lanes walk
// the channels of ONE pixel, so its NCHW stores are 64 different channel planes per instruction.  Here a workgroup sums a tile of
// 64 pixels x 64 channels (reads: float4s of channels, coalesced along the partial sums' rows), turns it in LDS and writes, per
// channel, 64 consecutive pixels.  Same additions in the same order, same epilogue arithmetic: bit-identical to split_reduce_kernel.
// grid (ceil(P / 64), Cout / 64)
static __global__ __launch_bounds__(kThreads) void split_reduce_nchw_kernel(Args a, int slices) {
  __shared__ float T[64][65];
  const int P = a.N * a.Ho * a.Wo, HW = a.Ho * a.Wo;
  const int p0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  const int t = threadIdx.x;
  const int cg = t & 15, row = t >> 4;
  const int ch = c0 + 4 * cg;
  const size_t n4 = (size_t)P * a.Cout / 4;
  const float4* part = reinterpret_cast<const float4*>(a.part);
  float bias[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) bias[k] = a.bias ? a.bias[ch + k] : 0.f;
  // slice 0 of the thread's four pixels first, then the other slices slice by slice with the four loads of a slice in flight together
  // (the additions stay in slice order per element)
  float4 sum[4];
  size_t idx[4];
  bool ok[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = p0 + row + 16 * j;
    ok[j] = p < P;
    idx[j] = ((size_t)(ok[j] ? p : 0) * a.Cout + ch) / 4;
    sum[j] = part[idx[j]];
  }
  for (int z = 1; z < slices; ++z) {
    float4 v[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = part[(size_t)z * n4 + idx[j]];
#pragma unroll
    for (int j = 0; j < 4; ++j) { sum[j].x = sum[j].x + v[j].x; sum[j].y = sum[j].y + v[j].y; sum[j].z = sum[j].z + v[j].z; sum[j].w = sum[j].w + v[j].w; }
  }
  uint32_t m = 0;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int pl = row + 16 * j;
    const float pre[4] = {sum[j].x + bias[0], sum[j].y + bias[1], sum[j].z + bias[2], sum[j].w + bias[3]};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float v = activate(pre[k], a.act);
      T[4 * cg + k][pl] = v;
      if (ok[j]) m = max(m, amax_bits(pre[k], v));
    }
  }
  publish_amax(m, a.amax_out, a.status, (blockIdx.y * gridDim.x + blockIdx.x) * 4 + (t >> 6));
  __syncthreads();
  // a wave writes 64 consecutive pixels of one channel per instruction (256 contiguous bytes), 16 channels in turn
  const int lane = t & 63, wv = t >> 6;
  const int p = p0 + lane;
  if (p < P) {
    const int pn = p / HW, r = p - pn * HW;
    float* dst = a.y + ((size_t)pn * a.Cout + c0) * HW + r;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int cl = wv * 16 + q;
      dst[(size_t)cl * HW] = T[cl][lane];
    }
  }
}

__device__ __attribute__((aligned(64))) const float g_zero_block[32] = {};

struct Geom {       // per lane / per wave constants of the loop
  size_t wstride;                            // uint4 between consecutive chunks
  int H, W, Cin, lda, kh, kw, stride, dil, chunks_per_tap, chunk0, k_order;
  // the four pixels this lane moves (DMA i: pixel 8*i + (lane >> 3)): top-left input coordinate of its window
  // (hugely negative when the pixel does not exist) and the float offset of (that coordinate, the lane's swizzled piece)
  int iy0[4], ix0[4], off0[4];
  int frag[4];                               // uint4 index, inside the wave's A image, of the lane's four fragment pieces
  int wave, lane;
};

// the 7 LDS-DMA instructions of one chunk into stage ST (compile-time LDS offsets).  x / wblock (this workgroup's
// 12 KB of chunk 0) are __restrict__ parameters on purpose (see the header comment).
// which (tap row, tap column, channel chunk) a chunk index is, walked incrementally (an integer division per chunk
// cost ~50 scalar instructions in the loop)
// r6: two orders.  k_order 0 walks a tap's channel chunks before the next tap (weights are packed that way: block (tap, chunk) at index
// tap * chunks_per_tap + chunk); k_order 1 walks the taps of one channel chunk before the next chunk, so that a tap's rows are the previous
// tap's rows shifted by `dil` pixels (or rows) and come out of the CU's L1 instead of L2.  `bidx` is the weight block of the walk's
// position, `bprev` that of the position before it (the pipelined ring holds B(v) beside A(v + 1)); both clamped to the array.
struct Walk {
  int ty, tx, kc, gch;
  int bidx, bprev;
  __device__ __forceinline__ void start(int pos, int kh, int kw, int chunks_per_tap, int k_order) {
    gch = pos;
    const int taps = kh * kw;
    int tap;
    // once per workgroup (pos = 0 unless K is sliced); fdiv with approximate reciprocals: its fix-up absorbs their last-bit error
    if (k_order) { kc = fdiv(pos, taps, __builtin_amdgcn_rcpf((float)taps)); tap = pos - kc * taps; }
    else { tap = fdiv(pos, chunks_per_tap, __builtin_amdgcn_rcpf((float)chunks_per_tap)); kc = pos - tap * chunks_per_tap; }
    ty = fdiv(tap, kw, __builtin_amdgcn_rcpf((float)kw)); tx = tap - ty * kw;
    bidx = bprev = min(tap * chunks_per_tap + kc, taps * chunks_per_tap - 1);
  }
  __device__ __forceinline__ void next(int kh, int kw, int chunks_per_tap, int k_order) {
    ++gch;
    bprev = bidx;
    if (k_order) { if (++tx == kw) { tx = 0; if (++ty == kh) { ty = 0; ++kc; } } }
    else { if (++kc == chunks_per_tap) { kc = 0; if (++tx == kw) { tx = 0; ++ty; } } }
    bidx = min((ty * kw + tx) * chunks_per_tap + kc, kh * kw * chunks_per_tap - 1);
  }
};

// Workgroup -> tile, XCD-aware.  The hardware deals consecutive workgroup ids round-robin to the 8 XCDs (id % 8),
// each with its own 4 MB L2.  Tiles are numbered (slice, channel tile, pixel tile) with the pixel tile fastest and
// XCD k takes the k-th eighth of that list, so the workgroups that share an L2 share their weights (one or two
// (slice, channel tile) blocks, a few hundred KB) and sweep the activations once — instead of every XCD streaming
// all weights AND all activations through its L2 (6 MB for a res4 conv2: it does not fit, and the misses go to the
// Infinity Cache).
struct Tile { int x, y, z; };
__device__ __forceinline__ Tile xcd_tile(int id, int nx, int ny, int nz, int order = 0, float inv_nx = 0.f, float inv_ny = 0.f) {
  const int total = nx * ny * nz;
  const int per = (total + 7) >> 3;
  const int t = (id & 7) * per + (id >> 3);
  Tile r;
  if (t >= total) { r.x = r.y = r.z = -1; return r; }      // the grid is 8 * per >= total: the surplus ids have no tile
  const bool fast = inv_nx > 0.f && inv_ny > 0.f && total < (1 << 24);
  if (order == 1) {      // channel tile fastest: the workgroups of one XCD are ALL channel tiles of an eighth of the (slice, pixel tile) list
    const int q = fast ? fdiv(t, ny, inv_ny) : t / ny;
    r.y = t - q * ny;
    r.z = fast ? fdiv(q, nx, inv_nx) : q / nx;
    r.x = q - r.z * nx;
    return r;
  }
  const int q = fast ? fdiv(t, nx, inv_nx) : t / nx;
  r.x = t - q * nx;
  r.z = fast ? fdiv(q, ny, inv_ny) : q / ny;
  r.y = q - r.z * ny;
  return r;
}

// ---------------------------------------------------------------------------------------------------------------
// r3: the DIRECT form, for convolutions whose weights are small (the small net's units on the 38 x 63 map: 64 -> 64 3x3, 64 / 256 ->
// 64 and 64 -> 256 1x1: 0.08 - 0.35 GFLOP each, run on EVERY non-key frame).  The tiled kernels above need 128-pixel tiles and K
// slices (+ a reduce pass) to occupy the chip and then pay ~1 us of DMA latency per chunk: 16 + 8 us for 0.35 GFLOP.  Here a
// wave owns a 32-pixel x 64-channel output tile and loads its operands straight into registers in fragment shape — A: the lane's
// pixel, 16 consecutive channels (4 x 16 B); B: 12 fragments of 1 KB, coalesced — with the next chunk's 16 loads in flight while
// the current one is cut and multiplied; no LDS staging, no barrier in the loop.  The K range (taps x chunks) of a tile is dealt
// to the waves of the workgroup in contiguous runs; their accumulators meet in LDS and are added in wave order
// (reproducible), then the shared epilogue (bias / residual / activation / second output).  Weights are re-read by every pixel
// tile (75 x 216 KB from L2 for the 3x3): fine for small weights, which is what the launch plan checks.

template <int PC> struct DirectOperands { uint4 a[4]; uint4 b[4 * PC]; };

template <int PC>
__device__ __forceinline__ void direct_load(DirectOperands<PC>& o, const float* __restrict__ x, const uint4* __restrict__ wtile, size_t wstride,
                                            int gch, int a_off, bool a_ok, int lane, int kstride = 0) {
  if (kstride) {
    // K-major input: the lane's 16 channels are 16 planes apart by `kstride` floats (a wave instruction reads 32 consecutive pixels of
    // two planes: coalesced 128-byte runs)
    const float* xp = a_ok ? x + a_off : reinterpret_cast<const float*>(g_zero_block);
    const int ks = a_ok ? kstride : 0;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      o.a[j] = make_uint4(__float_as_uint(xp[(4 * j + 0) * ks]), __float_as_uint(xp[(4 * j + 1) * ks]), __float_as_uint(xp[(4 * j + 2) * ks]),
                          __float_as_uint(xp[(4 * j + 3) * ks]));
  } else {
  const uint4* ap = reinterpret_cast<const uint4*>(a_ok ? x + a_off : g_zero_block);
#pragma unroll
  for (int j = 0; j < 4; ++j) o.a[j] = a_ok ? ap[j] : make_uint4(0u, 0u, 0u, 0u);
  }
  const uint4* bp = wtile + (size_t)gch * wstride + lane;
#pragma unroll
  for (int j = 0; j < 4 * PC; ++j) o.b[j] = bp[j * 64];   // ((t*2 + s)*PC + piece) * 64 + lane, t = 0..1: two column tiles are 256 * PC uint4 in a row
}

template <int PC>
__device__ __forceinline__ void direct_mma(const DirectOperands<PC>& o, f32x16& acc0, f32x16& acc1, float a_scale) {
  const PiecesN s0 = cut8<PC>(as_f4(o.a[0]), as_f4(o.a[1]), a_scale), s1 = cut8<PC>(as_f4(o.a[2]), as_f4(o.a[3]), a_scale);
  acc0 = mma_pc<PC>(s0, o.b[0 * PC], o.b[0 * PC + (PC > 1 ? 1 : 0)], o.b[0 * PC + (PC > 2 ? 2 : 0)], acc0);
  acc1 = mma_pc<PC>(s0, o.b[2 * PC], o.b[2 * PC + (PC > 1 ? 1 : 0)], o.b[2 * PC + (PC > 2 ? 2 : 0)], acc1);
  acc0 = mma_pc<PC>(s1, o.b[1 * PC], o.b[1 * PC + (PC > 1 ? 1 : 0)], o.b[1 * PC + (PC > 2 ? 2 : 0)], acc0);
  acc1 = mma_pc<PC>(s1, o.b[3 * PC], o.b[3 * PC + (PC > 1 ? 1 : 0)], o.b[3 * PC + (PC > 2 ? 2 : 0)], acc1);
}

// grid (ceil(P / 32), Cout / 64); block 64 * nw, nw = 1 .. kDirectMaxWaves waves; dynamic LDS (nw - 1) * 8 KB.  stride 1 (the plan's
// condition); any kh x kw, pads, dilation.  (Measured on the small net's 64 -> 64 3x3, 18 chunks: 3 waves x 6 chunks 12.3-14.0 us;
// 9 waves x 2 chunks 17.8-18.3 us — the launch bound for 9 waves caps the registers below the two operand sets.)
constexpr int kDirectMaxWaves = 3;

// One 32-pixel x 64-channel output tile (bx, by) of convolution `a` by the `nw` waves of the workgroup (every wave must call it: there
// is a workgroup barrier inside).  red_dyn: max(nw - 1, 1) * 8 KB of LDS.
template <int PC>
__device__ __forceinline__ void direct_tile(const Args& a, int bx, int by, int nx, int nw, float* red_dyn) {
  float (*red)[32 * 64] = reinterpret_cast<float (*)[32 * 64]>(red_dyn);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int P = a.N * a.Ho * a.Wo;
  const int m0 = bx * 32;
  const int cpt = a.Cin / kChunk, nchunks = a.kh * a.kw * cpt;
  // this wave's run of the (tap, chunk) list (waves past nw: an empty run)
  const float inv_nw = 1.0f / (float)nw;
  const int c_begin = wave < nw ? fdiv(nchunks * wave, nw, inv_nw) : 0, c_end = wave < nw ? fdiv(nchunks * (wave + 1), nw, inv_nw) : 0;
  const int col_tiles = a.Cout / 32;
  const size_t wstride = (size_t)col_tiles * (128 * PC);
  const uint4* wtile = a.wfrag + (size_t)(2 * by) * (128 * PC);
  // the lane's pixel and the 16 channels of a chunk it feeds (fragment role: row = lane & 31, k half = lane >> 5)
  const int pix = m0 + (lane & 31);
  const bool pix_ok = pix < P;
  int iy0 = 0, ix0 = 0, base = 0;
  if (pix_ok) {
    const int pn = fdiv(pix, a.Ho * a.Wo, a.inv_howo), r = pix - pn * a.Ho * a.Wo, py = fdiv(r, a.Wo, a.inv_wo), px = r - py * a.Wo;
    iy0 = py * a.stride - a.pad_h; ix0 = px * a.stride - a.pad_w;
    base = ((pn * a.H + iy0) * a.W + ix0) * a.lda + 16 * (lane >> 5);
    if (a.x_kmajor) base = (pn * a.lda + 16 * (lane >> 5)) * (a.H * a.W) + r;       // 1x1, stride 1, no padding: input pixel = output pixel
  }
  const int kstride = a.x_kmajor ? a.H * a.W : 0;
  f32x16 acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }

  auto operand_of = [&](int gch, int& off, bool& ok) {
    const int tap = fdiv(gch, cpt, a.inv_cpt), kc = gch - tap * cpt, ty = fdiv(tap, a.kw, a.inv_kw), tx = tap - ty * a.kw;
    const int dy = ty * a.dil, dx = tx * a.dil;
    ok = pix_ok && (unsigned)(iy0 + dy) < (unsigned)a.H && (unsigned)(ix0 + dx) < (unsigned)a.W;
    off = kstride ? base + kc * kChunk * kstride : base + (dy * a.W + dx) * a.lda + kc * kChunk;
  };
  const bool rows = rows_path_ok(a);
  DirectOperands<PC> cur, nxt;
  if (c_begin < c_end) {
    int off; bool ok;
    operand_of(c_begin, off, ok);
    direct_load(cur, a.x, wtile, wstride, c_begin, off, ok, lane, kstride);
  }
  // the fp16 form's scale, read AFTER the first operands are on their way (one wait for both: a launch this short is a chain of
  // memory round trips, and every one taken out of the chain is ~1 us of its ~10)
  float a_scale = 1.f;
  int s_exp = 0;
  if (PC == 2) {
    s_exp = 13 - amax_exponent_asm(a.amax, lane, a.status);
    a_scale = ldexpf(1.f, s_exp);
  }
  for (int c = c_begin; c < c_end; c += 2) {         // two register sets take turns (no copies)
    if (c + 1 < c_end) {
      int off; bool ok;
      operand_of(c + 1, off, ok);
      direct_load(nxt, a.x, wtile, wstride, c + 1, off, ok, lane, kstride);
    }
    direct_mma<PC>(cur, acc0, acc1, a_scale);
    if (c + 1 < c_end) {
      if (c + 2 < c_end) {
        int off; bool ok;
        operand_of(c + 2, off, ok);
        direct_load(cur, a.x, wtile, wstride, c + 2, off, ok, lane, kstride);
      }
      direct_mma<PC>(nxt, acc0, acc1, a_scale);
    }
  }
  // the waves' partial sums meet in LDS: [wave - 1][reg][lane] (conflict-free), added in wave order by wave 0
  if (wave > 0 && wave < nw) {
#pragma unroll
    for (int r = 0; r < 16; ++r) { red[wave - 1][r * 64 + lane] = acc0[r]; red[wave - 1][(16 + r) * 64 + lane] = acc1[r]; }
  }
  __syncthreads();
  if (wave > 0) return;
  for (int w = 0; w < nw - 1; ++w)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = acc0[r] + red[w][r * 64 + lane]; acc1[r] = acc1[r] + red[w][(16 + r) * 64 + lane]; }
  if (PC == 2) {
    float os[2];
    column_scales<2>(a, s_exp, by * kWgCh, lane, os);
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc0[r] = acc0[r] * os[0]; acc1[r] = acc1[r] * os[1]; }
  }
  if (rows) {
    // the two 32 x 32 tiles through the first 8 KB of the LDS block (the other waves' sums have been read) as float4 rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    float* T = red_dyn;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      T[row * 32 + (lane & 31)] = acc0[r];
      T[1024 + row * 32 + (lane & 31)] = acc1[r];
    }
    RowsIn<2> in;
    tile_rows_in<2>(a, m0, P, by * kWgCh, false, lane, in);
    const uint32_t mr = tile_rows_out<2>(a, T, m0, P, by * kWgCh, nullptr, lane, in);
    publish_amax(mr, a.amax_out, a.status, by * nx + bx);
    return;
  }
  RowOut ro;
  ro.valid = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int p = m0 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    ro.base[r] = 0;
    if (p < P) { ro.valid |= 1u << r; ro.base[r] = out_pixel_base(a, p); }
  }
  const uint32_t m = max(tile_store_max(a, ro, by * kWgCh + (lane & 31), acc0), tile_store_max(a, ro, by * kWgCh + 32 + (lane & 31), acc1));
  publish_amax(m, a.amax_out, a.status, by * nx + bx);
}

template <int PC>
static __global__ __launch_bounds__(64 * kDirectMaxWaves) void conv_split_direct_kernel(Args a) {
  extern __shared__ __attribute__((aligned(16))) float red_dyn[];
  direct_tile<PC>(a, blockIdx.x, blockIdx.y, gridDim.x, (int)(blockDim.x >> 6), red_dyn);
}

// partial maxima of |x| for the fp16 form's scale: kAmaxSlots workgroups, slot b = max over its grid-stride share (0 for an empty share)
static __global__ __launch_bounds__(256) void amax_partial_kernel(const float4* __restrict__ x, long n4, float* __restrict__ out) {
  __shared__ float red[4];
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = x[i];
    m = fmaxf(m, fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))));
  }
#pragma unroll
  for (int d = 32; d >= 1; d >>= 1) m = fmaxf(m, __shfl_xor(m, d, 64));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

}  // namespace convsplit
}  // namespace lsfa
