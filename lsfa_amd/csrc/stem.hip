// The stem of the ResNets (dff_rfcn/symbols/resnet.py:151-163: bn_data -> conv0 7x7/2 + bn0 + relu0 -> pool0 3x3/2 max)
// and the average pooling that shrinks the frame in front of the small net (resnet_v1_101_flownet_rfcn.py:216, `resize_data`,
// 4x4/4 avg).  Through the libraries this is six launches per frame (avg pool, bn_data pass, layout copy, convolution,
// bias + ReLU pass, max pool: ~70 us for the small net's 150x250 input, ~215 us for the backbone's 600x1000); here three:
//   avgpool_kernel        NCHW k x k / k average
//   stem_conv_kernel      bn_data (per-channel affine, applied while the input patch is staged: padding stays 0) + 7x7 stride-2
//                         convolution with bn0 folded + bias + ReLU, 3 -> 64 channels, NCHW in, channels-last out
//   maxpool3x3s2_kernel   channels-last 3x3 / 2, pad 1
// The convolution (r4) runs on the matrix pipe like every other one: fp32 operands as two fp16 pieces (hi = fp16(v s),
// lo = fp16(v s - hi), s a power of two), three v_mfma_f32_32x32x16_f16 per k-step (lo hi, hi lo, hi hi; the lo lo term is
// below fp32's own rounding), one fp32 accumulator.  K = (ci, ky) x 8: a lane's eight k values are the SEVEN taps of one
// (ci, ky) row preceded by a zero-weight tap, so that the window of output column p starts at the even patch column 2p and
// is four consecutive dwords of fp16 in LDS (no im2col); ky is padded to 8 the same way (a lane half = the parity of ky, so
// that one LDS address serves every k-step): 3 x 4 = 12 k-steps of 16 (147 / 192 of the MACs are real).
// A workgroup computes 4 rows x 32 columns x 64 channels per tile and walks `tiles_per_wg` tiles along x; wave = (2 rows) x
// (32 channels).  Its weight fragments (12 x 2 pieces, cut once by lsfa_stem_weights with one scale per output channel) stay
// in registers; the input patch (3 x 14 x 70, bn_data applied) is cut with the scale of ITS OWN maximum (an image's result
// does not depend on what else is in the batch); the next tile's patch is in flight while the matrix instructions of this
// one run, and is cut into LDS before this tile's stores are issued.
// Measured (tools/lab/stem_probe.py, hipGraph): 6 images of 1000x600 in 116 us (19 us per image; r3's kernel, an fp32 FMA
// chain per output at 60 % of the vector peak, took 77), 27.7 us for one, 19 us for the small net's nine 150x250 inputs (was
// 57).  Counters for the 6-image case: matrix pipe 29 % busy, vector ALU 33 %, 646 vector instructions per wave and tile
// (72 of them matrix instructions): two waves per SIMD (250 registers: 96 of them weights) leave the phases of a tile
// (stage, multiply, store) mostly unoverlapped.  Weights in LDS and eight waves per workgroup are the next step.
// The oracle's conv is a library stage: compared by tolerance, like every convolution.
#include "common.h"

using namespace lsfa;

namespace {

constexpr int kStemCout = 64, kStemCin = 3, kK = 7;
constexpr int kTileRows = 4, kTileCols = 32;
constexpr int kInRows = 2 * (kTileRows - 1) + 8;        // 14 patch rows per input channel (ky = 0..6 and the zero-weight ky = 7)
constexpr int kInCols = 2 * (kTileCols - 1) + 8;        // 70: output column p reads patch columns 2p .. 2p+7 (the first one with weight 0)
constexpr int kPitchD = 36;                             // dwords (pairs of fp16) per patch row
// Rows of even and of odd height sit in two regions 32 banks apart: a k-step reads ky = 2j from lanes 0-31 and ky = 2j + 1 from
// lanes 32-63 (rows of different parity, whatever the output row), 32 consecutive dwords each: the halves of a wave never share a bank.
constexpr int kRegionD = 800;                           // >= 3 * 7 * 36 = 756, = 32 mod 64
constexpr int kPatchD = 2 * kRegionD;                   // dwords per piece
__device__ __forceinline__ constexpr int patch_row_d(int ci, int row) { return (row & 1) * kRegionD + (ci * (kInRows / 2) + (row >> 1)) * kPitchD; }
constexpr int kPatchRows = kStemCin * kInRows;          // 42: wave w stages rows w, w + 4, ... (columns 0..63, a lane each); columns 64..69
constexpr int kRowsPerWave = (kPatchRows + 3) / 4;      // of row t / 6 go to thread t < 252
constexpr int kPerThread = kRowsPerWave + 1;            // 12
constexpr int kSteps = 12;                              // k-step s: input channel s / 4, ky = 2 (s % 4) + (lane half); ky = 7 has zero weights

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16x8 as_h(const uint4& u) {
  union { uint4 u; f16x8 v; } c;
  c.u = u;
  return c.v;
}

// (hi, lo) of v0 s and v1 s, packed (v0 in the low half)
__device__ __forceinline__ void cut_pair(float v0, float v1, float s, uint32_t& h, uint32_t& l) {
  const float a0 = v0 * s, a1 = v1 * s;           // exact: s is a power of two
  f16x2 hh;
  hh[0] = (_Float16)a0; hh[1] = (_Float16)a1;
  f16x2 ll;
  ll[0] = (_Float16)(a0 - (float)hh[0]);          // the differences are exact
  ll[1] = (_Float16)(a1 - (float)hh[1]);
  h = __builtin_bit_cast(uint32_t, hh);
  l = __builtin_bit_cast(uint32_t, ll);
}

// s = 2^(13 - floor(log2 m)) puts m into [2^13, 2^14) (fp16's top binades, clear of its overflow); 1 for m = 0, tiny or non-finite.
// inv = 1 / s.
__device__ __forceinline__ void scale_of(float m, float& s, float& inv) {
  const int be = (int)((__float_as_uint(m) >> 23) & 255u);
  const bool ok = be >= 32 && be <= 240;
  s = ok ? __uint_as_float((uint32_t)(267 - be) << 23) : 1.f;
  inv = ok ? __uint_as_float((uint32_t)(be - 13) << 23) : 1.f;
}

// r6: `tbl` (or NULL) = a device table of per-image base pointers: image n is read at tbl[n] instead of x + n * C * H * W, so that frames
// handed over as separate tensors need no staging copy into one batch (lsfa_avgpool_nchw_tbl; C = channels per image then)
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ x, const float* const* __restrict__ tbl, int C, int NC, int H, int W,
                                                      int k, int Ho, int Wo, float* __restrict__ y) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)NC * Ho * Wo) return;
  const int ox = (int)(i % Wo);
  const long r = i / Wo;
  const int oy = (int)(r % Ho);
  const long nc = r / Ho;
  const int y0 = oy * k, x0 = ox * k, y1 = min(y0 + k, H), x1 = min(x0 + k, W);
  const float* p = tbl ? tbl[nc / C] + (nc % C) * (long)H * W : x + nc * (long)H * W;
  float s = 0.f;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) s = s + p[(long)yy * W + xx];
  y[i] = s / (float)((y1 - y0) * (x1 - x0));       // ceil-mode windows are clipped to the image (no padding)
}

// The weight fragments of the convolution, cut once when the weights are loaded.  w_l: (3, 7, 7, 64) floats = [ci][ky][kx][co] with bn0's
// scale folded in.  wfrag: [channel half ni][k-step s][piece][lane] uint4, then 64 floats 1 / scale.  Lane l of step s holds, for output
// channel 32 ni + l % 32, the taps (0, w[ci][ky][0..6]) of (ci, ky) = (s / 4, 2 (s % 4) + l / 32) (zeros for ky = 7) as hi / lo fp16 of
// w * scale, the scale a power of two per OUTPUT CHANNEL (its largest |w| into [2^13, 2^14)).  One block of 128 threads.
constexpr int kFragVecs = 2 * kSteps * 2 * 64;          // uint4s in front of the 64 inverse scales
__global__ __launch_bounds__(128) void stem_weights_kernel(const float* __restrict__ w_l, uint4* __restrict__ wfrag) {
  const int lane = threadIdx.x & 63, ni = threadIdx.x >> 6, half = lane >> 5, co = (lane & 31) + 32 * ni;
  float wmax = 0.f;
  for (int k = 0; k < kStemCin * kK * kK; ++k) wmax = fmaxf(wmax, fabsf(w_l[k * kStemCout + co]));
  float sw, inv_w;
  scale_of(wmax, sw, inv_w);
  for (int s = 0; s < kSteps; ++s) {
    const int ky = 2 * (s & 3) + half;
    float w[kK];
    for (int j = 0; j < kK; ++j) w[j] = ky < kK ? w_l[(((s >> 2) * kK + ky) * kK + j) * kStemCout + co] : 0.f;
    uint4 h, l;
    cut_pair(0.f, w[0], sw, h.x, l.x);
    cut_pair(w[1], w[2], sw, h.y, l.y);
    cut_pair(w[3], w[4], sw, h.z, l.z);
    cut_pair(w[5], w[6], sw, h.w, l.w);
    wfrag[((ni * kSteps + s) * 2 + 0) * 64 + lane] = h;
    wfrag[((ni * kSteps + s) * 2 + 1) * 64 + lane] = l;
  }
  if (half == 0) reinterpret_cast<float*>(wfrag + kFragVecs)[co] = inv_w;
}

// grid (ceil(ceil(Wo / 32) / tiles_per_wg), ceil(Ho / 4), N); block 256.  wfrag: what stem_weights_kernel wrote.
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const float* __restrict__ x, const float* const* __restrict__ tbl, int H, int W, const float* __restrict__ in_scale,
                                                           const float* __restrict__ in_shift, const uint4* __restrict__ wfrag,
                                                           const float* __restrict__ bias, int Ho, int Wo, int tiles_per_wg,
                                                           const float* accum, int act, float* y, unsigned* __restrict__ amax_out) {
  __shared__ uint32_t hi_s[kPatchD], lo_s[kPatchD];
  __shared__ uint32_t red_s[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6), mi = wv >> 1, ni = wv & 1;
  const int half = lane >> 5, px = lane & 31, co = px + 32 * ni;
  const int n = blockIdx.z, oy0 = blockIdx.y * kTileRows;
  const int iy0 = 2 * oy0 - 3;
  const float* xin = tbl ? tbl[n] : x + (size_t)n * kStemCin * H * W;      // (tbl: lsfa_stem_conv7x7s2_tbl, per-image base pointers)

  // ---- this wave's weight fragments (stem_weights_kernel cut them): 24 coalesced 16-byte loads ----
  uint4 bhi[kSteps], blo[kSteps];
#pragma unroll
  for (int s = 0; s < kSteps; ++s) {
    bhi[s] = wfrag[((ni * kSteps + s) * 2 + 0) * 64 + lane];
    blo[s] = wfrag[((ni * kSteps + s) * 2 + 1) * 64 + lane];
  }
  const float inv_w = reinterpret_cast<const float*>(wfrag + kFragVecs)[co];
  const float b = bias ? bias[co] : 0.f;
  const float sc0 = in_scale ? in_scale[0] : 1.f, sc1 = in_scale ? in_scale[1] : 1.f, sc2 = in_scale ? in_scale[2] : 1.f;
  const float sh0 = in_shift ? in_shift[0] : 0.f, sh1 = in_shift ? in_shift[1] : 0.f, sh2 = in_shift ? in_shift[2] : 0.f;

  // the patch of tile t, kPerThread elements per thread, bn_data applied (the zero padding is applied to ITS output).  A wave's rows
  // are wave-uniform: channel, image row and their validity are scalar work; a lane adds its column.
  const int xr = tid / 6, xc = 64 + tid - 6 * xr;           // the extra element: row xr (< 42 for tid < 252), column 64..69
  const int xci = xr / kInRows, xiy = iy0 + xr - xci * kInRows;
  const bool xrow = tid < 6 * kPatchRows && xiy >= 0 && xiy < H;
  const float xsc = in_scale ? in_scale[min(xci, 2)] : 1.f, xsh = in_shift ? in_shift[min(xci, 2)] : 0.f;
  float pv[kPerThread];
#define LSFA_STEM_LOAD_PATCH(t)                                                                                           \
  {                                                                                                                       \
    const int ix0 = 2 * ((blockIdx.x * tiles_per_wg + (t)) * kTileCols) - 4;                                              \
    const bool cok = ix0 + lane >= 0 && ix0 + lane < W;                                                                   \
    _Pragma("unroll") for (int j = 0; j < kRowsPerWave; ++j) {                                                            \
      const int r = wv + 4 * j, ci = r / kInRows, iy = iy0 + r - ci * kInRows;                                            \
      const bool in = r < kPatchRows && iy >= 0 && iy < H && cok;                                                         \
      const float v = xin[in ? (ci * H + iy) * W + ix0 + lane : 0];                                                       \
      const float sc = ci == 0 ? sc0 : (ci == 1 ? sc1 : sc2), sh = ci == 0 ? sh0 : (ci == 1 ? sh1 : sh2);                \
      pv[j] = in ? v * sc + sh : 0.f;                                                                                     \
    }                                                                                                                     \
    const bool in = xrow && ix0 + xc < W;                                                                                 \
    const float v = xin[in ? (xci * H + xiy) * W + ix0 + xc : 0];                                                         \
    pv[kRowsPerWave] = in ? v * xsc + xsh : 0.f;                                                                          \
  }

  float top = 0.f;                        // max |y| this thread stored
  const float slope = act == 1 ? 0.f : (act == 2 ? 0.1f : 1.f);
  const int abase = half * kRegionD + 2 * mi * kPitchD + px;      // patch row 2 (2 mi + mm) + ky, ky = 2 (s % 4) + half: region half, row 2 mi + mm + s % 4
  const int lo = 4 * half * kStemCout + co;

  // Stage the patch held in pv: its scale from its own maximum (parity `par` of the exchange slots), pieces into LDS.
  // The caller has a barrier between the last read of the previous patch and the exchange.
#define LSFA_STEM_PATCH_MAX(par)                                                                                          \
  {                                                                                                                       \
    float m = 0.f;                                                                                                        \
    _Pragma("unroll") for (int q = 0; q < kPerThread; ++q) m = fmaxf(m, fabsf(pv[q]));                                    \
    uint32_t mb = __float_as_uint(m); /* bit patterns of non-negative floats order like the floats (a NaN ranks above everything) */ \
    _Pragma("unroll") for (int d = 32; d >= 1; d >>= 1) mb = max(mb, (uint32_t)__shfl_xor((int)mb, d, 64));              \
    if (lane == 0) red_s[par][wv] = mb;                                                                                   \
  }
#define LSFA_STEM_PATCH_CUT(par, inv_out)                                                                                 \
  {                                                                                                                       \
    const uint32_t mb = max(max(red_s[par][0], red_s[par][1]), max(red_s[par][2], red_s[par][3]));                        \
    float sx;                                                                                                             \
    scale_of(__uint_as_float(mb), sx, inv_out);                                                                           \
    uint16_t* hs = reinterpret_cast<uint16_t*>(hi_s);                                                                     \
    uint16_t* ls = reinterpret_cast<uint16_t*>(lo_s);                                                                     \
    _Pragma("unroll") for (int j = 0; j < kPerThread; ++j) {                                                              \
      const int r = j < kRowsPerWave ? wv + 4 * j : xr, c = j < kRowsPerWave ? lane : xc;                                 \
      if (j < kRowsPerWave ? r < kPatchRows : tid < 6 * kPatchRows) {                                                     \
        const float a = pv[j] * sx;                                                                                       \
        const _Float16 hh = (_Float16)a;                                                                                  \
        const _Float16 ll = (_Float16)(a - (float)hh);                                                                    \
        const int ci = r / kInRows, d = 2 * patch_row_d(ci, r - ci * kInRows) + c;                                        \
        hs[d] = __builtin_bit_cast(uint16_t, hh);                                                                         \
        ls[d] = __builtin_bit_cast(uint16_t, ll);                                                                         \
      }                                                                                                                   \
    }                                                                                                                     \
  }

  // Per tile: [loads of the NEXT patch issued] matrix instructions on this one | barrier | next patch cut into LDS | this tile's
  // stores | barrier.  The stores drain while the next tile multiplies: nothing waits on them (vmcnt counts stores too; the
  // wait for the patch loads comes after the matrix phase, when the previous tile's stores are long done).
  float inv_x;
  LSFA_STEM_LOAD_PATCH(0)
  LSFA_STEM_PATCH_MAX(0)
  __syncthreads();
  LSFA_STEM_PATCH_CUT(0, inv_x)
  __syncthreads();
  for (int t = 0; t < tiles_per_wg; ++t) {
    const int ox0 = (blockIdx.x * tiles_per_wg + t) * kTileCols;
    if (ox0 >= Wo) break;
    const bool more = t + 1 < tiles_per_wg && ox0 + kTileCols < Wo;
    if (more) LSFA_STEM_LOAD_PATCH(t + 1)

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
    // Patch row k of input channel ci (this lane half's parity) serves output row 2 mi with ky pair k and output row 2 mi + 1 with
    // ky pair k - 1: 15 row reads per tile, the next one in flight while the (up to) six matrix instructions of this one issue.
    uint4 ah[2], al[2];
#define LSFA_STEM_ROW(q, buf)                                                                                             \
  {                                                                                                                       \
    const int d = abase + (((q) / 5) * (kInRows / 2) + (q) % 5) * kPitchD;                                                \
    ah[buf] = make_uint4(hi_s[d], hi_s[d + 1], hi_s[d + 2], hi_s[d + 3]);                                                 \
    al[buf] = make_uint4(lo_s[d], lo_s[d + 1], lo_s[d + 2], lo_s[d + 3]);                                                 \
  }
    LSFA_STEM_ROW(0, 0)
#pragma unroll
    for (int q = 0; q < 15; ++q) {
      if (q + 1 < 15) LSFA_STEM_ROW(q + 1, (q + 1) & 1)
      __builtin_amdgcn_sched_barrier(0);        // the reads of row q + 1 are issued before row q's matrix instructions, not after
      const int ci = q / 5, k = q % 5, u = q & 1;
      if (k <= 3) {
        const int s = ci * 4 + k;
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(al[u]), as_h(bhi[s]), acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah[u]), as_h(blo[s]), acc[0], 0, 0, 0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah[u]), as_h(bhi[s]), acc[0], 0, 0, 0);
      }
      if (k >= 1) {
        const int s = ci * 4 + k - 1;
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(al[u]), as_h(bhi[s]), acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah[u]), as_h(blo[s]), acc[1], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(as_h(ah[u]), as_h(bhi[s]), acc[1], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef LSFA_STEM_ROW
    const float inv = inv_x * inv_w;      // of THIS tile's patch
    if (more) LSFA_STEM_PATCH_MAX((t + 1) & 1)
    __syncthreads();                      // every wave is done reading this tile's patch; the maxima of the next one are exchanged
    if (more) LSFA_STEM_PATCH_CUT((t + 1) & 1, inv_x)

    // accumulator register i of a lane: pixel (i & 3) + 8 (i >> 2) + 4 half of the row, channel co.
    // act(v) = max(v, 0) + slope min(v, 0), slope 0 / 0.1 / 1: the same values as the branches (one of the terms is a zero)
    const bool whole = ox0 + kTileCols <= Wo;
#pragma unroll
    for (int mm = 0; mm < 2; ++mm) {
      const int oy = oy0 + 2 * mi + mm;
      if (oy >= Ho) continue;
      float* yrow = y + (((size_t)n * Ho + oy) * Wo + ox0) * kStemCout + lo;           // a wave-uniform row, a 32-bit lane offset
      const float* arow = accum ? accum + (((size_t)n * Ho + oy) * Wo + ox0) * kStemCout + lo : nullptr;
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = acc[mm][i] * inv + b;
      if (whole) {
        if (arow) {
#pragma unroll
          for (int i = 0; i < 16; ++i) v[i] = v[i] + arow[((i & 3) + 8 * (i >> 2)) * kStemCout];
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const float o = fmaxf(v[i], 0.f) + slope * fminf(v[i], 0.f);
          yrow[((i & 3) + 8 * (i >> 2)) * kStemCout] = o;
          top = fmaxf(top, fabsf(o));
        }
      } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int p = (i & 3) + 8 * (i >> 2);
          if (ox0 + 4 * half + p < Wo) {
            if (arow) v[i] = v[i] + arow[p * kStemCout];
            const float o = fmaxf(v[i], 0.f) + slope * fminf(v[i], 0.f);
            yrow[p * kStemCout] = o;
            top = fmaxf(top, fabsf(o));
          }
        }
      }
    }
    __syncthreads();                      // the next patch is in LDS
  }
#undef LSFA_STEM_PATCH_MAX
#undef LSFA_STEM_PATCH_CUT
#undef LSFA_STEM_LOAD_PATCH
  if (amax_out) {      // the slots lsfa_conv_fwd reads as amax_in
    uint32_t m = __float_as_uint(top);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, d, 64));
    if (lane == 0) atomicMax(amax_out + ((((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wv) & 255), m);
  }
}

// channels-last 3x3 / 2 max pooling, pad 1 (the padding never wins: windows are clipped); a thread = 4 channels of one output.
// y2 (optional): max(y * scale2[c] + shift2[c], 0), the first unit's bn1 + relu1 of the pooled map, as a second output.
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const float4* __restrict__ x, int N, int H, int W, int C4, int Ho, int Wo,
                                                           float4* __restrict__ y, float4* __restrict__ y2,
                                                           const float4* __restrict__ scale2, const float4* __restrict__ shift2,
                                                           unsigned* __restrict__ amax_out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  float top = 0.f;        // max |.| of what this thread hands to the next convolution (y2 when present, else y)
  if (i < (long)N * Ho * Wo * C4) {
  const int c = (int)(i % C4);
  long r = i / C4;
  const int ox = (int)(r % Wo); r /= Wo;
  const int oy = (int)(r % Ho);
  const int n = (int)(r / Ho);
  const int y0 = max(2 * oy - 1, 0), y1 = min(2 * oy + 2, H), x0 = max(2 * ox - 1, 0), x1 = min(2 * ox + 2, W);
  float4 m = x[(((size_t)n * H + y0) * W + x0) * C4 + c];
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      const float4 v = x[(((size_t)n * H + yy) * W + xx) * C4 + c];
      m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
    }
  y[i] = m;
  if (y2) {
    const float4 sc = scale2[c], sh = shift2[c];
    m = make_float4(fmaxf(m.x * sc.x + sh.x, 0.f), fmaxf(m.y * sc.y + sh.y, 0.f), fmaxf(m.z * sc.z + sh.z, 0.f),
                    fmaxf(m.w * sc.w + sh.w, 0.f));
    y2[i] = m;
  }
  top = fmaxf(fmaxf(fabsf(m.x), fabsf(m.y)), fmaxf(fabsf(m.z), fabsf(m.w)));
  }
  if (amax_out) {      // the slots lsfa_conv_fwd reads as amax_in (bit patterns of non-negative floats order like the floats)
    uint32_t b = __float_as_uint(top);
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) b = max(b, (uint32_t)__shfl_xor((int)b, d, 64));
    if ((threadIdx.x & 63) == 0) atomicMax(amax_out + ((blockIdx.x * 4 + (threadIdx.x >> 6)) & 255), b);
  }
}

// r5: transform (lib/utils/image.py:296-308): a decoded frame - (N, H, W, 3) uint8, BGR - into the network's `data` tensor,
// (N, 3, H, W) float32, channel i = (im[:, :, 2 - i] - pixel_means[2 - i]) * pixel_scale.  A thread converts four consecutive pixels
// (12 bytes in, one float4 per plane out); the subtraction and the multiplication run in float64 like the reference's two numpy
// statements (np.zeros is float64) and the result is rounded to float32 once, where the reference hands the array to the executor:
// bit-identical to it for any means / scale.
__global__ __launch_bounds__(256) void image_transform_u8_kernel(const unsigned char* __restrict__ im, long npix4, int HW, double m0, double m1, double m2,
                                                                 double scale, float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= npix4) return;
  const long hw4 = HW / 4;
  const long n = i / hw4, q = i - n * hw4;
  const uint32_t* src = reinterpret_cast<const uint32_t*>(im + (n * HW + q * 4) * 3);      // 12-byte groups: 4-byte aligned (HW % 4 == 0)
  const uint32_t w0 = src[0], w1 = src[1], w2 = src[2];
  // bytes b0 g0 r0 b1 | g1 r1 b2 g2 | r2 b3 g3 r3
  const double b[4] = {(double)(w0 & 255u), (double)(w0 >> 24), (double)((w1 >> 16) & 255u), (double)((w2 >> 8) & 255u)};
  const double g[4] = {(double)((w0 >> 8) & 255u), (double)(w1 & 255u), (double)(w1 >> 24), (double)((w2 >> 16) & 255u)};
  const double r[4] = {(double)((w0 >> 16) & 255u), (double)((w1 >> 8) & 255u), (double)(w2 & 255u), (double)(w2 >> 24)};
  float* o = out + n * 3 * (long)HW + q * 4;
  // plane 0 = R - mean[2], plane 1 = G - mean[1], plane 2 = B - mean[0]   (m0, m1, m2 = pixel_means in B, G, R order)
  *reinterpret_cast<float4*>(o) = make_float4((float)((r[0] - m2) * scale), (float)((r[1] - m2) * scale), (float)((r[2] - m2) * scale), (float)((r[3] - m2) * scale));
  *reinterpret_cast<float4*>(o + HW) = make_float4((float)((g[0] - m1) * scale), (float)((g[1] - m1) * scale), (float)((g[2] - m1) * scale), (float)((g[3] - m1) * scale));
  *reinterpret_cast<float4*>(o + 2 * (long)HW) = make_float4((float)((b[0] - m0) * scale), (float)((b[1] - m0) * scale), (float)((b[2] - m0) * scale), (float)((b[3] - m0) * scale));
}

}  // namespace

extern "C" int lsfa_image_transform_u8(const unsigned char* im_hwc_bgr, int N, int H, int W, const double* pixel_means_bgr_host, double pixel_scale,
                                       float* data_nchw, void* stream) {
  LSFA_REQUIRE(im_hwc_bgr && pixel_means_bgr_host && data_nchw, "lsfa_image_transform_u8: NULL argument");
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0, "lsfa_image_transform_u8: bad shape");
  if (((long)H * W) % 4 != 0 || ((uintptr_t)im_hwc_bgr & 3) || ((uintptr_t)data_nchw & 15)) {
    set_error("lsfa_image_transform_u8: H*W = %ld must be a multiple of 4, the image 4-byte and the output 16-byte aligned", (long)H * W);
    return LSFA_ENOTSUP;
  }
  const long npix4 = (long)N * H * W / 4;
  ProfScope prof(LSFA_OP_STEM, (hipStream_t)stream);
  hipLaunchKernelGGL(image_transform_u8_kernel, dim3((unsigned)((npix4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, im_hwc_bgr, npix4, H * W,
                     pixel_means_bgr_host[0], pixel_means_bgr_host[1], pixel_means_bgr_host[2], pixel_scale, data_nchw);
  LSFA_LAUNCH_CHECK("lsfa_image_transform_u8");
  return LSFA_OK;
}

static int avgpool_launch(const float* x, const float* const* tbl, int N, int C, int H, int W, int k, float* y, void* stream, const char* who) {
  LSFA_REQUIRE((x || tbl) && y, "%s: NULL argument", who);
  LSFA_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && k > 0, "%s: bad shape", who);
  const int Ho = (H + k - 1) / k, Wo = (W + k - 1) / k;
  const long total = (long)N * C * Ho * Wo;
  ProfScope prof(LSFA_OP_STEM, (hipStream_t)stream);
  hipLaunchKernelGGL(avgpool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, tbl, C, N * C, H, W, k, Ho, Wo, y);
  LSFA_LAUNCH_CHECK(who);
  return LSFA_OK;
}

extern "C" int lsfa_avgpool_nchw(const float* x, int N, int C, int H, int W, int k, float* y, void* stream) {
  return avgpool_launch(x, nullptr, N, C, H, W, k, y, stream, "lsfa_avgpool_nchw");
}

extern "C" int lsfa_avgpool_nchw_tbl(const float* const* x_table, int N, int C, int H, int W, int k, float* y, void* stream) {
  return avgpool_launch(nullptr, x_table, N, C, H, W, k, y, stream, "lsfa_avgpool_nchw_tbl");
}

extern "C" size_t lsfa_stem_weight_bytes(void) { return (size_t)kFragVecs * sizeof(uint4) + kStemCout * sizeof(float); }

extern "C" int lsfa_stem_weights(const float* w_l, void* wfrag, void* stream) {
  LSFA_REQUIRE(w_l && wfrag, "lsfa_stem_weights: NULL argument");
  LSFA_REQUIRE(((uintptr_t)wfrag & 15) == 0, "lsfa_stem_weights: wfrag must be 16-byte aligned");
  hipLaunchKernelGGL(stem_weights_kernel, dim3(1), dim3(128), 0, (hipStream_t)stream, w_l, (uint4*)wfrag);
  LSFA_LAUNCH_CHECK("lsfa_stem_weights");
  return LSFA_OK;
}

extern "C" int lsfa_stem_conv7x7s2(const float* x, int N, int H, int W, const float* in_scale, const float* in_shift,
                                   const void* wfrag, const float* bias, float* y, void* stream) {
  return lsfa_stem_conv7x7s2_ex(x, N, H, W, in_scale, in_shift, wfrag, bias, nullptr, 1, y, nullptr, stream);
}

static int stem_launch(const float* x, const float* const* tbl, int N, int H, int W, const float* in_scale, const float* in_shift, const void* wfrag,
                       const float* bias, const float* accum, int act, float* y, unsigned* amax_out, void* stream);

extern "C" int lsfa_stem_conv7x7s2_ex(const float* x, int N, int H, int W, const float* in_scale, const float* in_shift,
                                      const void* wfrag, const float* bias, const float* accum, int act, float* y, unsigned* amax_out,
                                      void* stream) {
  return stem_launch(x, nullptr, N, H, W, in_scale, in_shift, wfrag, bias, accum, act, y, amax_out, stream);
}

extern "C" int lsfa_stem_conv7x7s2_tbl(const float* const* x_table, int N, int H, int W, const float* in_scale, const float* in_shift,
                                       const void* wfrag, const float* bias, const float* accum, int act, float* y, unsigned* amax_out,
                                       void* stream) {
  LSFA_REQUIRE(x_table, "lsfa_stem_conv7x7s2_tbl: NULL table");
  return stem_launch(nullptr, x_table, N, H, W, in_scale, in_shift, wfrag, bias, accum, act, y, amax_out, stream);
}

static int stem_launch(const float* x, const float* const* tbl, int N, int H, int W, const float* in_scale, const float* in_shift, const void* wfrag,
                       const float* bias, const float* accum, int act, float* y, unsigned* amax_out, void* stream) {
  LSFA_REQUIRE((x || tbl) && wfrag && y, "lsfa_stem_conv7x7s2: NULL argument");
  LSFA_REQUIRE(((uintptr_t)wfrag & 15) == 0, "lsfa_stem_conv7x7s2: wfrag must be 16-byte aligned (lsfa_stem_weights)");
  LSFA_REQUIRE((long)3 * H * W < (1L << 31), "lsfa_stem_conv7x7s2: an image of 2^31 elements or more");
  LSFA_REQUIRE(act >= 0 && act <= 2, "lsfa_stem_conv7x7s2_ex: act must be 0, 1 or 2");
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0, "lsfa_stem_conv7x7s2: bad shape");
  LSFA_REQUIRE((in_scale == nullptr) == (in_shift == nullptr), "lsfa_stem_conv7x7s2: in_scale and in_shift go together");
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  ProfScope prof(LSFA_OP_STEM, (hipStream_t)stream);
  // about two workgroups per CU (what fits side by side), each walking an equal share of its row of tiles
  const int xt = (Wo + kTileCols - 1) / kTileCols, yt = (Ho + kTileRows - 1) / kTileRows;
  long want = ((long)xt * yt * N) / 512;
  if (want < 1) want = 1;
  const int gx = (int)((xt + want - 1) / want);
  const int tpw = (xt + gx - 1) / gx;
  hipLaunchKernelGGL(stem_conv_kernel, dim3((xt + tpw - 1) / tpw, yt, N), dim3(256), 0, (hipStream_t)stream, x, tbl, H, W, in_scale, in_shift,
                     (const uint4*)wfrag, bias, Ho, Wo, tpw, accum, act, y, amax_out);
  LSFA_LAUNCH_CHECK("lsfa_stem_conv7x7s2");
  return LSFA_OK;
}

extern "C" int lsfa_maxpool3x3s2_nhwc(const float* x, int N, int H, int W, int C, float* y, float* y2, const float* scale2,
                                      const float* shift2, unsigned* amax_out, void* stream) {
  LSFA_REQUIRE(x && y, "lsfa_maxpool3x3s2_nhwc: NULL argument");
  LSFA_REQUIRE(!y2 || (scale2 && shift2 && y2 != y), "lsfa_maxpool3x3s2_nhwc: y2 needs scale2 / shift2 and must not alias y");
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0, "lsfa_maxpool3x3s2_nhwc: bad shape (C must be a multiple of 4)");
  const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const long total = (long)N * Ho * Wo * (C / 4);
  ProfScope prof(LSFA_OP_STEM, (hipStream_t)stream);
  hipLaunchKernelGGL(maxpool3x3s2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const float4*)x, N, H, W,
                     C / 4, Ho, Wo, (float4*)y, (float4*)y2, (const float4*)scale2, (const float4*)shift2, amax_out);
  LSFA_LAUNCH_CHECK("lsfa_maxpool3x3s2_nhwc");
  return LSFA_OK;
}
