// fp32 implicit-GEMM convolution on channels-last maps with the fp32 matrix cores
// (v_mfma_f32_32x32x2_f32: exact fp32 products and sums, the chip's fp32 peak).
//
// What it is for: conv2 of the pre-activation ResNet units (3x3, stride 1 or 2, dilation d, folded
// bn3 bias + ReLU; dff_rfcn/symbols/resnet.py:70-101, sym_common.py:92-135) on (H*W, C) rows.  At
// LSFA's size the stage-3 instance is a SMALL GEMM — 2394 pixels x 256 channels x K = 2304 — for a
// 256-CU part: 600 output tiles of 32x32 for 1024 SIMDs.  An fp32 MFMA occupies its SIMD for 64
// cycles whatever else is resident, so the time is (tile-tasks per SIMD, rounded up) x (MFMAs per
// task) x 64 cycles, and the lever is the task count, not occupancy:
//   * a workgroup computes a 64-pixel x 64-channel tile with 4 waves (one 32x32 accumulator tile each:
//     16 VGPRs), K walked tap by tap in 64-channel chunks (32 when Cin % 64 != 0), staged through LDS,
//     double-buffered;
//   * gridDim.z splits the TAPS over workgroups (3 x 3 taps for a 3x3 kernel) when the tile grid alone
//     would leave SIMDs idle; the slices write fp32 partial tiles to the workspace and a second kernel
//     adds them in a fixed order and applies bias + ReLU — deterministic, unlike the library's atomic
//     split-K (its `gkgs` kernels) which also needs a zero-fill launch;
//   * the MFMA sums over k in any order we like, as long as A and B agree: a lane reads 4 consecutive
//     k of its row/column with ONE ds_read_b128 (lanes 0-31 take k = 8c..8c+3, lanes 32-63 take
//     8c+4..8c+7) and feeds 4 MFMAs from it; LDS rows are padded by 4 floats, which spreads the 16
//     lanes of a b128 group over all 64 banks.
// Zero padding is realised when a chunk is staged (out-of-map pixels load zeros).
// Weight layout (prepared once at bind time): w[co][tap][ci], i.e. K contiguous per output channel.
#include "common.h"

#include <stdlib.h>

#include "conv_split_kernel.h"

using namespace lsfa;

namespace {

constexpr int kBM = 64, kBN = 64;
constexpr int kThreads = 256;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs {
  const float* x; const float* w; const float* bias; float* y; float* part;
  int N, H, W, Cin, Cout, kh, kw, stride, pad, dil, Ho, Wo, relu, taps_per_slice;
  // fused tail of a pre-activation unit's conv3 (resnet.py:93-101): y = conv + residual (in place allowed), and the
  // NEXT unit's bn1 + ReLU of that sum as a second output: y2 = max(y * scale2[c] + shift2[c], 0)
  const float* res; float* y2; const float* scale2; const float* shift2;
  int y_nchw;      // conv_reduce_kernel only: y / y2 / res are NCHW (the partial slices are always NHWC)
};

__device__ __forceinline__ void conv_epilogue(const ConvArgs& a, float v, size_t o, int ch, float bias, float sc2, float sh2) {
  v = v + bias;
  if (a.res) v = v + a.res[o];
  if (a.relu) v = fmaxf(v, 0.f);
  a.y[o] = v;
  if (a.y2) a.y2[o] = fmaxf(v * sc2 + sh2, 0.f);
  (void)ch;
}

// grid (ceil(P / 64), Cout / 64, slices); block 256.  P = N*Ho*Wo output pixels.  BK = channels per staged chunk
// (64 when Cin allows: 32 MFMAs per wave between barriers, long enough to cover the L2 latency of the next
// chunk's loads with the ~2 waves per SIMD these small grids leave; 32 otherwise).
template <int BK>
__global__ __launch_bounds__(kThreads) void conv_igemm_kernel(ConvArgs a) {
  constexpr int kLdk = BK + 4;                // padded LDS row (floats): conflict-free ds_read_b128 groups
  constexpr int NV = BK / 16;                 // float4 per thread and operand of a staged chunk
  __shared__ __attribute__((aligned(16))) float As[2][kBM * kLdk];
  __shared__ __attribute__((aligned(16))) float Bs[2][kBN * kLdk];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int P = a.N * a.Ho * a.Wo;
  const int m0 = blockIdx.x * kBM, n0 = blockIdx.y * kBN;
  const int taps = a.kh * a.kw;
  const int tap0 = blockIdx.z * a.taps_per_slice, tap1 = min(tap0 + a.taps_per_slice, taps);
  const int chunks_per_tap = a.Cin / BK;
  const int nchunks = (tap1 - tap0) * chunks_per_tap;

  // staging role of this thread: row (pixel of A / channel of B) and a (BK/4)-float column segment
  const int srow = tid >> 2, scol = (tid & 3) * (BK / 4);
  const int pix = m0 + srow;
  const bool pix_ok = pix < P;
  int py = 0, px = 0, pn = 0;
  if (pix_ok) { pn = pix / (a.Ho * a.Wo); const int r = pix - pn * a.Ho * a.Wo; py = r / a.Wo; px = r - py * a.Wo; }
  const float* wrow = a.w + ((size_t)(n0 + srow) * taps) * a.Cin + scol;

  // named registers: arrays (even with compile-time indices) and lambda captures ended up in scratch memory here
  float4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;      // NV = 2 uses the first two of each
  ra2 = ra3 = rb2 = rb3 = make_float4(0.f, 0.f, 0.f, 0.f);
  float a_keep = 0.f;
#define LSFA_CONV_FETCH(chunk_)                                                                                        \
  {                                                                                                                    \
    const int t_ = (chunk_) / chunks_per_tap;                                                                          \
    const int tap = tap0 + t_;                                                                                         \
    const int ci0 = ((chunk_) - t_ * chunks_per_tap) * BK;                                                             \
    const int ty = tap / a.kw, tx = tap - ty * a.kw;                                                                   \
    const int iy = py * a.stride - a.pad + ty * a.dil, ix = px * a.stride - a.pad + tx * a.dil;                        \
    const bool ok = pix_ok && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;                                              \
    const float4* src = reinterpret_cast<const float4*>(a.x + (((size_t)pn * a.H + (ok ? iy : 0)) * a.W + (ok ? ix : 0)) * a.Cin + ci0 + scol); \
    const float4* wsrc = reinterpret_cast<const float4*>(wrow + (size_t)tap * a.Cin + ci0);                            \
    ra0 = src[0]; ra1 = src[1]; rb0 = wsrc[0]; rb1 = wsrc[1];                                                          \
    if (NV > 2) { ra2 = src[2]; ra3 = src[3]; rb2 = wsrc[2]; rb3 = wsrc[3]; }                                          \
    a_keep = ok ? 1.0f : 0.0f;   /* zero padding = the (clamped, valid) load times 0, applied when the chunk is */     \
                                 /* written to LDS: any use of the loaded value here would stall the wave before its MFMAs */ \
  }
#define LSFA_CONV_STASH(buf_)                                                                                          \
  {                                                                                                                    \
    float4* da = reinterpret_cast<float4*>(&As[buf_][srow * kLdk + scol]);                                             \
    float4* db = reinterpret_cast<float4*>(&Bs[buf_][srow * kLdk + scol]);                                             \
    da[0] = make_float4(ra0.x * a_keep, ra0.y * a_keep, ra0.z * a_keep, ra0.w * a_keep);                               \
    da[1] = make_float4(ra1.x * a_keep, ra1.y * a_keep, ra1.z * a_keep, ra1.w * a_keep);                               \
    db[0] = rb0; db[1] = rb1;                                                                                          \
    if (NV > 2) {                                                                                                      \
      da[2] = make_float4(ra2.x * a_keep, ra2.y * a_keep, ra2.z * a_keep, ra2.w * a_keep);                             \
      da[3] = make_float4(ra3.x * a_keep, ra3.y * a_keep, ra3.z * a_keep, ra3.w * a_keep);                             \
      db[2] = rb2; db[3] = rb3;                                                                                        \
    }                                                                                                                  \
  }

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  LSFA_CONV_FETCH(0)
  LSFA_CONV_STASH(0)
  __syncthreads();
  const int arow = (wr * 32 + (lane & 31)) * kLdk + 4 * (lane >> 5);
  const int brow = (wc * 32 + (lane & 31)) * kLdk + 4 * (lane >> 5);
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int buf = chunk & 1;
    if (chunk + 1 < nchunks) LSFA_CONV_FETCH(chunk + 1)  // global loads in flight under the MFMAs
#pragma unroll
    for (int c = 0; c < BK / 8; ++c) {
      const float4 av = *reinterpret_cast<const float4*>(&As[buf][arow + 8 * c]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[buf][brow + 8 * c]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc, 0, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);   // keep every use of the freshly loaded registers behind the MFMAs
    if (chunk + 1 < nchunks) {
      LSFA_CONV_STASH(buf ^ 1)   // the other buffer: its last readers passed the barrier of the previous iteration
      __syncthreads();
    }
  }

#undef LSFA_CONV_FETCH
#undef LSFA_CONV_STASH
  // C/D layout of 32x32x2: column = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
  const int ch = n0 + wc * 32 + (lane & 31);
  const float bias = (a.part == nullptr && a.bias) ? a.bias[ch] : 0.f;
  const float sc2 = (a.part == nullptr && a.y2) ? a.scale2[ch] : 0.f, sh2 = (a.part == nullptr && a.y2) ? a.shift2[ch] : 0.f;
  float* part = a.part ? a.part + (size_t)blockIdx.z * P * a.Cout : nullptr;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
    const int p = m0 + wr * 32 + row;
    if (p < P) {
      const size_t o = (size_t)p * a.Cout + ch;
      if (part) part[o] = acc[r];
      else conv_epilogue(a, acc[r], o, ch, bias, sc2, sh2);
    }
  }
}

// the epilogue for the tap-split case: sum over slices of part, in slice order, then the same tail; float4 of channels per thread
__global__ __launch_bounds__(kThreads) void conv_reduce_kernel(ConvArgs a, long n4, int slices) {
  const long i = (long)blockIdx.x * kThreads + threadIdx.x;
  if (i >= n4) return;
  const float4* part = reinterpret_cast<const float4*>(a.part);
  float4 s = part[i];
  for (int z = 1; z < slices; ++z) {
    const float4 v = part[(size_t)z * n4 + i];
    s.x = s.x + v.x; s.y = s.y + v.y; s.z = s.z + v.z; s.w = s.w + v.w;
  }
  const int c4 = a.Cout / 4;
  const int ch = (int)(i % c4) * 4;
  const float sv[4] = {s.x, s.y, s.z, s.w};
  float o1[4], o2[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v = sv[k] + (a.bias ? a.bias[ch + k] : 0.f);
    if (a.res) {
      size_t ro = (size_t)i * 4 + k;
      if (a.y_nchw) { const int hw = a.Ho * a.Wo; const long p = (i * 4) / a.Cout, pn = p / hw; ro = ((size_t)pn * a.Cout + ch + k) * hw + (p - pn * hw); }
      v = v + a.res[ro];
    }
    if (a.relu) v = fmaxf(v, 0.f);
    o1[k] = v;
    o2[k] = a.y2 ? fmaxf(v * a.scale2[ch + k] + a.shift2[ch + k], 0.f) : 0.f;
  }
  if (a.y_nchw) {
    const long p = (i * 4) / a.Cout;
    const int hw = a.Ho * a.Wo;
    const long pn = p / hw, pr = p - pn * hw;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const size_t o = ((size_t)pn * a.Cout + ch + k) * hw + pr;
      a.y[o] = o1[k];
      if (a.y2) a.y2[o] = o2[k];
    }
    return;
  }
  reinterpret_cast<float4*>(a.y)[i] = make_float4(o1[0], o1[1], o1[2], o1[3]);
  if (a.y2) reinterpret_cast<float4*>(a.y2)[i] = make_float4(o2[0], o2[1], o2[2], o2[3]);
}

int pick_slices(long tiles, int taps) {
  // one 4-wave workgroup per tile: below ~1 workgroup per CU the SIMDs idle, so cut the taps into 3 (3x3 kernels)
  if (taps % 3 == 0 && tiles * 4 < 1024) return 3;
  return 1;
}

}  // namespace

extern "C" size_t lsfa_conv_nhwc_workspace_bytes(int N, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || stride <= 0) return 0;
  const int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  const long P = (long)N * Ho * Wo;
  const int slices = pick_slices(((P + kBM - 1) / kBM) * (Cout / kBN), kh * kw);
  return slices > 1 ? align_up((size_t)slices * P * Cout * sizeof(float), 256) : 256;
}

extern "C" int lsfa_conv_nhwc_fused_fwd(const float* x, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                                        int kh, int kw, int stride, int pad, int dil, int relu, const float* residual, float* y,
                                        float* y2, const float* scale2, const float* shift2, void* ws, size_t ws_bytes,
                                        void* stream) {
  LSFA_REQUIRE(x && w && y, "lsfa_conv_nhwc_fwd: NULL argument");
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && pad >= 0 && dil > 0, "lsfa_conv_nhwc_fwd: bad shape");
  LSFA_REQUIRE(!y2 || (scale2 && shift2), "lsfa_conv_nhwc_fused_fwd: y2 given without scale2 / shift2");
  LSFA_REQUIRE(!y2 || y2 != y, "lsfa_conv_nhwc_fused_fwd: y2 must not alias y");
  if (Cin % 32 != 0 || Cout % kBN != 0) {
    set_error("lsfa_conv_nhwc_fwd: Cin=%d must be a multiple of %d and Cout=%d of %d", Cin, 32, Cout, kBN);
    return LSFA_ENOTSUP;
  }
  const int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  LSFA_REQUIRE(Ho > 0 && Wo > 0, "lsfa_conv_nhwc_fwd: empty output");
  const long P = (long)N * Ho * Wo;
  LSFA_REQUIRE(P * Cout < (1L << 31) && (long)N * H * W * Cin < (1L << 33), "lsfa_conv_nhwc_fwd: tensor too large");
  const int taps = kh * kw;
  const long tiles = ((P + kBM - 1) / kBM) * (Cout / kBN);
  const int slices = pick_slices(tiles, taps);
  if (slices > 1 && (!ws || ws_bytes < lsfa_conv_nhwc_workspace_bytes(N, H, W, Cout, kh, kw, stride, pad, dil))) {
    set_error("lsfa_conv_nhwc_fwd: workspace %zu < %zu bytes", ws_bytes, lsfa_conv_nhwc_workspace_bytes(N, H, W, Cout, kh, kw, stride, pad, dil));
    return LSFA_EWORKSPACE;
  }
  hipStream_t s = (hipStream_t)stream;
  ConvArgs a = {x, w, bias, y, slices > 1 ? (float*)ws : nullptr, N, H, W, Cin, Cout, kh, kw, stride, pad, dil, Ho, Wo, relu,
                (taps + slices - 1) / slices, residual, y2, scale2, shift2, 0};
  ProfScope prof(LSFA_OP_CONV, s);
  const dim3 grid((unsigned)((P + kBM - 1) / kBM), Cout / kBN, slices);
  if (Cin % 64 == 0) hipLaunchKernelGGL(conv_igemm_kernel<64>, grid, dim3(kThreads), 0, s, a);
  else hipLaunchKernelGGL(conv_igemm_kernel<32>, grid, dim3(kThreads), 0, s, a);
  if (slices > 1) {
    const long n4 = P * Cout / 4;
    hipLaunchKernelGGL(conv_reduce_kernel, dim3((unsigned)((n4 + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, a, n4, slices);
  }
  LSFA_LAUNCH_CHECK("lsfa_conv_nhwc_fwd");
  return LSFA_OK;
}

extern "C" int lsfa_conv_nhwc_fwd(const float* x, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                                  int kh, int kw, int stride, int pad, int dil, int relu, float* y, void* ws, size_t ws_bytes,
                                  void* stream) {
  return lsfa_conv_nhwc_fused_fwd(x, N, H, W, Cin, w, bias, Cout, kh, kw, stride, pad, dil, relu, nullptr, y, nullptr, nullptr,
                                  nullptr, ws, ws_bytes, stream);
}

// ---- the same convolution on the bf16 matrix pipe with exactly split fp32 operands (conv_split_kernel.h) ----

namespace {
// how a split convolution is launched: which kernel, the tile grid, how K is cut
struct SplitPlan {
  int nt;               // general kernel: 32-channel column tiles per wave (2: conv_split_kernel, 4: conv_split_wide_kernel<4>)
  bool deep;            // general kernel, nt == 2: the four-stage ring (conv_split_deep_kernel), one workgroup per CU
  bool direct;          // conv_split_direct_kernel: operands straight into registers, a wave per 32 x 64 tile, no K slices
  bool halo;            // the 3x3 halo kernel (stride 1, pad = dilation 1 or 2), else the general one
  int dil;
  int patches_x, patches_y;
  int nx, ny, slices;   // tiles: nx pixel tiles x ny channel tiles x slices
  int per_slice;        // chunks (general: of taps*Cin/32; halo: of Cin/32) per slice
  int units_per_wg;     // halo, balanced mode: (tile, channel chunk) units per workgroup, else 0
  int max_pieces;       // ... and the most workgroups that can share one tile
};

// lab switch (lsfa_conv_split_set_variant): 0 = plan decides, 1 = the r2 kernel only (128 x 64 tiles, 2-stage ring), 2 = 128-channel tiles
// wherever Cout allows, 3 = plan decides but without the 4-stage ring
std::atomic<int> g_split_variant{0};

SplitPlan split_plan(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil) {
  SplitPlan p = {};
  p.nt = 2;
  const int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  p.ny = Cout / convsplit::kWgCh;
  p.dil = dil;
  p.halo = kh == 3 && kw == 3 && stride == 1 && pad == dil && (dil == 1 || dil == 2);
  // two 4-wave workgroups per CU is the design point: 512 workgroups fill the chip
  if (p.halo) {
    p.patches_x = (W + convsplit::kPatchCols - 1) / convsplit::kPatchCols;
    p.patches_y = (H + convsplit::kPatchRows - 1) / convsplit::kPatchRows;
    p.nx = N * p.patches_x * p.patches_y;
    const int cpt = Cin / 32;
    int s = 512 / (p.nx * p.ny > 0 ? p.nx * p.ny : 1);
    if (s < 1) s = 1;
    if (s > cpt) s = cpt;
    p.per_slice = (cpt + s - 1) / s;
    p.slices = (cpt + p.per_slice - 1) / p.per_slice;
    // measured (tools/lab/conv_split_lab.py): the halo form wins where its patches alone fill the chip (res2 conv2 30.7 vs
    // 34.6 us, the 256 -> 1024 fuse convolution 91 vs 106); where K must be cut anyway the general kernel's finer cut
    // (it slices taps x chunks) keeps more CUs busy (res4 conv2 32.7 vs 35.4, res3 conv2 31.4 vs 34.8)
    // (r3, tried: the small net's 64 -> 64 convolutions as 20 unsliced halo workgroups of 18 tap steps: 22-25 us against 16.5 + 8.4
    // for the sliced general tiles + reduce pass — a step is ~1.2 us of DMA latency when a workgroup has its CU to itself)
    if (p.slices > 1) p.halo = false;
    // balanced mode: fewer than 512 tiles but more than 512 (tile, chunk) units -> equal unit counts per workgroup
    // (fuse_reduce_add: 320 tiles x 8 chunks = 512 workgroups x 5 instead of one round of 320 x 8)
    const long tiles = (long)p.nx * p.ny, units = tiles * cpt;
    if (p.halo && tiles < 512 && units > 512) {
      const int per = (int)((units + 511) / 512);
      if (per * 4 <= cpt * 3) { p.units_per_wg = per; p.max_pieces = (cpt + per - 1) / per + 1; }     // worth it from 25 % shorter
    }
  }
  if (!p.halo) {
    const long P = (long)N * Ho * Wo;
    p.nx = (int)((P + convsplit::kWgPix - 1) / convsplit::kWgPix);
    const int chunk_total = kh * kw * (Cin / 32);
    const int variant = g_split_variant.load();
    // 128 x 128 workgroup tiles pay from ~128 chunks of K on (tools/lab/conv_split_lab.py, profiles/r3/conv_split_lab.txt: the DCN
    // contraction 92 -> 82 us, feat_conv_3x3 506 -> 491); on the short-K convolutions halving the tile count costs more
    // ... or when the 128 x 64 tiles alone overflow the 512 workgroup slots (the R-FCN score-map GEMM: 19 x 30 = 570 tiles of 16 chunks
    // = two rounds; 19 x 15 of the wide ones = one)
    if (Cout % 128 == 0 && variant != 1 && (variant == 2 || chunk_total >= 128 || (long)p.nx * (Cout / 64) > 512)) { p.nt = 4; p.ny = Cout / 128; }
    const double chunk_us = p.nt == 4 ? 1.6 : 1.2;
    // How many slices of K.  Two workgroups per CU run almost as fast as one each (measured: ~1.2 us per chunk either
    // way), so time ~ rounds of 512 workgroups x chunks per slice, plus the reduce pass over `slices` partial outputs
    // (~3 TB/s effective).  A 304-workgroup grid run as one round leaves 40 % of the slots empty for its whole length;
    // cut 5 ways it is 3 full rounds of a fifth each (feat_conv_3x3: 700 -> ~450 us).
    const long wgs = (long)p.nx * p.ny;
    const double out_mb = (double)P * Cout * 4.0 / 1e6;
    double best = 0;
    int best_s = 1;
    bool best_deep = false;
    // two ways to run a grid: two 2-stage workgroups per CU (512 slots; a chunk ~1.2 us each, the two overlapping) or one 4-stage
    // workgroup per CU (256 slots; ~0.95 us per chunk: three chunks in flight hide the DMA latency, what is left is one wave per
    // SIMD issuing 7 DMAs + ~110 cut instructions + 24 MFMAs).  The ring pays on grids that cannot fill 512 slots whichever way K
    // is cut (res4 conv1: 24.3 -> 20.9 us); measured per shape in profiles/r3/conv_split_lab.txt
    for (int deep = 0; deep <= ((p.nt == 2 && variant != 1 && variant != 3) ? 1 : 0); ++deep) {
      const double per_chunk = deep ? 0.95 : chunk_us;
      const long slots = deep ? 256 : 512;
      for (int s = 1; s <= 16; ++s) {
        if (s > 1 && chunk_total / s < (deep ? 6 : 8)) break;
        const int per = (chunk_total + s - 1) / s;
        const int used = (chunk_total + per - 1) / per;
        if (used != s) continue;
        const long rounds = (wgs * s + slots - 1) / slots;
        const double t = (double)rounds * per * per_chunk + (deep ? 1.5 : 0.0) + (s > 1 ? 3.0 + s * out_mb * 2.0 / 3.0 : 0.0);
        if ((s == 1 && deep == 0) || t < best * 0.97) { best = t; best_s = s; best_deep = deep != 0; }      // a finer cut must pay for itself
      }
    }
    p.deep = best_deep;
    p.per_slice = (chunk_total + best_s - 1) / best_s;
    p.slices = (chunk_total + p.per_slice - 1) / p.per_slice;
  }
  return p;
}

// the general kernel's plan for an output grid of Ho x Wo pixels (what a view launch falls back to)
SplitPlan split_plan_general(int N, int Ho, int Wo, int Cin, int Cout, int kh, int kw) {
  // stride 7 with a 1x1 input-independent shape: split_plan only looks at the OUTPUT grid for the general kernel, so describe
  // a convolution with that output: stride 1, no padding, input (Ho + kh - 1) x (Wo + kw - 1), and a stride that rules out the halo form
  SplitPlan p = split_plan(N, (Ho - 1) * 3 + kh, (Wo - 1) * 3 + kw, Cin, Cout, kh, kw, 3, 0, 1);
  return p;
}

// the plan of a launch whose operands may be views: the halo form needs the plain geometry
SplitPlan view_plan(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad_h, int pad_w, int dil, int lda, int Ho_grid,
                    int Wo_grid) {
  const int Ho = (H + 2 * pad_h - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad_w - dil * (kw - 1) - 1) / stride + 1;
  SplitPlan p = split_plan(N, H, W, Cin, Cout, kh, kw, stride, pad_h, dil);
  const bool plain = pad_h == pad_w && lda == Cin && Ho_grid == Ho && Wo_grid == Wo;
  if (!plain) p = split_plan_general(N, Ho_grid, Wo_grid, Cin, Cout, kh, kw);
  return p;
}

size_t split_workspace(const SplitPlan& p, long P, int Cout) {
  if (p.units_per_wg > 0)      // one 32 KB accumulator slot per (tile, piece)
    return (size_t)p.nx * p.ny * p.max_pieces * convsplit::kThreads * 32 * sizeof(float);
  return p.slices > 1 ? align_up((size_t)p.slices * P * Cout * sizeof(float), 256) : 256;
}
}  // namespace

extern "C" int lsfa_conv_split_set_variant(int variant) {
  LSFA_REQUIRE(variant >= 0 && variant <= 3, "lsfa_conv_split_set_variant: unknown variant %d", variant);
  g_split_variant.store(variant);
  return LSFA_OK;
}

extern "C" size_t lsfa_conv_split_weight_bytes(int Cout, int kh, int kw, int Cin) {
  if (Cout <= 0 || kh <= 0 || kw <= 0 || Cin <= 0 || Cin % 32 != 0 || Cout % 64 != 0) return 0;
  return (size_t)Cout * kh * kw * Cin * 6;      // three bf16 pieces per weight
}

extern "C" int lsfa_conv_split_weights(const float* w, int Cout, int kh, int kw, int Cin, void* wfrag, void* stream) {
  LSFA_REQUIRE(w && wfrag, "lsfa_conv_split_weights: NULL argument");
  if (lsfa_conv_split_weight_bytes(Cout, kh, kw, Cin) == 0) {
    set_error("lsfa_conv_split_weights: Cin=%d must be a multiple of 32 and Cout=%d of 64", Cin, Cout);
    return LSFA_ENOTSUP;
  }
  const long total = (long)kh * kw * (Cin / 32) * (Cout / 32) * 2 * 64;
  hipLaunchKernelGGL(convsplit::split_weights_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (uint4*)wfrag, Cout, kh * kw, Cin);
  LSFA_LAUNCH_CHECK("lsfa_conv_split_weights");
  return LSFA_OK;
}

namespace {
// every split convolution goes through here; the public entry points fill in what they expose
// the fp16 two-piece form runs on the 128 x 128 tiles only: K slices by the same cost model (a chunk ~1.1 us: 24 instead of 48 matrix
// instructions per wave next to the same DMAs)
SplitPlan wide_h_plan(long P, int Cin, int Cout, int kh, int kw) {
  SplitPlan p = {};
  p.nt = 4;
  p.ny = Cout / 128;
  p.nx = (int)((P + convsplit::kWgPix - 1) / convsplit::kWgPix);
  const int chunk_total = kh * kw * (Cin / 32);
  const long wgs = (long)p.nx * p.ny;
  const double out_mb = (double)P * Cout * 4.0 / 1e6;
  double best = 0;
  int best_s = 1;
  for (int s = 1; s <= 16; ++s) {
    if (s > 1 && chunk_total / s < 8) break;
    const int per = (chunk_total + s - 1) / s;
    if ((chunk_total + per - 1) / per != s) continue;
    const long rounds = (wgs * s + 511) / 512;
    const double t = (double)rounds * per * 1.1 + (s > 1 ? 3.0 + s * out_mb * 2.0 / 3.0 : 0.0);
    if (s == 1 || t < best * 0.97) { best = t; best_s = s; }
  }
  p.per_slice = (chunk_total + best_s - 1) / best_s;
  p.slices = (chunk_total + p.per_slice - 1) / p.per_slice;
  return p;
}

// LSFA_CONV_TILE_ORDER (lab): how workgroup ids map to (slice, channel tile, pixel tile), see xcd_tile
static int tile_order_from_env() {
  const char* e = getenv("LSFA_CONV_TILE_ORDER");
  return e ? atoi(e) : 0;
}

int conv_split_launch(convsplit::Args a, void* ws, size_t ws_bytes, void* stream, const char* who, int prof_op = LSFA_OP_CONV) {
  static const int tile_order = tile_order_from_env();
  a.tile_order = tile_order;
  const int N = a.N, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, kh = a.kh, kw = a.kw, stride = a.stride, dil = a.dil;
  LSFA_REQUIRE(a.x && a.wfrag && a.y, "%s: NULL argument", who);
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && a.pad_h >= 0 && a.pad_w >= 0 && dil > 0, "%s: bad shape", who);
  LSFA_REQUIRE(!a.y2 || (a.scale2 && a.shift2), "%s: y2 given without scale2 / shift2", who);
  LSFA_REQUIRE(!a.y2 || a.y2 != a.y, "%s: y2 must not alias y", who);
  if (Cin % 32 != 0 || Cout % convsplit::kWgCh != 0) {
    set_error("%s: Cin=%d must be a multiple of 32 and Cout=%d of %d", who, Cin, Cout, convsplit::kWgCh);
    return LSFA_ENOTSUP;
  }
  const int Ho = (H + 2 * a.pad_h - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * a.pad_w - dil * (kw - 1) - 1) / stride + 1;
  LSFA_REQUIRE(Ho > 0 && Wo > 0, "%s: empty output", who);
  if (a.Ho <= 0) a.Ho = Ho;        // a transposed convolution's phase passes its own (smaller) output grid
  if (a.Wo <= 0) a.Wo = Wo;
  // taps that fall outside the image read zeros wherever they are, so a grid may extend past the symmetric-padding output
  // (a transposed convolution's odd phase needs one more column on the right: padding 0 on the left, 1 on the right)
  LSFA_REQUIRE(a.Ho <= Ho + kh && a.Wo <= Wo + kw, "%s: output grid %dx%d far larger than the convolution's %dx%d", who, a.Ho, a.Wo, Ho, Wo);
  if (a.lda <= 0) a.lda = Cin;
  if (a.ldy <= 0) a.ldy = Cout;
  LSFA_REQUIRE(a.lda >= Cin && a.ldy >= Cout && a.lda % 4 == 0, "%s: lda %d / ldy %d smaller than the channel counts (or lda not a multiple of 4)", who, a.lda, a.ldy);
  LSFA_REQUIRE(!(a.y_nchw && (a.view || a.ldy != Cout)), "%s: an NCHW output cannot be a view", who);
  LSFA_REQUIRE(((uintptr_t)a.x & 15) == 0, "%s: x must be 16-byte aligned", who);
  if (!a.view) { a.out_H = a.Ho; a.out_W = a.Wo; a.out_sy = a.out_sx = 1; }
  const long P = (long)N * a.Ho * a.Wo;
  LSFA_REQUIRE(((long)N * a.out_H * a.out_W + 1) * (long)(a.y_nchw ? Cout : a.ldy) < (1L << 31) && P * Cout < (1L << 31) &&
               ((long)N * H * W + (long)(a.pad_h + 1) * (W + 1)) * a.lda < (1L << 31), "%s: tensor too large", who);
  SplitPlan p = view_plan(N, H, W, Cin, Cout, kh, kw, stride, a.pad_h, a.pad_w, dil, a.lda, a.Ho, a.Wo);
  // small weights on a small map: the direct kernel (a 64-channel group's weights <= 256 KB, re-read by every 32-pixel tile)
  // and the weights' re-reads stay modest: (P / 32) tiles x all weights <= 48 MB through L2
  if (a.nphase <= 1 && stride == 1 && (size_t)kh * kw * Cin * 64 * 6 <= (256u << 10) && P <= 16384 &&
      (size_t)((P + 31) / 32) * kh * kw * Cin * Cout * 6 <= (48u << 20) && g_split_variant.load() != 1) {
    p = SplitPlan{};
    p.direct = true;
    p.slices = 1;
  }
  const int nph = a.nphase > 1 ? a.nphase : 1;
  if (a.amax) {      // the fp16 two-piece form: 128 x 128 tiles, plain launches
    LSFA_REQUIRE(Cout % 128 == 0 && nph == 1, "%s: the fp16 form needs Cout %% 128 == 0 and no phases", who);
    p = wide_h_plan(P, Cin, Cout, kh, kw);
  }
  const size_t need = split_workspace(p, P, Cout) * (size_t)nph;
  if ((p.slices > 1 || p.units_per_wg > 0) && (!ws || ws_bytes < need)) {
    set_error("%s: workspace %zu < %zu bytes", who, ws_bytes, need);
    return LSFA_EWORKSPACE;
  }
  LSFA_REQUIRE(nph == 1 || (!p.halo && p.units_per_wg == 0), "%s: phases need the general kernel", who);
  hipStream_t s = (hipStream_t)stream;
  a.part = p.slices > 1 ? (float*)ws : nullptr;
  a.part_stride = P * Cout;
  a.chunks_per_slice = p.per_slice;
  a.units_per_wg = a.max_pieces = 0;
  ProfScope prof(prof_op, s);
  int tiles = p.nx * p.ny * p.slices * nph;
  if (p.units_per_wg > 0) {
    a.part = (float*)ws;
    a.units_per_wg = p.units_per_wg;
    a.max_pieces = p.max_pieces;
    tiles = (p.nx * p.ny * (Cin / 32) + p.units_per_wg - 1) / p.units_per_wg;      // workgroups
  }
  const dim3 grid((unsigned)(8 * ((tiles + 7) / 8)));
  if (p.direct) {
    const int nchunks = kh * kw * (Cin / 32);
    int nw = (nchunks + 1) / 2;            // at least two chunks per wave, at most kDirectMaxWaves waves
    if (nw > convsplit::kDirectMaxWaves) nw = convsplit::kDirectMaxWaves;
    if (nw < 1) nw = 1;
    hipLaunchKernelGGL(convsplit::conv_split_direct_kernel, dim3((unsigned)((P + 31) / 32), Cout / 64), dim3(64 * nw),
                       (size_t)(nw - 1) * 32 * 64 * sizeof(float), s, a);
  }
  else if (p.halo && p.dil == 1)
    hipLaunchKernelGGL(convsplit::conv_split3x3_kernel<1>, grid, dim3(convsplit::kThreads), 0, s, a, p.patches_x, p.patches_y, p.nx, p.ny, p.slices);
  else if (p.halo)
    hipLaunchKernelGGL(convsplit::conv_split3x3_kernel<2>, grid, dim3(convsplit::kThreads), 0, s, a, p.patches_x, p.patches_y, p.nx, p.ny, p.slices);
  else if (a.amax)
    hipLaunchKernelGGL((convsplit::conv_split_wide_kernel<4, 2>), grid, dim3(convsplit::kThreads), 0, s, a, p.nx, p.ny, p.slices * nph);
  else if (p.nt == 4)
    hipLaunchKernelGGL(convsplit::conv_split_wide_kernel<4>, grid, dim3(convsplit::kThreads), 0, s, a, p.nx, p.ny, p.slices * nph);
  else if (p.deep)
    hipLaunchKernelGGL(convsplit::conv_split_deep_kernel, grid, dim3(convsplit::kThreads), 0, s, a, p.nx, p.ny, p.slices * nph);
  else
    hipLaunchKernelGGL(convsplit::conv_split_kernel, grid, dim3(convsplit::kThreads), 0, s, a, p.nx, p.ny, p.slices * nph);
  if (p.units_per_wg > 0) {
    hipLaunchKernelGGL(convsplit::conv_split3x3_fixup_kernel, dim3((unsigned)(p.nx * p.ny)), dim3(convsplit::kThreads), 0, s, a, p.patches_x,
                       p.patches_y, p.nx);
  }
  if (p.slices > 1 && a.y_nchw && !a.res && !a.y2 && nph == 1 && Cout % 64 == 0) {
    hipLaunchKernelGGL(convsplit::split_reduce_nchw_kernel, dim3((unsigned)((P + 63) / 64), (unsigned)(Cout / 64)), dim3(convsplit::kThreads), 0,
                       s, a, p.slices);
  } else if (p.slices > 1) {
    const long n4 = P * Cout / 4;
    hipLaunchKernelGGL(convsplit::split_reduce_kernel, dim3((unsigned)((n4 + convsplit::kThreads - 1) / convsplit::kThreads), nph),
                       dim3(convsplit::kThreads), 0, s, a, n4, p.slices);
  }
  LSFA_LAUNCH_CHECK(who);
  return LSFA_OK;
}
}  // namespace

extern "C" size_t lsfa_conv_split_workspace_bytes(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0 || stride <= 0 || kh <= 0 || kw <= 0 || dil <= 0 || pad < 0) return 0;
  const int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const SplitPlan p = split_plan(N, H, W, Cin, Cout, kh, kw, stride, pad, dil);
  return split_workspace(p, (long)N * Ho * Wo, Cout);
}

// ---- r3 (opt-in): the fp16 two-piece form (three matrix instructions per product; conv_split_kernel.h) ------------------------------
extern "C" size_t lsfa_conv_split_h_weight_bytes(int Cout, int kh, int kw, int Cin) {
  if (Cout <= 0 || kh <= 0 || kw <= 0 || Cin <= 0) return 0;
  return (size_t)Cout * kh * kw * Cin * 4;        // two fp16 pieces per weight
}

extern "C" int lsfa_conv_split_h_weights(const float* w, int Cout, int kh, int kw, int Cin, int w_exp, void* out, void* stream) {
  LSFA_REQUIRE(w && out, "lsfa_conv_split_h_weights: NULL argument");
  LSFA_REQUIRE(Cout > 0 && Cout % 128 == 0 && Cin > 0 && Cin % 32 == 0 && kh > 0 && kw > 0,
               "lsfa_conv_split_h_weights: Cout=%d must be a multiple of 128 and Cin=%d of 32", Cout, Cin);
  LSFA_REQUIRE(w_exp > -120 && w_exp < 120, "lsfa_conv_split_h_weights: w_exp %d out of range", w_exp);
  const long total = (long)kh * kw * (Cin / 32) * (Cout / 32) * 2 * 64;
  hipLaunchKernelGGL(convsplit::split_weights_h_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (uint4*)out, Cout, kh * kw, Cin, w_exp);
  LSFA_LAUNCH_CHECK("lsfa_conv_split_h_weights");
  return LSFA_OK;
}

extern "C" int lsfa_amax_partial(const float* x, long long n, float* out, void* stream) {
  LSFA_REQUIRE(x && out && n > 0 && n % 4 == 0 && ((uintptr_t)x & 15) == 0, "lsfa_amax_partial: x must be 16-byte aligned, n a positive multiple of 4");
  hipLaunchKernelGGL(convsplit::amax_partial_kernel, dim3(convsplit::kAmaxSlots), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                     (long)(n / 4), out);
  LSFA_LAUNCH_CHECK("lsfa_amax_partial");
  return LSFA_OK;
}

extern "C" size_t lsfa_conv_split_h_workspace_bytes(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cout % 128 || Cin <= 0 || stride <= 0 || kh <= 0 || kw <= 0 || dil <= 0 || pad < 0) return 0;
  const int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 0;
  const SplitPlan p = wide_h_plan((long)N * Ho * Wo, Cin, Cout, kh, kw);
  return split_workspace(p, (long)N * Ho * Wo, Cout);
}

extern "C" int lsfa_conv_split_h_fwd(const float* x, const void* wfrag_h, int w_exp, const float* amax, const float* bias, int N, int H,
                                     int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil, int act, int y_nchw,
                                     float* y, void* ws, size_t ws_bytes, void* stream) {
  LSFA_REQUIRE(amax, "lsfa_conv_split_h_fwd: amax (lsfa_amax_partial of x, or of a map that bounds it) must be given");
  convsplit::Args a = {};
  a.x = x; a.wfrag = (const uint4*)wfrag_h; a.bias = bias; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.kh = kh; a.kw = kw; a.stride = stride; a.pad_h = a.pad_w = pad; a.dil = dil;
  a.act = act; a.y_nchw = y_nchw;
  a.amax = amax; a.w_exp = w_exp;
  return conv_split_launch(a, ws, ws_bytes, stream, "lsfa_conv_split_h_fwd");
}

// Deconvolution(kernel 4, stride 2, pad 0) + Crop(offset (1,1)) to Hc x Wc as ONE launch of four 2x2-tap phase convolutions.
// Output row 2m + py of the cropped map reads input rows (m - 1, m) through taps ky = (3, 1) when py = 0 and rows (m, m + 1)
// through ky = (2, 0) when py = 1 (columns alike): phase (py, px) is an ordinary 2x2 convolution with padding (1 - py, 1 - px)
// whose weights wfrag[py * 2 + px] the caller cut with lsfa_conv_split_weights from w[:, :, kys, kxs] (Cout, 2, 2, Cin).
extern "C" size_t lsfa_deconv4x4s2_crop_workspace_bytes(int N, int Hi, int Wi, int Cin, int Cout, int Hc, int Wc) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || Hc <= 0 || Wc <= 0) return 0;
  const int gh = (Hc + 1) / 2, gw = (Wc + 1) / 2;
  const SplitPlan p = split_plan_general(N, gh, gw, Cin, Cout, 2, 2);
  return split_workspace(p, (long)N * gh * gw, Cout) * 4;
}

extern "C" int lsfa_deconv4x4s2_crop_fwd(const float* x, int lda, int N, int Hi, int Wi, int Cin, const void* wfrag4,
                                         const float* bias, int Cout, int act, float* y, int ldy, int Hc, int Wc, void* ws,
                                         size_t ws_bytes, void* stream) {
  LSFA_REQUIRE(x && wfrag4 && y, "lsfa_deconv4x4s2_crop_fwd: NULL argument");
  LSFA_REQUIRE(act >= 0 && act <= 2 && Hc > 0 && Wc > 0 && Hc <= 2 * Hi + 1 && Wc <= 2 * Wi + 1, "lsfa_deconv4x4s2_crop_fwd: bad shape");
  convsplit::Args a = {};
  a.x = x; a.bias = bias; a.wfrag = (const uint4*)wfrag4; a.y = y;
  a.N = N; a.H = Hi; a.W = Wi; a.Cin = Cin; a.Cout = Cout; a.kh = a.kw = 2; a.stride = 1; a.dil = 1;
  a.act = act; a.lda = lda; a.ldy = ldy;
  a.view = 1; a.out_H = Hc; a.out_W = Wc; a.out_sy = a.out_sx = 2;
  a.nphase = 4;
  a.ph_wstride = (long)(lsfa_conv_split_weight_bytes(Cout, 2, 2, Cin) / 16);
  // the launch is sized for phase (0, 0), the largest grid
  a.pad_h = a.pad_w = 1; a.Ho = (Hc + 1) / 2; a.Wo = (Wc + 1) / 2;
  return conv_split_launch(a, ws, ws_bytes, stream, "lsfa_deconv4x4s2_crop_fwd", LSFA_OP_FLOWNET);
}

extern "C" size_t lsfa_conv_split_view_workspace_bytes(int lda, int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad_h,
                                                       int pad_w, int dil, int Ho, int Wo) {
  if (N <= 0 || H <= 0 || W <= 0 || Cout <= 0 || Cin <= 0 || stride <= 0 || kh <= 0 || kw <= 0 || dil <= 0 || pad_h < 0 || pad_w < 0) return 0;
  const int Hn = (H + 2 * pad_h - dil * (kh - 1) - 1) / stride + 1, Wn = (W + 2 * pad_w - dil * (kw - 1) - 1) / stride + 1;
  if (Ho <= 0) Ho = Hn;
  if (Wo <= 0) Wo = Wn;
  if (Ho <= 0 || Wo <= 0) return 0;
  const SplitPlan p = view_plan(N, H, W, Cin, Cout, kh, kw, stride, pad_h, pad_w, dil, lda > 0 ? lda : Cin, Ho, Wo);
  return split_workspace(p, (long)N * Ho * Wo, Cout);
}

extern "C" int lsfa_conv_split_fwd(const float* x, int N, int H, int W, int Cin, const void* wfrag, const float* bias, int Cout,
                                   int kh, int kw, int stride, int pad, int dil, int relu, int y_nchw, const float* residual, float* y,
                                   float* y2, const float* scale2, const float* shift2, void* ws, size_t ws_bytes, void* stream) {
  convsplit::Args a = {};
  a.x = x; a.wfrag = (const uint4*)wfrag; a.bias = bias; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.kh = kh; a.kw = kw; a.stride = stride; a.pad_h = a.pad_w = pad; a.dil = dil;
  a.act = relu ? 1 : 0; a.res = residual; a.y2 = y2; a.scale2 = scale2; a.shift2 = shift2; a.y_nchw = y_nchw;
  return conv_split_launch(a, ws, ws_bytes, stream, "lsfa_conv_split_fwd");
}

extern "C" int lsfa_conv_split_view_fwd(const float* x, int lda, int N, int H, int W, int Cin, const void* wfrag, const float* bias,
                                        int Cout, int kh, int kw, int stride, int pad_h, int pad_w, int dil, int act, float* y, int ldy,
                                        int Ho, int Wo, int out_H, int out_W, int out_sy, int out_sx, void* ws, size_t ws_bytes,
                                        void* stream) {
  LSFA_REQUIRE(act >= 0 && act <= 2, "lsfa_conv_split_view_fwd: act must be 0 (none), 1 (ReLU) or 2 (LeakyReLU 0.1)");
  convsplit::Args a = {};
  a.x = x; a.wfrag = (const uint4*)wfrag; a.bias = bias; a.y = y;
  a.N = N; a.H = H; a.W = W; a.Cin = Cin; a.Cout = Cout; a.kh = kh; a.kw = kw; a.stride = stride; a.pad_h = pad_h; a.pad_w = pad_w; a.dil = dil;
  a.act = act; a.lda = lda; a.ldy = ldy; a.Ho = Ho; a.Wo = Wo;
  if (out_H > 0) {
    LSFA_REQUIRE(out_W > 0 && out_sy > 0 && out_sx > 0, "lsfa_conv_split_view_fwd: bad output view");
    a.view = 1; a.out_H = out_H; a.out_W = out_W; a.out_sy = out_sy; a.out_sx = out_sx;
  }
  return conv_split_launch(a, ws, ws_bytes, stream, "lsfa_conv_split_view_fwd", LSFA_OP_FLOWNET);
}
