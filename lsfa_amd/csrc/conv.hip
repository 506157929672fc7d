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

#include <algorithm>

#include "conv_ring_kernel.h"

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

// ---- the split-operand convolution family (conv_split_kernel.h, conv_ring_kernel.h) -----------------------------------------------------
// fp32 in, fp32 accumulate, fp32 out; every fp32 product is formed on the bf16 / fp16 matrix pipe from `pieces` pieces per operand:
//   3  three bf16 pieces, six products (exact cut, no scale needed)
//   2  two fp16 pieces + a power-of-two scale per map (amax_in), three products: the default of the fp32 path since r4
//   1  one bf16 piece, one product: the bf16 mode (BASELINE configs[2])

namespace {
// how a convolution is launched: which kernel, the tile grid, how K is cut
struct SplitPlan {
  int nt;               // ring kernel: 32-channel column tiles per wave (2: 128 x 64 workgroup tiles, 4: 128 x 128)
  int st;               // ring kernel: stages of the LDS ring
  bool sp;              // ring kernel: split roles (512-thread workgroups: four loader waves + four consumer waves)
  int wv;               // ring kernel: waves that multiply (4: 128-pixel tiles; 8: 256-pixel tiles, eight mixed-role waves)
  bool direct;          // conv_split_direct_kernel: operands straight into registers, a wave per 32 x 64 tile, no K slices
  int nx, ny, slices;   // tiles: nx pixel tiles x ny channel tiles x slices
  int per_slice;        // chunks (of taps * Cin / 32) per slice
};

// lab override (lsfa_conv_plan_override): 0 = the plan decides
std::atomic<int> g_force_nt{0}, g_force_st{0}, g_force_slices{0}, g_force_kernel{0};
std::atomic<int> g_force_tile_order{-1}, g_force_k_order{-1};      // lsfa_conv_order_override: -1 = the default / environment

size_t ring_lds_bytes(int nt, int pieces, int st) { return (size_t)st * (16384 + (size_t)nt * pieces * 2048); }
bool ring_ok(int nt, int pieces, int st) {
  if (st < 2 || st > 4 || (nt != 2 && nt != 4) || pieces < 1 || pieces > 3) return false;
  if ((nt * pieces) % 2) return false;
  if (nt == 4 && pieces == 3 && st == 4) return false;       // 160 KB exactly: no room for anything else
  return ring_lds_bytes(nt, pieces, st) <= 160 * 1024 && (st - 1) * (4 + nt * pieces / 2) <= 63;
}

constexpr int kPlanDefault = 0;      // which r5 rules are on by default (bits of LSFA_CONV_PLAN_LAB)

// The ring kernel's plan, from the sweeps of tools/lab/conv_ring_lab.py over the network's shapes (profiles/r4/conv_ring_lab.txt: every
// tile width x ring depth x K cut x wave roles, hipGraph-timed):
//   * wave roles: loader / consumer waves (512-thread workgroups) win wherever a workgroup has at least four chunks of K to walk
//     (res4 conv1 21.7 -> 19.9 us, res3 conv2 31.9 -> 26.4, the DCN contraction 59.8 -> 56.2); on the two-chunk launches (res2
//     conv3: 37,500 pixels, K = 64) the extra waves only add launch weight (38 vs 48 us): mixed-role, two-stage;
//   * ring depth 3 (two chunks in flight) everywhere else: depth 4 never measured better, depth 2 loses 5-10 % on one-workgroup-per-CU grids;
//   * 128 x 128 tiles when there are >= 512 output channels and >= 64 chunks of K (the A tile's cut and copies feed twice the MFMAs),
//     128 x 64 otherwise (more tiles for the small maps of this network);
//   * K slices: by a small cost model over rounds of resident workgroups (below).
// r5: LSFA_CONV_PLAN_LAB (bit mask; in-situ A/B of the r5 variants against r4's plan, see profiles/r5/plan_ab.txt)
static int plan_lab_from_env() {
  const char* e = getenv("LSFA_CONV_PLAN_LAB");
  return e ? atoi(e) : -1;
}

// lab: LSFA_CONV_PLAN_LONGK="kernel,nt,st,slices" forces that plan on launches of >= 512 chunks of K only (feat_conv_3x3: 576) - an in-situ A/B of
// one layer's plan inside a whole pass (tools/lab/key_batch_probe.py)
struct LongKForce { int k, nt, st, s; };
static LongKForce longk_from_env() {
  LongKForce f = {0, 0, 0, 0};
  const char* e = getenv("LSFA_CONV_PLAN_LONGK");
  if (e) sscanf(e, "%d,%d,%d,%d", &f.k, &f.nt, &f.st, &f.s);
  return f;
}

// lab: LSFA_CONV_PLAN_AT="chunks,cout,kernel,nt,st,slices" forces that plan on the launches of exactly that K (in chunks of 32) and channel count
// (an in-situ A/B of one layer: the small net's fuse convolution is 72,1024; the R-FCN convolution 16,1920)
struct AtForce { int chunks, cout; LongKForce f; };
static AtForce at_from_env() {
  AtForce a = {0, 0, {0, 0, 0, 0}};
  const char* e = getenv("LSFA_CONV_PLAN_AT");
  if (e) sscanf(e, "%d,%d,%d,%d,%d,%d", &a.chunks, &a.cout, &a.f.k, &a.f.nt, &a.f.st, &a.f.s);
  return a;
}

void ring_plan(SplitPlan& p, long P, int chunk_total, int Cout, int pieces) {
  static const int lab_env = plan_lab_from_env();
  static const LongKForce longk_env = longk_from_env();
  static const AtForce at = at_from_env();
  const bool at_hit = at.chunks > 0 && at.chunks == chunk_total && at.cout == Cout;
  const LongKForce longk = at_hit ? at.f : longk_env;
  const int lab = lab_env >= 0 ? lab_env : kPlanDefault;
  p.wv = 4;
  const bool lk = (at_hit || chunk_total >= 512) && (longk.k || longk.nt || longk.st || longk.s);
  const int f_nt = lk ? longk.nt : g_force_nt.load(), f_st = lk ? longk.st : g_force_st.load(), f_s = lk ? longk.s : g_force_slices.load(),
            f_k = lk ? longk.k : g_force_kernel.load();
  p.nx = (int)((P + convsplit::kWgPix - 1) / convsplit::kWgPix);
  // 128 x 128 tiles: long K with >= 512 output channels (r4, one image), or - maps of several images, the batched pipeline - wherever the
  // wider tiles alone still give the chip well over a wave of workgroups (profiles/r4/conv_ring_lab_batch3.txt: res4 conv3 44.1 -> 39.3 us,
  // res5 conv3 108 -> 97, res5 sc 153 -> 127, res3 conv2 40.9 -> 38.6 at three images); never on launches of a few chunks (res2 conv3: 144 ->
  // 168; res3 conv3 at six images: 158 -> 179)
  const long tiles4 = (long)p.nx * (Cout / 128);
  int nt = (Cout % 128 == 0 &&
            ((Cout >= 512 && chunk_total >= 64) || (tiles4 >= 400 && chunk_total >= 8) || (tiles4 >= 200 && chunk_total >= 32 && Cout <= 256))) ? 4 : 2;
  if (f_nt && Cout % (32 * f_nt) == 0) nt = f_nt;
  const long tiles = (long)p.nx * (Cout / (32 * nt));
  // K slices: rounds of resident workgroups x chunks per slice (~0.9 us per chunk and workgroup with two chunks in flight: feat_conv_3x3 as
  // 456 workgroups of 192 chunks = two rounds on 256 one-workgroup CUs = 329 us; as 304 of 288 it is also two rounds: 431 us) + ~3.5 us
  // per round for dispatch, first arrival and epilogue + the reduce pass (measured 5.5 us + 0.125 us per MB of partial sums:
  // L2 / Infinity-Cache resident); never fewer than four chunks per slice
  const double out_mb = (double)P * Cout * 4.0 / 1e6;
  int s = 1;
  double best = 1e30;
  for (int c = 1; c <= 16; ++c) {
    if (c > 1 && chunk_total / c < 4) break;
    const int per_c = (chunk_total + c - 1) / c;
    if ((chunk_total + per_c - 1) / per_c != c) continue;
    // a CU's chunk rate is shared by the workgroups resident on it (456 workgroups on two slots per CU measured like two rounds:
    // res4 conv1 cut six ways 22.9 us against 19.9 cut three ways): rounds are counted per CU
    const long rounds = (tiles * c + 255) / 256;
    const double t = rounds * (per_c * 0.9 + 3.5) + (c > 1 ? 5.5 + 0.125 * c * out_mb : 0.0);
    if (t < best * 0.97) { best = t; s = c; }      // a finer cut must pay for itself
  }
  if (f_s) s = f_s;
  int per = (chunk_total + s - 1) / s;
  s = (chunk_total + per - 1) / per;                    // every slice non-empty
  bool sp = per >= 4;
  // the expanding 1x1s (conv3 of a bottleneck: K = Cout / 4, at most 16 chunks, an epilogue of residual + two outputs per tile): the four
  // loader waves only add launch weight; mixed roles, two stages (res5 conv3 42.7 -> 40.4 us, res3 conv3 31.1 -> 24.9; at three images
  // res5 conv3 113 -> 97, res4 conv3 44.1 -> 39.3)
  if (s == 1 && per <= 16 && Cout >= 128 * chunk_total) sp = false;
  // 128 x 128 tiles with at least ~1.5 workgroups per CU: two 256-thread mixed-role workgroups share a CU (two stages: 64 KB of LDS each)
  // and overlap each other better than one 512-thread workgroup's loader and consumer waves do (six images, conv_ring_lab_batch6.txt:
  // res5 shortcut 275 -> 257 us, res3 conv2 75 -> 67, res3 conv1 51 -> 44, feat_conv_3x3 1852 -> 1783; the DCN contraction 265 -> 270)
  if (s == 1 && nt == 4 && tiles4 >= 400) sp = false;
  // r5 candidates, OFF by default (kPlanDefault): the isolated-layer sweeps (profiles/r5/conv_ring_lab_batch*.txt) favour loader / consumer
  // waves everywhere and the 256 x 128 tiles for the wide short-K outputs, by 5-45 % per layer; IN the six-image backbone pass each of these
  // rules costs 2-4 % (profiles/r5/plan_ab.txt: 9499 us with r4's plan, 9690 / 9855 / 9781 / 9795 with rule 1 / 2 / 1+2 / all) - a layer
  // whose operands the previous layer left in L2 / the Infinity Cache, under the clocks of a 10 ms matrix-heavy pass, is not the layer
  // the lab times back to back on rotating buffers.  The plan follows the in-situ numbers.
  if ((lab & 2) && s == 1 && nt == 4 && tiles4 >= 400) sp = per >= 4;
  if ((lab & 4) && s == 1 && per <= 16 && Cout >= 128 * chunk_total) sp = per >= 4;
  // 256 x 128 tiles, eight mixed-role waves (r5): wide outputs, short or medium K, at least ~a wave of such tiles
  const long tiles8 = ((P + 255) / 256) * (Cout / 128);
  bool wv8 = (lab & 1) && s == 1 && pieces < 3 && Cout % 128 == 0 && Cout >= 1024 && chunk_total <= 72 && tiles8 >= 200 && f_k == 0 && f_nt == 0;
  if (f_k == 1) sp = false;
  if (f_k == 2) sp = true;
  if (f_k == 4 && nt == 4 && pieces < 3) wv8 = true;
  int st = sp ? 3 : 2;
  // the one-piece (bf16) form's 128 x 128 stage is 24 KB: two mixed-role workgroups per CU fit THREE stages each (144 KB), and the second
  // chunk in flight is worth 10-28 % at six images (profiles/r4/conv_ring_lab_batch6_bf16.txt: res4 conv3 67.8 -> 48.9 us, res5 conv3
  // 189 -> 141, res5 conv1 92 -> 79, the DCN contraction 187 -> 164); the 128 x 64 tiles and the two-piece form measured no better with it
  if (!sp && pieces == 1 && nt == 4) st = 3;
  if (f_st) st = f_st;
  while (st > 2 && !ring_ok(nt, pieces, st)) --st;
  if (wv8) { p.wv = 8; sp = false; nt = 4; st = (f_st == 2) ? 2 : 3; p.nx = (int)((P + 255) / 256); }
  p.nt = nt; p.st = st; p.slices = s; p.per_slice = per; p.sp = sp;
  p.ny = Cout / (32 * nt);
}

// (r2-r5 also had a 3x3 halo form - a workgroup staging a 4 x 32 output patch's input halo once per channel chunk; with loader / consumer
// waves the ring kernel passed it on its last shapes in r4 (res2 conv2 24.0 vs 27.1 us) and it was removed in r6; profiles/r4/conv_ring_lab.txt
// has its last numbers)
SplitPlan split_plan(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad, int dil, int pieces) {
  SplitPlan p = {};
  p.nt = 2; p.st = 2;
  const int Ho = (H + 2 * pad - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad - dil * (kw - 1) - 1) / stride + 1;
  ring_plan(p, (long)N * Ho * Wo, kh * kw * (Cin / 32), Cout, pieces);
  return p;
}

// the ring kernel's plan for an output grid of Ho x Wo pixels (what a view launch falls back to)
SplitPlan split_plan_general(int N, int Ho, int Wo, int Cin, int Cout, int kh, int kw, int pieces) {
  SplitPlan p = {};
  ring_plan(p, (long)N * Ho * Wo, kh * kw * (Cin / 32), Cout, pieces);
  return p;
}

// the plan of a launch whose operands may be views
SplitPlan view_plan(int N, int H, int W, int Cin, int Cout, int kh, int kw, int stride, int pad_h, int pad_w, int dil, int lda, int Ho_grid,
                    int Wo_grid, int pieces) {
  const int Ho = (H + 2 * pad_h - dil * (kh - 1) - 1) / stride + 1, Wo = (W + 2 * pad_w - dil * (kw - 1) - 1) / stride + 1;
  const bool plain = pad_h == pad_w && lda == Cin && Ho_grid == Ho && Wo_grid == Wo;
  if (!plain) return split_plan_general(N, Ho_grid, Wo_grid, Cin, Cout, kh, kw, pieces);
  return split_plan(N, H, W, Cin, Cout, kh, kw, stride, pad_h, dil, pieces);
}

size_t split_workspace(const SplitPlan& p, long P, int Cout) {
  return p.slices > 1 ? align_up((size_t)p.slices * P * Cout * sizeof(float), 256) : 256;
}

bool direct_fits(const convsplit::Args& a, long P, int pieces) {
  const size_t wbytes = (size_t)a.kh * a.kw * a.Cin * 2 * pieces;      // per output channel
  // r6: a narrow exact-cut launch (the RPN head: 512 -> 64 channels, three bf16 pieces) stays on the direct kernel for a whole segment's maps
  // (21,546 pixels) too - what its K-major form did until r5 whatever the size: on the ring kernel it would be 169 workgroups of 128 x 64 tiles,
  // and its K would be summed in one chain per output instead of the direct kernel's three (the ROI coordinates' margin against float64: 1.25 / 2.0)
  const bool narrow = pieces == 3 && a.Cout <= 64;
  return a.nphase <= 1 && a.stride == 1 && wbytes * 64 <= (256u << 10) && P <= (narrow ? 32768 : 16384) &&
         (size_t)((P + 31) / 32) * wbytes * a.Cout <= ((narrow ? 160u : 48u) << 20) && g_force_kernel.load() == 0;
}

// waves per tile of the direct kernel: at least two chunks per wave, at most kDirectMaxWaves waves
int direct_waves(const convsplit::Args& a) {
  const int nchunks = a.kh * a.kw * (a.Cin / 32);
  int nw = (nchunks + 1) / 2;
  if (nw > convsplit::kDirectMaxWaves) nw = convsplit::kDirectMaxWaves;
  return nw < 1 ? 1 : nw;
}

template <int PC, int ST>
void launch_ring8(const convsplit::Args& a, dim3 grid, hipStream_t s, int nx, int ny, int nz) {      // 256-pixel tiles: eight mixed-role waves
  if (a.in_scale) hipLaunchKernelGGL((convsplit::conv_ring_kernel<4, PC, ST, false, true, 8>), grid, dim3(512), 0, s, a, nx, ny, nz);
  else hipLaunchKernelGGL((convsplit::conv_ring_kernel<4, PC, ST, false, false, 8>), grid, dim3(512), 0, s, a, nx, ny, nz);
}
template <int NT, int PC, int ST>
void launch_ring(bool sp, const convsplit::Args& a, dim3 grid, hipStream_t s, int nx, int ny, int nz) {
  if (a.in_scale) {      // the input's bn + ReLU applied at the cut (pieces 1 and 2: the frame path's two modes)
    if (sp) hipLaunchKernelGGL((convsplit::conv_ring_kernel<NT, (PC < 3 ? PC : 2), ST, true, true>), grid, dim3(2 * convsplit::kThreads), 0, s, a, nx, ny, nz);
    else hipLaunchKernelGGL((convsplit::conv_ring_kernel<NT, (PC < 3 ? PC : 2), ST, false, true>), grid, dim3(convsplit::kThreads), 0, s, a, nx, ny, nz);
    return;
  }
  if (sp) hipLaunchKernelGGL((convsplit::conv_ring_kernel<NT, PC, ST, true>), grid, dim3(2 * convsplit::kThreads), 0, s, a, nx, ny, nz);
  else hipLaunchKernelGGL((convsplit::conv_ring_kernel<NT, PC, ST, false>), grid, dim3(convsplit::kThreads), 0, s, a, nx, ny, nz);
}
template <int PC>
bool launch_ring_pc(int nt, int st, bool sp, const convsplit::Args& a, dim3 grid, hipStream_t s, int nx, int ny, int nz) {
  if (nt == 2 && st == 2) launch_ring<2, PC, 2>(sp, a, grid, s, nx, ny, nz);
  else if (nt == 2 && st == 3) launch_ring<2, PC, 3>(sp, a, grid, s, nx, ny, nz);
  else if (nt == 2 && st == 4) launch_ring<2, PC, 4>(sp, a, grid, s, nx, ny, nz);
  else if (nt == 4 && st == 2) launch_ring<4, PC, 2>(sp, a, grid, s, nx, ny, nz);
  else if (nt == 4 && st == 3) launch_ring<4, PC, 3>(sp, a, grid, s, nx, ny, nz);
  else if (nt == 4 && st == 4 && PC < 3) launch_ring<4, (PC < 3 ? PC : 2), 4>(sp, a, grid, s, nx, ny, nz);
  else return false;
  return true;
}
template <int PC>
void launch_direct(const convsplit::Args& a, hipStream_t s, long P) {
  const int nw = direct_waves(a);
  hipLaunchKernelGGL(convsplit::conv_split_direct_kernel<PC>, dim3((unsigned)((P + 31) / 32), a.Cout / 64), dim3(64 * nw),
                     (size_t)(nw > 1 ? nw - 1 : 1) * 32 * 64 * sizeof(float), s, a);      // the waves' sums; at least the 8 KB the row epilogue uses
}

// LSFA_CONV_TILE_ORDER (lab): how workgroup ids map to (slice, channel tile, pixel tile), see xcd_tile
static int tile_order_from_env() {
  const char* e = getenv("LSFA_CONV_TILE_ORDER");
  return e ? atoi(e) : 1;
}

// LSFA_CONV_K_ORDER (lab): how the ring kernel walks K, see Walk (conv_split_kernel.h)
static int k_order_from_env() {
  const char* e = getenv("LSFA_CONV_K_ORDER");
  return e ? atoi(e) : 0;
}

// every split-operand convolution goes through here; the public entry points fill in what they expose
// validation, the launch plan, and every derived field of the argument block
int conv_split_prepare(convsplit::Args& a, int pieces, SplitPlan& p, long& P_out, const char* who) {
  static const int tile_order = tile_order_from_env();
  static const int k_order = k_order_from_env();
  const int f_to = g_force_tile_order.load(), f_ko = g_force_k_order.load();
  a.tile_order = f_to >= 0 ? f_to : tile_order;
  a.k_order = f_ko >= 0 ? f_ko : k_order;
  const int N = a.N, H = a.H, W = a.W, Cin = a.Cin, Cout = a.Cout, kh = a.kh, kw = a.kw, stride = a.stride, dil = a.dil;
  LSFA_REQUIRE(a.x && a.wfrag && a.y, "%s: NULL argument", who);
  LSFA_REQUIRE(pieces >= 1 && pieces <= 3, "%s: pieces must be 1 (bf16), 2 (fp16 hi / lo) or 3 (bf16 x 3), not %d", who, pieces);
  LSFA_REQUIRE(pieces != 2 || a.amax, "%s: the fp16 two-piece form needs amax_in (lsfa_amax_partial of x, a producer's amax_out, or a bound)", who);
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && kh > 0 && kw > 0 && stride > 0 && a.pad_h >= 0 && a.pad_w >= 0 && dil > 0, "%s: bad shape", who);
  LSFA_REQUIRE(!a.y2 || (a.scale2 && a.shift2), "%s: y2 given without scale2 / shift2", who);
  LSFA_REQUIRE((a.scale2 == nullptr) == (a.shift2 == nullptr), "%s: scale2 and shift2 go together", who);
  LSFA_REQUIRE(a.y2 || !a.scale2 || a.amax_out, "%s: scale2 / shift2 without y2 only publish the second output's maximum: amax_out is missing", who);
  LSFA_REQUIRE((a.in_scale == nullptr) == (a.in_shift == nullptr), "%s: in_scale and in_shift go together", who);
  if (a.in_scale && (a.kh != 1 || a.kw != 1 || a.pad_h || a.pad_w || a.Cin > convsplit::kAffineMaxCin || a.x_kmajor || a.nphase > 1 || pieces == 3)) {
    set_error("%s: in_scale / in_shift need a 1x1 convolution without padding on at most %d channels of a channels-last map, pieces 1 or 2", who,
              convsplit::kAffineMaxCin);
    return LSFA_ENOTSUP;
  }
  LSFA_REQUIRE(!a.y2 || a.y2 != a.y, "%s: y2 must not alias y", who);
  LSFA_REQUIRE(a.act >= 0 && a.act <= 2, "%s: act must be 0 (none), 1 (ReLU) or 2 (LeakyReLU 0.1)", who);
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
  LSFA_REQUIRE(P + 256 < (1L << 24), "%s: more than 2^24 output pixels", who);      // fdiv's range (conv_split_kernel.h)
  LSFA_REQUIRE(((long)N * a.out_H * a.out_W + 1) * (long)(a.y_nchw ? Cout : a.ldy) < (1L << 31) && P * Cout < (1L << 31) &&
               ((long)N * H * W + (long)(a.pad_h + 1) * (W + 1)) * a.lda < (1L << 31), "%s: tensor too large", who);
  p = view_plan(N, H, W, Cin, Cout, kh, kw, stride, a.pad_h, a.pad_w, dil, a.lda, a.Ho, a.Wo, pieces);
  if (a.x_kmajor) {
    // a K-major (NCHW) input exists in the direct kernel only: 1x1, stride 1, no padding, plain output grid, small weights
    const size_t wbytes = (size_t)Cin * 2 * pieces * 64;
    if (kh != 1 || kw != 1 || stride != 1 || a.pad_h || a.pad_w || a.nphase > 1 || a.lda < Cin || wbytes > (256u << 10) || a.Ho != H || a.Wo != W) {
      set_error("%s: an NCHW input (x_nchw) needs a 1x1 / stride 1 / pad 0 convolution with at most 256 KB of weights per 64 output channels", who);
      return LSFA_ENOTSUP;
    }
    LSFA_REQUIRE((long)N * a.lda * H * W < (1L << 31), "%s: tensor too large", who);
    p = SplitPlan{};
    p.direct = true;
    p.slices = 1;
  } else
  // small weights on a small map: the direct kernel (a 64-channel group's weights <= 256 KB at three pieces, re-read by every 32-pixel
  // tile) while the weights' re-reads stay modest: (P / 32) tiles x all weights <= 48 MB through L2
  if (direct_fits(a, P, pieces) && !g_force_nt.load() && !a.in_scale) {
    p = SplitPlan{};
    p.direct = true;
    p.slices = 1;
  }
  // ... except with the input's activation table in LDS (16 KB more per workgroup: three stages would leave one workgroup per CU)
  if (a.in_scale && !p.direct && !p.sp && p.wv == 4 && pieces == 1 && p.nt == 4 && p.st == 3 && g_force_st.load() == 0) p.st = 2;
  a.part_stride = P * Cout;
  a.chunks_per_slice = p.per_slice;
  a.inv_wo = 1.0f / (float)a.Wo;
  a.inv_howo = 1.0f / (float)(a.Ho * a.Wo);
  a.inv_nx = 1.0f / (float)(p.nx > 0 ? p.nx : 1);
  a.inv_ny = 1.0f / (float)(p.ny > 0 ? p.ny : 1);
  a.inv_cpt = 1.0f / (float)(Cin / 32);
  a.inv_kw = 1.0f / (float)kw;
  P_out = P;
  return LSFA_OK;
}

int conv_split_launch(convsplit::Args a, int pieces, void* ws, size_t ws_bytes, void* stream, const char* who, int prof_op = LSFA_OP_CONV) {
  SplitPlan p;
  long P = 0;
  const int rc = conv_split_prepare(a, pieces, p, P, who);
  if (rc != LSFA_OK) return rc;
  const int Cin = a.Cin, Cout = a.Cout;
  (void)Cin;
  const int nph = a.nphase > 1 ? a.nphase : 1;
  const size_t need = split_workspace(p, P, Cout) * (size_t)nph;
  if (p.slices > 1 && (!ws || ws_bytes < need)) {
    set_error("%s: workspace %zu < %zu bytes", who, ws_bytes, need);
    return LSFA_EWORKSPACE;
  }
  LSFA_REQUIRE(nph == 1 || !p.direct, "%s: phases need the ring kernel", who);
  hipStream_t s = (hipStream_t)stream;
  a.part = p.slices > 1 ? (float*)ws : nullptr;
  ProfScope prof(prof_op, s);
  const int tiles = p.nx * p.ny * p.slices * nph;
  const dim3 grid((unsigned)(8 * ((tiles + 7) / 8)));
  if (p.direct) {
    if (pieces == 3) launch_direct<3>(a, s, P);
    else if (pieces == 2) launch_direct<2>(a, s, P);
    else launch_direct<1>(a, s, P);
  } else {
    if (p.wv == 8) {
      if (pieces == 2 && p.st == 3) launch_ring8<2, 3>(a, grid, s, p.nx, p.ny, p.slices * nph);
      else if (pieces == 2) launch_ring8<2, 2>(a, grid, s, p.nx, p.ny, p.slices * nph);
      else if (p.st == 3) launch_ring8<1, 3>(a, grid, s, p.nx, p.ny, p.slices * nph);
      else launch_ring8<1, 2>(a, grid, s, p.nx, p.ny, p.slices * nph);
    } else {
    const bool ok = pieces == 3 ? launch_ring_pc<3>(p.nt, p.st, p.sp, a, grid, s, p.nx, p.ny, p.slices * nph)
                  : pieces == 2 ? launch_ring_pc<2>(p.nt, p.st, p.sp, a, grid, s, p.nx, p.ny, p.slices * nph)
                                : launch_ring_pc<1>(p.nt, p.st, p.sp, a, grid, s, p.nx, p.ny, p.slices * nph);
    LSFA_REQUIRE(ok, "%s: no ring kernel for nt=%d st=%d pieces=%d", who, p.nt, p.st, pieces);
    }
  }
  if (p.slices > 1 && a.y_nchw && !a.res && !a.scale2 && nph == 1 && Cout % 64 == 0) {
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

convsplit::Args args_of(const lsfa_conv_desc& d) {
  convsplit::Args a = {};
  a.x = d.x; a.wfrag = (const uint4*)d.wfrag; a.bias = d.bias; a.y = d.y;
  a.N = d.N; a.H = d.H; a.W = d.W; a.Cin = d.Cin; a.Cout = d.Cout; a.kh = d.kh; a.kw = d.kw; a.stride = d.stride;
  a.pad_h = d.pad_h; a.pad_w = d.pad_w; a.dil = d.dil; a.act = d.act; a.y_nchw = d.y_nchw;
  a.res = d.residual; a.y2 = d.y2; a.scale2 = d.scale2; a.shift2 = d.shift2;
  a.lda = d.lda; a.ldy = d.ldy; a.Ho = d.Ho; a.Wo = d.Wo;
  if (d.out_H > 0) { a.view = 1; a.out_H = d.out_H; a.out_W = d.out_W; a.out_sy = d.out_sy; a.out_sx = d.out_sx; }
  a.amax = d.amax_in; a.w_exp = d.w_exp; a.amax_out = d.amax_out; a.status = d.status;
  a.wscale = d.w_scale;
  a.x_kmajor = d.x_nchw ? 1 : 0;
  a.in_scale = d.in_scale; a.in_shift = d.in_shift;
  return a;
}
}  // namespace

extern "C" int lsfa_conv_plan_override(int kernel, int nt, int st, int slices) {
  LSFA_REQUIRE(kernel >= 0 && kernel <= 4 && kernel != 3 && (nt == 0 || nt == 2 || nt == 4) && (st == 0 || (st >= 2 && st <= 4)) && slices >= 0 && slices <= 16,
               "lsfa_conv_plan_override: kernel 0, 1, 2 or 4 (3 was the halo form, removed in r6), nt 0/2/4, st 0/2..4, slices 0..16");
  g_force_kernel.store(kernel); g_force_nt.store(nt); g_force_st.store(st); g_force_slices.store(slices);
  return LSFA_OK;
}

extern "C" int lsfa_conv_order_override(int tile_order, int k_order) {
  LSFA_REQUIRE(tile_order >= -1 && tile_order <= 1 && k_order >= -1 && k_order <= 1, "lsfa_conv_order_override: tile_order and k_order are -1 (default), 0 or 1");
  g_force_tile_order.store(tile_order); g_force_k_order.store(k_order);
  return LSFA_OK;
}

extern "C" size_t lsfa_conv_weight_bytes(int Cout, int kh, int kw, int Cin, int pieces) {
  if (Cout <= 0 || kh <= 0 || kw <= 0 || Cin <= 0 || Cin % 32 != 0 || Cout % 64 != 0 || pieces < 1 || pieces > 3) return 0;
  return (size_t)Cout * kh * kw * Cin * 2 * pieces;
}

extern "C" int lsfa_conv_weights(const float* w, int Cout, int kh, int kw, int Cin, int pieces, int w_exp, void* wfrag, void* stream) {
  LSFA_REQUIRE(w && wfrag, "lsfa_conv_weights: NULL argument");
  if (lsfa_conv_weight_bytes(Cout, kh, kw, Cin, pieces) == 0) {
    set_error("lsfa_conv_weights: Cin=%d must be a multiple of 32, Cout=%d of 64, pieces=%d one of 1, 2, 3", Cin, Cout, pieces);
    return LSFA_ENOTSUP;
  }
  LSFA_REQUIRE(w_exp > -120 && w_exp < 120 && (pieces == 2 || w_exp == 0), "lsfa_conv_weights: w_exp %d out of range (it is 0 unless pieces == 2)", w_exp);
  const long total = (long)kh * kw * (Cin / 32) * (Cout / 32) * 2 * 64;
  const dim3 grid((unsigned)((total + 255) / 256));
  hipStream_t s = (hipStream_t)stream;
  if (pieces == 3) hipLaunchKernelGGL(convsplit::pack_weights_kernel<3>, grid, dim3(256), 0, s, w, (uint4*)wfrag, Cout, kh * kw, Cin, 0);
  else if (pieces == 2) hipLaunchKernelGGL(convsplit::pack_weights_kernel<2>, grid, dim3(256), 0, s, w, (uint4*)wfrag, Cout, kh * kw, Cin, w_exp);
  else hipLaunchKernelGGL(convsplit::pack_weights_kernel<1>, grid, dim3(256), 0, s, w, (uint4*)wfrag, Cout, kh * kw, Cin, 0);
  LSFA_LAUNCH_CHECK("lsfa_conv_weights");
  return LSFA_OK;
}

// r5: the two-piece form with one power-of-two scale per OUTPUT channel: w_exp_pc[co] (device, Cout ints) scales channel co's weights where
// they are cut; the caller passes wscale[co] = 2^-w_exp_pc[co] (floats, device) as lsfa_conv_desc::w_scale.
extern "C" int lsfa_conv_weights_pc(const float* w, int Cout, int kh, int kw, int Cin, const int* w_exp_pc, void* wfrag, void* stream) {
  LSFA_REQUIRE(w && wfrag && w_exp_pc, "lsfa_conv_weights_pc: NULL argument");
  if (lsfa_conv_weight_bytes(Cout, kh, kw, Cin, 2) == 0) {
    set_error("lsfa_conv_weights_pc: Cin=%d must be a multiple of 32, Cout=%d of 64", Cin, Cout);
    return LSFA_ENOTSUP;
  }
  const long total = (long)kh * kw * (Cin / 32) * (Cout / 32) * 2 * 64;
  hipLaunchKernelGGL(convsplit::pack_weights_kernel<2>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, (uint4*)wfrag, Cout,
                     kh * kw, Cin, 0, w_exp_pc);
  LSFA_LAUNCH_CHECK("lsfa_conv_weights_pc");
  return LSFA_OK;
}

extern "C" size_t lsfa_conv_workspace_bytes(const lsfa_conv_desc* d) {
  if (!d || d->N <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0 || d->Cin <= 0 || d->stride <= 0 || d->kh <= 0 || d->kw <= 0 || d->dil <= 0 ||
      d->pad_h < 0 || d->pad_w < 0 || d->pieces < 1 || d->pieces > 3 || d->Cin % 32 || d->Cout % 64)
    return 0;
  const int Hn = (d->H + 2 * d->pad_h - d->dil * (d->kh - 1) - 1) / d->stride + 1, Wn = (d->W + 2 * d->pad_w - d->dil * (d->kw - 1) - 1) / d->stride + 1;
  const int Ho = d->Ho > 0 ? d->Ho : Hn, Wo = d->Wo > 0 ? d->Wo : Wn;
  if (Ho <= 0 || Wo <= 0) return 0;
  const SplitPlan p = view_plan(d->N, d->H, d->W, d->Cin, d->Cout, d->kh, d->kw, d->stride, d->pad_h, d->pad_w, d->dil, d->lda > 0 ? d->lda : d->Cin, Ho, Wo,
                                d->pieces);
  return split_workspace(p, (long)d->N * Ho * Wo, d->Cout);
}

extern "C" int lsfa_conv_fwd(const lsfa_conv_desc* d, void* ws, size_t ws_bytes, void* stream) {
  LSFA_REQUIRE(d, "lsfa_conv_fwd: NULL descriptor");
  return conv_split_launch(args_of(*d), d->pieces, ws, ws_bytes, stream, "lsfa_conv_fwd", LSFA_OP_CONV);
}

extern "C" int lsfa_conv_plan_query(const lsfa_conv_desc* d, int* out8) {
  LSFA_REQUIRE(d && out8, "lsfa_conv_plan_query: NULL argument");
  convsplit::Args a = args_of(*d);
  SplitPlan p;
  long P = 0;
  const int rc = conv_split_prepare(a, d->pieces, p, P, "lsfa_conv_plan_query");
  if (rc != LSFA_OK) return rc;
  out8[0] = p.direct ? 2 : 1;
  out8[1] = p.direct ? 2 : p.nt; out8[2] = p.st; out8[3] = p.sp ? 1 : 0; out8[4] = p.wv ? p.wv : 4; out8[5] = p.slices;
  out8[6] = a.in_scale ? 1 : 0; out8[7] = d->pieces;
  return LSFA_OK;
}

extern "C" int lsfa_amax_partial(const float* x, long long n, float* out, void* stream) {
  LSFA_REQUIRE(x && out && n > 0 && n % 4 == 0 && ((uintptr_t)x & 15) == 0, "lsfa_amax_partial: x must be 16-byte aligned, n a positive multiple of 4");
  hipLaunchKernelGGL(convsplit::amax_partial_kernel, dim3(convsplit::kAmaxSlots), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                     (long)(n / 4), out);
  LSFA_LAUNCH_CHECK("lsfa_amax_partial");
  return LSFA_OK;
}

// The device's status word (bit 0: a convolution wrote a non-finite value - with the fp16 form that is what an under-estimated amax
// produces; bit 1: a convolution's INPUT maximum was already inf / NaN): read it (synchronising `stream`), clear it, and turn a set bit
// into an error.
extern "C" int lsfa_status_check(unsigned* status_dev, void* stream) {
  LSFA_REQUIRE(status_dev, "lsfa_status_check: NULL status word");
  unsigned h = 0;
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemcpyAsync(&h, status_dev, sizeof(h), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return hip_fail(e, "lsfa_status_check");
  if (h == 0) return LSFA_OK;
  e = hipMemsetAsync(status_dev, 0, sizeof(unsigned), s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) return hip_fail(e, "lsfa_status_check");
  set_error("lsfa_status_check: status 0x%x:%s%s", h,
            (h & 1u) ? " a convolution produced a non-finite output (fp16 two-piece form: amax_in under-estimates max|x|, or the input held inf / NaN)" : "",
            (h & 2u) ? " a convolution's input maximum was inf / NaN" : "");
  return LSFA_EOVERFLOW;
}

// Deconvolution(kernel 4, stride 2, pad 0) + Crop(offset (1,1)) to Hc x Wc as ONE launch of four 2x2-tap phase convolutions.
// Output row 2m + py of the cropped map reads input rows (m - 1, m) through taps ky = (3, 1) when py = 0 and rows (m, m + 1)
// through ky = (2, 0) when py = 1 (columns alike): phase (py, px) is an ordinary 2x2 convolution with padding (1 - py, 1 - px)
// whose weights wfrag[py * 2 + px] the caller cut with lsfa_conv_weights from w[:, :, kys, kxs] (Cout, 2, 2, Cin).
extern "C" size_t lsfa_deconv4x4s2_crop_workspace_bytes(int N, int Hi, int Wi, int Cin, int Cout, int Hc, int Wc, int pieces) {
  if (N <= 0 || Hi <= 0 || Wi <= 0 || Cin <= 0 || Cout <= 0 || Hc <= 0 || Wc <= 0 || pieces < 1 || pieces > 3) return 0;
  const int gh = (Hc + 1) / 2, gw = (Wc + 1) / 2;
  const SplitPlan p = split_plan_general(N, gh, gw, Cin, Cout, 2, 2, pieces);
  return split_workspace(p, (long)N * gh * gw, Cout) * 4;
}

extern "C" int lsfa_deconv4x4s2_crop_fwd(const float* x, int lda, int N, int Hi, int Wi, int Cin, const void* wfrag4, int pieces, int w_exp,
                                         const float* amax_in, const float* bias, int Cout, int act, float* y, int ldy, int Hc, int Wc,
                                         unsigned* amax_out, unsigned* status, void* ws, size_t ws_bytes, void* stream) {
  LSFA_REQUIRE(x && wfrag4 && y, "lsfa_deconv4x4s2_crop_fwd: NULL argument");
  LSFA_REQUIRE(act >= 0 && act <= 2 && Hc > 0 && Wc > 0 && Hc <= 2 * Hi + 1 && Wc <= 2 * Wi + 1, "lsfa_deconv4x4s2_crop_fwd: bad shape");
  convsplit::Args a = {};
  a.x = x; a.bias = bias; a.wfrag = (const uint4*)wfrag4; a.y = y;
  a.N = N; a.H = Hi; a.W = Wi; a.Cin = Cin; a.Cout = Cout; a.kh = a.kw = 2; a.stride = 1; a.dil = 1;
  a.act = act; a.lda = lda; a.ldy = ldy;
  a.view = 1; a.out_H = Hc; a.out_W = Wc; a.out_sy = a.out_sx = 2;
  a.nphase = 4;
  a.ph_wstride = (long)(lsfa_conv_weight_bytes(Cout, 2, 2, Cin, pieces) / 16);
  a.amax = amax_in; a.w_exp = w_exp; a.amax_out = amax_out; a.status = status;
  // the launch is sized for phase (0, 0), the largest grid
  a.pad_h = a.pad_w = 1; a.Ho = (Hc + 1) / 2; a.Wo = (Wc + 1) / 2;
  return conv_split_launch(a, pieces, ws, ws_bytes, stream, "lsfa_deconv4x4s2_crop_fwd", LSFA_OP_CONV);
}
