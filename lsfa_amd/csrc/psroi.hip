// Position-sensitive ROI pooling (drop-in operator) and the fused R-FCN head tail
// (PSROI pooling of both maps + 7x7 average + class softmax in one launch).
// Reference interfaces: include/lsfa_hip.h.  Arithmetic: oracle orc_psroi_pool,
// orc_global_avg, orc_softmax_rows, operation for operation (-ffp-contract=off, fmaf
// exactly where the oracle has it).
#include <float.h>

#include "common.h"

namespace {

constexpr int kThreads = 256;

struct RoiGeom {
  int batch;
  float start_w, start_h, bin_w, bin_h;
};

// psroi_pooling.cu:50-64
__device__ __forceinline__ RoiGeom roi_geom(const float* roi, float scale, int P) {
  RoiGeom g;
  g.batch = (int)roi[0];
  g.start_w = roundf(roi[1]) * scale;
  g.start_h = roundf(roi[2]) * scale;
  const float end_w = (float)((double)roundf(roi[3]) + 1.) * scale;
  const float end_h = (float)((double)roundf(roi[4]) + 1.) * scale;
  const float roi_width = (float)fmax((double)(end_w - g.start_w), 0.1);
  const float roi_height = (float)fmax((double)(end_h - g.start_h), 0.1);
  g.bin_h = roi_height / (float)P;
  g.bin_w = roi_width / (float)P;
  return g;
}

// one bin: psroi_pooling.cu:66-99.  Returns the average; *c_out = source channel.
__device__ __forceinline__ float pool_bin(const float* __restrict__ data, const RoiGeom& g, int C, int H,
                                          int W, int ctop, int ph, int pw, int P, int group, int* c_out) {
  int hstart = (int)floorf(fmaf((float)ph, g.bin_h, g.start_h));
  int wstart = (int)floorf(fmaf((float)pw, g.bin_w, g.start_w));
  int hend = (int)ceilf(fmaf((float)(ph + 1), g.bin_h, g.start_h));
  int wend = (int)ceilf(fmaf((float)(pw + 1), g.bin_w, g.start_w));
  hstart = min(max(hstart, 0), H); hend = min(max(hend, 0), H);
  wstart = min(max(wstart, 0), W); wend = min(max(wend, 0), W);
  const bool is_empty = (hend <= hstart) || (wend <= wstart);
  int gw = (int)floorf((float)pw * (float)group / (float)P);
  int gh = (int)floorf((float)ph * (float)group / (float)P);
  gw = min(max(gw, 0), group - 1);
  gh = min(max(gh, 0), group - 1);
  const int c = (ctop * group + gh) * group + gw;
  if (c_out) *c_out = c;
  const float* plane = data + ((size_t)g.batch * C + c) * H * W;
  float out_sum = 0.f;
  for (int h = hstart; h < hend; ++h)
    for (int w = wstart; w < wend; ++w) out_sum += plane[h * W + w];
  const float bin_area = (float)((hend - hstart) * (wend - wstart));
  return is_empty ? 0.f : out_sum / bin_area;
}

// Drop-in operator: one thread per output bin, (n, ctop, ph, pw) order like the reference.
__global__ __launch_bounds__(kThreads) void psroi_kernel(const float* __restrict__ data,
                                                         const float* __restrict__ rois, int C, int H, int W,
                                                         int R, float scale, int output_dim, int P, int group,
                                                         float* __restrict__ out, float* __restrict__ mapping) {
  const size_t count = (size_t)R * output_dim * P * P;
  for (size_t index = (size_t)blockIdx.x * kThreads + threadIdx.x; index < count;
       index += (size_t)gridDim.x * kThreads) {
    const int pw = (int)(index % P);
    const int ph = (int)((index / P) % P);
    const int ctop = (int)((index / P / P) % output_dim);
    const int n = (int)(index / P / P / output_dim);
    const RoiGeom g = roi_geom(rois + (size_t)n * 5, scale, P);
    int c;
    out[index] = pool_bin(data, g, C, H, W, ctop, ph, pw, P, group, &c);
    if (mapping) mapping[index] = (float)c;
  }
}

// Fused head: one 16-wave workgroup per ROI (the ~2 bins per thread keep ~20k scattered loads in
// flight per CU; the bins are latency- not bandwidth-bound).  Bins of both maps -> LDS, 49-bin
// averages in (ph,pw) order -> LDS, then the class softmax (sequential sum = oracle order).
constexpr int kHeadThreads = 1024;
__global__ __launch_bounds__(kHeadThreads) void rfcn_head_kernel(
    const float* __restrict__ cls_map, const float* __restrict__ box_map, const float* __restrict__ rois,
    int H, int W, int ncls, int nbox, float scale, int P, int group, float* __restrict__ cls_prob,
    float* __restrict__ cls_score, float* __restrict__ bbox_pred) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int r = blockIdx.x;
  const int PP = P * P;
  const int ndim = ncls + nbox;
  float* bins = smem;               // ndim * PP
  float* avg = smem + ndim * PP;    // ndim
  const RoiGeom g = roi_geom(rois + (size_t)r * 5, scale, P);
  const int Ccls = ncls * group * group, Cbox = nbox * group * group;
  for (int i = threadIdx.x; i < ndim * PP; i += kHeadThreads) {
    const int d = i / PP, k = i - d * PP;
    const int ph = k / P, pw = k - ph * P;
    bins[i] = d < ncls ? pool_bin(cls_map, g, Ccls, H, W, d, ph, pw, P, group, nullptr)
                       : pool_bin(box_map, g, Cbox, H, W, d - ncls, ph, pw, P, group, nullptr);
  }
  __syncthreads();
  for (int d = threadIdx.x; d < ndim; d += kHeadThreads) {
    float s = 0.f;
    for (int k = 0; k < PP; ++k) s += bins[d * PP + k];
    const float a = s / (float)PP;
    avg[d] = a;
    if (d >= ncls) bbox_pred[(size_t)r * nbox + (d - ncls)] = a;
    else if (cls_score) cls_score[(size_t)r * ncls + d] = a;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    // lanes compute the exps in parallel; every lane then sums them in class order
    float m = avg[0];
    for (int j = 1; j < ncls; ++j) m = fmaxf(m, avg[j]);
    for (int j = threadIdx.x; j < ncls; j += 64) bins[j] = expf_cr(avg[j] - m);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int j = 0; j < ncls; ++j) s += bins[j];
    for (int j = threadIdx.x; j < ncls; j += 64) cls_prob[(size_t)r * ncls + j] = bins[j] / s;
  }
}


// Same head on the position-sensitive layout ps_map[n][h][w][gh*G+gw][dim], dim = classes then box
// coordinates: what the two R-FCN 1x1 convolutions produce when run as ONE GEMM
// [H*W, 512] x [512, G*G*(ncls+nbox)] with permuted weight rows.  Thread i -> (bin = i / D,
// dim = i % D): the D = ncls+nbox values of a (cell, bin) are contiguous, so a wave reads whole
// 156-byte segments instead of 64 scattered words from 64 channel planes.  Sum order inside a bin
// is unchanged (h outer, w inner), so the result is bit-identical to the NCHW kernel's.
__global__ __launch_bounds__(kHeadThreads) void rfcn_head_ps_kernel(
    const float* __restrict__ ps_map, const float* __restrict__ rois, int H, int W, int ncls, int nbox,
    float scale, int P, int group, int cell_ld, float* __restrict__ cls_prob, float* __restrict__ cls_score,
    float* __restrict__ bbox_pred) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int r = blockIdx.x;
  const int PP = P * P;
  const int D = ncls + nbox;
  float* bins = smem;            // D * PP, index d*PP + k like the NCHW kernel
  float* avg = smem + D * PP;    // D
  const RoiGeom g = roi_geom(rois + (size_t)r * 5, scale, P);
  const size_t cell_stride = (size_t)cell_ld;        // floats between consecutive cells (>= group*group*D: rows may be padded)
  const float* base = ps_map + (size_t)g.batch * H * W * cell_stride;
  for (int i = threadIdx.x; i < D * PP; i += kHeadThreads) {
    const int k = i / D, d = i - k * D;
    const int ph = k / P, pw = k - ph * P;
    int hstart = (int)floorf(fmaf((float)ph, g.bin_h, g.start_h));
    int wstart = (int)floorf(fmaf((float)pw, g.bin_w, g.start_w));
    int hend = (int)ceilf(fmaf((float)(ph + 1), g.bin_h, g.start_h));
    int wend = (int)ceilf(fmaf((float)(pw + 1), g.bin_w, g.start_w));
    hstart = min(max(hstart, 0), H); hend = min(max(hend, 0), H);
    wstart = min(max(wstart, 0), W); wend = min(max(wend, 0), W);
    const bool is_empty = (hend <= hstart) || (wend <= wstart);
    int gw = (int)floorf((float)pw * (float)group / (float)P);
    int gh = (int)floorf((float)ph * (float)group / (float)P);
    gw = min(max(gw, 0), group - 1);
    gh = min(max(gh, 0), group - 1);
    const float* p = base + (size_t)(gh * group + gw) * D + d;
    // the reference's order of additions (row by row, left to right) over the bin's pixels taken as ONE run, so that eight loads are in
    // flight before the first of them is added whatever the bin's width
    float out_sum = 0.f;
    {
      const int bw = wend - wstart, n = is_empty ? 0 : (hend - hstart) * bw;
      int h = hstart, w = wstart, i8 = 0;
      for (; i8 + 8 <= n; i8 += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          v[u] = p[((size_t)h * W + w) * cell_stride];
          if (++w == wend) { w = wstart; ++h; }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) out_sum += v[u];
      }
      for (; i8 < n; ++i8) {
        out_sum += p[((size_t)h * W + w) * cell_stride];
        if (++w == wend) { w = wstart; ++h; }
      }
    }
    const float bin_area = (float)((hend - hstart) * (wend - wstart));
    bins[d * PP + k] = is_empty ? 0.f : out_sum / bin_area;
  }
  __syncthreads();
  for (int d = threadIdx.x; d < D; d += kHeadThreads) {
    float s = 0.f;
    for (int k = 0; k < PP; ++k) s += bins[d * PP + k];
    const float a = s / (float)PP;
    avg[d] = a;
    if (d >= ncls) bbox_pred[(size_t)r * nbox + (d - ncls)] = a;
    else if (cls_score) cls_score[(size_t)r * ncls + d] = a;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float m = avg[0];
    for (int j = 1; j < ncls; ++j) m = fmaxf(m, avg[j]);
    for (int j = threadIdx.x; j < ncls; j += 64) bins[j] = expf_cr(avg[j] - m);
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    float s = 0.f;
    for (int j = 0; j < ncls; ++j) s += bins[j];
    for (int j = threadIdx.x; j < ncls; j += 64) cls_prob[(size_t)r * ncls + j] = bins[j] / s;
  }
}

}  // namespace

extern "C" int lsfa_psroi_pool_fwd(const float* data, const float* rois, int N, int C, int H, int W, int R,
                                   float spatial_scale, int output_dim, int pooled_size, int group_size,
                                   float* out, float* mapping_channel, void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(N > 0 && C > 0 && H > 0 && W > 0 && R >= 0, "lsfa_psroi_pool_fwd: bad shape");
  LSFA_REQUIRE(output_dim > 0 && pooled_size > 0 && group_size > 0, "lsfa_psroi_pool_fwd: bad parameters");
  // psroi_pooling-inl.h:167-169 (InferShape): channels must be output_dim * group^2
  LSFA_REQUIRE(C == output_dim * group_size * group_size,
               "lsfa_psroi_pool_fwd: channels %d != output_dim %d * group_size^2 %d", C, output_dim,
               group_size * group_size);
  if (R == 0) return LSFA_OK;
  LSFA_REQUIRE(data && rois && out, "lsfa_psroi_pool_fwd: NULL argument");
  hipStream_t s = (hipStream_t)stream;
  const size_t count = (size_t)R * output_dim * pooled_size * pooled_size;
  size_t nb = (count + kThreads - 1) / kThreads;
  if (nb > 262144) nb = 262144;
  ProfScope prof(LSFA_OP_PSROI, s);
  hipLaunchKernelGGL(psroi_kernel, dim3((unsigned)nb), dim3(kThreads), 0, s, data, rois, C, H, W, R,
                     spatial_scale, output_dim, pooled_size, group_size, out, mapping_channel);
  LSFA_LAUNCH_CHECK("lsfa_psroi_pool_fwd");
  return LSFA_OK;
}

extern "C" int lsfa_rfcn_head_fwd(const float* cls_map, const float* box_map, const float* rois, int N, int H,
                                  int W, int R, int ncls, int nbox, float spatial_scale, int pooled_size,
                                  int group_size, float* cls_prob, float* cls_score, float* bbox_pred,
                                  void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && R >= 0 && ncls > 0 && nbox > 0 && pooled_size > 0 && group_size > 0,
               "lsfa_rfcn_head_fwd: bad shape");
  if (R == 0) return LSFA_OK;
  LSFA_REQUIRE(cls_map && box_map && rois && cls_prob && bbox_pred, "lsfa_rfcn_head_fwd: NULL argument");
  const size_t lds = sizeof(float) * ((size_t)(ncls + nbox) * pooled_size * pooled_size + (ncls + nbox));
  if (lds > 64 * 1024) {
    set_error("lsfa_rfcn_head_fwd: (ncls+nbox)*pooled^2 = %zu floats does not fit the LDS budget", lds / 4);
    return LSFA_ENOTSUP;
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(LSFA_OP_RFCN_HEAD, s);
  hipLaunchKernelGGL(rfcn_head_kernel, dim3(R), dim3(kHeadThreads), lds, s, cls_map, box_map, rois, H, W, ncls,
                     nbox, spatial_scale, pooled_size, group_size, cls_prob, cls_score, bbox_pred);
  LSFA_LAUNCH_CHECK("lsfa_rfcn_head_fwd");
  return LSFA_OK;
}

extern "C" int lsfa_rfcn_head_ps_fwd(const float* ps_map, const float* rois, int N, int H, int W, int R, int ncls,
                                     int nbox, float spatial_scale, int pooled_size, int group_size, float* cls_prob,
                                     float* cls_score, float* bbox_pred, void* stream) {
  return lsfa_rfcn_head_ps_ld_fwd(ps_map, group_size * group_size * (ncls + nbox), rois, N, H, W, R, ncls, nbox, spatial_scale, pooled_size,
                                  group_size, cls_prob, cls_score, bbox_pred, stream);
}

extern "C" int lsfa_rfcn_head_ps_ld_fwd(const float* ps_map, int cell_ld, const float* rois, int N, int H, int W, int R, int ncls,
                                        int nbox, float spatial_scale, int pooled_size, int group_size, float* cls_prob,
                                        float* cls_score, float* bbox_pred, void* stream) {
  using namespace lsfa;
  LSFA_REQUIRE(cell_ld >= group_size * group_size * (ncls + nbox), "lsfa_rfcn_head_ps_ld_fwd: cell_ld %d < %d", cell_ld,
               group_size * group_size * (ncls + nbox));
  LSFA_REQUIRE(N > 0 && H > 0 && W > 0 && R >= 0 && ncls > 0 && nbox > 0 && pooled_size > 0 && group_size > 0,
               "lsfa_rfcn_head_ps_fwd: bad shape");
  if (R == 0) return LSFA_OK;
  LSFA_REQUIRE(ps_map && rois && cls_prob && bbox_pred, "lsfa_rfcn_head_ps_fwd: NULL argument");
  const size_t lds = sizeof(float) * ((size_t)(ncls + nbox) * pooled_size * pooled_size + (ncls + nbox));
  if (lds > 64 * 1024) {
    set_error("lsfa_rfcn_head_ps_fwd: (ncls+nbox)*pooled^2 = %zu floats does not fit the LDS budget", lds / 4);
    return LSFA_ENOTSUP;
  }
  hipStream_t s = (hipStream_t)stream;
  ProfScope prof(LSFA_OP_RFCN_HEAD, s);
  hipLaunchKernelGGL(rfcn_head_ps_kernel, dim3(R), dim3(kHeadThreads), lds, s, ps_map, rois, H, W, ncls, nbox,
                     spatial_scale, pooled_size, group_size, cell_ld, cls_prob, cls_score, bbox_pred);
  LSFA_LAUNCH_CHECK("lsfa_rfcn_head_ps_fwd");
  return LSFA_OK;
}
