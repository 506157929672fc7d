/*
 * lsfa_oracle.c — CPU restatement of LSFA's per-frame hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lsfa_amd/ may import, link or call
 * this file; it exists so that tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg can check (and time) the HIP path against an independent
 * scalar statement of the reference's algorithm.
 *
 * Provenance: the reference (hustvl/LSFA) has NO CPU implementation of its
 * native operators (psroi_pooling.cc:22-32 is an empty stub, multi_proposal.cc:27
 * is LOG(FATAL)), so each function below follows the reference's CUDA kernel
 * or Python helper line by line; the file:line it follows is cited on each
 * function.  Pinning: the anchor table, IoU/NMS and box decode are checked
 * against golden vectors captured from the reference's importable numpy
 * helpers (tests/golden/make_golden.py -> tests/golden/ npz files).  The warp
 * (GridGenerator + BilinearSampler) lives in un-vendored MXNet@75a9e187d and
 * has no golden vector in the reference: PARITY UNPINNED for orc_warp_bilinear,
 * orc_psroi_pool and the non-NMS stages of orc_proposal (they restate the
 * published kernels; nothing in the reference pins their outputs).
 *
 * Floating-point contract shared with the HIP kernels (so that results are
 * bit-identical and index decisions cannot flip):
 *   - compiled with -ffp-contract=off; fused multiply-adds appear ONLY where
 *     written as fmaf(), at the places where nvcc's default -fmad=true
 *     contracts the reference's CUDA expression `a*b + c` inside one statement;
 *   - exp() of a float is the correctly rounded float exponential, obtained
 *     as (float)exp((double)x)  (CUDA's expf is within 2 ulp of this; no
 *     bit-level statement of it exists);
 *   - sums run in the reference's loop order.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

static inline float expf_cr(float x) { return (float)exp((double)x); }
static inline float fminf_(float a, float b) { return a < b ? a : b; }
static inline float fmaxf_(float a, float b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* ------------------------------------------------------------------------ *
 * Base anchors.  Follows utils::GenerateAnchors / _Transform / _MakeAnchor,
 * dff_rfcn/operator_cxx/multi_proposal-inl.h:256-295 (host code: no fma).
 * anchors: (nr*ns, 4), ratios outer, scales inner.
 * ------------------------------------------------------------------------ */
void orc_generate_anchors(int feature_stride, const float* ratios, int nr,
                          const float* scales, int ns, float* anchors) {
  float base[4] = {0.0f, 0.0f, (float)(feature_stride - 1.0), (float)(feature_stride - 1.0)};
  int n = 0;
  for (int j = 0; j < nr; ++j) {
    for (int k = 0; k < ns; ++k) {
      float scale = scales[k], ratio = ratios[j];
      /* :272-273 — the reference subtracts base_anchor[1] for the width too */
      float w = base[2] - base[1] + 1.0f;
      float h = base[3] - base[1] + 1.0f;
      float x_ctr = (float)(base[0] + 0.5 * (w - 1.0f));
      float y_ctr = (float)(base[1] + 0.5 * (h - 1.0f));
      float size = w * h;
      float size_ratios = floorf(size / ratio);
      float new_w = floorf(sqrtf(size_ratios) + 0.5f) * scale;
      float new_h = floorf((new_w / scale * ratio) + 0.5f) * scale;
      anchors[n * 4 + 0] = x_ctr - 0.5f * (new_w - 1.0f);
      anchors[n * 4 + 1] = y_ctr - 0.5f * (new_h - 1.0f);
      anchors[n * 4 + 2] = x_ctr + 0.5f * (new_w - 1.0f);
      anchors[n * 4 + 3] = y_ctr + 0.5f * (new_h - 1.0f);
      ++n;
    }
  }
}

/* devIoU, lib/nms/nms_kernel.cu:30-38 == multi_proposal.cu:252-260 */
static inline float dev_iou(const float* a, const float* b) {
  float left = fmaxf_(a[0], b[0]), right = fminf_(a[2], b[2]);
  float top = fmaxf_(a[1], b[1]), bottom = fminf_(a[3], b[3]);
  float width = fmaxf_(right - left + 1, 0.f), height = fmaxf_(bottom - top + 1, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

float orc_dev_iou(const float* a, const float* b) { return dev_iou(a, b); }

/* ------------------------------------------------------------------------ *
 * Bitmask NMS on score-sorted boxes.  The mask kernel (nms_kernel.cu:40-84:
 * bit j of row i set iff IoU(i,j) > thresh, j > i) followed by the host sweep
 * (:133-146) is equivalent to this greedy loop: box i survives iff no earlier
 * survivor k has IoU(k,i) > thresh.
 * boxes (n, box_dim) sorted by score desc; keep (n) -> indices; returns count.
 * ------------------------------------------------------------------------ */
int orc_nms_sorted(const float* boxes, int n, int box_dim, float thresh, int* keep) {
  unsigned char* removed = (unsigned char*)calloc((size_t)(n > 0 ? n : 1), 1);
  int num = 0;
  for (int i = 0; i < n; ++i) {
    if (removed[i]) continue;
    keep[num++] = i;
    const float* bi = boxes + (size_t)i * box_dim;
    for (int j = i + 1; j < n; ++j) {
      if (!removed[j] && dev_iou(bi, boxes + (size_t)j * box_dim) > thresh) removed[j] = 1;
    }
  }
  free(removed);
  return num;
}

/* Same, but materialising the uint64 mask exactly as the kernel writes it
 * (used to check the HIP mask kernel word for word).  mask: (n, ceil(n/64)). */
void orc_nms_mask(const float* boxes, int n, int box_dim, float thresh, uint64_t* mask) {
  int col_blocks = (n + 63) / 64;
  for (int i = 0; i < n; ++i) {
    for (int cb = 0; cb < col_blocks; ++cb) {
      uint64_t t = 0;
      int col_size = imin(n - cb * 64, 64);
      int start = (i / 64 == cb) ? (i % 64) + 1 : 0;
      for (int k = start; k < col_size; ++k) {
        if (dev_iou(boxes + (size_t)i * box_dim, boxes + (size_t)(cb * 64 + k) * box_dim) > thresh)
          t |= 1ULL << k;
      }
      mask[(size_t)i * col_blocks + cb] = t;
    }
  }
}

/* ------------------------------------------------------------------------ *
 * Proposal.  Follows MultiProposalGPUOp::Forward, multi_proposal.cu:403-558.
 * ------------------------------------------------------------------------ */
typedef struct { float score; int idx; } orc_sk_t;
static int orc_sk_cmp(const void* pa, const void* pb) {
  const orc_sk_t* a = (const orc_sk_t*)pa; const orc_sk_t* b = (const orc_sk_t*)pb;
  /* thrust::stable_sort_by_key(..., greater) :517-521: score descending, equal
   * scores keep ascending index */
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx);
}

/* Stage 1 only: ProposalGridKernel :47-70 + BBoxPredKernel :78-135 +
 * FilterBoxKernel :196-216.  proposals (B, H*W*A, 5) rows [x1,y1,x2,y2,score]. */
void orc_proposal_decode(const float* cls_prob, const float* bbox_pred, const float* im_info,
                         int B, int A, int H, int W, int feature_stride,
                         const float* scales, int n_scales, const float* ratios, int n_ratios,
                         int rpn_min_size, float* proposals) {
  float* anchors = (float*)malloc(sizeof(float) * 4 * (size_t)A);
  orc_generate_anchors(feature_stride, ratios, n_ratios, scales, n_scales, anchors);
  const int count_anchors = A * H * W;
  for (int b = 0; b < B; ++b) {
    float im_height = im_info[b * 3];
    float im_width = im_info[b * 3 + 1];
    int real_height = (int)(im_height / feature_stride);
    int real_width = (int)(im_width / feature_stride);
    float min_size = (float)rpn_min_size * im_info[b * 3 + 2];
    for (int h = 0; h < H; ++h) for (int w = 0; w < W; ++w) for (int a = 0; a < A; ++a) {
      size_t index = (size_t)b * count_anchors + ((size_t)h * W + w) * A + a;
      float* p = proposals + index * 5;
      /* ProposalGridKernel :62-67 */
      float x1 = anchors[a * 4 + 0] + (float)(w * feature_stride);
      float y1 = anchors[a * 4 + 1] + (float)(h * feature_stride);
      float x2 = anchors[a * 4 + 2] + (float)(w * feature_stride);
      float y2 = anchors[a * 4 + 3] + (float)(h * feature_stride);
      float score = cls_prob[(((size_t)b * (2 * A) + a + A) * H + h) * W + w];
      /* BBoxPredKernel :100-124 (fmaf where nvcc contracts a*b+c) */
      float width = x2 - x1 + 1.0f;
      float height = y2 - y1 + 1.0f;
      float ctr_x = fmaf(0.5f, width - 1.0f, x1);
      float ctr_y = fmaf(0.5f, height - 1.0f, y1);
      size_t ba = (size_t)b * A + a;
      float dx = bbox_pred[((ba * 4 + 0) * H + h) * W + w];
      float dy = bbox_pred[((ba * 4 + 1) * H + h) * W + w];
      float dw = bbox_pred[((ba * 4 + 2) * H + h) * W + w];
      float dh = bbox_pred[((ba * 4 + 3) * H + h) * W + w];
      float pred_ctr_x = fmaf(dx, width, ctr_x);
      float pred_ctr_y = fmaf(dy, height, ctr_y);
      float pred_w = expf_cr(dw) * width;
      float pred_h = expf_cr(dh) * height;
      float pred_x1 = fmaf(-0.5f, pred_w - 1.0f, pred_ctr_x);
      float pred_y1 = fmaf(-0.5f, pred_h - 1.0f, pred_ctr_y);
      float pred_x2 = fmaf(0.5f, pred_w - 1.0f, pred_ctr_x);
      float pred_y2 = fmaf(0.5f, pred_h - 1.0f, pred_ctr_y);
      pred_x1 = fmaxf_(fminf_(pred_x1, im_width - 1.0f), 0.0f);
      pred_y1 = fmaxf_(fminf_(pred_y1, im_height - 1.0f), 0.0f);
      pred_x2 = fmaxf_(fminf_(pred_x2, im_width - 1.0f), 0.0f);
      pred_y2 = fmaxf_(fminf_(pred_y2, im_height - 1.0f), 0.0f);
      if (h >= real_height || w >= real_width) score = -1.0f;   /* :131-133 */
      /* FilterBoxKernel :204-214 */
      float iw = pred_x2 - pred_x1 + 1.0f;
      float ih = pred_y2 - pred_y1 + 1.0f;
      if (iw < min_size || ih < min_size) {
        pred_x1 -= min_size / 2; pred_y1 -= min_size / 2;
        pred_x2 += min_size / 2; pred_y2 += min_size / 2;
        score = -1.0f;
      }
      p[0] = pred_x1; p[1] = pred_y1; p[2] = pred_x2; p[3] = pred_y2; p[4] = score;
    }
  }
  free(anchors);
}

/* Full op.  rois (B*post_n, 5), scores (B*post_n) [may be NULL].
 * Optional debug outputs (may be NULL): order_out (B, pre_n) = anchor index of
 * each sorted candidate; keep_out (B, pre_n) = NMS survivors (indices into the
 * sorted list), num_keep_out (B). */
void orc_proposal(const float* cls_prob, const float* bbox_pred, const float* im_info,
                  int B, int A, int H, int W, int feature_stride,
                  const float* scales, int n_scales, const float* ratios, int n_ratios,
                  int rpn_pre_nms_top_n, int rpn_post_nms_top_n, float threshold, int rpn_min_size,
                  float* rois, float* scores, int* order_out, int* keep_out, int* num_keep_out) {
  const int count_anchors = A * H * W;
  /* :435-437 */
  int pre_n = rpn_pre_nms_top_n > 0 ? rpn_pre_nms_top_n : count_anchors;
  pre_n = imin(pre_n, count_anchors);
  int post_n = imin(rpn_post_nms_top_n, pre_n);
  float* proposals = (float*)malloc(sizeof(float) * 5 * (size_t)B * count_anchors);
  orc_proposal_decode(cls_prob, bbox_pred, im_info, B, A, H, W, feature_stride, scales, n_scales,
                      ratios, n_ratios, rpn_min_size, proposals);
  orc_sk_t* sk = (orc_sk_t*)malloc(sizeof(orc_sk_t) * (size_t)count_anchors);
  float* ordered = (float*)malloc(sizeof(float) * 5 * (size_t)pre_n);
  int* keep = (int*)malloc(sizeof(int) * (size_t)pre_n);
  for (int b = 0; b < B; ++b) {
    const float* pb = proposals + (size_t)b * count_anchors * 5;
    for (int i = 0; i < count_anchors; ++i) { sk[i].score = pb[i * 5 + 4]; sk[i].idx = i; }  /* CopyScore :222-232 */
    qsort(sk, (size_t)count_anchors, sizeof(orc_sk_t), orc_sk_cmp);                          /* :517-521 */
    for (int i = 0; i < pre_n; ++i) {                                                         /* Reorder :238-250 */
      memcpy(ordered + (size_t)i * 5, pb + (size_t)sk[i].idx * 5, sizeof(float) * 5);
      if (order_out) order_out[(size_t)b * pre_n + i] = sk[i].idx;
    }
    int out_size = orc_nms_sorted(ordered, pre_n, 5, threshold, keep);                        /* _nms :309-357 */
    if (keep_out) memcpy(keep_out + (size_t)b * pre_n, keep, sizeof(int) * (size_t)out_size);
    if (num_keep_out) num_keep_out[b] = out_size;
    for (int index = 0; index < post_n; ++index) {                                            /* PrepareOutput :363-388 */
      int keep_i = index < out_size ? keep[index] : keep[index % out_size];
      float* o = rois + ((size_t)b * post_n + index) * 5;
      o[0] = (float)b;
      for (int j = 0; j < 4; ++j) o[j + 1] = ordered[(size_t)keep_i * 5 + j];
      if (scores) scores[(size_t)b * post_n + index] = ordered[(size_t)keep_i * 5 + 4];
    }
  }
  free(keep); free(ordered); free(sk); free(proposals);
}

/* ------------------------------------------------------------------------ *
 * PSROI pooling forward.  Follows PSROIPoolForwardKernel,
 * dff_rfcn/operator_cxx/psroi_pooling.cu:44-100, one output element at a time.
 * out / mapping_channel: (R, output_dim, P, P); mapping_channel may be NULL.
 * The op fills out with -FLT_MAX first (psroi_pooling-inl.h:77); every element
 * is then overwritten, so the fill is not observable.
 * ------------------------------------------------------------------------ */
void orc_psroi_pool(const float* data, const float* rois, int N, int C, int H, int W, int R,
                    float spatial_scale, int output_dim, int pooled_size, int group_size,
                    float* out, float* mapping_channel) {
  (void)N;
  const int P = pooled_size;
  const size_t count = (size_t)R * output_dim * P * P;
  for (size_t index = 0; index < count; ++index) {
    int pw = (int)(index % P);
    int ph = (int)((index / P) % P);
    int ctop = (int)((index / P / P) % output_dim);
    int n = (int)(index / P / P / output_dim);
    const float* roi = rois + (size_t)n * 5;
    int roi_batch_ind = (int)roi[0];
    float roi_start_w = roundf(roi[1]) * spatial_scale;                     /* :53 */
    float roi_start_h = roundf(roi[2]) * spatial_scale;
    float roi_end_w = (float)((double)roundf(roi[3]) + 1.) * spatial_scale; /* :55 */
    float roi_end_h = (float)((double)roundf(roi[4]) + 1.) * spatial_scale;
    float roi_width = (float)fmax((double)(roi_end_w - roi_start_w), 0.1);  /* :59-60 */
    float roi_height = (float)fmax((double)(roi_end_h - roi_start_h), 0.1);
    float bin_size_h = roi_height / (float)P;                                /* :63-64 */
    float bin_size_w = roi_width / (float)P;
    int hstart = (int)floorf(fmaf((float)ph, bin_size_h, roi_start_h));      /* :66-73 */
    int wstart = (int)floorf(fmaf((float)pw, bin_size_w, roi_start_w));
    int hend = (int)ceilf(fmaf((float)(ph + 1), bin_size_h, roi_start_h));
    int wend = (int)ceilf(fmaf((float)(pw + 1), bin_size_w, roi_start_w));
    hstart = imin(imax(hstart, 0), H); hend = imin(imax(hend, 0), H);       /* :75-78 */
    wstart = imin(imax(wstart, 0), W); wend = imin(imax(wend, 0), W);
    int is_empty = (hend <= hstart) || (wend <= wstart);
    int gw = (int)floorf((float)pw * (float)group_size / (float)P);         /* :81-84 */
    int gh = (int)floorf((float)ph * (float)group_size / (float)P);
    gw = imin(imax(gw, 0), group_size - 1);
    gh = imin(imax(gh, 0), group_size - 1);
    int c = (ctop * group_size + gh) * group_size + gw;
    const float* plane = data + ((size_t)roi_batch_ind * C + c) * H * W;
    float out_sum = 0;
    for (int h = hstart; h < hend; ++h)
      for (int w = wstart; w < wend; ++w) out_sum += plane[h * W + w];
    float bin_area = (float)((hend - hstart) * (wend - wstart));
    out[index] = is_empty ? 0.f : out_sum / bin_area;
    if (mapping_channel) mapping_channel[index] = (float)c;
  }
}

/* Global average over the PxP bins (Pooling global_pool avg,
 * resnet_v1_101_flownet_rfcn.py:535-536): sequential sum in (ph,pw) order, / (P*P). */
void orc_global_avg(const float* pooled, int R, int D, int P, float* avg) {
  for (int i = 0; i < R * D; ++i) {
    float s = 0.f;
    for (int k = 0; k < P * P; ++k) s += pooled[(size_t)i * P * P + k];
    avg[i] = s / (float)(P * P);
  }
}

/* SoftmaxActivation over classes (:540): max, exp(x-max), sum in class order, divide. */
void orc_softmax_rows(const float* x, int R, int D, float* y) {
  for (int r = 0; r < R; ++r) {
    const float* xr = x + (size_t)r * D; float* yr = y + (size_t)r * D;
    float m = xr[0];
    for (int j = 1; j < D; ++j) m = fmaxf_(m, xr[j]);
    float s = 0.f;
    for (int j = 0; j < D; ++j) { yr[j] = expf_cr(xr[j] - m); s += yr[j]; }
    for (int j = 0; j < D; ++j) yr[j] = yr[j] / s;
  }
}

/* ------------------------------------------------------------------------ *
 * Warp.  GridGenerator(transform_type='warp') + BilinearSampler as called at
 * resnet_v1_101_flownet_rfcn.py:468-470, :571-576 and :230-236; arithmetic per
 * MXNet@75a9e187d src/operator/grid_generator-inl.h (warp: (flow + grid_dst) /
 * ((size-1)/2) - 1) and bilinear_sampler.cc (x_real = (gx+1)*(W-1)/2, four taps,
 * each tap 0 when outside the map) [MXNet, un-vendored — PARITY UNPINNED].
 * Epilogue order: ((bilerp * mul) + ((w0*r0 + w1*r1 + w2*r2 ...) + bias)) + add.
 * ------------------------------------------------------------------------ */
void orc_warp_bilinear(const float* feat, int feat_n, const float* flow, int N, int C, int H, int W,
                       const float* mul, const float* add,
                       const float* res, int res_c, const float* res_w, const float* res_b,
                       float* out) {
  const size_t HW = (size_t)H * W;
  const float half_w = (float)((W - 1) / 2.0), half_h = (float)((H - 1) / 2.0);
  for (int n = 0; n < N; ++n) {
    const float* fbase = feat + (feat_n == 1 ? 0 : (size_t)n * C * HW);
    for (int y = 0; y < H; ++y) for (int x = 0; x < W; ++x) {
      size_t p = (size_t)y * W + x;
      float fx = flow[((size_t)n * 2 + 0) * HW + p], fy = flow[((size_t)n * 2 + 1) * HW + p];
      float gx = ((float)x + fx) / half_w - 1.0f;
      float gy = ((float)y + fy) / half_h - 1.0f;
      float x_real = (gx + 1.0f) * (float)(W - 1) / 2.0f;
      float y_real = (gy + 1.0f) * (float)(H - 1) / 2.0f;
      float fx0 = floorf(x_real), fy0 = floorf(y_real);
      int x0 = (int)fx0, y0 = (int)fy0;
      float wx0 = 1.0f - (x_real - fx0), wy0 = 1.0f - (y_real - fy0);
      float wx1 = 1.0f - wx0, wy1 = 1.0f - wy0;
      int vx0 = (x0 >= 0 && x0 <= W - 1), vx1 = (x0 + 1 >= 0 && x0 + 1 <= W - 1);
      int vy0 = (y0 >= 0 && y0 <= H - 1), vy1 = (y0 + 1 >= 0 && y0 + 1 <= H - 1);
      for (int c = 0; c < C; ++c) {
        const float* plane = fbase + (size_t)c * HW;
        float tl = (vx0 && vy0) ? plane[(size_t)y0 * W + x0] : 0.f;
        float tr = (vx1 && vy0) ? plane[(size_t)y0 * W + x0 + 1] : 0.f;
        float bl = (vx0 && vy1) ? plane[(size_t)(y0 + 1) * W + x0] : 0.f;
        float br = (vx1 && vy1) ? plane[(size_t)(y0 + 1) * W + x0 + 1] : 0.f;
        float v = tl * wy0 * wx0 + tr * wy0 * wx1 + bl * wy1 * wx0 + br * wy1 * wx1;
        size_t o = ((size_t)n * C + c) * HW + p;
        if (mul) v = v * mul[o];
        if (res) {
          float r = res_w[(size_t)c * res_c] * res[((size_t)n * res_c) * HW + p];
          for (int k = 1; k < res_c; ++k) r = r + res_w[(size_t)c * res_c + k] * res[((size_t)n * res_c + k) * HW + p];
          r = r + res_b[c];
          v = v + r;
        }
        if (add) v = v + add[o];
        out[o] = v;
      }
    }
  }
}

/* ------------------------------------------------------------------------ *
 * Nq aggregation combine, resnet_v1_101_flownet_rfcn.py:104-108:
 * softmax over axis 0 of the (2,1,H,W) logits, out = w0*a + w1*b.
 * ------------------------------------------------------------------------ */
void orc_aggregate_softmax2(const float* a, const float* b, const float* logits, int C, int H, int W, float* out) {
  const size_t HW = (size_t)H * W;
  for (size_t p = 0; p < HW; ++p) {
    float l0 = logits[p], l1 = logits[HW + p];
    float m = fmaxf_(l0, l1);
    float e0 = expf_cr(l0 - m), e1 = expf_cr(l1 - m);
    float s = e0 + e1;
    float w0 = e0 / s, w1 = e1 / s;
    for (int c = 0; c < C; ++c) {
      size_t o = (size_t)c * HW + p;
      out[o] = w0 * a[o] + w1 * b[o];
    }
  }
}

/* Fgfa combine, resnet_v1_101_flownet_rfcn.py:111-116, :136-147.
 * L2Normalization(mode='channel'): x / sqrt(sum_c x^2 + 1e-10). */
/* Sum over the E embedding channels of one pixel in the FIXED order the HIP kernel uses (the reference's
 * MXNet `sum(axis=1)` has no specified order [MXNet, un-vendored]; any order is a valid reading of
 * resnet_v1_101_flownet_rfcn.py:111-116, so the order is defined here once for both sides): 64 partial
 * sums, partial j taking channels e = j, j+64, j+128, ... in increasing order, then a butterfly
 * partial[j] += partial[j ^ d] for d = 32, 16, 8, 4, 2, 1 (float addition is commutative, so every j
 * ends with the same total). */
#define ORC_TREE_LANES 64
static float orc_tree_total(float* partial) {
  float next[ORC_TREE_LANES];
  for (int d = ORC_TREE_LANES / 2; d >= 1; d >>= 1) {
    for (int j = 0; j < ORC_TREE_LANES; ++j) next[j] = partial[j] + partial[j ^ d];
    for (int j = 0; j < ORC_TREE_LANES; ++j) partial[j] = next[j];
  }
  return partial[0];
}

void orc_aggregate_cosine(const float* a, const float* b, const float* emb_warp, const float* emb_cur,
                          int C, int E, int H, int W, float* out) {
  const size_t HW = (size_t)H * W;
  for (size_t p = 0; p < HW; ++p) {
    float pw[ORC_TREE_LANES], pc[ORC_TREE_LANES];
    for (int j = 0; j < ORC_TREE_LANES; ++j) { pw[j] = 0.f; pc[j] = 0.f; }
    for (int e = 0; e < E; ++e) {
      float vw = emb_warp[(size_t)e * HW + p], vc = emb_cur[(size_t)e * HW + p];
      pw[e % ORC_TREE_LANES] += vw * vw;
      pc[e % ORC_TREE_LANES] += vc * vc;
    }
    float sw = orc_tree_total(pw), sc = orc_tree_total(pc);
    float nw = sqrtf(sw + 1e-10f), nc = sqrtf(sc + 1e-10f);
    for (int j = 0; j < ORC_TREE_LANES; ++j) { pw[j] = 0.f; pc[j] = 0.f; }
    for (int e = 0; e < E; ++e) {
      float vw = emb_warp[(size_t)e * HW + p] / nw, vc = emb_cur[(size_t)e * HW + p] / nc;
      pw[e % ORC_TREE_LANES] += vw * vc;
      pc[e % ORC_TREE_LANES] += vc * vc;
    }
    float l0 = orc_tree_total(pw), l1 = orc_tree_total(pc);
    float m = fmaxf_(l0, l1);
    float e0 = expf_cr(l0 - m), e1 = expf_cr(l1 - m);
    float s = e0 + e1;
    float w0 = e0 / s, w1 = e1 / s;
    for (int c = 0; c < C; ++c) {
      size_t o = (size_t)c * HW + p;
      out[o] = w0 * a[o] + w1 * b[o];
    }
  }
}

/* ------------------------------------------------------------------------ *
 * Box decode + clip + rescale in float64.  Follows nonlinear_pred
 * (lib/bbox/bbox_transform.py:103-140), clip_boxes (:45-60) and `/ scale`
 * (dff_rfcn/core/tester.py:148-152).  pred (R, 4*nreg).
 * ------------------------------------------------------------------------ */
void orc_bbox_pred_clip(const float* rois, const float* deltas, int R, int nreg,
                        double im_h, double im_w, double scale, double* pred) {
  for (int r = 0; r < R; ++r) {
    double x1 = rois[r * 5 + 1], y1 = rois[r * 5 + 2], x2 = rois[r * 5 + 3], y2 = rois[r * 5 + 4];
    double widths = x2 - x1 + 1.0, heights = y2 - y1 + 1.0;
    double ctr_x = x1 + 0.5 * (widths - 1.0), ctr_y = y1 + 0.5 * (heights - 1.0);
    for (int k = 0; k < nreg; ++k) {
      const float* d = deltas + ((size_t)r * nreg + k) * 4;
      double pcx = (double)d[0] * widths + ctr_x;
      double pcy = (double)d[1] * heights + ctr_y;
      /* np.exp on a float32 array yields float32, then the product promotes to float64 */
      double pw = (double)expf_cr(d[2]) * widths;
      double ph = (double)expf_cr(d[3]) * heights;
      double bx1 = pcx - 0.5 * (pw - 1.0), by1 = pcy - 0.5 * (ph - 1.0);
      double bx2 = pcx + 0.5 * (pw - 1.0), by2 = pcy + 0.5 * (ph - 1.0);
      bx1 = fmax(fmin(bx1, im_w - 1), 0); by1 = fmax(fmin(by1, im_h - 1), 0);
      bx2 = fmax(fmin(bx2, im_w - 1), 0); by2 = fmax(fmin(by2, im_h - 1), 0);
      double* o = pred + ((size_t)r * nreg + k) * 4;
      o[0] = bx1 / scale; o[1] = by1 / scale; o[2] = bx2 / scale; o[3] = by2 / scale;
    }
  }
}

/* numpy nms (lib/nms/nms.py:37-74) on float64 dets (n,5) [x1,y1,x2,y2,score].
 * Order = score descending; ties broken by ascending index (the reference's
 * argsort()[::-1] leaves tie order to numpy's unstable sort).  Keeps when
 * ovr <= thresh.  keep -> indices into dets; returns count. */
typedef struct { double score; int idx; } orc_dk_t;
static int orc_dk_cmp(const void* pa, const void* pb) {
  const orc_dk_t* a = (const orc_dk_t*)pa; const orc_dk_t* b = (const orc_dk_t*)pb;
  if (a->score > b->score) return -1;
  if (a->score < b->score) return 1;
  return (a->idx > b->idx) - (a->idx < b->idx);
}
int orc_nms_f64(const double* dets, int n, double thresh, int* keep) {
  if (n == 0) return 0;
  orc_dk_t* order = (orc_dk_t*)malloc(sizeof(orc_dk_t) * (size_t)n);
  unsigned char* removed = (unsigned char*)calloc((size_t)n, 1);
  for (int i = 0; i < n; ++i) { order[i].score = dets[i * 5 + 4]; order[i].idx = i; }
  qsort(order, (size_t)n, sizeof(orc_dk_t), orc_dk_cmp);
  int num = 0;
  for (int oi = 0; oi < n; ++oi) {
    if (removed[oi]) continue;
    int i = order[oi].idx;
    keep[num++] = i;
    double area_i = (dets[i * 5 + 2] - dets[i * 5 + 0] + 1) * (dets[i * 5 + 3] - dets[i * 5 + 1] + 1);
    for (int oj = oi + 1; oj < n; ++oj) {
      if (removed[oj]) continue;
      int j = order[oj].idx;
      double xx1 = fmax(dets[i * 5 + 0], dets[j * 5 + 0]), yy1 = fmax(dets[i * 5 + 1], dets[j * 5 + 1]);
      double xx2 = fmin(dets[i * 5 + 2], dets[j * 5 + 2]), yy2 = fmin(dets[i * 5 + 3], dets[j * 5 + 3]);
      double w = fmax(0.0, xx2 - xx1 + 1), h = fmax(0.0, yy2 - yy1 + 1);
      double inter = w * h;
      double area_j = (dets[j * 5 + 2] - dets[j * 5 + 0] + 1) * (dets[j * 5 + 3] - dets[j * 5 + 1] + 1);
      double ovr = inter / (area_i + area_j - inter);
      if (!(ovr <= thresh)) removed[oj] = 1;
    }
  }
  free(removed); free(order);
  return num;
}

/* Per-frame post-processing, dff_rfcn/core/tester.py:265-281: for each class
 * j>=1 keep rois with prob > score_thresh, NMS, then cap the image at
 * max_per_image detections by the global score threshold.
 * dets (ncls, R, 5) float64, counts (ncls), keep_idx (ncls, R) roi index. */
static int orc_dbl_cmp(const void* a, const void* b) {
  double x = *(const double*)a, y = *(const double*)b; return (x > y) - (x < y);
}
void orc_det_postprocess(const float* rois, const float* deltas, const float* probs,
                         int R, int ncls, int nreg, int class_agnostic,
                         double im_h, double im_w, double scale,
                         double score_thresh, double nms_thresh, int max_per_image,
                         double* dets, int* counts, int* keep_idx) {
  double* pred = (double*)malloc(sizeof(double) * 4 * (size_t)nreg * R);
  double* cls_dets = (double*)malloc(sizeof(double) * 5 * (size_t)R);
  int* src = (int*)malloc(sizeof(int) * (size_t)R);
  int* keep = (int*)malloc(sizeof(int) * (size_t)R);
  orc_bbox_pred_clip(rois, deltas, R, nreg, im_h, im_w, scale, pred);
  counts[0] = 0;
  int total = 0;
  for (int j = 1; j < ncls; ++j) {
    int m = 0;
    for (int r = 0; r < R; ++r) {
      double s = (double)probs[(size_t)r * ncls + j];
      if (s > score_thresh) {
        int col = class_agnostic ? 1 : j;
        memcpy(cls_dets + (size_t)m * 5, pred + ((size_t)r * nreg + col) * 4, sizeof(double) * 4);
        cls_dets[(size_t)m * 5 + 4] = s;
        src[m++] = r;
      }
    }
    int nk = orc_nms_f64(cls_dets, m, nms_thresh, keep);
    for (int k = 0; k < nk; ++k) {
      memcpy(dets + ((size_t)j * R + k) * 5, cls_dets + (size_t)keep[k] * 5, sizeof(double) * 5);
      if (keep_idx) keep_idx[(size_t)j * R + k] = src[keep[k]];
    }
    counts[j] = nk; total += nk;
  }
  if (max_per_image > 0 && total > max_per_image) {     /* tester.py:274-281 */
    double* all = (double*)malloc(sizeof(double) * (size_t)total);
    int t = 0;
    for (int j = 1; j < ncls; ++j) for (int k = 0; k < counts[j]; ++k) all[t++] = dets[((size_t)j * R + k) * 5 + 4];
    qsort(all, (size_t)total, sizeof(double), orc_dbl_cmp);
    double image_thresh = all[total - max_per_image];
    free(all);
    for (int j = 1; j < ncls; ++j) {
      int m = 0;
      for (int k = 0; k < counts[j]; ++k) {
        if (dets[((size_t)j * R + k) * 5 + 4] >= image_thresh) {
          if (m != k) {
            memcpy(dets + ((size_t)j * R + m) * 5, dets + ((size_t)j * R + k) * 5, sizeof(double) * 5);
            if (keep_idx) keep_idx[(size_t)j * R + m] = keep_idx[(size_t)j * R + k];
          }
          ++m;
        }
      }
      counts[j] = m;
    }
  }
  free(keep); free(src); free(cls_dets); free(pred);
}

/* ------------------------------------------------------------------------ *
 * Deformable im2col, MXNet@75a9e187d src/operator/contrib/nn/deformable_im2col.cuh
 * (deformable_im2col_gpu_kernel + deformable_im2col_bilinear) as called through
 * sym_common.py:138-157 [MXNet, un-vendored — PARITY UNPINNED].
 * col (N, C*kh*kw, Ho*Wo).
 * ------------------------------------------------------------------------ */
static float dcn_bilinear(const float* plane, int data_width, int height, int width, float h, float w) {
  int h_low = (int)floorf(h), w_low = (int)floorf(w);
  int h_high, w_high;
  if (h_low >= height - 1) { h_high = h_low = height - 1; h = (float)h_low; } else { h_high = h_low + 1; }
  if (w_low >= width - 1) { w_high = w_low = width - 1; w = (float)w_low; } else { w_high = w_low + 1; }
  float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  float v1 = plane[h_low * data_width + w_low], v2 = plane[h_low * data_width + w_high];
  float v3 = plane[h_high * data_width + w_low], v4 = plane[h_high * data_width + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  return w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
}
void orc_deform_im2col(const float* data, const float* offset, int N, int C, int H, int W,
                       int kh, int kw, int pad, int stride, int dilate, int dg, int Ho, int Wo, float* col) {
  const int cpg = C / dg;
  for (int n = 0; n < N; ++n) for (int c = 0; c < C; ++c) {
    int g = c / cpg;
    const float* plane = data + ((size_t)n * C + c) * H * W;
    const float* off = offset + ((size_t)n * dg + g) * 2 * kh * kw * Ho * Wo;
    for (int i = 0; i < kh; ++i) for (int j = 0; j < kw; ++j)
      for (int ho = 0; ho < Ho; ++ho) for (int wo = 0; wo < Wo; ++wo) {
        int h_in = ho * stride - pad, w_in = wo * stride - pad;
        float oh = off[((size_t)(2 * (i * kw + j)) * Ho + ho) * Wo + wo];
        float ow = off[((size_t)(2 * (i * kw + j) + 1) * Ho + ho) * Wo + wo];
        float h_im = (float)(h_in + i * dilate) + oh;
        float w_im = (float)(w_in + j * dilate) + ow;
        float val = 0.f;
        if (h_im >= 0 && w_im >= 0 && h_im < H && w_im < W) val = dcn_bilinear(plane, W, H, W, h_im, w_im);
        col[(((size_t)n * C + c) * kh * kw + (i * kw + j)) * Ho * Wo + (size_t)ho * Wo + wo] = val;
      }
  }
}

/* y = x*scale[c] + shift[c], optional ReLU (BatchNorm use_global_stats + relu,
 * sym_common.py:92-102; scale/shift precomputed by the caller). */
void orc_scale_shift_leaky(const float* x, const float* scale, const float* shift, int N, int C, int HW, float slope, float* y) {
  for (int n = 0; n < N; ++n) for (int c = 0; c < C; ++c) for (int p = 0; p < HW; ++p) {
    size_t o = ((size_t)n * C + c) * HW + p;
    float v = x[o] * scale[c] + shift[c];
    y[o] = v > 0.f ? v : v * slope;        /* mx LeakyReLU 'leaky': x > 0 ? x : slope * x */
  }
}

void orc_scale_shift_relu(const float* x, const float* scale, const float* shift, int N, int C, int HW, int relu, float* y) {
  for (int n = 0; n < N; ++n) for (int c = 0; c < C; ++c) for (int p = 0; p < HW; ++p) {
    size_t o = ((size_t)n * C + c) * HW + p;
    float v = x[o] * scale[c] + shift[c];
    y[o] = (relu && v < 0.f) ? 0.f : v;
  }
}


/* ------------------------------------------------------------------------ *
 * Compressed-domain motion vectors: accumulation back to the key frame and the residual
 * (external/data_loader_py2/coviar_data_loader.c:71-177, create_and_load_mv_residual with
 * accumulate = 1).  mvs (n, 7) int32 rows = AVMotionVector's {source, w, h, src_x, src_y, dst_x, dst_y}
 * in the order the decoder lists them.  accu_* hold, per pixel, the (x, y) of the pixel of the GOP's
 * first frame it came from; HERE they are (H, W, 2) row-major (the reference keeps them x-major,
 * accu[x*height*2 + y*2 + c], :111-113 — only the index arithmetic differs).  Start from identity
 * (:316-323).  Integer work: bit-exact by definition.
 * ------------------------------------------------------------------------ */
void orc_coviar_identity(int* accu, int width, int height) {
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) { accu[((size_t)y * width + x) * 2] = x; accu[((size_t)y * width + x) * 2 + 1] = y; }
}

/* one P-frame: :89-124.  accu_new must equal accu_old on entry (the reference memcpy's new -> old after every
 * frame, :123-125, so pixels no block writes keep their value); later blocks overwrite earlier ones. */
void orc_coviar_accumulate(const int* mvs, int n, const int* accu_old, int* accu_new, int width, int height) {
  for (int i = 0; i < n; ++i) {
    const int* mv = mvs + (size_t)i * 7;
    const int w = mv[1], h = mv[2], src_x = mv[3], src_y = mv[4], dst_x = mv[5], dst_y = mv[6];
    if (dst_x - src_x != 0 || dst_y - src_y != 0) {
      for (int x_start = (-1 * w / 2); x_start < w / 2; ++x_start) {
        for (int y_start = (-1 * h / 2); y_start < h / 2; ++y_start) {
          const int p_dst_x = dst_x + x_start, p_dst_y = dst_y + y_start;
          const int p_src_x = src_x + x_start, p_src_y = src_y + y_start;
          if (p_dst_y >= 0 && p_dst_y < height && p_dst_x >= 0 && p_dst_x < width &&
              p_src_y >= 0 && p_src_y < height && p_src_x >= 0 && p_src_x < width) {
            for (int c = 0; c < 2; ++c)
              accu_new[((size_t)p_dst_y * width + p_dst_x) * 2 + c] = accu_old[((size_t)p_src_y * width + p_src_x) * 2 + c];
          }
        }
      }
    }
  }
}

/* :131-141: mv[y, x] = (x, y) - accu[y, x]   -> (H, W, 2) int32 */
void orc_coviar_mv(const int* accu, int width, int height, int* mv) {
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const size_t o = ((size_t)y * width + x) * 2;
      mv[o] = x - accu[o];
      mv[o + 1] = y - accu[o + 1];
    }
}

/* :144-171: res[y, x, c] = cur[y, x, c] - ref[accu_y, accu_x, c]; bgr (H, W, 3) uint8, res (H, W, 3) int32 */
void orc_coviar_residual(const unsigned char* bgr_cur, const unsigned char* bgr_ref, const int* accu, int width, int height,
                         int* res) {
  for (int y = 0; y < height; ++y)
    for (int x = 0; x < width; ++x) {
      const size_t o = ((size_t)y * width + x);
      const int src_x = accu[o * 2], src_y = accu[o * 2 + 1];
      for (int c = 0; c < 3; ++c)
        res[o * 3 + c] = (int)bgr_cur[o * 3 + c] - (int)bgr_ref[((size_t)src_y * width + src_x) * 3 + c];
    }
}
