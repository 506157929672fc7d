/*
 * lsfa_hip.h — C ABI of liblsfa_hip.so, the MI355X (gfx950) implementation of
 * LSFA's per-frame inference hot path.
 *
 * Conventions (all entry points):
 *   - plain C ABI: pointers + sizes, no C++/torch types.  Pointers are DEVICE
 *     pointers unless the parameter name ends in `_host`.
 *   - tensors are contiguous fp32 NCHW exactly as the reference operators
 *     receive them from MXNet (psroi_pooling-inl.h:73-76 CheckContiguous).
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  No
 *     entry point allocates, frees or synchronises unless documented; scratch
 *     memory comes from the caller (`ws`, sized by lsfa_*_workspace_bytes), so
 *     every launch sequence can be captured in a hipGraph.
 *   - return value: 0 on success, a negative LSFA_E* code for argument errors,
 *     or a positive hipError_t.  Nothing throws.  lsfa_last_error() returns a
 *     thread-local message (the counterpart of MXGetLastError, which is how
 *     the reference's CHECK_xx / LOG(FATAL) reach Python as MXNetError).
 *
 * Each declaration cites the reference interface it replaces
 * (paths relative to the hustvl/LSFA tree).
 */
#ifndef LSFA_HIP_H_
#define LSFA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LSFA_OK 0
#define LSFA_EINVAL (-1)    /* bad argument (the reference's CHECK_EQ / CHECK_GT) */
#define LSFA_EWORKSPACE (-2) /* workspace too small */
#define LSFA_ENOTSUP (-3)   /* shape outside what the kernels support */

const char* lsfa_last_error(void);
/* ABI version of this header; bumped on any signature change. */
int lsfa_abi_version(void);

/* ------------------------------------------------------------------------ *
 * Position-sensitive ROI pooling.
 * Replaces: PSROIPoolingOp<gpu>::Forward   dff_rfcn/operator_cxx/psroi_pooling-inl.h:55-80
 *           PSROIPoolForward / PSROIPoolForwardKernel  psroi_pooling.cu:32-128
 *           params (spatial_scale, output_dim, pooled_size, group_size)  psroi_pooling-inl.h:33-47
 * data  (N, output_dim*group*group, H, W);  rois (R,5) = [batch_idx,x1,y1,x2,y2]
 * out   (R, output_dim, pooled, pooled)
 * mapping_channel (R, output_dim, pooled, pooled) or NULL (the reference's
 *       second, invisible output "maxidx"; written as float like the reference).
 * ------------------------------------------------------------------------ */
int lsfa_psroi_pool_fwd(const float* data, const float* rois,
                        int N, int C, int H, int W, int R,
                        float spatial_scale, int output_dim, int pooled_size, int group_size,
                        float* out, float* mapping_channel, void* stream);

/* Fused R-FCN head tail: PSROI pooling of the class map and of the box map,
 * the global 7x7 average (Pooling global_pool avg) and the class softmax, in
 * one launch, without materialising the (R,dim,7,7) tensors.
 * Replaces: psroipooled_cls_rois / psroipooled_loc_rois / ave_cls_scors_rois /
 *           ave_bbox_pred_rois / cls_prob   dff_rfcn/symbols/resnet_v1_101_flownet_rfcn.py:520-540 (key), :628-648 (cur)
 * cls_map (N, ncls*g*g, H, W), box_map (N, nbox*g*g, H, W), rois (R,5)
 * cls_prob (R, ncls)  softmax over classes;  cls_score (R,ncls) pre-softmax average or NULL
 * bbox_pred (R, nbox)
 * ------------------------------------------------------------------------ */
int lsfa_rfcn_head_fwd(const float* cls_map, const float* box_map, const float* rois,
                       int N, int H, int W, int R, int ncls, int nbox,
                       float spatial_scale, int pooled_size, int group_size,
                       float* cls_prob, float* cls_score, float* bbox_pred, void* stream);

/* The same head on a position-sensitive layout: ps_map (N, H, W, group*group, ncls+nbox), i.e. for
 * every cell and bin the ncls class scores then the nbox box values are contiguous.  That is what
 * rfcn_cls and rfcn_bbox (resnet_v1_101_flownet_rfcn.py:517-518) produce when both 1x1 convolutions
 * run as ONE GEMM [H*W,512] x [512, group^2*(ncls+nbox)] with the weight rows permuted
 * (row (gh*G+gw)*(ncls+nbox)+d  <-  rfcn_cls row (d*G+gh)*G+gw, or rfcn_bbox row ((d-ncls)*G+gh)*G+gw).
 * Results are bit-identical to lsfa_rfcn_head_fwd on the equivalent NCHW maps. */
int lsfa_rfcn_head_ps_fwd(const float* ps_map, const float* rois,
                          int N, int H, int W, int R, int ncls, int nbox,
                          float spatial_scale, int pooled_size, int group_size,
                          float* cls_prob, float* cls_score, float* bbox_pred, void* stream);
/* ... with `cell_ld` floats between consecutive cells of ps_map (>= group*group*(ncls+nbox)): the map as lsfa_conv_split_fwd
 * writes it, its 1911 columns padded to 1920 = 30 x 64 output channels. */
int lsfa_rfcn_head_ps_ld_fwd(const float* ps_map, int cell_ld, const float* rois, int N, int H, int W, int R, int ncls,
                             int nbox, float spatial_scale, int pooled_size, int group_size, float* cls_prob,
                             float* cls_score, float* bbox_pred, void* stream);

/* ------------------------------------------------------------------------ *
 * Motion-vector / flow guided bilinear feature warp with fused epilogue.
 * Replaces: mx.sym.GridGenerator(transform_type='warp') + mx.sym.BilinearSampler
 *           at resnet_v1_101_flownet_rfcn.py:468-469 (key), :571-572 (cur), :678-679 (batch)
 *           `* scale_map` :470 ; res_diff_ada (rnet_conv0 1x1 3->C, +bias) :57-67 and
 *           `conv_feat + res_diff` :576 ; `cur_feat + warp_conv_feat` :236.
 * out[n,c,y,x] = bilerp(feat[n,c], x+flow[n,0,y,x], y+flow[n,1,y,x])   (taps outside the map are 0)
 *                [* mul[n,c,y,x]]                                         if mul  != NULL
 *                [+ (res_w[c,:] . res[n,:,y,x] + res_b[c])]               if res  != NULL (res has res_c channels)
 *                [+ add[n,c,y,x]]                                         if add  != NULL
 * feat may have a smaller batch than flow (feat_n divides N): map n samples feature n mod feat_n.  feat_n = 1 broadcasts the
 * key-frame feature as tile_as does (operator_py/tile_as.py:16-19); feat_n = B < N serves the F frames of a segment of B lock-step
 * clips laid out frame-major (map f*B + b reads clip b's key feature).
 * ------------------------------------------------------------------------ */
int lsfa_warp_bilinear(const float* feat, int feat_n, const float* flow,
                       int N, int C, int H, int W,
                       const float* mul, const float* add,
                       const float* res, int res_c, const float* res_w, const float* res_b,
                       float* out, void* stream);
/* r6: the same operator on CHANNELS-LAST maps - feat_cl (feat_n, H, W, C), add_cl / out_cl (N, H, W, C), C a multiple of 4, 16-byte aligned;
 * flow (N, 2, H, W) and res (N, res_c, H, W) as above; no `mul` (the non-key path's epilogue: + rnet_conv0(res) + add).  Same arithmetic,
 * same bits as lsfa_warp_bilinear on the transposed maps.  amax_out (or NULL): 256 zeroed slots that receive max|out| over channels
 * [amax_c0, C) (a multiple of 4) - the scale of the convolution that reads them: the R-FCN score maps take channels [512, 1024), the RPN head
 * [0, 512) with an exact three-piece cut; both are GEMMs over the channel axis, so with this layout the non-key path needs no
 * lsfa_nchw_to_nhwc in front of them. */
int lsfa_warp_bilinear_cl(const float* feat_cl, int feat_n, const float* flow, int N, int C, int H, int W, const float* add_cl,
                          const float* res, int res_c, const float* res_w, const float* res_b, float* out_cl, unsigned* amax_out,
                          int amax_c0, void* stream);
/* Kernel choice (process-wide; results are identical bit for bit): 0 = by shape (default), 1 = the gather kernel only (rounds 1-2),
 * 2 = the LDS-staged kernel (round 3: whole planes copied into LDS by DMA, taps read from LDS) or LSFA_ENOTSUP when the shape or
 * alignment does not allow it (H*W even and <= 4096, 16-byte aligned maps, (C*H*W) % 4 == 0). */
int lsfa_warp_set_variant(int variant);

/* ------------------------------------------------------------------------ *
 * Long-term aggregation combine (Nq_net tail).
 * Replaces: softmax(axis=0) + SliceChannel + tile x2 + mul x2 + add
 *           resnet_v1_101_flownet_rfcn.py:104-108
 * logits (2,1,H,W): index 0 weights `a` (warped old key feature), 1 weights `b`.
 * out = w0*a + w1*b,  (w0,w1) = softmax(logits[:, 0, y, x]).
 * ------------------------------------------------------------------------ */
int lsfa_aggregate_softmax2(const float* a, const float* b, const float* logits,
                            int C, int H, int W, float* out, void* stream);

/* N maps per launch: a, b, out (N, C, H, W); logits (2, N, H, W) = the Nq convolutions' output for
 * Concat(warp x N, cur x N) on the batch axis (:95), row n weighting a[n], row N+n weighting b[n].
 * N = 1 is lsfa_aggregate_softmax2.  Used when several clips advance together (BASELINE configs[2]) and for
 * the HBM-resident roofline measurement (configs[4]). */
int lsfa_aggregate_softmax2_batched(const float* a, const float* b, const float* logits,
                                    int N, int C, int H, int W, float* out, void* stream);
/* r5: the same with the 2N logit rows `logit_row_stride` floats apart (>= H*W): channel 0 of the Nq net's last convolution
 * written NCHW with its output channels padded to 64 is read in place (row stride 64*H*W) instead of through a strided copy */
int lsfa_aggregate_softmax2_rows(const float* a, const float* b, const float* logits, long logit_row_stride,
                                 int N, int C, int H, int W, float* out, void* stream);

/* Fgfa variant: weights from cosine similarity of 2 embeddings
 * Replaces: compute_weight + softmax + tile/mul/add  resnet_v1_101_flownet_rfcn.py:111-116, :136-147
 * emb_cur, emb_warp (1,E,H,W); a = warped feature, b = current feature (1,C,H,W).
 * w_a = <norm(emb_warp), norm(emb_cur)>, w_b = <norm(emb_cur), norm(emb_cur)>, softmax over the two. */
int lsfa_aggregate_cosine(const float* a, const float* b, const float* emb_warp, const float* emb_cur,
                          int C, int E, int H, int W, float* out, void* stream);

/* ------------------------------------------------------------------------ *
 * RPN proposal generation (anchors -> decode -> clip -> filter -> stable sort
 * -> top pre_n -> NMS -> first post_n, cyclic pad).
 * Replaces: MultiProposalGPUOp::Forward  dff_rfcn/operator_cxx/multi_proposal.cu:403-558
 *           (+ kernels :47-388, anchors multi_proposal-inl.h:256-295, params :124-159);
 *           this is also the in-repo specification of mx.contrib.sym.Proposal as called at
 *           resnet_v1_101_flownet_rfcn.py:501, :609.
 * cls_prob (B, 2A, H, W), bbox_pred (B, 4A, H, W), im_info (B,3) = [im_h, im_w, im_scale]
 * rois (B*post_n, 5), scores (B*post_n, 1) or NULL.
 * `ws` must hold lsfa_proposal_workspace_bytes(...) bytes.
 * ------------------------------------------------------------------------ */
size_t lsfa_proposal_workspace_bytes(int B, int A, int H, int W, int pre_nms_top_n);
/* Launch plan (process-wide; results are identical bit for bit): AUTO picks the chip-wide plan (six short
 * kernels spread over the CUs) for count <= 32768 anchors and pre_nms_top_n <= 8192, else the single-workgroup
 * plan (one 16-wave workgroup per image, everything in its CU's LDS).  r6: inside the chip-wide plan AUTO builds the full
 * suppression mask for ONE image (lowest latency) and takes the mask-free box sweep for B >= 2 (a batched pass: one workgroup
 * per image instead of every CU; +2-4 % frames/s in the pipeline).  The workspace size does not depend on it. */
enum { LSFA_PROPOSAL_PLAN_AUTO = 0, LSFA_PROPOSAL_PLAN_SINGLE_WORKGROUP = 1, LSFA_PROPOSAL_PLAN_CHIP_WIDE = 2,
       /* the chip-wide plan with a mask-free NMS stage (r3 experiment, same results): only the diagonal tiles of the suppression mask are
        * built and one 16-wave workgroup decides everything across blocks from the survivors' boxes in LDS.  No dependent trips to
        * L2, but 64 x (survivors so far) IoU tests per block on ONE CU's four SIMDs: 160 us against 18 + 48 us for mask + sweep on the
        * benchmark's RPN, so it is not what AUTO picks */
       LSFA_PROPOSAL_PLAN_CHIP_WIDE_BOX_SWEEP = 3,
       /* r6, ABLATION ONLY (tools/lab/tail_ablation.sh, LSFA_LAB_SKIP_TAIL=1): the chip-wide plan WITHOUT its suppression stage - rois and
        * scores are not written.  What the NMS launches cost the frame pipeline is the difference to a normal run; never a product setting. */
       LSFA_PROPOSAL_PLAN_LAB_NO_NMS = 99 };
int lsfa_proposal_set_plan(int plan);
int lsfa_proposal(const float* cls_prob, const float* bbox_pred, const float* im_info,
                  int B, int A, int H, int W, int feature_stride,
                  const float* scales_host, int n_scales, const float* ratios_host, int n_ratios,
                  int rpn_pre_nms_top_n, int rpn_post_nms_top_n, float threshold, int rpn_min_size,
                  float* rois, float* scores, void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------ *
 * NMS on score-sorted boxes, device resident.
 * Replaces: nms_kernel + the host sweep of _nms   lib/nms/nms_kernel.cu:40-84, :97-150
 *           and multi_proposal.cu:262-357.
 * boxes (n, box_dim>=4) sorted by score descending; keep (n) int32 indices into
 * boxes, num_keep (1) int32; both device.  ws >= lsfa_nms_workspace_bytes(n).
 * ------------------------------------------------------------------------ */
size_t lsfa_nms_workspace_bytes(int n);
int lsfa_nms_sorted(const float* boxes, int n, int box_dim, float thresh,
                    int* keep, int* num_keep, void* ws, size_t ws_bytes, void* stream);

/* The same sweep on float64 boxes with numpy's arithmetic and keep rule (`ovr <= thresh` survives).
 * Replaces: the loop of nms()   lib/nms/nms.py:47-72   after its `scores.argsort()[::-1]` (:45), which the
 * caller does (the host side mirrors the reference: lsfa_amd/nms/nms.py).  pred_eval's per-class NMS runs on
 * float64 dets (dff_rfcn/core/tester.py:270-271); this is that path for callers of py_nms_wrapper. */
int lsfa_nms_sorted_f64(const double* boxes, int n, int box_dim, double thresh,
                        int* keep, int* num_keep, void* ws, size_t ws_bytes, void* stream);

/* The reference's exact host-pointer entry point (lib/nms/gpu_nms.hpp:14-15):
 * boxes_host (boxes_num, boxes_dim) already sorted by score; keep_out has room
 * for boxes_num ints.  Allocates, copies and synchronises like the original. */
void _nms(int* keep_out, int* num_out, const float* boxes_host, int boxes_num,
          int boxes_dim, float nms_overlap_thresh, int device_id);

/* ------------------------------------------------------------------------ *
 * Per-frame detection post-processing, all classes in one launch.
 * Replaces: bbox_pred (nonlinear_pred) lib/bbox/bbox_transform.py:103-140, clip_boxes :45-60,
 *           `/ scale` dff_rfcn/core/tester.py:148-152, and the per-class threshold + NMS +
 *           max_per_image loop tester.py:265-281 (py_nms_wrapper -> lib/nms/nms.py:37-74).
 * rois (R,5) [batch,x1,y1,x2,y2]; deltas (R, 4*nreg); probs (R, ncls).
 * class_agnostic != 0: every class uses delta columns 4:8 (tester.py:269).
 * Arithmetic is float64 like the numpy reference (nonlinear_pred upcasts, :114).
 * Outputs: dets (ncls, R, 5) float64 rows [x1,y1,x2,y2,score] for the survivors of
 * class j in NMS (score-descending) order, counts (ncls) int32 (class 0 = background: count 0),
 * keep_idx (ncls, R) int32 = roi index of each survivor (may be NULL).
 * max_per_image <= 0 disables the cap.
 * ------------------------------------------------------------------------ */
size_t lsfa_det_workspace_bytes(int R, int ncls);
int lsfa_det_postprocess(const float* rois, const float* deltas, const float* probs,
                         int R, int ncls, int nreg, int class_agnostic,
                         double im_h, double im_w, double scale,
                         double score_thresh, double nms_thresh, int max_per_image,
                         double* dets, int* counts, int* keep_idx,
                         void* ws, size_t ws_bytes, void* stream);

/* The same for the B images of a batch in one launch pair (the non-key frames of a segment, the clips of a lock-step batch): image b's rois /
 * deltas / probs are rows [b*R, (b+1)*R) (MultiProposal's layout, multi_proposal.cu:560-575), its outputs dets[b] (ncls, R, 5), counts[b] (ncls),
 * keep_idx[b] (ncls, R); one image size and scale for all (frames of one clip). */
int lsfa_det_postprocess_batch(const float* rois, const float* deltas, const float* probs,
                               int B, int R, int ncls, int nreg, int class_agnostic,
                               double im_h, double im_w, double scale,
                               double score_thresh, double nms_thresh, int max_per_image,
                               double* dets, int* counts, int* keep_idx,
                               void* ws, size_t ws_bytes, void* stream);

/* ------------------------------------------------------------------------ *
 * RPN head.  rpn_cls_score + rpn_bbox_pred (1x1 convolutions of channels [0, 512) of the NCHW feature, both as ONE convolution: lsfa_conv_fwd
 * with x_nchw = 1, output channel o < 2A = score channel o, 2A <= o < 6A = box delta o - 2A, padded to 64) leave `logits` (N, H*W, ld)
 * channels-last; lsfa_rpn_softmax_split applies the per-anchor two-way softmax (Reshape (2, A*H, W) -> SoftmaxActivation(channel) ->
 * Reshape) and splits them into the NCHW maps MultiProposal takes.
 * Replaces: dff_rfcn/symbols/resnet_v1_101_flownet_rfcn.py:479-494 (and the rpn_inv_normalize the caller folds into the weights).
 * cls_prob (N, 2A, H, W): channel a = background, A + a = foreground of anchor a; bbox_pred (N, 4A, H, W).
 * ------------------------------------------------------------------------ */
int lsfa_rpn_softmax_split(const float* logits, int N, int H, int W, int ld, int A, float* cls_prob, float* bbox_pred, void* stream);

/* Box decode + clip + rescale only (float64 out), for callers that keep the
 * reference's im_detect() signature:  tester.py:143-152. pred_boxes (R, 4*nreg) */
int lsfa_bbox_pred_clip(const float* rois, const float* deltas, int R, int nreg,
                        double im_h, double im_w, double scale, double* pred_boxes, void* stream);

/* ------------------------------------------------------------------------ *
 * Deformable convolution support: bilinear im2col of the DCN units
 * (mx.contrib.symbol.DeformableConvolution, dff_rfcn/symbols/sym_common.py:138-157).
 * data (N,C,H,W), offset (N, 2*kh*kw*dg, Ho, Wo) -> col (N, C*kh*kw, Ho*Wo);
 * the GEMM with the (Cout, C*kh*kw) weight is the caller's (MFMA library GEMM).
 * ------------------------------------------------------------------------ */
int lsfa_deform_im2col(const float* data, const float* offset,
                       int N, int C, int H, int W, int kh, int kw,
                       int pad, int stride, int dilate, int deform_groups,
                       int Ho, int Wo, float* col, void* stream);

/* The same sampling for channels-last tensors: data (N,H,W,C), offset (N,Ho,Wo,2*kh*kw*dg),
 * col (N, Ho*Wo, kh*kw, C) — a row of col is one output pixel, ordered (tap, channel), so the
 * contraction is rows x (kh*kw*C, Cout).  C/deform_groups must be a multiple of 4. */
int lsfa_deform_im2col_cl(const float* data, const float* offset,
                          int N, int C, int H, int W, int kh, int kw,
                          int pad, int stride, int dilate, int deform_groups,
                          int Ho, int Wo, float* col, void* stream);
/* ... with the offsets of a pixel offset_ld floats apart (>= 2*kh*kw*dg): the offset branch of a DCN unit
 * (sym_common.py:249-257, 72 channels) computed by lsfa_conv_split_fwd with its output channels padded to 128. */
int lsfa_deform_im2col_cl_ld(const float* data, const float* offset, int offset_ld,
                             int N, int C, int H, int W, int kh, int kw,
                             int pad, int stride, int dilate, int deform_groups,
                             int Ho, int Wo, float* col, void* stream);

/* ------------------------------------------------------------------------ *
 * Convolution + bias + ReLU on channels-last maps with the fp32 matrix cores (implicit GEMM).
 * Replaces: mx.sym.Convolution (no_bias) + the BatchNorm folded into it + Activation('relu') of the
 *           pre-activation ResNet units' conv2   dff_rfcn/symbols/resnet.py:84-93, sym_common.py:92-135
 *           (any kh x kw, stride, dilation; zero padding `pad` on every side).
 * x (N, H, W, Cin), y (N, Ho, Wo, Cout) channels-last; w (Cout, kh*kw, Cin): K contiguous per output channel
 * (the reference's (Cout, Cin, kh, kw) weight permuted once at bind time); bias (Cout) or NULL.
 * Cin % 32 == 0, Cout % 64 == 0.  Small tile grids split the taps over workgroups and need
 * lsfa_conv_nhwc_workspace_bytes(...) of workspace (partial tiles, added in a fixed order: deterministic).
 * ------------------------------------------------------------------------ */
size_t lsfa_conv_nhwc_workspace_bytes(int N, int H, int W, int Cout, int kh, int kw, int stride, int pad, int dil);
int lsfa_conv_nhwc_fwd(const float* x, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                       int kh, int kw, int stride, int pad, int dil, int relu, float* y,
                       void* ws, size_t ws_bytes, void* stream);

/* The same convolution with the tail of a pre-activation unit fused in (resnet.py:93-101 + the next unit's :78-80):
 * y = conv(x) [+ bias] + residual (residual may be y itself: in place), and, when y2 != NULL, the NEXT unit's
 * BatchNorm + ReLU of that sum as a second output, y2 = max(y*scale2[c] + shift2[c], 0) — so conv3 + the shortcut add
 * + bn1/relu1 of the following unit are one launch. */
int lsfa_conv_nhwc_fused_fwd(const float* x, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                             int kh, int kw, int stride, int pad, int dil, int relu, const float* residual, float* y,
                             float* y2, const float* scale2, const float* shift2,
                             void* ws, size_t ws_bytes, void* stream);

/* The convolutions of the frame path (and, with kh = kw = 1, the 1x1 convolutions = plain GEMMs of the channels-last units):
 * fp32 in, fp32 accumulate, fp32 out, with every fp32 product formed on the bf16 / fp16 matrix pipe from `pieces` pieces per operand
 * (gfx950 has no xf32 MFMA and its fp32 MFMA runs at 1/16 of the bf16 / fp16 rate):
 *   pieces = 3  three bf16 pieces (8 + 8 + 8 mantissa bits, an exact cut) and the six partial products of weight >= 2^-16;
 *   pieces = 2  two fp16 pieces of x * s and w * 2^w_exp (hi = fp16(v), lo = fp16(v - hi)) and the three products hi hi + hi lo + lo hi:
 *               half the matrix-pipe cycles of the six-product form and at least as close to a float64 convolution (tests/test_hip_ops.py).
 *               fp16's exponent range makes it need the map's scale: `amax_in` = 256 floats whose maximum is >= max|x| (lsfa_amax_partial
 *               of x, a producing convolution's `amax_out`, or of a map that bounds x); s is the power of two that puts that maximum
 *               into [2^13, 2^14).  An UNDER-estimate makes fp16(x s) overflow: the output turns non-finite and bit 0 of *status is
 *               raised - lsfa_status_check() turns it into an error instead of a silently wrong feature map.
 *   pieces = 1  one bf16 piece (round to nearest even), one product: the bf16 mode (BASELINE configs[2]).
 * Replaces mx.sym.Convolution (+ the BatchNorm / ReLU / residual add around it) of dff_rfcn/symbols/resnet.py:70-101,
 * sym_common.py:92-135, resnet_v1_101_flownet_rfcn.py:44-55 (feat_conv_3x3), :150-207 (FlowNet), :94-109 (Nq), :209-236 (small net),
 * :479-546 (RPN / R-FCN score maps).
 * Weights are cut and laid out once at bind time: lsfa_conv_weights writes lsfa_conv_weight_bytes(...) bytes (2 * pieces per weight, in
 * MFMA fragment order) from w (Cout, kh, kw, Cin); Cin % 32 == 0, Cout % 64 == 0; w_exp (pieces == 2 only) = 13 - floor(log2 max|w|).
 * Operands may be VIEWS of wider channels-last maps (FlowNet's Concat / Crop / Deconvolution nodes then need no copies):
 *   x   (N, H, W, lda) with the Cin input channels first in each pixel's row (lda = 0: Cin; a multiple of 4);
 *   y   points at the first output element; ldy floats between output pixels (0: Cout); out_H > 0 places output pixel (oy, ox) of
 *       image n at y + (((n*out_H + oy*out_sy)*out_W + ox*out_sx) * ldy) - a channel slice of a concatenated map, or every other pixel
 *       of it (one parity of a stride-2 transposed convolution); Ho, Wo > 0: the output grid of this launch when it is not the
 *       convolution's own;
 *   y_nchw != 0: y, y2 and residual are (N, Cout, Ho, Wo) - the layout the reference's operators take - instead of channels-last.
 * Epilogue: + bias[c], + residual (may alias y), act (0 none, 1 ReLU, 2 LeakyReLU 0.1), and optionally a second output
 * y2 = max(y * scale2[c] + shift2[c], 0) - the NEXT pre-activation unit's bn1 + relu1 (resnet.py:77-79) - and `amax_out`: 256 unsigned
 * slots that receive (atomicMax on the bit pattern; the caller zeroes them once per frame) max|y2| if scale2 / shift2 are given (with
 * y2 == NULL that maximum is all the second output leaves behind), else max|y|.
 * K may be cut into slices whose partial sums go through `ws` and are added in a fixed order (bit-reproducible). */
typedef struct lsfa_conv_desc {
  const float* x; int lda; int N, H, W, Cin;
  const void* wfrag; int pieces; int w_exp; const float* amax_in;
  const float* bias; int Cout, kh, kw, stride, pad_h, pad_w, dil;
  int act; int y_nchw; const float* residual; float* y; int ldy;
  float* y2; const float* scale2; const float* shift2;
  unsigned* amax_out; unsigned* status;
  int Ho, Wo, out_H, out_W, out_sy, out_sx;
  int prof_tag;          /* lsfa_prof_*: r5: ignored - every call of the family is timed as "conv" (FLOPs and time over the same calls) */
  int x_nchw;            /* != 0: x is an NCHW map (N, lda, H, W) whose channels [0, Cin) are the input (K-major for the contraction): 1x1 /
                          * stride 1 / no padding and small weights only (Cout * Cin * 2 * pieces <= 256 KB per 64 output channels) -
                          * the RPN's 1x1 convolutions on the feature map the reference's operators exchange; else LSFA_ENOTSUP */
  const float* in_scale; /* with in_shift (Cin floats each, or both NULL): the convolution's input is max(x * in_scale[k] + in_shift[k], 0), */
  const float* in_shift; /* k the input channel, applied where the operand is cut - a pre-activation ResNet unit's bn1 + relu1
                          * (dff_rfcn/symbols/resnet.py:78-80) on the previous unit's sum without that map ever being stored: the
                          * previous conv3 is given scale2 / shift2 but y2 = NULL and only publishes the activated map's maximum in its
                          * amax_out, which is this call's amax_in.  1x1, no padding, Cin <= 2048, pieces 1 or 2; else LSFA_ENOTSUP.  The
                          * values multiplied are bit for bit the ones y2 would have held. */
  const float* w_scale;  /* r5, pieces == 2: Cout floats 2^-w_exp[co] for weights cut by lsfa_conv_weights_pc (one power-of-two scale per
                          * OUTPUT channel: a BatchNorm folded into trained weights spreads the channels' magnitudes over many octaves, and
                          * every channel keeps its 22 bits); NULL: one scale 2^w_exp for the whole tensor (lsfa_conv_weights) */
} lsfa_conv_desc;
size_t lsfa_conv_weight_bytes(int Cout, int kh, int kw, int Cin, int pieces);
int lsfa_conv_weights(const float* w, int Cout, int kh, int kw, int Cin, int pieces, int w_exp, void* wfrag, void* stream);
/* r5: the two-piece cut with w_exp_pc[co] (device, Cout ints) per output channel; pass 2^-w_exp_pc[co] as lsfa_conv_desc::w_scale */
int lsfa_conv_weights_pc(const float* w, int Cout, int kh, int kw, int Cin, const int* w_exp_pc, void* wfrag, void* stream);
size_t lsfa_conv_workspace_bytes(const lsfa_conv_desc* d);
int lsfa_conv_fwd(const lsfa_conv_desc* d, void* ws, size_t ws_bytes, void* stream);
/* 256 partial maxima of |x| (n floats, n % 4 == 0, 16-byte aligned): an `amax_in` for maps no convolution of this library produced */
int lsfa_amax_partial(const float* x, long long n, float* out256, void* stream);
/* Reads and clears the status word the convolutions raise (synchronises `stream`): LSFA_OK, or LSFA_EOVERFLOW with lsfa_last_error()
 * saying which bit was set.  The frame loop calls it where it synchronises anyway (end of a video, after a benchmark region). */
#define LSFA_EOVERFLOW (-4)
int lsfa_status_check(unsigned* status_dev, void* stream);
/* measurement hook (tools/lab/conv_ring_lab.py): force the kernel (1: the ring kernel, mixed-role waves; 2: the ring kernel with
 * loader / consumer waves; never the direct form then; 3 was the 3x3 halo form, removed in r6: refused), the tile width nt (2 | 4), the
 * ring depth st (2..4) and the number of K slices; 0 = the launch plan decides.  Process-wide; results stay bit-reproducible per setting. */
int lsfa_conv_plan_override(int kernel, int nt, int st, int slices);
/* r6, measurement / test hook: the two orders the ring kernel's launch can be laid out in, process-wide like the override above; -1 = back to
 * the default (or the LSFA_CONV_TILE_ORDER / LSFA_CONV_K_ORDER environment variables).  tile_order: 0 = workgroups numbered with the pixel tile
 * fastest, 1 = the channel tile fastest (all channel tiles of a pixel tile on one XCD); results do not depend on it.  k_order: 0 = K walked tap by
 * tap (a tap's channel chunks, then the next tap), 1 = channel chunk by channel chunk (a chunk's taps, then the next chunk: a tap's rows are the
 * previous tap's rows shifted, served from the CU's L1); the summation order differs between the two, within a setting every plan agrees. */
int lsfa_conv_order_override(int tile_order, int k_order);
/* r5: 4 = the ring kernel on 256-pixel tiles (eight mixed-role waves; nt 4, pieces 1 | 2).
 * measurement hook (bench.py's per-instantiation roofline table): which kernel lsfa_conv_fwd would launch for `d` -
 * out[0] kernel (1 ring, 2 direct), out[1] nt, out[2] st, out[3] loader / consumer waves, out[4] waves that multiply (4 | 8),
 * out[5] K slices, out[6] the input's activation applied at the cut, out[7] pieces.  Launches nothing. */
int lsfa_conv_plan_query(const lsfa_conv_desc* d, int* out8);
/* Deconvolution(kernel 4, stride 2) + Crop(offset (1,1)) to Hc x Wc (+ bias + activation) as ONE launch of four 2x2-tap phase
 * convolutions (resnet_v1_101_flownet_rfcn.py:170-176 `deconv5` ... `deconv2`): x (N, Hi, Wi, lda) with Cin channels used,
 * wfrag4 = four consecutive blocks of lsfa_conv_weight_bytes(Cout, 2, 2, Cin, pieces) bytes, block py*2 + px =
 * lsfa_conv_weights of the (Cout, 2, 2, Cin) weight w[:, :, kys, kxs] transposed to (out, in, ky, kx), kys = (3, 1) for py = 0
 * and (2, 0) for py = 1 (kxs alike); y points at channel c0 of pixel (0, 0) of the (N, Hc, Wc, ldy) map. */
size_t lsfa_deconv4x4s2_crop_workspace_bytes(int N, int Hi, int Wi, int Cin, int Cout, int Hc, int Wc, int pieces);
int lsfa_deconv4x4s2_crop_fwd(const float* x, int lda, int N, int Hi, int Wi, int Cin, const void* wfrag4, int pieces, int w_exp,
                              const float* amax_in, const float* bias, int Cout, int act, float* y, int ldy, int Hc, int Wc,
                              unsigned* amax_out, unsigned* status, void* ws, size_t ws_bytes, void* stream);

/* ---------------------------------------------------------------------------
 * The stem of the ResNets and the frame shrink in front of the small net, three launches instead of six library ones.
 *   lsfa_avgpool_nchw      mx.symbol.Pooling(kernel=(k,k), stride=(k,k), pool_type='avg', pooling_convention='full')
 *                          (dff_rfcn/symbols/resnet_v1_101_flownet_rfcn.py:216 `resize_data`, k = 4): x (N,C,H,W) -> y (N,C,ceil(H/k),
 *                          ceil(W/k)); edge windows are clipped to the image and averaged over what they hold.
 *   lsfa_stem_weights      conv0's weight as w_l (3,7,7,64) floats = [ci][ky][kx][co] with bn0's scale folded in -> wfrag, the
 *                          lsfa_stem_weight_bytes() (16-byte aligned) bytes lsfa_stem_conv7x7s2 reads: the matrix-instruction
 *                          fragments of the weights as two fp16 pieces with a power-of-two scale per output channel.  Once per
 *                          weight load.
 *   lsfa_stem_conv7x7s2    bn_data + conv0 + bn0 + relu0 (dff_rfcn/symbols/resnet.py:151, :162): x (N,3,H,W) NCHW; in_scale/in_shift (3)
 *                          = bn_data as a per-channel affine (NULL: none), applied before the zero padding; wfrag from
 *                          lsfa_stem_weights; bias (64) = bn0's shift; y (N,Ho,Wo,64) channels-last, Ho = (H-1)/2+1.  fp32 operands
 *                          as two fp16 pieces on the matrix pipe (three products, fp32 accumulation: fp32 accuracy, like
 *                          lsfa_conv_fwd's pieces = 2); the input's scale is taken per 4 x 32 output tile from the tile's own
 *                          patch, so an image's result does not depend on what else is in the batch.
 *   lsfa_maxpool3x3s2_nhwc pool0 (resnet.py:163: 3x3, stride 2, pad 1, max): x (N,H,W,C) -> y (N,(H-1)/2+1,(W-1)/2+1,C), C % 4 == 0;
 *                          y2 != NULL: also max(y*scale2[c] + shift2[c], 0), the first unit's bn1 + relu1 (resnet.py:78-80).
 *                          amax_out != NULL: 256 slots (zeroed by the caller) that receive max|y2| (max|y| without y2) the way
 *                          lsfa_conv_fwd's amax_out does, for the next convolution's amax_in.
 * ------------------------------------------------------------------------ */
int lsfa_avgpool_nchw(const float* x, int N, int C, int H, int W, int k, float* y, void* stream);
/* r6: the same with the N images given by a DEVICE table of N base pointers (image n = C x H x W floats at x_table[n]): frames that arrive as
 * separate tensors (dff_rfcn/core/loader.py:131-141 hands one array per frame) go through a batched pass without a staging copy into one
 * (N, C, H, W) buffer.  The table's address is what a captured graph bakes in; lsfa_ptr_table_set rewrites its entries before a replay. */
int lsfa_avgpool_nchw_tbl(const float* const* x_table, int N, int C, int H, int W, int k, float* y, void* stream);
/* r5: transform (lib/utils/image.py:296-308), the last host-side step of a frame moved behind its upload: decoded frames
 * (N, H, W, 3) uint8 BGR on the device -> `data` (N, 3, H, W) float32, channel i = (im[..., 2 - i] - pixel_means[2 - i]) * pixel_scale.
 * computed in float64 like the reference's numpy statements and rounded to float32 once.
 * pixel_means_bgr_host: three doubles in host memory, B, G, R order (config.network.PIXEL_MEANS); H*W % 4 == 0. */
int lsfa_image_transform_u8(const unsigned char* im_hwc_bgr, int N, int H, int W, const double* pixel_means_bgr_host, double pixel_scale,
                            float* data_nchw, void* stream);
size_t lsfa_stem_weight_bytes(void);
int lsfa_stem_weights(const float* w_l, void* wfrag, void* stream);
int lsfa_stem_conv7x7s2(const float* x, int N, int H, int W, const float* in_scale, const float* in_shift,
                        const void* wfrag, const float* bias, float* y, void* stream);
int lsfa_maxpool3x3s2_nhwc(const float* x, int N, int H, int W, int C, float* y, float* y2, const float* scale2,
                           const float* shift2, unsigned* amax_out, void* stream);
/* lsfa_stem_conv7x7s2 with an accumulation input and a choice of activation: y = act(conv(x) + bias + accum), accum (N,Ho,Wo,64)
 * or NULL (may be y itself), act 0 none / 1 ReLU / 2 LeakyReLU(0.1).  FlowNet's flow_conv1 (7x7 / 2, 6 -> 64 channels,
 * resnet_v1_101_flownet_rfcn.py:153) is two such passes, one per image of the pair, with the weight's input channels 0-2 / 3-5.
 * amax_out (or NULL): 256 zeroed slots that receive max|y|, like lsfa_conv_fwd's. */
int lsfa_stem_conv7x7s2_ex(const float* x, int N, int H, int W, const float* in_scale, const float* in_shift,
                           const void* wfrag, const float* bias, const float* accum, int act, float* y, unsigned* amax_out,
                           void* stream);
/* r6: lsfa_stem_conv7x7s2_ex with the N images given by a device table of base pointers (3 x H x W floats each): see lsfa_avgpool_nchw_tbl. */
int lsfa_stem_conv7x7s2_tbl(const float* const* x_table, int N, int H, int W, const float* in_scale, const float* in_shift,
                            const void* wfrag, const float* bias, const float* accum, int act, float* y, unsigned* amax_out, void* stream);

/* ---------------------------------------------------------------------------
 * FlowNet-S pieces that are not MFMA-sized (lsfa_amd/csrc/flownet.hip); with lsfa_conv_split_view_fwd and the two-pass stem
 * convolution they make get_flownet (resnet_v1_101_flownet_rfcn.py:150-207) free of library calls.
 *   lsfa_head_conv3x3    Convolution1..5 (:178-203): 3x3, pad 1, stride 1, Cin -> Cout <= 4 channels.  x (N,H,W,lda) channels-last
 *                        (the first Cin channels of each pixel), w (Cout,3,3,Cin), y = (conv + bias) * mul either NCHW
 *                        (N,Cout,H,W) (out_nchw != 0: the `flow` output, mul = 2.5, :204) or channels [c0, c0+Cout) of (N,H,W,ldy).
 *   lsfa_upsample_flow   upsample_flow*to* (:180, :185, :190, :195): Deconvolution(kernel 4, stride 2, C -> C, C <= 8) + Crop(offset 1)
 *                        to Hc x Wc; in (N,Hi,Wi,C), w (C,C,4,4) (MXNet's (in, out, kh, kw)), out channels [c0, c0+C) of (N,Hc,Wc,ldy);
 *                        amax_out (or NULL): the 256 slots of the destination map that receive max|result| (see lsfa_conv_fwd).
 *   lsfa_avgpool2_nhwc   Pooling(kernel 2, stride 2, avg, 'full') on (N,H,W,C), C % 4 == 0 -> (N,ceil(H/2),ceil(W/2),C), edge
 *                        windows clipped (:201).
 * ------------------------------------------------------------------------ */
int lsfa_head_conv3x3(const float* x, int lda, int N, int H, int W, int Cin, const float* w, const float* bias, int Cout,
                      float mul, float* y, int out_nchw, int ldy, int c0, void* stream);
int lsfa_upsample_flow(const float* in, int N, int Hi, int Wi, int C, const float* w, const float* bias, int Hc, int Wc,
                       float* out, int ldy, int c0, unsigned* amax_out, void* stream);
int lsfa_avgpool2_nhwc(const float* x, int N, int H, int W, int C, float* y, void* stream);
/* channels [c0, c0 + C) of an (N, Ctot, HW) map as (N, HW, C) rows: the NCHW feature the reference's operators exchange
 * (`conv_feat`, resnet_v1_101_flownet_rfcn.py:479-481 SliceChannel; Concat(warp, feat), :94-133) in the channels-last form lsfa_conv_fwd
 * reads (the R-FCN score maps, the Nq / embedding nets).  amax_out (or NULL): 256 zeroed slots that receive max|x| of the copied values
 * (lsfa_conv_fwd's amax_in). */
int lsfa_nchw_to_nhwc(const float* x, int N, int Ctot, int HW, int c0, int C, float* y, unsigned* amax_out, void* stream);

/* Inference BatchNorm (use_global_stats) + ReLU as one pass: y = max(x*scale[c]+shift[c], 0)
 * (sym_common.py:92-102 bn + relu of every pre-activation unit, resnet.py:70-101).
 * relu != 0 applies the ReLU.  In-place (y == x) allowed. */
int lsfa_scale_shift_relu(const float* x, const float* scale, const float* shift,
                          int N, int C, int HW, int relu, float* y, void* stream);

/* y = leaky(x*scale[c]+shift[c]) with leaky(v) = v > 0 ? v : v*slope — a convolution's bias and the
 * LeakyReLU(0.1) that follows it in FlowNet (resnet_v1_101_flownet_rfcn.py:153-176) as one pass. */
int lsfa_scale_shift_leaky(const float* x, const float* scale, const float* shift,
                           int N, int C, int HW, float slope, float* y, void* stream);

/* The same pass for a channels-last map: x, y are (rows, C) with the channel the fastest axis
 * (rows = N*H*W), C a multiple of 4, all pointers 16-byte aligned.  In-place allowed. */
int lsfa_scale_shift_relu_cl(const float* x, const float* scale, const float* shift,
                             long long rows, int C, int relu, float* y, void* stream);

/* ------------------------------------------------------------------------ *
 * Compressed-domain motion vectors: accumulation back to the key frame, the accumulated field and
 * the residual.
 * Replaces: create_and_load_mv_residual (accumulate = 1)   external/data_loader_py2/coviar_data_loader.c:71-177
 *           and the identity initialisation of accu_src    :316-323.
 * mvs (n_mvs, 7) int32 DEVICE rows = AVMotionVector's {source, w, h, src_x, src_y, dst_x, dst_y} in decoder
 * order (later blocks overwrite earlier ones, as the reference's serial loop does); max_block_area >= w*h of
 * every block.  accu_* (H, W, 2) int32 = per pixel the (x, y) it came from in the GOP's first frame
 * (row-major here; the reference's buffer is x-major).  lsfa_mv_accumulate writes EVERY pixel of accu_new
 * (accu_new != accu_old): ping-pong the two buffers from frame to frame.
 * mv (H, W, 2) int32 = (x, y) - accu  (:131-141);  res (H, W, 3) int32 = cur - ref[accu]  (:144-171),
 * bgr_* (H, W, 3) uint8.  Outputs feed transform_mv_res (lib/utils/image.py:202-263).
 * ------------------------------------------------------------------------ */
size_t lsfa_mv_workspace_bytes(int width, int height);
int lsfa_mv_identity(int* accu, int width, int height, void* stream);
int lsfa_mv_accumulate(const int* mvs, int n_mvs, int max_block_area, const int* accu_old, int* accu_new,
                       int width, int height, void* ws, size_t ws_bytes, void* stream);
int lsfa_mv_field(const int* accu, int width, int height, int* mv, void* stream);
int lsfa_mv_residual(const unsigned char* bgr_cur, const unsigned char* bgr_ref, const int* accu,
                     int width, int height, int* res, void* stream);
/* r5: transform_mv_res (lib/utils/image.py:202-228) on the device: the decoded frame's motion vectors (H, W, 2) and residual (H, W, 3) -
 * int32 as lsfa_mv_field / lsfa_mv_residual leave them (flags bit 0) or float32 - to the network's `motion_vector` (1, 2, h, w) and `res_diff`
 * (1, 3, h, w): cv2.resize by im_scale (INTER_LINEAR, float32 work type), zero padding to rcnn_stride, the residual's in-place channel loop
 * (BGR -> RGB, minus pixel_means, times pixel_scale, channel 2 from the rewritten channel 0 as in the reference), cv2.resize by 1 / rcnn_stride
 * in float64, motion vectors times im_scale / rcnn_stride; rounded to float32 once, where the reference hands its float64 arrays to the
 * executor.  One launch, nothing materialised at full resolution.  h1, w1 = cvRound(H * im_scale), cvRound(W * im_scale) (the caller rounds as
 * numpy does); out_h, out_w = ceil(h1 / stride), ceil(w1 / stride) - checked.  pixel_means_bgr_host: three doubles in host memory.
 * flags bit 1: the motion vectors are negated first, the `motion_vector = - motion_vector` of get_image (lib/utils/image.py:54).
 * OpenCV is not in the reference tree: the interpolation arithmetic follows OpenCV 3.2's resize.cpp as restated in oracle/np_ref.py
 * (parity unpinned for that part; everything around it is pinned by golden G6).  One case of resize() is NOT restated: at a scale of
 * exactly 1/2 in both directions OpenCV turns INTER_LINEAR into INTER_AREA (resizeAreaFast: the mean of the 2 x 2 block as (a + b + c + d) * 0.25f,
 * in an order its SIMD and scalar paths do not share).  The two-tap form used here weighs the same four pixels by 0.25 each: identical for
 * integer-valued sources (decoder frames, motion vectors, residuals: every partial sum is exact), up to an ulp apart for fractional float32
 * maps.  The second resize (1 / rcnn_stride = 1/16) never takes that switch. */
int lsfa_transform_mv_res(const void* motion_vector, const void* res_diff, int flags, int H, int W, double im_scale, int h1, int w1,
                          int rcnn_stride, const double* pixel_means_bgr_host, double pixel_scale, float* out_mv, float* out_res,
                          int out_h, int out_w, void* stream);
/* r5: resize + transform (lib/utils/image.py:266-308) of decoded frames in one launch: N frames (H, W, 3) BGR, uint8 (is_u8) or float32, on the
 * device -> `data` (N, 3, out_h, out_w) float32: cv2.resize by im_scale on the float image (get_image's .astype(np.float32), :52; INTER_LINEAR,
 * float32 work type), zero padding to `stride` (config.network.IMAGE_STRIDE; 0: none), channel i = (im[..., 2 - i] - pixel_means[2 - i]) *
 * pixel_scale: the subtraction in float32 when stride == 0 (a float32 image minus a Python float) and in float64 when stride > 0 (the padded copy is
 * np.zeros(...), a float64 image: image.py:288-293), the product in float64, rounded to float32 once.  h1, w1 = cvRound(H * im_scale), cvRound(W * im_scale); out_h, out_w = h1, w1 rounded up to
 * the stride - checked.  (lsfa_image_transform_u8 is the uint8-image form of `transform`: float64 subtraction; the two agree for zero means.)
 * is_u8: 0 = float32 frames; 1 = uint8 frames converted to float and interpolated in float (a decoder's frame after get_image's .astype); 2 (r6) = a
 * uint8 frame as cv2.imread hands it over - the LAST frame of a video, :45 - interpolated on OpenCV's fixed-point uint8 path (2048-scaled short
 * coefficients, int32 horizontal pass, `(((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2` vertically: OpenCV 3.2 imgwarp.cpp, restated as
 * oracle/np_ref.py::cv2_resize_linear_u8 - parity unpinned like the float path), then transformed with a float64 subtraction (a uint8 image's rule).
 * The result is about one intensity level from the float interpolation at most (measured <= 0.81).  NOT restated (either path): OpenCV's switch to INTER_AREA at im_scale 0.5. */
int lsfa_image_resize_transform(const void* im_hwc_bgr, int is_u8, int N, int H, int W, double im_scale, int h1, int w1, int stride,
                                const double* pixel_means_bgr_host, double pixel_scale, float* data_nchw, int out_h, int out_w, void* stream);

/* ------------------------------------------------------------------------ *
 * Plumbing without a reference counterpart: a hipStream_t that is nobody else's (non-blocking; PyTorch's
 * streams come from a small round-robin pool, and two graphs captured on pool twins share one BLAS
 * workspace — lsfa_amd/core/streams.py wraps these in torch.cuda.ExternalStream for capture and replay).
 * ------------------------------------------------------------------------ */
int lsfa_stream_create(void** stream_out, int high_priority);
int lsfa_stream_destroy(void* stream);
/* up to 32 device-to-device copies of 4-byte elements (elems4[k] of them, dst[k] <- src[k]) as one launch: a frame's - or a whole
 * segment's - images, motion vectors and residuals into the static buffers a captured graph reads.  r5: src[k] == NULL zero-fills
 * dst[k] (amax slots, the padding channels of a Concat map): the frame path issues no PyTorch fill or copy kernel */
int lsfa_copy_many(int njobs, void* const* dst, const void* const* src, const long* elems4, void* stream);
/* r6: table_dev[i] = ptrs_host[i] for i < n, as one tiny launch per 64 entries on `stream` (the values travel as kernel arguments): how the
 * per-image pointer tables of the *_tbl entry points are rewritten between replays of a captured graph. */
int lsfa_ptr_table_set(void** table_dev, int n, const void* const* ptrs_host, void* stream);

/* ------------------------------------------------------------------------ *
 * Live per-op timing with HIP events on the launch stream (bench.py's roofline leg).
 * lsfa_prof_enable(mask) makes the entry points whose op id bit is set in `mask`
 * (-1 = all, 0 = off) bracket their launches with hipEventRecord on `stream`;
 * lsfa_prof_read synchronises the events and returns accumulated milliseconds and
 * launch counts per op id, then clears them.  Do not enable during graph capture.
 * ------------------------------------------------------------------------ */
enum {
  LSFA_OP_PSROI = 0, LSFA_OP_RFCN_HEAD = 1, LSFA_OP_WARP = 2, LSFA_OP_AGG = 3,
  LSFA_OP_PROPOSAL = 4, LSFA_OP_NMS = 5, LSFA_OP_DET = 6, LSFA_OP_DCN_IM2COL = 7,
  LSFA_OP_BNRELU = 8, LSFA_OP_CONV = 9, LSFA_OP_STEM = 10, LSFA_OP_FLOWNET = 11, LSFA_OP_COUNT = 12
};
int lsfa_prof_enable(int mask);
int lsfa_prof_read(double* ms_host /*LSFA_OP_COUNT*/, int* launches_host /*LSFA_OP_COUNT*/);
const char* lsfa_op_name(int op_id);

#ifdef __cplusplus
}
#endif
#endif  /* LSFA_HIP_H_ */
