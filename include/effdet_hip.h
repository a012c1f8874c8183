/* libeffdet_hip.so - C ABI of the MI355X (gfx950) EfficientDet inference + OOD-scoring hot path (and of the
 * pretrain step's loss / backward / optimizer operators).
 *
 * The reference (DavidPetrus/ood_object_detection, a fork of rwightman/efficientdet-pytorch 0.2.3) is
 * pure Python with no FFI layer: its "operator API" for this path is a handful of Python callables and
 * the PyTorch ops under them.  Each entry point below names the reference call site (file:line under
 * the reference tree) whose arithmetic it replaces; INTEGRATION.md shows the ctypes binding a
 * maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it is an array-of-parameters argument documented as
 *     host memory ("host:" below); buffers are caller-owned; nothing is allocated internally;
 *   - `stream` is a hipStream_t; all work is enqueued on it; no call synchronises (graph-capturable);
 *   - activations are NHWC, channels a multiple of 8; `dtype` 0 = float32 (parity mode),
 *     1 = bfloat16 (throughput mode, fp32 accumulate); weights of the MFMA GEMMs use the same dtype,
 *     every per-channel vector (scale/shift/bias/depthwise taps/SE weights) is float32; the effdet_train_* and
 *     effdet_eval_* entry points are float32 only (channels a multiple of 4);
 *   - `dtype` 2 = EFFDET_BF16X2, the "accurate" mode (accepted by effdet_stem_dw_fused[_u8], effdet_mbconv_expand_dw[_gated],
 *     effdet_pw_gemm_bn_act / _group, effdet_sepconv_fused, effdet_maxpool_same and the *_parts / *_tiles queries): every
 *     activation and GEMM weight is the unevaluated sum of two bfloat16 numbers, x ~ hi + lo with hi = bf16(x),
 *     lo = bf16(x - hi) (16 significand bits), and a multiply is three bf16 MFMAs (hi*hi + hi*lo + lo*hi, fp32 accumulate) -
 *     float32-grade results (north_star's 1e-3 against the reference's float32 CPU path) without the 16x slower fp32 MFMA.
 *     Layout: 8 consecutive channels (or 8 consecutive K of a weight row) are 32 bytes, [8 x bf16 hi][8 x bf16 lo]; a tensor
 *     therefore has the shape, pitch and byte size of its float32 counterpart.  Pointers 16-byte aligned, channel counts
 *     multiples of 8.  The kernel that PRODUCES a value splits it; ood_object_detection_amd/pairfmt.py packs weights.  The
 *     reference computes this path in float32 (effdet/anchors.py:136, efficientdet.py:895-933): dtype 0 and dtype 2 both
 *     stand for it, at different speeds;
 *   - return value: 0 on success, -22 (EINVAL) for a rejected argument, -5 (EIO) if the launch failed or if a kernel
 *     of an EARLIER call flagged a failure on the device (effdet_device_error);
 *   - thread-safe per stream; the only global mutable state is the device-side failure word.
 */
#ifndef EFFDET_HIP_H
#define EFFDET_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define EFFDET_F32 0
#define EFFDET_BF16 1
#define EFFDET_BF16X2 2          /* two-term bfloat16 values ("accurate" mode), see Conventions */
/* Padding convention of strided convolutions / pools (timm `padding=`, effdet config.pad_type; reference call sites
 * effdet/efficientdet.py:46,66,70,165): default TF-"SAME" (the tf_ model family); OR this flag into the `dtype` argument of an
 * inference entry point that pads (effdet_stem_*, effdet_mbconv_expand_dw[_gated], effdet_dwconv_bn_act, effdet_maxpool_same,
 * effdet_sepconv_fused) - or into the first selector argument of a training entry point that pads (`k` of effdet_train_dwconv_*,
 * `op` of effdet_train_spatial, `method` of effdet_train_fpn_combine / _wgrad, `idx` of effdet_train_fpn_input_bwd, `B` of
 * effdet_train_im2col_stem) - for pad_type='' (efficientdet_d0 / d1 on efficientnet_b0 / b1: static symmetric padding
 * ((s-1)+(k-1))/2).  Output sizes are the same; the window of a stride-2 layer on an even map starts one pixel earlier. */
#define EFFDET_PAD_SYMMETRIC (1 << 24)

/* ABI version of this header (bumped on any signature change). */
int effdet_abi_version(void);
/* Text of the HIP error behind the calling thread's most recent -5 return. */
const char* effdet_last_error(void);
/* Device-side failure word: kernels OR a bit into it when they detect a failure they cannot return (bit 0: a wave of the
 * wide MBConv kernel ran out of spins waiting for its workgroup's X-ring hand-off - its output tile is invalid).  The word is
 * host-coherent memory: no copy, no synchronisation.  While it is non-zero every launching entry point returns -5.
 * Returns the word; clear != 0 resets it.  (The reference has no counterpart: PyTorch raises from the failing op itself.) */
int effdet_device_error(int clear);

/* ---- backbone (timm EfficientNet; timm call sites effdet/efficientdet.py:17-18,837) ---------------- */

/* conv_stem 3x3/s2 TF-SAME + bn1 + SiLU.  X: NCHW [B,3,H,W] (in_dtype), Wt: [27][Cout] tap-major
 * (ky,kx,ci), Y: NHWC [B,ceil(H/2),ceil(W/2),Cout] (out_dtype). */
int effdet_stem_conv(void* stream, int in_dtype, int out_dtype, const void* X, const float* Wt,
                     const float* scale, const float* shift, void* Y, int B, int H, int W, int Cout);

/* Fused conv_stem + bn1 + SiLU -> blocks.0.0 depthwise 3x3/s1 + BN + SiLU (+ SE pool partials); the stem
 * output stays in LDS.  Wk: [C][32] im2col weights (dtype), k = (ky*3+kx)*3+ci zero-padded from 27 to 32;
 * taps [9][C] fp32; Y NHWC [B,ceil(H/2),ceil(W/2),C]; pool_partial [B][effdet_stem_dw_parts(dtype,H,W,C)][C]
 * or NULL: the launch writes exactly that many rows per image (bf16 with C = 32 and an even width takes the
 * rolling-window form, whose row count differs from the tile form's) - size AND sum the buffer by
 * effdet_stem_dw_parts, never by effdet_stem_dw_tiles_per_image (the float32 tile form's count, kept for
 * callers that run float32 only).  The same rule holds for effdet_mbconv_expand_dw[_gated]: its pool_partial
 * has effdet_mbconv[_gated]_tiles_per_image(dtype, ...) rows per image, which already answers for the form
 * the given dtype will run.  C <= 64. */
int effdet_stem_dw_fused(void* stream, int in_dtype, int dtype, const void* X, const void* Wk,
                         const float* s1, const float* t1, const float* taps, const float* s2, const float* t2,
                         void* Y, float* pool_partial, int B, int H, int W, int C);
int effdet_stem_dw_parts(int dtype, int H, int W, int C);      /* pool partial rows per image of the kernel that will run (bf16: rolling-window form) */
int effdet_stem_dw_tiles_per_image(int H, int W);

/* Input normalisation of the reference's PrefetchLoader (effdet/data/loader.py:114-128):
 * y = (float(x) - mean[c]) / std[c], x uint8 NCHW [B,C,hw], mean/std = 255 * the dataset constants
 * (host arrays of C <= 4 floats), y in out_dtype.  The *_u8 stem entry points apply the same arithmetic
 * (result rounded to the model dtype, exactly what feeding the normalised tensor would give) while loading
 * the input patch, so a uint8 batch needs no separate pass. */
/* ResizePad (effdet/data/transforms.py:75-107): Pillow's 8-bit BILINEAR resize of an HWC uint8 image [h][w][3] to
 * sw x sh, pasted top-left on an S x S canvas of fill_rgb (host array of 3 ints), written planar [3][S][S].  bounds_*
 * [out][2] = (first source index, count) and coef_* [out][ksize] are Pillow's integer coefficient tables (22 fractional
 * bits; device pointers, built by the caller as Pillow's precompute_coeffs / normalize_coeffs_8bpc do); workspace:
 * h*sw*3 bytes.  Bit-identical to Image.resize(.., BILINEAR) + paste. */
int effdet_resize_pad_u8(void* stream, const unsigned char* src, int h, int w, unsigned char* dst, int S, int sw, int sh,
                         const int* bounds_x, const int* coef_x, int ksize_x,
                         const int* bounds_y, const int* coef_y, int ksize_y,
                         const int* fill_rgb, unsigned char* workspace);
int effdet_normalize_u8(void* stream, int out_dtype, const unsigned char* X, const float* mean, const float* stdv,
                        void* Y, int B, int C, long long hw);
int effdet_stem_conv_u8(void* stream, int out_dtype, const unsigned char* X, const float* mean, const float* stdv,
                        const float* Wt, const float* scale, const float* shift, void* Y, int B, int H, int W, int Cout);
int effdet_stem_dw_fused_u8(void* stream, int dtype, const unsigned char* X, const float* mean, const float* stdv,
                            const void* Wk, const float* s1, const float* t1, const float* taps, const float* s2, const float* t2,
                            void* Y, float* pool_partial, int B, int H, int W, int C);

/* 1x1 conv as GEMM with folded BN / bias, optional SiLU (act=1) or ReLU (act=2), optional SE gate on A
 * (gate [B,K] fp32, rows_per_image = H*W), optional residual [M,N].  A: [M,K], W: [N,K].
 * Output row m goes to C + (m / rows_per_image) * c_image_stride + (m % rows_per_image) * ldc
 * (pass 0 for rows_per_image / c_image_stride / ldc to get a plain [M,N] matrix).
 * Replaces conv_pw / conv_pwl + BatchNorm2d (+Swish) of timm's MBConv blocks and the BiFPN lateral
 * ConvBnAct2d (effdet/efficientdet.py:42-57, :155-158). */
int effdet_pw_gemm_bn_act(void* stream, int dtype, const void* A, long long M, int K, const void* W, int N,
                          const float* scale, const float* shift, int act, const void* residual,
                          const float* gate, int rows_per_image,
                          void* C, long long c_image_stride, long long ldc);
/* n <= 8 independent GEMMs of this form (no gate, no residual, dense [M][N] outputs) in ONE launch; every N must select the
 * same output-channel tile (N <= 96).  The BiFPN's lateral 1x1 convs of the backbone features (ResampleFeatureMap.conv,
 * effdet/efficientdet.py:155-158): six small problems that each fill a fraction of the chip. */
int effdet_pw_gemm_group(void* stream, int dtype, int n, const void* const* A, const long long* M, const int* K,
                         const void* const* W, const int* N, const float* const* scale, const float* const* shift,
                         int act, void* const* C);

/* depthwise k x k (k = 3|5, stride 1|2, TF-SAME) + folded BN + SiLU.  Wt: [k*k][C] fp32.
 * If pool_partial != NULL it receives [B][effdet_dwconv_blocks_per_image(Ho,Wo,C)][C] partial sums of
 * the output for the squeeze-excite average. */
int effdet_dwconv_bn_act(void* stream, int dtype, const void* X, void* Y, const float* Wt,
                         const float* scale, const float* shift, int act, float* pool_partial,
                         int B, int H, int W, int C, int k, int stride);
int effdet_dwconv_blocks_per_image(int Ho, int Wo, int C);

/* Fused front half of an inverted-residual block: expand 1x1 (MFMA) + BN + SiLU -> depthwise k x k
 * (TF-SAME, stride 1|2) + BN + SiLU; the expanded activation never leaves LDS.  W1: [mid][Cin] (dtype),
 * taps: [k*k][mid] fp32.  pool_partial (optional): [B][effdet_mbconv_tiles_per_image(...)][mid] partial
 * sums for the SE average.  Early stages run as spatial tiles, late stages (wide inputs, small maps) as
 * (row band x 64-channel slice) workgroups; the choice is internal.  Replaces conv_pw/bn1/act1/conv_dw/bn2/act2 of timm's InvertedResidual. */
int effdet_mbconv_expand_dw(void* stream, int dtype, const void* X, void* Y, const void* W1,
                            const float* s1, const float* t1, const float* taps,
                            const float* s2, const float* t2, float* pool_partial,
                            int B, int H, int W, int Cin, int mid, int k, int stride);
int effdet_mbconv_tiles_per_image(int dtype, int H, int W, int Cin, int mid, int k, int stride);
/* The same with X multiplied by a per-image channel gate [B][Cin] (rounded to dtype) while it is loaded.  Lets a
 * squeeze-excited, residual-free block hand its depthwise output straight to the next block: its project conv +
 * BN (both linear) are folded into W1 / t1 by the caller, so the narrow tensor in between is never written.
 * Always the spatial-tile geometry: size pool_partial with effdet_mbconv_gated_tiles_per_image. */
int effdet_mbconv_expand_dw_gated(void* stream, int dtype, const void* X, const float* in_gate, void* Y, const void* W1,
                                  const float* s1, const float* t1, const float* taps,
                                  const float* s2, const float* t2, float* pool_partial,
                                  int B, int H, int W, int Cin, int mid, int k, int stride);
int effdet_mbconv_gated_tiles_per_image(int dtype, int H, int W, int Cin, int mid, int k, int stride);

/* SqueezeExcite gate: mean -> fc(C->R)+SiLU -> fc(R->C) -> sigmoid.  W1: [R][C] (conv_reduce weight),
 * W2t: [R][C] (conv_expand weight TRANSPOSED, so that consecutive threads read consecutive channels). */
int effdet_se_gate(void* stream, const float* partial, int nblk, int hw, const float* W1, const float* b1,
                   const float* W2t, const float* b2, float* gate, int B, int C, int R);

/* ---- BiFPN + heads (effdet/efficientdet.py:140-469) ---------------------------------------------- */

/* create_pool2d('max', 3, 2, 'same') (effdet/efficientdet.py:165-166): P6, P7.  image strides in
 * elements (0 = dense). */
int effdet_maxpool_same(void* stream, int dtype, const void* X, long long x_image_stride,
                        void* Y, long long y_image_stride, int B, int H, int W, int C);

/* Fused FpnCombine -> Swish -> SeparableConv2d -> BN for one BiFPN node (nlevels = 1,
 * effdet/efficientdet.py:224-245, :281-292), or one HeadNet layer over all pyramid levels
 * (nlevels <= 5, effdet/efficientdet.py:438-452) with the per-anchor OOD energy / max-logit epilogue
 * when ood_classes > 0.  All array arguments are host: memory.
 *   level_hw[l] = {H,W};  per (level, input): pointer, image stride (elements), {H,W}, mode
 *   (0 same size, 1 nearest x2 upsample, 2 max-pool 3x3/s2 SAME);
 *   fuse_mode 0: single input, 1: sum_i (x_i*w_i)/den ('fastattn'), 2: sum_i x_i*w_i ('attn','sum');
 *   dw_w [9][F] fp32; pw_w [N][F] (dtype); scale/shift [rows][N] fp32 indexed by affine_row[l]
 *   (scale may be NULL); out_ptr[l] + b*out_image_stride[l] + (y*W+x)*N.
 *   dtype 3 (= 1 | 2): bfloat16 inputs / weights, float32 OUTPUTS written straight from the accumulators (out pointers and
 *   strides then address float32 elements) - used for the box regressions, which decode reads as float32 (anchors.py:136).
 *   dtype 6 (= 2 | 4): two-term bf16 inputs / weights, plain float32 OUTPUTS (the accurate mode's class logits and box
 *   regressions: what _post_process and decode read). */
int effdet_sepconv_fused(void* stream, int dtype, int B, int nlevels, const int* level_hw, int n_in,
                         const void* const* in_ptr, const long long* in_image_stride, const int* in_hw,
                         const int* in_mode, int fuse_mode, const float* fuse_w, float fuse_den, int pre_act,
                         const float* dw_w, const void* pw_w, const float* scale, const float* shift,
                         const int* affine_row, int post_act, int F, int N,
                         void* const* out_ptr, const long long* out_image_stride,
                         int ood_classes, int num_anchors, float* ood_energy, float* ood_maxlogit,
                         long long ood_image_stride, const long long* ood_level_off);

/* MetaHead (effdet/efficientdet.py:569-695): the fork's functional class head - SeparableConv layers shared by the
 * levels, each followed by F.batch_norm(training=True) (statistics of the current batch, per level) and Swish.
 * effdet_sepconv_meta runs one layer for all levels: the previous layer's batch-norm arrives folded into
 * in_scale / in_shift [rows][F] (row in_affine_row[level]; NULL for the first layer) applied before the SiLU (pre_act),
 * the conv output + bias is written raw and its per-channel sums / sums of squares go to stat_partial
 * [B][effdet_sepconv_tiles][2][N] (NULL: not needed).  dw_out (optional, per level [B, H*W, F]) receives the
 * depthwise output (`x_pred`, what ret_activs returns).  effdet_bn_batch_stats turns the partial sums into the next
 * layer's scale / shift [nlevels][N]: scale = w / sqrt(var_biased + eps), shift = b - mean * scale with
 * weight / bias [rows][N] (row param_row[level]). */
int effdet_sepconv_meta(void* stream, int dtype, int B, int nlevels, const int* level_hw,
                        const void* const* in_ptr, const long long* in_image_stride,
                        const float* in_scale, const float* in_shift, const int* in_affine_row, int pre_act,
                        const float* dw_w, const void* pw_w, const float* bias, int F, int N,
                        void* const* out_ptr, const long long* out_image_stride,
                        float* stat_partial, void* const* dw_out, const long long* dw_out_image_stride);
int effdet_sepconv_tiles(int dtype, int nlevels, const int* level_hw, int* level_tile_begin);
int effdet_bn_batch_stats(void* stream, int dtype, const float* partial, int B, int nlevels, const int* level_hw, int N,
                          const float* weight, const float* bias, const int* param_row, float eps,
                          float* out_scale, float* out_shift);

/* ---- post-processing ------------------------------------------------------------------------------ */

/* _post_process (effdet/bench.py:12-56).  cls_all [B, n_anchors, C], box_all [B, n_anchors, 4] (dtype);
 * outputs out_cls [B,k], out_box [B,k,4] (dtype), out_indices / out_classes [B,k] int64.
 * Descending by logit, ties by lower flat index.  k <= 16384.
 * anchor_max: optional [B, n_anchors] fp32 maximum logit of every anchor (the class head's OOD max-logit
 * output); lets the select skip the anchors that cannot reach the top k.  NULL: computed internally. */
long long effdet_topk_workspace_bytes(int B, long long n_anchors);
int effdet_topk_select(void* stream, int dtype, const void* cls_all, const float* anchor_max, int B, long long n_anchors, int C,
                       const void* box_all, int k, void* out_cls, void* out_box,
                       long long* out_indices, long long* out_classes, void* workspace, long long workspace_bytes);

/* generate_detections, first half (effdet/anchors.py:132-144): decode, clip (when img_scale and img_size
 * are given), sigmoid, keep score > 0.01 (order kept).  Outputs are [B,k(,4)] with count[b] valid rows.
 * dtype 0: logits and box regressions float32; 1: both bfloat16; 3 (= 1 | 2): bfloat16 logits, float32 box regressions
 * (what a bfloat16 model's box head writes, see effdet_sepconv_fused). */
int effdet_decode_threshold(void* stream, int dtype, const void* cls_topk, const void* box_topk,
                            const float* anchors, const long long* indices, const long long* classes,
                            const float* img_scale, const float* img_size, int B, int k,
                            float* boxes, float* scores, int* classes_out, int* src, int* count, float* maxcoord);
/* The same with the box regressions taken from the box head's full output box_all [B, n_anchors, 4] through `indices`
 * (effdet_topk_select may then be called with box_all = NULL and run concurrently with the box head). */
int effdet_decode_threshold_gather(void* stream, int dtype, const void* cls_topk, const void* box_all, long long n_anchors,
                                   const float* anchors, const long long* indices, const long long* classes,
                                   const float* img_scale, const float* img_size, int B, int k,
                                   float* boxes, float* scores, int* classes_out, int* src, int* count, float* maxcoord);

/* generate_detections, second half (effdet/anchors.py:145-166).  det [B,max_det,6] zero padded rows
 * x1,y1,x2,y2,score,class+1; det_count [B]; keep_src [B,max_det] = position in the top-k list or -1.
 * Hard: torchvision batched_nms semantics (anchors.py:150).  Soft: effdet/soft_nms.py:42-169. */
int effdet_nms_hard(void* stream, const float* boxes, const float* scores, const int* classes, const int* src,
                    const int* count, const float* maxcoord, int B, int k, double iou_threshold, int max_det,
                    const float* img_scale, float* det, int* det_count, int* keep_src);

/* effdet_decode_threshold[_gather] followed by effdet_nms_hard in ONE launch (the hard-NMS path of generate_detections,
 * effdet/anchors.py:132-166): n_anchors > 0 reads the box regressions from the head's [B, n_anchors, 4] output through `indices`
 * (0: `box` is the gathered [B, k, 4] tensor); img_scale_clip / img_size as in effdet_decode_threshold (clipping, both or neither),
 * img_scale_out as effdet_nms_hard's img_scale (output scaling).  The compacted intermediate arrays are still written. */
int effdet_detections_hard(void* stream, int dtype, const void* cls_topk, const void* box, long long n_anchors,
                           const float* anchors, const long long* indices, const long long* classes,
                           const float* img_scale_clip, const float* img_size, int B, int k,
                           float* boxes, float* scores, int* classes_out, int* src, int* count, float* maxcoord,
                           double iou_threshold, int max_det, const float* img_scale_out,
                           float* det, int* det_count, int* keep_src);

int effdet_nms_soft(void* stream, const float* boxes, const float* scores, const int* classes, const int* src,
                    const int* count, const float* maxcoord, int B, int k, int method_gaussian, float sigma,
                    float iou_threshold, float score_threshold, int max_det,
                    const float* img_scale, float* det, int* det_count, int* keep_src);

/* The same soft-NMS for any k (the stand-alone soft_nms / batched_soft_nms API of effdet/soft_nms.py:42-169 has no size
 * limit and returns every pick): working scores in `score_scratch` [B,k] floats, max_det <= k. effdet_nms_soft itself
 * keeps the candidates in registers (k <= 8192, max_det <= k). */
int effdet_nms_soft_large(void* stream, const float* boxes, const float* scores, const int* classes, const int* src,
                          const int* count, const float* maxcoord, int B, int k, int method_gaussian, float sigma,
                          float iou_threshold, float score_threshold, int max_det,
                          const float* img_scale, float* det, int* det_count, int* keep_src, float* score_scratch);

/* OOD scores of the kept detections: energy/maxlogit [B,n_anchors] -> [B,max_det] (0 where padded); out_anchor
 * (optional) receives the anchor index each kept detection came from (-1 where padded). */
int effdet_gather_ood(void* stream, const int* keep_src, const long long* indices, const float* energy,
                      const float* maxlogit, long long n_anchors, int B, int k, int max_det,
                      float* out_energy, float* out_maxlogit, long long* out_anchor);

/* ---- training-side operators -------------------------------------------------- */

/* loss_fn (effdet/loss.py:224-298): alpha-weighted BCE-with-logits on one-hot targets (gamma unused, as in the
 * fork), label smoothing, `target != -2` mask, Huber box loss on targets != 0, normaliser sum(num_positives)+1.
 * cls [B,N,C], box [B,N,4] (dtype), cls_t [B,N] int64 (class index, -1 background, -2 ignore), box_t [B,N,4] fp32.
 * out3 = {total, class_loss, box_loss}; grad_cls / grad_box (dtype, optional) = d total / d head outputs. */
long long effdet_detection_loss_workspace_floats(int B, long long N, int C);
int effdet_detection_loss(void* stream, int dtype, const void* cls, const void* box, const long long* cls_t,
                          const float* box_t, const float* num_positives, int B, long long N, int C,
                          float alpha, float delta, float box_loss_weight, float label_smoothing,
                          float* out3, void* grad_cls, void* grad_box, float* workspace, long long workspace_floats);

/* AnchorLabeler.batch_label_anchors (effdet/anchors.py:384-438): IoU (region_similarity_calculator.py:24-73),
 * ArgMaxMatcher thresholds (t, t) with force_match_for_each_row (argmax_matcher.py:116-146), class target =
 * label - 1 (background -1), FasterRcnnBoxCoder.encode (box_coder.py:81-110).  gt_boxes [B,Mmax,4] yxyx,
 * gt_cls [B,Mmax] int64 (rows with class <= -1 are padding / filtered), anchors [N,4] yxyx.
 * Outputs cls_t [B,N] int64, box_t [B,N,4], num_positives [B], match [B,N] int64 (optional; index into the
 * valid rows, -1 unmatched).  Mmax <= 512. */
long long effdet_label_anchors_workspace_bytes(int B, int Mmax, long long N);
int effdet_label_anchors(void* stream, const float* anchors, const float* gt_boxes, const long long* gt_cls,
                         int B, int Mmax, long long N, float match_threshold, long long* cls_t, float* box_t,
                         float* num_positives, long long* match, void* workspace, long long workspace_bytes);

/* ProjectionNet.weighted_median (effdet/efficientdet.py:748-760): per column of embds [n][d] (n <= 1024), the value at
 * which the cumulative confidence (in ascending value order) first reaches half of sum(confs); conf_sum[0] = that sum. */
int effdet_weighted_median(void* stream, const float* embds, const float* confs, int n, int d, float* med, float* conf_sum);

/* ---- network backward of the pretrain step (pretrain.py:226-236 `qry_loss.backward()`) ------------
 * float32, NHWC.  Generic operators the host sequences into forward-with-saved-activations and backward
 * (ood_object_detection_amd/train_engine.py).  "Row maps": row m of a matrix lives at
 * base + (m / rpi) * img_stride + (m % rpi) * ld  (pass rpi = img_stride = 0 for a dense matrix; ld = 0 -> #columns):
 * lets the class / box predict layers read and write the packed [B, N, C] head outputs per pyramid level.
 * Every reduction is two-stage in a fixed order (bitwise reproducible gradients); workspaces are caller-owned. */

/* C[M,N] (+)= A[M,K] W[N,K]^T + bias[N] (bias may be NULL).  1x1-conv forward (replaces the nn.Conv2d(1x1) calls of
 * timm's blocks and effdet/efficientdet.py:42-83 in training), and its input gradient dX = dY W when W is passed
 * transposed.  C2 (optional, addressed like C) receives silu(C): the pre-activation and the activation in one pass. */
int effdet_train_gemm_nt(void* stream, const float* A, long long a_rpi, long long a_img_stride, long long a_ld,
                         const float* W, const float* bias, float* C, long long c_rpi, long long c_img_stride,
                         long long c_ld, long long M, int K, int N, int accumulate, float* C2);
/* dY[M,N]^T [X[M,K] | 1] in one pass: out = the 1x1-conv weight gradient as a dense [N][K] matrix followed by the N sums
 * sum_m dY[m,n] (bias / BN shift gradient); out holds N*K + N floats (autograd of conv2d 1x1). */
long long effdet_train_gemm_tn_workspace_floats(long long M, int N, int K);
int effdet_train_gemm_tn(void* stream, const float* dY, long long y_rpi, long long y_img_stride, long long y_ld,
                         const float* X, long long x_rpi, long long x_img_stride, long long x_ld,
                         long long M, int N, int K, float* out, float* workspace, long long workspace_floats);
/* out[g][l] (+)= sum_s in[g][s][l], s ascending. */
int effdet_train_reduce_mid(void* stream, const float* in, int G, int S, long long L, float* out, int accumulate);
/* depthwise k x k (k = 3|5, stride 1|2, TF-SAME) backward: dX [B,H,W,C] from dY [B,Ho,Wo,C] and taps [k*k][C];
 * out [(k*k+1)][C] = tap gradients (rows 0..k*k-1) and sum of dY (last row); cmajor != 0: the tap gradients in the
 * parameter's layout [C][k*k], followed by the C sums. */
int effdet_train_dwconv_bwd_dx(void* stream, const float* dY, const float* taps, float* dX,
                               int B, int H, int W, int C, int k, int stride);
long long effdet_train_dwconv_bwd_dw_workspace_floats(int B, int H, int W, int C, int k, int stride);
int effdet_train_dwconv_bwd_dw(void* stream, const float* dY, const float* X, float* out,
                               int B, int H, int W, int C, int k, int stride, float* workspace, long long workspace_floats, int cmajor);
/* Element-wise family over n floats (n, C multiples of 4; channel = i % C, image = i / (hw*C)):
 *  0 silu(a)   1 b*silu'(a)   2 a+b   3 a*v0[c]+v1[c] (v1 optional)   4 a*v0[img,c]   5 a*v0[img,c]+v1[img,c]*s0
 *  6 v0[c]*(a - v1[c] - (b - v2[c])*v3[c])  (batch-statistics BN backward)
 *  7 (a*s0)/s3 + (b*s1)/s3 [+ (c*s2)/s3]  (FpnCombine 'fastattn', effdet/efficientdet.py:240-242)   8 a*s0
 *  9 a*s0 + b*s1 [+ c*s2]   10 a*b   11 a*b*silu''(c)  (double backward of SiLU: the MetaHead's second-order MAML terms)
 * sdev (optional): device float[4] that replaces s0..s3 at run time, so that a captured hipGraph of the training step
 * sees the current BiFPN edge weights.  out2 (optional): silu(out), written in the same pass. */
int effdet_train_ew(void* stream, int op, float* out, const float* a, const float* b, const float* c,
                    const float* v0, const float* v1, const float* v2, const float* v3,
                    float s0, float s1, float s2, float s3, long long n, int C, long long hw, const float* sdev, float* out2);
/* Per-channel reductions over the rows of dense [G][R][C] tensors -> out [G][C]:
 * mode 0 sum a; 1 sum a*b; 2 sum (a - v[c])^2; 3 sum a*(b - v[c]); 4: modes 0 and 3 in one pass, out [G][2][C];
 * every result is multiplied by alpha (1/M gives means). */
long long effdet_train_col_reduce_workspace_floats(int G, long long R, int C);
int effdet_train_col_reduce(void* stream, int mode, const float* a, const float* b, const float* v,
                            int G, long long R, int C, float* out, float* workspace, long long workspace_floats, float alpha);
/* op 0: nearest x2 upsample in [B,H,W,C] -> out [B,2H,2W,C];  op 1: its backward, in = d out [B,2H,2W,C] -> [B,H,W,C];
 * op 2: 3x3/s2 TF-SAME max-pool backward, in = pool input [B,H,W,C], aux = dY [B,ceil(H/2),ceil(W/2),C] -> dX
 * (gradient goes to the first maximum of a window in row-major order, as torch's max_pool2d does). */
int effdet_train_spatial(void* stream, int op, const float* in, const float* aux, float* out, int B, int H, int W, int C);
/* conv_stem patches: X NCHW [B,3,H,W] fp32 -> col [B*ceil(H/2)*ceil(W/2)][32], k = (ky*3+kx)*3+ci, columns 27..31 zero. */
int effdet_train_im2col_stem(void* stream, const float* X, float* col, int B, int H, int W);
/* SqueezeExcite backward: pool_sum / gate / dgate [B,C], W1 [R][C], W2t [R][C] (as effdet_se_gate) -> ds [B,C]
 * (gradient w.r.t. the pooled mean) and per-image parameter gradients pgrad [B][R*C + R + R*C + C] =
 * {d conv_reduce.weight, d conv_reduce.bias, d conv_expand.weight^T, d conv_expand.bias}. */
int effdet_train_se_bwd(void* stream, const float* pool_sum, int hw, const float* gate, const float* dgate,
                        const float* W1, const float* b1, const float* W2t, float* ds, float* pgrad, int B, int C, int R);

/* Parameter-sized helpers of conv + BatchNorm on running statistics (one launch each instead of a dozen tensor ops):
 * fold: scale = gamma / sqrt(var + eps), shift = beta - mean * scale, rstd; Wf [N][K] = W * scale[n], WfT [K][N] = Wf^T,
 * WT [K][N] = W^T (each matrix optional).  grads: from dWext = effdet_train_gemm_tn's output ([N][K] then N sums; or the
 * depthwise [(K+1)][N] layout when transposed != 0): dW [N][K] = scale * dWraw, d gamma = rstd * (sum_k W * dWraw - mean * dsum),
 * d beta = dsum. */
int effdet_train_fold_bn(void* stream, const float* W, int N, int K, const float* gamma, const float* beta,
                         const float* mean, const float* var, float eps,
                         float* Wf, float* WfT, float* WT, float* scale, float* shift, float* rstd);
int effdet_train_convbn_grads(void* stream, const float* dWext, int N, int K, int transposed, const float* W,
                              const float* scale, const float* rstd, const float* mean,
                              float* dW, float* dgamma, float* dbeta);

/* nn.BatchNorm2d bookkeeping of the layers that are not folded (BiFPN / heads), one launch each.  finalize: from the batch
 * (train != 0: running stats updated with `momentum`, running_var with var * unbias, num_batches_tracked += 1) or running
 * (train == 0; pass them as mean / var) statistics -> rstd, scale = gamma * rstd, shift = beta - mean * scale.
 * bwd_prep: s1 = sum(dy), s2c = sum(dy (c - mean)) -> d gamma, d beta and the two vectors v1 = s1/M, v3 = rstd^2 s2c/M that
 * op 6 of effdet_train_ew needs. */
int effdet_train_bn_finalize(void* stream, const float* mean, const float* var, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, long long* num_batches_tracked, int C, int train,
                             float momentum, float unbias, float eps, float* scale, float* shift, float* rstd);
int effdet_train_bn_bwd_prep(void* stream, const float* s1, const float* s2c, const float* rstd, int C, float inv_m,
                             float* dgamma, float* dbeta, float* v1, float* v3);
/* The two halves above fused with the column reductions in front of them (one launch less per BatchNorm and direction):
 * bn_var_finalize = effdet_train_col_reduce mode 2 over a [R][C] with the batch mean, then finalize in training mode;
 * bn_bwd_sums     = effdet_train_col_reduce mode 4 (sum dy, sum dy (c - mean)), then bwd_prep -> out [4][C] = d gamma, d beta, v1, v3.
 * workspace: effdet_train_col_reduce_workspace_floats(1, R, C) floats. */
int effdet_train_bn_var_finalize(void* stream, const float* a, const float* mean, long long R, int C, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, long long* num_batches_tracked,
                                 float momentum, float unbias, float eps, float* scale, float* shift, float* rstd,
                                 float* workspace, long long workspace_floats);
int effdet_train_bn_bwd_sums(void* stream, const float* dy, const float* c, const float* mean, const float* rstd,
                             long long R, int C, float* out, float* workspace, long long workspace_floats);

/* Table-driven forms of effdet_train_fold_bn / plain transposes and of effdet_train_convbn_grads: one launch for all convs of
 * a stage.  `table` = n records in DEVICE memory:
 *   prep  { int kind, rows, cols; float eps; const float *src, *gamma, *beta, *mean, *var; float *dst0, *dst1, *dst2, *scale,
 *           *shift, *rstd; }   kind 0: dst0 [cols][rows] = src [rows][cols] transposed; kind 1: fold_bn with W = src [N = rows][K = cols],
 *           dst0 = Wf, dst1 = WfT, dst2 = WT (each optional); kind 2: effdet_train_fpn_weights with src = edge_weights [rows],
 *           method = (int) eps ('fastattn' 0 | 'attn' 1), dst0 = {w0, w1, w2, den}
 *   grads { const float *dWext, *W, *scale, *rstd, *mean; float *dW, *dgamma, *dbeta; int N, K, transposed, pad; } */
int effdet_train_prep_table(void* stream, const void* table, int n, long long max_elems);
int effdet_train_grads_table(void* stream, const void* table, int n, int max_n);

/* ---- fused forms of one MBConv block of the backbone (timm InvertedResidual / DepthwiseSeparableConv, BN in eval mode) --------
 * train_dwconv_fwd:     conv_dw + folded BN: Z = pre-activation (kept for the backward), A = silu(Z) (optional) and the SE pool partial
 *                       rows of A ([B][effdet_train_dwconv_fwd_parts(H,W,C,k,stride)][C], optional) in one pass
 * train_se_gate:        effdet_se_gate that also writes the pooled sums [B][C] effdet_train_se_bwd reads
 * train_gemm_nt_fused:  C = (A * a_scale[row / a_scale_rows]) W^T + bias + R, C2 = silu(C): the SE gate [B][K] applied while A
 *                       is loaded, the shortcut R [M][N] added in the epilogue (a_scale, bias, R, C2 optional; dense rows)
 * train_gemm_tn_scaled: out = dY^T [X * x_scale[row / x_scale_rows] | 1] (weight gradient of the gated project conv)
 * train_dwconv_bwd_dx_silu: effdet_train_dwconv_bwd_dx followed by * silu'(Z) in the same pass
 * (effdet_train_ew op 12: (a * v0[img][c] + v1[img][c] * s0) * silu'(c) - SE gate backward and SiLU backward in one pass) */
int effdet_train_dwconv_fwd_parts(int H, int W, int C, int k, int stride);
int effdet_train_dwconv_fwd(void* stream, const float* X, float* Z, float* A, const float* Wt, const float* scale,
                            const float* shift, float* pool_partial, int B, int H, int W, int C, int k, int stride);
int effdet_train_se_gate(void* stream, const float* partial, int nblk, int hw, const float* W1, const float* b1,
                         const float* W2t, const float* b2, float* gate, float* pool_sum, int B, int C, int R);
int effdet_train_gemm_nt_fused(void* stream, const float* A, const float* a_scale, long long a_scale_rows, const float* W,
                               const float* bias, const float* R, float* C, float* C2, long long M, int K, int N);
int effdet_train_gemm_tn_scaled(void* stream, const float* dY, const float* X, const float* x_scale, long long x_scale_rows,
                                long long M, int N, int K, float* out, float* workspace, long long workspace_floats);
int effdet_train_dwconv_bwd_dx_silu(void* stream, const float* dY, const float* taps, const float* Z, float* dX,
                                    int B, int H, int W, int C, int k, int stride);

/* ---- the head towers over the whole pyramid in one launch per layer (effdet/efficientdet.py:438-452: conv weights shared by
 * the levels, BatchNorm per level).  "Packed pyramid" = float32 [sum_l B*Hs[l]*Ws[l]][C], level-major: the NHWC tensors of the L
 * levels (L <= 8) one behind the other.  A 1x1 conv over it is ONE GEMM: effdet_train_gemm_nt_levels / _tn_levels are
 * effdet_train_gemm_nt / _tn over those rows, where the side flagged `*_packed` is instead the image-major head tensor of
 * effdet/loss.py's packing, [B][sum_l Hs[l]*Ws[l]][pk_ld] with pk_img_stride floats per image (a_packed and c_packed exclude each
 * other; C2 needs a dense C).  Reductions are two-stage in a fixed order (bitwise reproducible); `workspace`:
 * effdet_train_levels_workspace_floats floats.
 *   levels_dw          depthwise 3x3 / s1 / TF-SAME, taps [9][C]; flip != 0: taps mirrored = gradient w.r.t. the input
 *   levels_dw_bwd_dw   out [9][C] (cmajor != 0: [C][9], the parameter's layout) = gradient of the (shared) taps, summed over all levels
 *   levels_col_reduce  out [L][C] (mode 0: sum a; mode 2: sum (a - v[l][c] * vscale[l])^2) or [L][2][C] (mode 4: sum a',
 *                      sum a' (b - v[l][c]), a' = a * silu'(pre) when pre is given - the SiLU backward of the layer above)
 *   levels_bn_finalize nn.BatchNorm2d bookkeeping of the L layers (pointer tables of their parameters / buffers; train[l] != 0:
 *                      batch statistics mean = sum * inv_m, var = sq * inv_m, running stats updated; else running statistics)
 *                      -> mean, scale = gamma * rstd, shift = beta - mean * scale, rstd, all [L][C]
 *   levels_bn_bwd_prep sums [L][2][C] of mode 4 -> d gamma, d beta, v1, v3 ([L][C]; as effdet_train_bn_bwd_prep)
 *   levels_ew          op 3: out = a * v0[l][c] + v1[l][c] (out2 = silu(out) when given); op 6: BatchNorm backward
 *                      v0 (a' - v1 - (b - v2) v3) for levels with train[l] != 0, a' v0 otherwise; a' as in mode 4 */
int effdet_train_gemm_nt_levels(void* stream, const float* A, int a_packed, const float* W, const float* bias, float* C,
                                int c_packed, int B, int L, const int* Hs, const int* Ws, long long pk_img_stride,
                                long long pk_ld, int K, int N, float* C2);
int effdet_train_gemm_tn_levels(void* stream, const float* dY, int y_packed, const float* X, int B, int L, const int* Hs,
                                const int* Ws, long long pk_img_stride, long long pk_ld, int N, int K, float* out,
                                float* workspace, long long workspace_floats);
long long effdet_train_levels_workspace_floats(int B, int L, const int* Hs, const int* Ws, int C);
int effdet_train_levels_dw(void* stream, const float* X, const float* taps, float* Y, int B, int L, const int* Hs,
                           const int* Ws, int C, int flip);
int effdet_train_levels_dw_bwd_dw(void* stream, const float* dY, const float* X, float* out, int B, int L,
                                  const int* Hs, const int* Ws, int C, float* workspace, long long workspace_floats,
                                  int cmajor);
int effdet_train_levels_col_reduce(void* stream, int mode, const float* a, const float* b, const float* v,
                                   const float* pre, const float* vscale, int B, int L, const int* Hs, const int* Ws, int C,
                                   float* out, float* workspace, long long workspace_floats);
int effdet_train_levels_bn_finalize(void* stream, const float* sum, const float* sq, int L, int C, const void* const* gamma,
                                    const void* const* beta, void* const* running_mean, void* const* running_var,
                                    void* const* num_batches_tracked, const int* train, const float* inv_m,
                                    const float* unbias, const float* momentum, const float* eps,
                                    float* mean, float* scale, float* shift, float* rstd);
int effdet_train_levels_bn_bwd_prep(void* stream, const float* sums, const float* rstd, const float* inv_m, int L, int C,
                                    float* dgamma, float* dbeta, float* v1, float* v3);
int effdet_train_levels_ew(void* stream, int op, float* out, float* out2, const float* a, const float* b, const float* pre,
                           const float* v0, const float* v1, const float* v2, const float* v3, const int* train,
                           int B, int L, const int* Hs, const int* Ws, int C);

/* ---- one BiFPN node's FpnCombine without materialising the resampled inputs (effdet/efficientdet.py:180-245).  The n (2 or 3)
 * source tensors are float32 NHWC [B][hs[i]][ws[i]][C]; each is the node's size (identity), half of it (nearest x2 upsample) or
 * about twice it (3x3 / s2 TF-SAME max-pool) - told from the shapes.  method: 0 'fastattn', 1 'attn', 2 'sum'.
 *   fpn_weights   edge_weights -> wdev = {w0, w1, w2, den}: relu and den = sum + 1e-4 | softmax, den 1 | ones, den 1
 *   fpn_combine   fused = sum_i (R_i(x_i) * w_i) / den ('fastattn' arithmetic; else sum_i R_i(x_i) * w_i), act = silu(fused)
 *   fpn_wgrad     dots [n][C] = sum over pixels of dfused * R_i(x_i), dfused = dact * silu'(fused); grad [n] = d edge_weights
 *                 (workspace: effdet_train_fpn_dots_workspace_floats floats; method 2: dots only)
 *   fpn_input_bwd out = (w_idx / den) * R_idx^T(dfused) + acc (acc optional): gradient of source tensor idx, [B][h][w][C] */
int effdet_train_fpn_weights(void* stream, const float* edge_weights, int n, int method, float* wdev);
int effdet_train_fpn_combine(void* stream, int n, const void* const* srcs, const int* hs, const int* ws, int method,
                             const float* wdev, float* fused, float* act, int B, int H, int W, int C);
long long effdet_train_fpn_dots_workspace_floats(int B, int H, int W, int C);
int effdet_train_fpn_wgrad(void* stream, int n, const void* const* srcs, const int* hs, const int* ws, int method,
                           const float* wdev, const float* edge_weights, const float* dact, const float* fused,
                           float* dots, float* grad, int B, int H, int W, int C, float* workspace, long long workspace_floats);
int effdet_train_fpn_input_bwd(void* stream, int idx, const float* src, int h, int w, const float* wdev, const float* dact,
                               const float* fused, const float* acc, float* out, int B, int H, int W, int C);

/* ---- optimizer half of the pretrain step (pretrain.py:272-276) ------------------------------------ */

/* torch.nn.utils.clip_grad_norm_(params, max_norm) + torch.optim.Adam.step() on flat float32 buffers.
 * effdet_sqnorm: out[0] (+)= sum(g^2) (two-stage, fixed order; workspace of effdet_sqnorm_workspace_floats floats).
 * effdet_adam_clip_step: g' = g * min(1, max_norm / (sqrt(*sqnorm) + 1e-6)) (sqnorm NULL: no clipping), then Adam
 * in torch's operation order with bias corrections for `step` (1-based). */
long long effdet_sqnorm_workspace_floats(void);
int effdet_sqnorm(void* stream, const float* g, long long n, float* workspace, float* out, int accumulate);
int effdet_adam_clip_step(void* stream, float* p, const float* g, float* m, float* v, long long n,
                          float lr, float beta1, float beta2, float eps, int step, float max_norm, const float* sqnorm);
/* The same with the two bias corrections read from device memory (bc_dev[0] = 1 - beta1^step, bc_dev[1] = sqrt(1 - beta2^step)):
 * the launch can sit in a captured hipGraph while the host refreshes bc_dev before every replay. */
int effdet_adam_clip_step_dev(void* stream, float* p, const float* g, float* m, float* v, long long n,
                              float lr, float beta1, float beta2, float eps, const float* bc_dev, float max_norm, const float* sqnorm);

/* ---- detection evaluation (the effdet/evaluation package, driven by pretrain.py:246-252) ----------------------- */

/* Per-image greedy matching (per_image_evaluation.py:377-405) + CorLoc (:143-176).  det [B,max_det,6] rows
 * x1,y1,x2,y2,score,class (1-based) in descending score order, det_count [B]; gt_boxes [B,M,4] yxyx, gt_cls [B,M]
 * 1-based (<= 0: padding).  tp [B,max_det]: 1 true positive, 0 false positive, -1 dropped (padding row, box with
 * ymax <= ymin or xmax <= xmin, class out of range).  gt_count / gt_imgs / correct_imgs [num_classes] int32 are
 * accumulated (+=): ground-truth instances, images containing the class, images whose top-scoring detection of the
 * class is correctly localised. */
int effdet_eval_match(void* stream, const float* det, const int* det_count, const float* gt_boxes, const long long* gt_cls,
                      int B, int max_det, int M, int num_classes, float iou_threshold,
                      int* tp, int* gt_count, int* gt_imgs, int* correct_imgs);
/* VOC all-points average precision per class (metrics.py:4-90) over n accumulated detections: scores [n], classes [n]
 * 0-based, tp [n] as above; ap [num_classes] float64, NaN for classes without ground truth.  n <= 65536. */
long long effdet_eval_ap_workspace_bytes(int n);
int effdet_eval_ap(void* stream, const float* scores, const int* classes, const int* tp, int n, int num_classes,
                   const int* gt_count, double* ap, void* workspace, long long workspace_bytes);

/* ---- OOD evaluation helpers -------------------------------------------- */

/* The fork's novelty score (infer.py:425-427, 465-471, 607-616), reported beside energy / max-logit:
 * embds [n,d] fp32 = ProjectionNet outputs (normalised inside, F.normalize p=2), confs [n] = anchor confidences (logits),
 * proto_idx [m] int64 = rows of embds that form the cluster (`max_idxs` of the episode code).
 * soft_thresh = sigmoid(dot_mult * (conf + dot_add)); sim = mean_j (use_max = 0) or max_j (use_max = 1) of the cosine
 * similarity to the prototypes; score = soft_thresh * sim.  m * d <= 16384.
 * A proto_idx entry outside [0, n) is never dereferenced: every row of `score` and `sim` then comes back NaN (checked on
 * the device, no host synchronisation; `soft_thresh` stays valid). */
int effdet_novelty_score(void* stream, const float* embds, const float* confs, const long long* proto_idx, int n, int d, int m,
                         float dot_mult, float dot_add, int use_max, float* score, float* soft_thresh, float* sim);

/* Image-level OOD score out[b] = max_a(-energy[b, a]) over the per-anchor energies [B, N]. */
int effdet_ood_image_score(void* stream, const float* energy, int B, long long N, float* out);
/* AUROC of in-distribution scores `pos` against OOD scores `neg` by exact pair counting:
 * counts[0] = #{(i,j): pos_i > neg_j}, counts[1] = #{pos_i == neg_j};  AUROC = (counts[0] + counts[1]/2) / (n_pos*n_neg). */
int effdet_auroc_counts(void* stream, const float* pos, const float* neg, int n_pos, int n_neg, unsigned long long* counts);

#ifdef __cplusplus
}
#endif
#endif /* EFFDET_HIP_H */
