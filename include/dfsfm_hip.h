/*
 * dfsfm_hip.h -- C ABI of libdfsfm_hip.so: the MI355X (gfx950) hot kernels of the
 * DetectorFreeSfM dense-matching path.
 *
 * The reference (zju3dv/DetectorFreeSfM) has no FFI of its own for this path: the boundary is
 * a duck-typed Python nn.Module contract.  This library sits UNDER that
 * contract: `detectorfreesfm_amd` keeps the reference's Python plugin surface and calls these
 * entry points through ctypes on raw device pointers.  Every entry point cites the reference
 * code it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless the name ends in `_host`
 *   - all tensors are dense row-major fp32 unless stated; indices are int64 like torch's
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); every call is
 *     asynchronous on that stream and re-entrant across streams (no global state, no
 *     allocation: scratch comes from the caller through `workspace`)
 *   - return value: 0 on success, a negative DFSFM_E_* code otherwise; nothing is launched
 *     when an argument check fails
 */
#ifndef DFSFM_HIP_H
#define DFSFM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DFSFM_OK 0
#define DFSFM_E_BADARG (-1)      /* null pointer / non-positive size                       */
#define DFSFM_E_UNSUPPORTED (-2) /* shape outside what the kernels are built for           */
#define DFSFM_E_WORKSPACE (-3)   /* workspace too small (query the *_workspace function)   */
#define DFSFM_E_LAUNCH (-4)      /* hipLaunch / runtime error; see dfsfm_last_error_string */

/* Library / device introspection. */
int dfsfm_version(void);                    /* ABI version, currently 1 */
const char* dfsfm_last_error_string(void);  /* thread-local text for the last DFSFM_E_LAUNCH */

/* ------------------------------------------------------------------------------------------
 * K1  Linear attention  phi(Q) (phi(K)^T V)
 * Replaces LinearAttention.forward
 *   third_party/LoFTR/src/loftr/loftr_module/linear_attention.py:20-47   (coarse, H=8, D=32)
 *   src/MultiviewMatcher/matcher_module/linear_attention.py:28-60        (refine, H=8, D=16)
 * out[n,l,h,:] = (Q[n,l,h,:] . KV[n,h]) * Z[n,l,h] * S,  Q=elu(q)+1, K=elu(k)+1,
 * KV[n,h]=sum_s K[n,s,h,:]^T (v[n,s,h,:]/S),  Z = 1/(Q . sum_s K + eps); masks multiply Q, K, v.
 *
 * q [N,L,H*D] with row stride ldq floats (>= H*D), k/v [N,S,H*D] with row strides ldk/ldv,
 * out [N,L,H*D] with row stride ldo; batch strides are L*ldq, S*ldk, S*ldv, L*ldo.
 * q_mask [N, L/q_group], kv_mask [N, S/kv_group] are uint8 (0/1) or NULL; a mask entry covers
 * `group` consecutive tokens (the refinement head repeats a per-view mask over W*W tokens,
 * matcher_module/transformer.py:151).  D must be 16 or 32.
 * Output: fp32 `out` (may be NULL) and/or split fp16 planes out_hi/out_lo with row stride ldo_s
 * (value = hi + lo/2048, see dfsfm_conv2d_nhwc_f32) for the GEMM that consumes the message.
 * ---------------------------------------------------------------------------------------- */
size_t dfsfm_linear_attention_workspace(int N, int S, int H, int D);
int dfsfm_linear_attention_f32(const float* q, const float* k, const float* v,
                               const uint8_t* q_mask, int q_group,
                               const uint8_t* kv_mask, int kv_group,
                               float* out, int N, int L, int S, int H, int D,
                               int ldq, int ldk, int ldv, int ldo, float eps,
                               void* out_hi, void* out_lo, int ldo_s,
                               void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K3+K4+K5  Coarse correlation + dual-softmax + mutual-NN selection + keypoint epilogue
 * Replaces CoarseMatching.forward / get_coarse_match (eval, dual_softmax, no padding masks)
 *   third_party/LoFTR/src/loftr/utils/coarse_matching.py:84-145, 148-258, mask_border :8-22
 * feat0 [N,L,C], feat1 [N,S,C] (L=h0c*w0c, S=h1c*w1c); the L x S matrices are never written.
 * A match (b,i,j) is kept iff conf>thr, conf is the max of its row AND of its column, and
 * neither cell lies in the first `border` rows/cols of its grid (the reference's high-side
 * slices are empty, coarse_matching.py:19-22 -- reproduced).  Output rows are in ascending
 * (b,i) order like torch.where.
 *   mkpts0[m] = (i % w0c, i / w0c) * (coarse_scale * scale0[b][{1,0}])   (:239-247)
 * scale0/scale1 [N,2] = (h_scale, w_scale) or NULL (treated as 1).
 * Outputs must hold N*L rows; *count (device int32) receives M.
 * ---------------------------------------------------------------------------------------- */
size_t dfsfm_coarse_match_workspace(int N, int L, int S);
int dfsfm_coarse_match_f32(const float* feat0, const float* feat1, int N, int L, int S, int C,
                           float temperature, float thr, int border,
                           int h0c, int w0c, int h1c, int w1c,
                           const float* scale0, const float* scale1, float coarse_scale,
                           int64_t* b_ids, int64_t* i_ids, int64_t* j_ids, float* mconf,
                           float* mkpts0, float* mkpts1, int32_t* count,
                           void* workspace, size_t workspace_bytes, void* stream);

/* Same operation on features that arrive as fp16 split planes (feat = hi + lo/2048, the format the last
 * encoder layer's LayerNorm writes -- see dfsfm_conv2d_nhwc_f32): the correlation runs as three fp16 MFMA
 * products per k-step (fp32-class accuracy, ~6e-7 relative) on the LDS-DMA main loop of the linear layers
 * instead of the fp32 matrix path.  C must be a power of 4 (the 1/sqrt(C) scaling of
 * coarse_matching.py:103-104 then folds into one exact multiply) and a multiple of 32.
 * Same workspace query, outputs and ordering as dfsfm_coarse_match_f32. */
int dfsfm_coarse_match_split(const void* feat0_hi, const void* feat0_lo, const void* feat1_hi,
                             const void* feat1_lo, int N, int L, int S, int C, float temperature, float thr,
                             int border, int h0c, int w0c, int h1c, int w1c, const float* scale0,
                             const float* scale1, float coarse_scale, int64_t* b_ids, int64_t* i_ids,
                             int64_t* j_ids, float* mconf, float* mkpts0, float* mkpts1, int32_t* count,
                             void* workspace, size_t workspace_bytes, void* stream);

/* The same with padding masks (mask0 [N,L], mask1 [N,S] uint8, 1 = valid): rows / columns of padded coarse cells take
 * no part in the softmaxes and produce no match, as `sim_matrix.masked_fill_(~(mask_c0[..., None] * mask_c1[:, None]), -INF)`
 * does in the reference (third_party/LoFTR/src/loftr/utils/coarse_matching.py:110-113; the MatchFormer copy
 * third_party/MatchFormer/model/backbone/coarse_matching.py:104-107).  With `border` > 0 the border rule is
 * `mask_border_with_padding` (coarse_matching.py:25-41): the low `border` rows / columns of both grids as without masks,
 * and per pair the last `border` rows / columns of each frame's valid extent (h = max column sum, w = max row sum of
 * the mask) plus everything beyond them. */
int dfsfm_coarse_match_split_masked(const void* feat0_hi, const void* feat0_lo, const void* feat1_hi,
                                    const void* feat1_lo, const uint8_t* mask0, const uint8_t* mask1, int N, int L, int S,
                                    int C, float temperature, float thr, int border, int h0c, int w0c, int h1c, int w1c,
                                    const float* scale0, const float* scale1, float coarse_scale, int64_t* b_ids,
                                    int64_t* i_ids, int64_t* j_ids, float* mconf, float* mkpts0, float* mkpts1,
                                    int32_t* count, void* workspace, size_t workspace_bytes, void* stream);

/* Dense confidence matrix conf[N,L,S] = softmax(sim,1)*softmax(sim,2) (coarse_matching.py:103-116).
 * Debug / parity aid only ("conf_matrix" is stored by the reference but no inference caller
 * reads it, src/coarse_match/coarse_match_worker.py:83-91). Same workspace as above. */
int dfsfm_coarse_conf_matrix_f32(const float* feat0, const float* feat1, int N, int L, int S, int C,
                                 float temperature, float* conf,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K8  RoIAlign patch extraction (TensorFlow crop_and_resize semantics, extrapolation value)
 * Replaces roi_align.RoIAlign(crop, crop, transform_fpcoor=False) as called from
 *   src/MultiviewMatcher/matcher_module/fine_preprocess.py:92-106 (un-vendored
 *   third_party/RoIAlign.pytorch; semantics in oracle/restate.py:roi_align_crop)
 * feat [Nimg,C,H,W]; boxes [M,4]=(x1,y1,x2,y2) pixels; box_ind [M] int32 or NULL (=0).
 * out_slot [M] int64 or NULL: patch m is written to out[out_slot[m]] (lets the caller scatter
 * one image's patches straight into (view,track) order, MultiviewMatcher.py:253-266).
 * mean/std [C] or NULL: when given, (value-mean[c])/std[c] is applied to every output
 * (fuses the ImageNet normalisation of S2DNet._forward, backbone/S2DNet/s2dnet.py:132-133).
 * out [*,C,crop_h,crop_w], or [*,crop_h,crop_w,C] when out_channels_last != 0 (the layout the
 * NHWC convolution kernels below consume).
 * ---------------------------------------------------------------------------------------- */
int dfsfm_roi_align_f32(const float* feat, int Nimg, int C, int H, int W,
                        const float* boxes, const int32_t* box_ind, const int64_t* out_slot, int M,
                        int crop_h, int crop_w, float extrapolation_value,
                        const float* mean, const float* std, float* out, int out_channels_last,
                        void* stream);

/* ------------------------------------------------------------------------------------------
 * K11+K12  Fine-window correlation, softmax expectation, best-candidate selection, keypoints
 * Replaces FineMatching.forward (test config: s2d heatmap + argsoftmax,
 * left_point_movement, best_left_strategy='smallest_mean_std')
 *   src/MultiviewMatcher/utils/fine_matching.py:36-98, 100-119, 129-179, 195-252, 258-285
 * ref [T,W*W,C], qry [T,Vq,W*W,C]; track_mask [T,Vq] uint8; movable [T] uint8 or NULL (=1).
 * For every track: the centre left x left window of `ref` gives L candidates; for each
 * candidate and view: heat = softmax_r(<ref_l, qry_r>/sqrt(C)); (ex,ey)=E[grid], std =
 * sqrt(max(var_x,1e-10))+sqrt(max(var_y,1e-10)); score_l = masked mean over views of std;
 * best = argmin_l (first minimum), or the centre candidate when not movable.
 * Outputs (any may be NULL):
 *   best_index [T] int32, left_norm [T,2], coords [T,Vq,2], std [T,Vq]
 *   query_refined [T,2]   = query_pts[t]  + left_norm * (left/2) * scale_q[t]
 *   ref_refined [T,Vq,2]  = ref_pts[t,n]  + coords    * (W/2)    * scale_r[t,n]
 * query_pts [T,2], scale_q [T,2], ref_pts / scale_r addressed as base[(t*rs_t + n*rs_n)*2]
 * so the caller can pass the reference's [V-1,T,2] tensors without a transpose
 * (rs_t=1, rs_n=T) or a [T,Vq,2] tensor (rs_t=Vq, rs_n=1).  C must be a multiple of 4,
 * W*W <= 256, left <= W, left*left <= 64.
 * ---------------------------------------------------------------------------------------- */
int dfsfm_fine_match_f32(const float* ref, const float* qry, const uint8_t* track_mask,
                         const uint8_t* movable, int T, int Vq, int W, int left, int C,
                         const float* query_pts, const float* scale_q,
                         const float* ref_pts, const float* scale_r, int64_t rs_t, int64_t rs_n,
                         int32_t* best_index, float* left_norm, float* coords, float* std,
                         float* query_refined, float* ref_refined, void* stream);
/* The same operation on features that arrive as fp16x2-split planes (value = hi + lo/2048, the form the encoder kernels write:
 * dfsfm_encoder_apply_f32 out_hi / out_lo): ref_hi / ref_lo [T,W*W,C], qry_hi / qry_lo [T,Vq,W*W,C] fp16, dense.  The planes
 * are streamed straight into the MFMA fragments; every other argument as above.  C = 64 or 128. */
int dfsfm_fine_match_split(const void* ref_hi, const void* ref_lo, const void* qry_hi, const void* qry_lo,
                           const uint8_t* track_mask, const uint8_t* movable, int T, int Vq, int W, int left, int C,
                           const float* query_pts, const float* scale_q,
                           const float* ref_pts, const float* scale_r, int64_t rs_t, int64_t rs_n,
                           int32_t* best_index, float* left_norm, float* coords, float* std,
                           float* query_refined, float* ref_refined, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2/K10 epilogues  LayerNorm (+ residual) with row strides
 * out[r, 0:C] = (residual ? residual[r, 0:C] : 0) + LayerNorm(x[r, 0:C]) * gamma + beta
 * Replaces norm1 / norm2 / the residual add of LoFTREncoderLayer.forward
 *   third_party/LoFTR/src/loftr/loftr_module/transformer.py:50,56-58
 *   src/MultiviewMatcher/matcher_module/transformer.py:82,88-95
 * and, through `ldo`, the torch.cat([x, message]) of :55 / :87: norm1 writes straight into the
 * second half of a [rows, 2C] buffer whose first half holds x.  ldx/ldr/ldo are row strides in
 * floats.  C must be 64, 128 or 256.  The residual is fp32 (`residual`) or split fp16 planes
 * (res_hi/res_lo, row stride ldr in halves) -- between encoder layers the token state only exists as
 * split planes.  The result is written as fp32 (`out`, may be NULL) and/or as split fp16 planes
 * out_hi/out_lo (row stride ldo_s) for the GEMMs that consume it.
 * ---------------------------------------------------------------------------------------- */
int dfsfm_layernorm_f32(const float* x, int64_t ldx, const float* gamma, const float* beta, float eps,
                        const float* residual, const void* res_hi, const void* res_lo, int64_t ldr,
                        float* out, int64_t ldo, void* out_hi, void* out_lo, int64_t ldo_s, int64_t rows,
                        int C, void* stream);

/* out[r, :] = x[r, :] (+ add[r % add_rows, :]) as fp32 (`out`, may be NULL) and/or split planes.
 * Used once per forward to add the positional encoding (LoFTR loftr.py:58-59: the [h*w, C] table
 * broadcasts over the batch) and hand the tokens to the first encoder layer.  C % 4 == 0. */
int dfsfm_split_rows_f32(const float* x, int64_t ldx, const float* add, int64_t add_rows, float* out,
                         int64_t ldo, void* out_hi, void* out_lo, int64_t ldo_s, int64_t rows, int C,
                         void* stream);

/* Split planes of rows that sit in uniformly strided BLOCKS of a larger tensor: row r is read at
 * x + (r / blk_rows) * blk_stride + (r % blk_rows) * ldx.  Used to hand the per-track token groups of the [T,V,WW,C] feature
 * tensor (src/MultiviewMatcher/MultiviewMatcher.py:240-270) to the first encoder layer without a contiguous copy. */
int dfsfm_split_rows_blocked_f32(const float* x, int64_t blk_rows, int64_t blk_stride, int64_t ldx, void* out_hi, void* out_lo,
                                 int64_t ldo_s, int64_t rows, int C, void* stream);

/* Depth-wise 3x3 convolution (groups = C, stride 1, pad 1, bias) on an NHWC fp32 map with its consumer fused:
 *   mode 0: dw(x) + b      mode 1: x * sigmoid(dw(x) + b)      mode 2: GELU(dw(x) + b)   (erf form)
 * Replaces Positional (pa_conv + sigmoid gate) and the DWConv + GELU of Mlp of MatchFormer-LA
 *   third_party/MatchFormer/model/backbone/match_LA_large.py:15-27, 39-41, 108-116
 * x [N,H,W,C] fp32; w9c [9][C] fp32 = weight[c,0,ky,kx] at [(ky*3+kx)][c]; bias [C]; C % 8 == 0.
 * out [N,H,W,C] fp32 and / or out_hi / out_lo [N,H,W,C] fp16 split planes (either may be NULL). */
int dfsfm_dwconv3x3_nhwc_f32(const float* x, int N, int H, int W, int C, const float* w9c, const float* bias, int mode,
                             float* out, void* out_hi, void* out_lo, void* stream);

/* F.interpolate(mode='bilinear', align_corners=True) of NHWC fp32 maps (FPN top-down path, match_LA_large.py:232-236).
 * x [N,hin,win,C] -> out [N,hout,wout,C]; C % 4 == 0. */
int dfsfm_bilinear_up_nhwc_f32(const float* x, int N, int hin, int win, int C, int hout, int wout, float* out, void* stream);

/* ---- ASpanFormer coarse matcher (third_party/aspantransformer/src/ASpanFormer/) ------------------
 * Maps are NHWC fp32 with a row (= token) pitch in floats, so column slices of wider buffers can be read and written. */

/* F.avg_pool2d(x, k, stride=k): x [N,H,W,C] (pitch ldx) -> out [N,H/k,W/k,C] (pitch ldo); the window is summed in (ky, kx)
 * order and divided by k*k like ATen.  aspan_module/transformer.py:163-167, attention.py:60-66. */
int dfsfm_avgpool_nhwc_f32(const float* x, int64_t ldx, int N, int H, int W, int C, int k, float* out, int64_t ldo,
                           void* stream);

/* FullAttention.forward (aspan_module/attention.py:141-165): out[n,l,h,:] = softmax_s(scale * <q[n,l,h,:], k[n',s,h,:]>) v[n',s,h,:]
 * with n' = n ^ kv_swap (kv_swap = 1 pairs image 0 with image 1 in a stacked batch).  q [N,L,H*D], k / v [N,S,H*D], pitches ld*
 * and batch strides s* in floats; D = 32.  scale = temp / sqrt(D). */
int dfsfm_full_attention_f32(const float* q, int64_t ldq, int64_t sq, const float* k, int64_t ldk, int64_t sk, const float* v,
                             int64_t ldv, int64_t sv, float* out, int64_t ldo, int64_t so, int N, int L, int S, int H, int D,
                             int kv_swap, float scale, void* stream);

/* One level of HierachicalAttention (aspan_module/attention.py:49-53, 64-66, 92-133) for N images: for every 2x2 group of the
 * query level map q [N, h*w, C] the mean flow offset / span of the group's full-resolution cells (flow [N, H0*W0, 4] = x, y,
 * var_x, var_y, dense; span = max(exp(var/2) * radius_scale * 2 / nsample1, 1)), nsample1^2 bilinear samples (grid_sample, zero
 * padding, align_corners False) of the level maps k, v [N, hk*wk, C] of image n ^ kv_swap at offset + sample_offset * span,
 * softmax attention of the group's 4 queries over the samples per head.  Row pitches ld*, batch strides s* in floats.
 * out [N, h*w, C] (dense batches of pitch ldo): row g*4 + n, the order the reference's view() produces.
 * C = 256, nhead = 8, nsample = (2, 8); no padding masks. */
int dfsfm_span_attention_f32(const float* q, int64_t ldq, int64_t sq, int h, int w, const float* k, int64_t ldk, int64_t sk,
                             const float* v, int64_t ldv, int64_t sv, int hk, int wk, const float* flow, int H0, int W0,
                             const float* sample_offset, int nhead, int C, int nsample0, int nsample1, float radius_scale,
                             float temp, float* out, int64_t ldo, int N, int kv_swap, void* stream);

/* layernorm2d (aspan_module/attention.py:7-19) on token rows: res + affine * (x - mean) / (std_unbiased + 1e-6) + bias; the
 * residual arrives as split planes (may be NULL); out fp32 and / or split planes.  C = 256 or 384. */
int dfsfm_layernorm2d_f32(const float* x, int64_t ldx, const float* affine, const float* bias, const void* res_hi,
                          const void* res_lo, int64_t ldr, float* out, int64_t ldo, void* out_hi, void* out_lo, int64_t ldo_s,
                          int64_t rows, int C, void* stream);

/* F.upsample(x, scale_factor=scale, mode='bilinear' (align_corners False) | 'nearest') of x [N,hin,win,C] (pitch ldx) into fp32
 * and / or split-plane rows (pitches ldo, ldo_s).  aspan_module/transformer.py:177-180, attention.py:85-86. */
int dfsfm_upsample_nhwc_f32(const float* x, int64_t ldx, int N, int hin, int win, int C, int scale, int bilinear, float* out,
                            int64_t ldo, void* out_hi, void* out_lo, int64_t ldo_s, void* stream);

/* messageLayer_gla.decode_flow after the flow_decoder convs (aspan_module/transformer.py:125-133):
 * out[r] = (sigmoid(x[r,0]) * wk, sigmoid(x[r,1]) * hk, x[r,2], x[r,3]). */
int dfsfm_flow_decode_f32(const float* x, int64_t ldx, int64_t rows, float wk, float hk, float* out, void* stream);

/* F.interpolate(x, size=(hout, wout), mode='bilinear', align_corners=False) of fp32 planes x [N,hin,win] -> out [N,hout,wout]:
 * the online resize of frames whose sides are not multiples of 32 (torchvision 0.9.1 transforms.Resize on a float tensor,
 * third_party/aspantransformer/src/ASpanFormer/aspanformer.py:131-139). */
int dfsfm_resize_bilinear_f32(const float* x, int N, int hin, int win, int hout, int wout, float* out, void* stream);

/* PIL's 8-bit fixed-point resampling (Image.resize on 'L' / 'RGB' images) + the tensor conversion of the reference's
 * image readers.  Replaces, for an already decoded frame,
 *   resize_image(image, (w_new, h_new), "pil_LANCZOS")           src/dataset/utils.py:160-177
 *   pad_bottom_right / grayscale2tensor / rgb2tensor / mask2tensor  src/dataset/utils.py:30-62, as called at :108-121, :147-160
 * Arithmetic (Pillow src/libImaging/Resample.c, ImagingResampleHorizontal_8bpc / Vertical_8bpc): a pixel is
 *   clip8((2^21 + sum_x src[first + x] * kk[x]) >> 22) in int32; horizontal pass, rounded to 8 bits, then vertical.
 * src [H,W,C] bytes with row pitch src_stride, C = 1 or 3.  bounds_x [Wn,2] / bounds_y [Hn,2] = (first source index, tap
 * count) and kk_x [Wn,ksize_x] / kk_y [Hn,ksize_y] = the 22-bit coefficients of precompute_coeffs + normalize_coeffs_8bpc
 * (any filter; identity tables skip a pass exactly).  tmp: H*Wn*C bytes of scratch.  Outputs, any subset:
 *   out_u8 [Hn,Wn,C];   out_f32 [C,pad_h,pad_w] = lut256[byte] inside the image (lut256[v] = v / 255.f for the reference's
 *   readers) and 0 in the bottom / right padding;   mask [pad_h,pad_w] = 1 inside, 0 in the padding (with out_f32 only). */
int dfsfm_resample_u8(const uint8_t* src, int64_t src_stride, int H, int W, int C, const int32_t* bounds_x,
                      const int32_t* kk_x, int ksize_x, int Wn, const int32_t* bounds_y, const int32_t* kk_y, int ksize_y,
                      int Hn, uint8_t* tmp, uint8_t* out_u8, float* out_f32, float* mask, int pad_h, int pad_w,
                      const float* lut256, void* stream);

/* Separable resampling of NHWC patch feature maps:
 *   out[m, oy, ox, c] = sum_{qy,qx} By[oy, qy] * Bx[ox, qx] * y[m, qy, qx, c]
 * Replaces nn.Upsample(mode='bicubic', align_corners=True) followed by the centre-window crop of S2DNet's coarse
 * adaptation map (src/MultiviewMatcher/backbone/S2DNet/s2dnet.py:164-193): By [hout,hin] / Bx [wout,win] are the rows
 * of the interpolation matrix that fall inside the window (the caller obtains them from the framework's own bicubic
 * kernel, so the coefficients are the reference's).  y [M,hin,win,C], out [M,hout*wout,C] fp32, C % 64 == 0. */
int dfsfm_resample_separable_f32(const float* y, int M, int hin, int win, int C, const float* By, const float* Bx,
                                 int hout, int wout, float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * K9 -> K10 hand-off  dst[slot[m], p, c] = a[m, c, p] (+ b[m, c, p])
 * a, b [M, C, P] (NCHW patch features, P = W*W), slot [M] int64 or NULL (identity), dst [*, P, C].
 * Replaces the hypercolumn sum (backbone/S2DNet/s2dnet.py:164-171), the 'm c h w -> m (h w) c'
 * rearrange and the original-order gather of src/MultiviewMatcher/MultiviewMatcher.py:240-270.
 * ---------------------------------------------------------------------------------------- */
int dfsfm_add_scatter_tokens_f32(const float* a, const float* b, const int64_t* slot, float* dst,
                                 int M, int C, int P, void* stream);

/* ------------------------------------------------------------------------------------------
 * K6/K9 first layers  Direct convolution for tiny input depth (fp32 FMA, no operand splitting)
 * Replaces conv1 + bn1 + relu of ResNetFPN_8_2 (7x7 stride 2, 1 -> 128;
 *   third_party/LoFTR/src/loftr/backbone/resnet_fpn.py:100-104) and conv1_1 + relu of the S2DNet VGG
 *   encoder (3x3, 3 -> 64; src/MultiviewMatcher/backbone/S2DNet/s2dnet.py:127-175): K = kh*kw*Cin is
 *   49 / 27, too shallow for the matrix cores.
 * x fp32 NHWC view (strides in elements as for dfsfm_conv2d_nhwc_f32); w fp32 [kh*kw*Cin][Cout] in
 * (ky,kx,ci) order, 256-byte aligned; bias [Cout] or NULL (folded BN); fp32 and/or split outputs
 * (ldo_s >= Cout, Cout % 64 == 0).  Supported (Cin,kh,kw,stride,pad,Cout): (1,7,7,2,3,128) and
 * (3,3,3,1,1,64); anything else returns DFSFM_E_UNSUPPORTED (use dfsfm_conv2d_nhwc_f32).
 * ---------------------------------------------------------------------------------------- */
int dfsfm_conv2d_direct_f32(const float* x, int64_t sxn, int64_t sxh, int64_t ldx, int Nimg, int H, int W,
                            int Cin, const float* w, int Cout, int kh, int kw, int stride, int pad,
                            const float* bias, int relu, float* out, int64_t ldo, void* out_hi, void* out_lo,
                            int64_t ldo_s, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2/K6/K9  Convolution / linear layer, NHWC, implicit GEMM on the fp16 matrix cores with an
 * fp16x2 operand split (fp32-class accuracy, see csrc/conv_gemm.hip), fused epilogue:
 *   out[m, co] = act( sum_{ky,kx,ci} x[n, oy*stride+ky-pad, ox*stride+kx-pad, ci] * w[co,ky,kx,ci]
 *                     + bias[co] + residual[m, co] ),   m = (n*Ho + oy)*Wo + ox
 * Replaces nn.Conv2d (+ folded eval BatchNorm, ReLU, residual add) of
 *   third_party/LoFTR/src/loftr/backbone/resnet_fpn.py:15-40,100-118
 *   src/MultiviewMatcher/backbone/S2DNet/s2dnet.py:24-52,127-175
 * and, as the 1x1 case with Nimg=1,H=1,W=rows, nn.Linear of the encoder layers
 *   third_party/LoFTR/src/loftr/loftr_module/transformer.py:21-31,42-55.
 *
 * "Split" tensors: a value v is stored as two fp16 planes, v = hi + lo/2048, with
 * hi = (|v| >= 2^-14 ? fp16(v) : 0) and lo = fp16((v - hi) * 2048) (22 significant bits, same
 * 4 bytes per element as fp32).  Activations between convolutions travel in this form so the
 * kernel can DMA them straight into LDS.
 *
 * Input : either x (fp32) or x_hi/x_lo (split planes, fp16); element (n,y,x,c) at
 *         [n*sxn + y*sxh + x*ldx + c] (element strides; any NHWC view, e.g. a centre crop).
 *         Split input needs Cin, ldx, sxh, sxn multiples of 8 and planes < 4 GiB.
 * Weights: w_hi/w_lo fp16 [ceil128(Cout)][Kpad], K = kh*kw*Cin in (ky,kx,ci) order, zero padded,
 *         Kpad % 32 == 0, same hi/lo split.  bias [Cout] or NULL.
 * Residual: fp32 [M, Cout] (row stride ldr) or split planes res_hi/res_lo (row stride ldr), or none.
 * Output: fp32 `out` (row stride ldo) and/or split planes out_hi/out_lo (row stride ldo_s, Cout_s >=
 *         Cout channels, Cout_s % 8 == 0; channels >= Cout are written as zeros).
 * tap_padded != 0 (split input, kh == kw in {3,5}, stride 1, pad kw/2 only): the weights are laid out
 *         with K = (ky, kx, ceil32(Cin)) -- Kpad = kh*kw*ceil32(Cin) -- and the kernel variant that
 *         loads each activation row once per (ky, 32-channel chunk) and reuses it for the kw taps
 *         is used (1/kw of the activation traffic).
 * ln_gamma/ln_beta [Cout] (or both NULL), ln_eps: LayerNorm fused into the epilogue of a linear layer
 *         (split input, 1x1, Cout = 64, 128 or 256, no ReLU):  out = residual + LN(x.W^T + bias) * gamma + beta
 *         -- the merge -> norm1 and mlp -> norm2 (+x) pairs of LoFTREncoderLayer.forward
 *         (third_party/LoFTR/src/loftr/loftr_module/transformer.py:50-58, d_model 256;
 *         src/MultiviewMatcher/matcher_module/transformer.py:82-95, d_model 128) in one pass.
 * ---------------------------------------------------------------------------------------- */
int dfsfm_conv2d_nhwc_f32(const float* x, const void* x_hi, const void* x_lo, int64_t sxn, int64_t sxh,
                          int64_t ldx, int Nimg, int H, int W, int Cin, const void* w_hi, const void* w_lo,
                          int Cout, int Kpad, int kh, int kw, int stride, int pad, const float* bias,
                          const float* residual, const void* res_hi, const void* res_lo, int64_t ldr,
                          int relu, float* out, int64_t ldo, void* out_hi, void* out_lo, int64_t ldo_s,
                          int Cout_s, int tap_padded, const float* ln_gamma, const float* ln_beta, float ln_eps,
                          void* stream);

/* ------------------------------------------------------------------------------------------
 * K9 front end in one launch: conv1_1 (3 -> 64, 3x3, pad 1) -> ReLU -> conv1_2 (64 -> 64, 3x3, pad 1) -> ReLU ->
 * { centre window, MaxPool2d(3, stride 2, padding 1) } of S2DNet's VGG encoder on `patch` x `patch` RGB patches
 *   src/MultiviewMatcher/backbone/S2DNet/s2dnet.py:86-92,127-175 (features[0..4] with the substituted pooling layer),
 *   backbone/S2DNet/vggnet.py:12-44
 * Neither relu1_1 nor relu1_2 is written to memory (csrc/s2d_front.hip).  Both layers are the fp16x2-split MFMA product of
 * dfsfm_conv2d_nhwc_f32 (fp32-class: operands hi + lo/2048, fp32 accumulation).
 * patches [n][patch][patch][3] fp32 NHWC (normalised); w1_frag: conv1_1 weights as MFMA A fragments, fp16
 * [2 halves of 32 channels][2 blocks of 16][hi, lo][64 lanes][8]: element j of lane l = plane(w1[c][k]) with
 * c = 32 half + 16 block + (l & 15), k = 8 (l >> 4) + j = 3 (3 ky + kx) + ci (k >= 27: 0); b1 [64]; w2_hi / w2_lo [w2_rows >= 64][kpad] fp16 tap-padded split planes of conv1_2
 * (k = (ky*3 + kx)*64 + ci, kpad = 576); b2 [64].  Outputs, split planes (value = hi + lo/2048), 64 channels:
 * crop_* [n][c1-c0][c1-c0][64] = relu1_2[:, c0:c1, c0:c1, :], pool_* [n][(patch+1)/2][(patch+1)/2][64].
 * patch must be 35 (else DFSFM_E_UNSUPPORTED: callers run the three separate layers).
 * ---------------------------------------------------------------------------------------- */
int dfsfm_s2d_front_f32(const float* patches, int64_t n_patches, int patch, const void* w1_frag, const float* b1,
                        const void* w2_hi, const void* w2_lo, int64_t w2_rows, int64_t kpad, const float* b2, int c0, int c1,
                        void* crop_hi, void* crop_lo, void* pool_hi, void* pool_lo, void* stream);

/* nn.MaxPool2d(3, stride=2, padding=1) on a dense NHWC tensor, fp32 (x -> out, C % 4 == 0) or split
 * planes (x_hi/x_lo -> out_hi/out_lo, C % 8 == 0)  (S2DNet with substitute_pooling_layers,
 * backbone/S2DNet/s2dnet.py:89-92). */
int dfsfm_maxpool3x3s2_nhwc_f32(const float* x, const void* x_hi, const void* x_lo, int Nimg, int H, int W,
                                int C, float* out, void* out_hi, void* out_lo, void* stream);

/* ------------------------------------------------------------------------------------------
 * Match-table consumers: scene-wide keypoint merge and match re-indexing
 * Replaces the per-image / per-pair Python loops of coarse_match.py:203-237:
 *   Match2Kpts.__getitem__ (src/coarse_match/utils/merge_kpts.py:36-61), agg_groupby_2d (:4-17),
 *   keypoint_worker / update_matches(merge=False) / transform_keypoints
 *   (src/coarse_match/coarse_match_worker.py:151-175, 182-243, 250-270).
 * rows [M,5] = (x0, y0, x1, y1, conf) of every match of the scene, pairs concatenated in table order;
 * img0/img1 [M] = image index of each side (an image never meets itself in a pair).
 * Every endpoint becomes the integer keypoint (int)x, (int)y of its image (0 <= x, y < 2^20); equal
 * keypoints of an image merge, their confidences summed in float64 in table order; an image's
 * keypoints are numbered by descending summed score, ties in (x, y) lexicographic order.
 *   kpts [2M,2] fp32, scores [2M] fp32: keypoints of image i at offsets[i] .. offsets[i+1], in id order
 *   offsets [n_images+1] int64;  match_ids [M,2] int64 = (id of side 0 in img0, id of side 1 in img1)
 *   *n_kpts (device int64) = total number of keypoints;  *status (device int32) != 0 if an image
 *   index or coordinate was out of range (outputs are then unspecified).
 * No host synchronisation.  Workspace: dfsfm_merge_keypoints_workspace(M) bytes.
 * ---------------------------------------------------------------------------------------- */
size_t dfsfm_merge_keypoints_workspace(int64_t M);
int dfsfm_merge_keypoints(const float* rows, const int32_t* img0, const int32_t* img1, int64_t M, int n_images,
                          float* kpts, float* scores, int64_t* offsets, int64_t* match_ids, int64_t* n_kpts,
                          int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2 + K1 (+ K10) fused  one LoFTREncoderLayer application, d_model 128, 8 heads (csrc/encoder_fused.hip)
 * Replaces LoFTREncoderLayer.forward + LinearAttention.forward of the refinement head
 *   src/MultiviewMatcher/matcher_module/transformer.py:66-95, matcher_module/linear_attention.py:28-60
 *   (the same layer as third_party/LoFTR/src/loftr/loftr_module/transformer.py:35-58, linear_attention.py:20-47)
 * in two launches that read every token row once each and write it once:
 *
 * dfsfm_encoder_kv_f32     source rows [N*S, 128] (split fp16 planes src_hi/src_lo, row stride ld_src halves) ->
 *                          k|v = W_k,v x (never stored) -> per sequence n the "apply image" kv_image[n]
 *                          (DFSFM_ENCODER_KV_IMAGE_BYTES bytes: KV^T = (sum_s phi(k_s)^T v_s / S)^T per head as fp16 hi/lo
 *                          MFMA fragments + Ksum = sum_s phi(k_s) as 128 floats).  kv_mask [N, ceil(S/kv_group)] uint8 or NULL.
 * dfsfm_encoder_apply_f32  x rows [N*L, 128] (split planes) + kv_image -> out = x + norm2(mlp([x | norm1(merge(attn))]))
 *                          as split planes (out_hi/out_lo, row stride ldo halves; may be NULL) and/or fp32 rows (out32, row
 *                          stride ldo32 floats; may be NULL; = hi + lo/2048).  q_mask [N, ceil(L/q_group)] uint8 or NULL.
 *                          S = tokens per sequence on the source side (the reference's v_length).  L >= 32.
 *                          debug / debug_stage: fp32 [N*L,128] dump of one intermediate (1 q, 2 message, 3 norm1, 4 mlp
 *                          output) for the tests; pass NULL / 0.
 * The weights travel as ready-made fragment streams built once on the host (detectorfreesfm_amd/ops.py::
 * EncoderFusedWeights: wstream_kv = 8 slabs, wstream = 32 slabs of 16 KB; layout documented there and in the kernel).
 * ---------------------------------------------------------------------------------------- */
#define DFSFM_ENCODER_KV_IMAGE_BYTES (16 * 1024 + 512)
int dfsfm_encoder_kv_f32(const void* src_hi, const void* src_lo, int64_t ld_src, int N, int S, const void* wstream_kv,
                         const uint8_t* kv_mask, int kv_group, void* kv_image, void* stream);
int dfsfm_encoder_apply_f32(const void* x_hi, const void* x_lo, int64_t ldx, int N, int L, int S, const void* wstream,
                            const void* kv_image, const uint8_t* q_mask, int q_group, const float* gamma1,
                            const float* beta1, float eps1, const float* gamma2, const float* beta2, float eps2,
                            float attn_eps, void* out_hi, void* out_lo, int64_t ldo, float* out32, int64_t ldo32,
                            float* debug, int debug_stage, void* stream);

/* ------------------------------------------------------------------------------------------
 * K2 + K1 fused for d_model 256, 8 heads of 32: the COARSE transformer's LoFTREncoderLayer (csrc/encoder256.hip)
 *   third_party/LoFTR/src/loftr/loftr_module/transformer.py:35-58, 80-101; linear_attention.py:20-47
 *
 * dfsfm_encoder256_state_f32  source side of a layer application: k, v [N*S, 256] fp32 rows (row strides ldk / ldv floats;
 *                             the two halves of the k|v projection) -> per sequence n the apply image kv_image[n]
 *                             (DFSFM_ENCODER256_KV_IMAGE_BYTES: per head and 16-row block the fp16 hi/lo MFMA A fragment
 *                             of KV_h^T = (sum_s phi(k_s)^T v_s / S)^T, then Ksum = sum_s phi(k_s) as 256 floats).
 *                             kv_mask [N, ceil(S/kv_group)] uint8 or NULL.  Two launches (chunked partial sums, a
 *                             fixed-order reduction: deterministic); workspace dfsfm_encoder256_state_workspace(N, S).
 * dfsfm_encoder256_kv_f32     the same state straight from the source TOKENS: source rows [N*S, 256] (split planes, row stride
 *                             ld_src halves) -> k | v = W_kv x (never stored; wstream_kv: 32 slabs of 16 KB, ops.Encoder256Weights)
 *                             -> chunk partials of phi(K)^T V / S and sum phi(K) -> the same image.  Replaces the k | v projection
 *                             GEMM + dfsfm_encoder256_state_f32; workspace dfsfm_encoder256_kv_workspace(N, S).
 * dfsfm_encoder256_apply_f32  query side in ONE launch: x rows [N*L, 256] (split planes) + kv_image ->
 *                             out = x + norm2(mlp.2(relu(mlp.0([x | norm1(merge(attention(W_q x)))]))))
 *                             as split planes and / or fp32 rows; arguments as dfsfm_encoder_apply_f32 (L >= 16;
 *                             debug dump [N*L, 256]).  wstream: 128 slabs of 16 KB (ops.Encoder256Weights).
 * ---------------------------------------------------------------------------------------- */
#define DFSFM_ENCODER256_KV_IMAGE_BYTES (32 * 1024 + 1024)
size_t dfsfm_encoder256_state_workspace(int N, int S);
int dfsfm_encoder256_state_f32(const float* k, const float* v, int ldk, int ldv, const uint8_t* kv_mask, int kv_group,
                               int N, int S, void* kv_image, void* workspace, size_t workspace_bytes, void* stream);
size_t dfsfm_encoder256_kv_workspace(int N, int S);
int dfsfm_encoder256_kv_f32(const void* src_hi, const void* src_lo, int64_t ld_src, int N, int S, const void* wstream_kv,
                            const uint8_t* kv_mask, int kv_group, void* kv_image, void* workspace, size_t workspace_bytes,
                            void* stream);
int dfsfm_encoder256_apply_f32(const void* x_hi, const void* x_lo, int64_t ldx, int N, int L, int S, const void* wstream,
                               const void* kv_image, const uint8_t* q_mask, int q_group, const float* gamma1,
                               const float* beta1, float eps1, const float* gamma2, const float* beta2, float eps2,
                               float attn_eps, void* out_hi, void* out_lo, int64_t ldo, float* out32, int64_t ldo32,
                               float* debug, int debug_stage, void* stream);

/* ------------------------------------------------------------------------------------------
 * Baseline JPEG decode (csrc/jpeg_decode.hip) -- the decode in front of dfsfm_resample_u8
 * Replaces, for baseline Huffman files (SOF0 / SOF1, 8 bit; grey or YCbCr 4:4:4 / 4:2:2 / 4:2:0; with or without DRI)
 *   cv2.imread(path, cv2.IMREAD_GRAYSCALE)         src/dataset/utils.py:127 (read_grayscale), :183
 *   cv2.imread(path, cv2.IMREAD_COLOR) + BGR2RGB   src/dataset/utils.py:86-92 (read_rgb)
 * = libjpeg-turbo's default decompression (jdhuff.c decode_mcu, jidctint.c jpeg_idct_islow, jdsample.c fancy
 * upsampling, jdcolor.c ycc_rgb_convert): out_channels 1 gives the luma plane (what IMREAD_GRAYSCALE returns, no
 * colour conversion), 3 gives RGB; bytes identical to the library.  EXIF orientation is the caller's (a transpose / flip).
 *
 * The host parses the marker segments (detectorfreesfm_amd/jpeg.py) and passes
 *   frame_host  the frame header as plain ints (below); nseg / nchunks / chunk_bytes describe how the scan is cut
 *   scan        the entropy-coded bytes of the single interleaved scan as in the file (stuffed zeros, RSTn markers and
 *               fill bytes in place); 16-byte aligned, the allocation readable up to the next multiple of 16
 *   block_base  [ceil(scan_bytes / 4096)] uint32: entropy bytes in front of each 4096-byte block of `scan` (a byte counts
 *               unless it is a 00 after FF, an FF not followed by 00, or D0..D7 after FF) -- the first kernel compacts
 *               the scan with them
 *   huff_tab    4 slots of 2384 bytes (slots named by dc_slot / ac_slot), per Huffman code: uint16[1024] = (length << 8) |
 *               symbol for every 10-bit prefix of a code of at most 10 bits, else 0; uint32[6][3] = for lengths 11..16 the
 *               left-aligned 16-bit (limit, first code, index of its first symbol); uint8[256] = HUFFVAL (T.81 Annex C)
 *   qt          [3][64] uint16 quantisation steps per component, natural (row-major) order
 *   seg_beg / seg_end [nseg] byte range of each restart interval in the COMPACTED scan; seg_chunk0 [nseg] its first
 *               chunk; chunk_seg [nchunks] the segment of a chunk; chunk i of a segment covers chunk_bytes compacted bytes
 * Decoding is a fixed-point iteration over the chunks (`sweeps` relaxation passes, see the kernel file); status[0] == 0
 * says the fixed point was reached -- otherwise call again with resume = 1 (the workspace keeps the state) and more
 * sweeps.  status[1] = invalid codes on the final path, status[2] = restart intervals with a wrong block count (a corrupt
 * file; `out` is then unspecified), status[3] = sweeps of this call that still decoded something.  status is 4 device
 * int32, read by the caller when convenient: no host sync here.
 * out [height][out_stride] bytes, out_stride >= width * out_channels.
 * ---------------------------------------------------------------------------------------- */
typedef struct dfsfm_jpeg_frame {
    int32_t width, height;
    int32_t ncomp;                   /* 1 (grey) or 3 (Y, Cb, Cr) */
    int32_t h[3], v[3];              /* sampling factors; chroma must be 1 x 1, luma 1x1 / 2x1 / 2x2 */
    int32_t dc_slot[3], ac_slot[3];  /* huff_lut slot (0..3) of each component's DC / AC table */
    int32_t restart;                 /* MCUs per restart interval (DRI), 0 = none */
    int32_t nseg, nchunks, chunk_bytes;
} dfsfm_jpeg_frame;
size_t dfsfm_jpeg_decode_workspace(const dfsfm_jpeg_frame* frame_host, int64_t scan_bytes, int out_channels);   /* 0 = unsupported */
int dfsfm_jpeg_decode_u8(const uint8_t* scan, int64_t scan_bytes, const dfsfm_jpeg_frame* frame_host,
                         const uint32_t* huff_tab, const uint16_t* qt, const uint32_t* block_base, const uint32_t* seg_beg,
                         const uint32_t* seg_end, const int32_t* seg_chunk0, const int32_t* chunk_seg, uint8_t* out, int64_t out_stride,
                         int out_channels, int sweeps, int resume, int32_t* status, void* workspace,
                         size_t workspace_bytes, void* stream);

/* The same decode for a BATCH of files with one set of launches per 7 files (grid.y = file): a scene's frames
 * (src/dataset/coarse_matching_dataset.py:41-88 reads them one by one through cv2.imread) cost ~15 launches per seven files
 * instead of per file.  jobs_host: host array of n_jobs argument sets of dfsfm_jpeg_decode_u8 (device pointers inside; each
 * job has its own output, status[4] and workspace of dfsfm_jpeg_decode_workspace bytes).  Every job is validated before
 * anything is launched; results per job are those of dfsfm_jpeg_decode_u8 with the same sweeps / resume. */
typedef struct dfsfm_jpeg_job {
    const uint8_t* scan;
    int64_t scan_bytes;
    const dfsfm_jpeg_frame* frame_host;
    const uint32_t* huff_tab;
    const uint16_t* qt;
    const uint32_t* block_base;
    const uint32_t* seg_beg;
    const uint32_t* seg_end;
    const int32_t* seg_chunk0;
    const int32_t* chunk_seg;
    uint8_t* out;
    int64_t out_stride;
    int32_t* status;
    void* workspace;
    size_t workspace_bytes;
} dfsfm_jpeg_job;
int dfsfm_jpeg_decode_batch_u8(const dfsfm_jpeg_job* jobs_host, int n_jobs, int out_channels, int sweeps, int resume, void* stream);

/* The colour stage alone, for multi-scan sequential files (each component in a scan of its own -- cv2.imread takes them,
 * /root/reference/src/dataset/utils.py:86-92): y / cb / cr are device planes of the components' REAL samples (width x height for y,
 * ceil(width / h0) x ceil(height / v0) for cb and cr; row strides in bytes) as three dfsfm_jpeg_decode_u8 calls on the grey frames of
 * jpeg.plan_components leave them; out = RGB [height][width][3], libjpeg-turbo's fancy upsampling + YCbCr -> RGB.  (h0, v0) = luma
 * sampling factors: 1x1, 2x1, 2x2, 1x2, 4x1. */
int dfsfm_jpeg_ycc_planes_to_rgb_u8(const uint8_t* y, int64_t y_stride, const uint8_t* cb, const uint8_t* cr, int64_t c_stride,
                                    int width, int height, int h0, int v0, uint8_t* out, int64_t out_stride, void* stream);

/* HOST function (no device is touched): the index of an entropy-coded scan that the decode calls above take as arguments --
 * what libjpeg's jdmarker.c / jdhuff.c learn byte by byte while cv2.imread decodes.  scan = the bytes behind the SOS header,
 * n_avail of them.  Returns scan_len = offset of the first marker that is not RSTn (n_avail if there is none) or a negative
 * DFSFM_E_*; *n_rst = restart markers inside the scan; block_base[ceil(scan_len / 4096)] = entropy bytes in front of each
 * 4096-byte block of the scan (stuffed zeros, marker / fill FFs and restart codes not counted); seg_beg / seg_end [n_rst + 1] =
 * byte range of each restart interval in that compacted numbering.  If an array is too small only the counts are returned. */
int64_t dfsfm_jpeg_scan_index(const uint8_t* scan, int64_t n_avail, uint32_t* block_base, int64_t block_cap, uint32_t* seg_beg,
                              uint32_t* seg_end, int64_t seg_cap, int64_t* n_rst);

#ifdef __cplusplus
}
#endif
#endif /* DFSFM_HIP_H */
