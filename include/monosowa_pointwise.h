/*
 * monosowa_pointwise.h -- C-ABI of the pointwise HIP kernels used around the library convolutions of the
 * MonoDETR backbone (frozen batch-norm folded into the convolution; reference:
 * MonoDETR/lib/models/monodetr/backbone.py:28-65 FrozenBatchNorm2d + torchvision's Bottleneck
 * `relu(bn(conv(x)))` / `relu(bn3(conv3(.)) + identity)`).  Device pointers, float32, NHWC ([rows, C] contiguous).
 * Return 0, a hipError_t (> 0), -1 (NULL pointer) or -2 (shape / alignment: C % 4 == 0, 16-byte aligned).
 */
#ifndef MONOSOWA_POINTWISE_H_
#define MONOSOWA_POINTWISE_H_
#ifdef __cplusplus
extern "C" {
#endif

/* y[r, c] = act(y[r, c] + bias[c] (+ residual[r, c])) in place; residual may be NULL; relu != 0 applies max(., 0). */
int mono_bias_act_f32(float *y, const float *bias, const float *residual, long long rows, int C, int relu, void *stream);
/* The frozen stem (reference backbone.py:72-74, 83; torchvision's ResNet stem): out = max_pool2d(relu(y + bias), 3, stride 2, padding 1)
 * in one pass over the channels-last convolution output y [N, H, W, C]; out [N, (H-1)/2+1, (W-1)/2+1, C].  Forward only. */
int mono_bias_relu_maxpool_nhwc_f32(const float *y, const float *bias, float *out, int N, int H, int W, int C, void *stream);
/* The tail of a FROZEN bottleneck (torchvision Bottleneck behind reference backbone.py:72-74, 83) in one pass:
 * y[M, 256] = relu(relu(x[M, 64] + b_in) w[64, 256] + b_out + res[M, 256]) -- x the 3 x 3 convolution's raw output, w conv3's weight
 * as [K][N] with bn3's scale folded in, rows = channels-last pixels.  K must be 64, N 256.  Forward only.  y must NOT alias res, x or
 * x0 (the kernels read res one channel block ahead of the y stores through __restrict__ pointers): an aliased call returns -2.
 * gfx950 only: the three conv1x1 kernels keep their weight matrices in more than 64 KB of static LDS (65 KB tail / head, 129 KB tail_ds;
 * CDNA4 has 160 KB per CU) -- the library is built for --offload-arch=gfx950 and nothing else. */
int mono_conv1x1_tail_f32(const float *x, const float *b_in, const float *w, const float *b_out, const float *res, float *y,
                          long long M, int K, int N, void *stream);
/* The tail of a stage's FIRST frozen bottleneck: as mono_conv1x1_tail_f32 with the identity = the stride-1 1 x 1 downsample convolution of
 * the block's input x0 [M, 64] (wd [64][256], its norm's scale folded in, its shift inside b_out), evaluated into the same accumulator. */
int mono_conv1x1_tail_ds_f32(const float *x, const float *b_in, const float *w, const float *x0, const float *wd, const float *b_out,
                             float *y, long long M, int K, int N, void *stream);
/* The head of a frozen bottleneck in one pass: y[M, 64] = relu(x[M, K] w[K, 64] + b_out), K = 64 or 256 (conv1 + bn1 + ReLU). */
int mono_conv1x1_head_f32(const float *x, const float *w, const float *b_out, float *y, long long M, int K, int N, void *stream);

/* grad_in[i] = y[i] > 0 ? grad_out[i] : 0   (n % 4 == 0; grad_in may alias grad_out). */
int mono_relu_grad_f32(const float *grad_out, const float *y, float *grad_in, long long n, void *stream);
/* grad_in = scale[c] * grad_out * (y > 0) on a channels-last tensor of n elements, C channels (C % 4 == 0, n % C == 0): the same with the
 * frozen norm's scale put on the gradient (trainable 1 x 1 convolution + frozen BN + ReLU without an identity branch, backbone.py:72-115). */
int mono_relu_grad_scale_f32(const float *grad_out, const float *y, const float *scale, float *grad_in, long long n, int C, void *stream);

/* ReLU with a byte mask: y = relu(y + bias (+ residual)) in place and mask[i] = sign bits of elements 4i..4i+3 (one byte
 * per float4); the backward below reads the mask instead of y (1/16 of the bytes).  grad_b may be NULL. */
int mono_bias_relu_mask_f32(float *y, const float *bias, const float *residual, unsigned char *mask, long long rows, int C,
                            void *stream);
int mono_relu_grad_mask_f32(const float *grad_a, const float *grad_b, const unsigned char *mask, float *grad_in, long long n,
                            void *stream);
/* the same for three gradients (a tensor with three consumers) */
int mono_relu_grad_mask3_f32(const float *grad_a, const float *grad_b, const float *grad_c, const unsigned char *mask, float *grad_in,
                             long long n, void *stream);

/* Frozen batch-norm + ReLU after a convolution without residual, applied here instead of folded into the weights
 * (backbone.py:28-65): y = relu(y * scale[c] + shift[c]) in place + byte mask; grad_in = grad * mask * scale[c]. */
int mono_affine_relu_mask_f32(float *y, const float *scale, const float *shift, unsigned char *mask, long long rows, int C,
                              void *stream);
int mono_affine_relu_grad_f32(const float *grad, const unsigned char *mask, const float *scale, float *grad_in, long long rows,
                              int C, void *stream);

/* grad_in = (grad_a + grad_b) * (y > 0): ReLU backward of a tensor with two consumers (ResNet block output -> next
 * convolution and identity branch, backbone.py:64-82 of the reference's torchvision ResNet) in one pass. */
int mono_relu_grad2_f32(const float *grad_a, const float *grad_b, const float *y, float *grad_in, long long n, void *stream);
/* the same for three consumers: grad_in[i] = y[i] > 0 ? (grad_a[i] + grad_b[i]) + grad_c[i] : 0 */
int mono_relu_grad3_f32(const float *grad_a, const float *grad_b, const float *grad_c, const float *y, float *grad_in, long long n,
                        void *stream);

/* y = LayerNorm_256(x + dropout_p(z)) over rows of C = 256 channels (reference: the post-norm residual blocks of
 * depthaware_transformer.py:339-354,500-515).  The keep mask is a hash of (seed, element index): the backward
 * recomputes it from the same seed.  Saves s = x + dropout(z), mean, rstd [rows] for the backward. */
int mono_dropout_add_layernorm_fwd_f32(const float *x, const float *z, const float *gamma, const float *beta, float *y,
                                       float *s, float *mean, float *rstd, long long rows, int C, float p,
                                       unsigned long long seed, float eps, void *stream);

/* y = dropout_p(relu(h)) over n contiguous floats (n % 4 == 0), hash mask from (seed, element index); the FFN hidden
 * activation `self.dropout2(F.relu(self.linear1(src)))` (depthaware_transformer.py:352,513, transformer.py:63). */
int mono_relu_dropout_fwd_f32(const float *h, float *y, long long n, float p, unsigned long long seed, void *stream);
/* grad_h = grad_y / (1 - p) where y > 0 (kept and h > 0), else 0. */
int mono_relu_dropout_bwd_f32(const float *grad_y, const float *y, float *grad_h, long long n, float p, void *stream);
/* The same over a [rows, 256] matrix, plus colsum[256] = column sums of grad_h (the bias gradient of the linear in front);
 * partials: mono_reduce_blocks(rows) * 256 floats of scratch. */
int mono_relu_dropout_bwd_colsum_f32(const float *grad_y, const float *y, float *grad_h, float *colsum, float *partials, long long rows,
                                     float p, void *stream);

/* The matched-pair losses of SetCriterion (monodetr.py:1010-1103: 3D-centre and l/r/t/b L1, GIoU, Laplacian depth,
 * dimension-aware size L1, 12-bin heading cross entropy + residual L1) for all decoder layers in one launch.
 * boxes/depth/dims/angle: [NL, B, Q, 6|2|3|24] contiguous predictions; idx: int64 [3, NL, K] = (image, query, flat target
 * index) of the K matched pairs per layer; t_*: targets concatenated over the batch.  out: [NL, 6] per-layer sums
 * {center, bbox, giou, depth, dim, angle} (divide by num_boxes); comp: [NL] (size compensation weights, for the backward). */
int mono_matched_losses_fwd_f32(const float *boxes, const float *depth, const float *dims, const float *angle,
                                const long long *idx, const float *t_box, const float *t_depth, const float *t_size,
                                const long long *t_bin, const float *t_res, float *out, float *comp, int NL, int B, int Q,
                                int K, void *stream);
/* g_boxes / g_depth / g_dims / g_angle: gradients of the predictions for grad_out [NL, 6]; ZERO on entry. */
int mono_matched_losses_bwd_f32(const float *boxes, const float *depth, const float *dims, const float *angle,
                                const long long *idx, const float *t_box, const float *t_depth, const float *t_size,
                                const long long *t_bin, const float *t_res, const float *comp, const float *grad_out,
                                float *g_boxes, float *g_depth, float *g_dims, float *g_angle, int NL, int B, int Q, int K,
                                void *stream);

/* One step of the reference's AdamW variant (lib/helpers/optimizer_helper.py:69-129: eps added to sqrt(v) before the
 * bias correction, decay scaled by the corrected step size) over all parameters in one launch.
 * table (device): p[n_chunks], g[n_chunks], m[n_chunks], v[n_chunks] as 64-bit device addresses, then n[n_chunks]
 * (int32 elements per chunk, any size), then wd[n_chunks] (float weight decay). */
int mono_adamw_step_f32(const void *table, int n_chunks, double beta1, double beta2, double eps, double step_size, void *stream);

/* dW[M, N] = dY[R, M]^T . X[R, N] and (db != NULL) db[M] = the column sums of dY, f32, exact products (v_mfma_f32_32x32x2_f32), summed in a
 * fixed order (no atomics): the weight and bias gradients of y = x W^T + b over a few thousand tokens -- autograd's AddmmBackward of the
 * nn.Linear layers of the decoder / depth-token encoder (reference depthaware_transformer.py:339-354,440-515) -- in two launches that
 * read dY once.  Row-major, leading dimensions ldy / ldx in floats and multiples of 4, every pointer 16-byte aligned.
 * mono_linear_wgrad_workspace: floats of scratch `ws` for the call; 0 = shape not served (M, N multiples of 64 up to 4096, R >= 64). */
long long mono_linear_wgrad_workspace(int R, int M, int N);
int mono_linear_wgrad_f32(const float *dy, long long ldy, const float *x, long long ldx, float *dw, float *db, float *ws, int R, int M,
                          int N, void *stream);

/* out[c] = sum_r g[r][c] for ANY width C <= 1024 (odd widths too: 81 depth bins); partials: mono_colsum_any_blocks(rows) * C floats. */
int mono_colsum_any_blocks(long long rows);
int mono_colsum_any_f32(const float *g, float *out, float *partials, long long rows, int C, void *stream);

/* out[c] = sum_k x[k][c] over a contiguous stack [n, C] of n slices (the partial products of a split-K weight gradient); C % 4 == 0,
 * 16-byte aligned pointers. */
int mono_sum_slices_f32(const float *x, float *out, int n, long long C, void *stream);

/* Number of workgroups, = rows of scratch the two row reductions below need, for `rows` input rows. */
int mono_reduce_blocks(long long rows);

/* gx, gz [rows, 256]; ggamma_gbeta [3, 256] (ggamma, gbeta, then the column sums of gz = the bias gradient of the linear whose
 * output z is), overwritten.
 * partials: scratch of mono_reduce_blocks(rows) * 768 floats (per-workgroup partial sums; no atomics: deterministic). */
int mono_dropout_add_layernorm_bwd_f32(const float *gy, const float *s, const float *mean, const float *rstd,
                                       const float *gamma, float *gx, float *gz, float *ggamma_gbeta, float *partials,
                                       long long rows, int C, float p, unsigned long long seed, void *stream);

/* out[c] = sum_r g[r][c] for a row-major [rows, C] matrix (C % 4 == 0, C <= 1024): the bias gradient of a linear
 * layer over tokens (autograd's `grad.sum(0)` for nn.Linear in the reference).  out is overwritten.
 * partials: scratch of mono_reduce_blocks(rows) * C floats. */
int mono_colsum_f32(const float *g, float *out, float *partials, long long rows, int C, void *stream);
/* The same over a [batch, rows, C] view with batch_stride floats between batches (C <= 512): sums of one pyramid level's
 * tokens over the batch (the level_embed gradient of the visual encoder, depthaware_transformer.py:232-240). */
int mono_colsum_strided_f32(const float *g, float *out, float *partials, int batch, long long rows, long long batch_stride,
                            int C, void *stream);

/* Column sums of several row ranges of a contiguous [batch, S, C] tensor in one launch pair: out[l][c] = sum over all batches and rows
 * [bounds[2 l], bounds[2 l + 1]) of g[., ., c], l < n_levels <= 8, C % 4 == 0, C <= 512; `bounds` is a HOST array of 2 n_levels ints.  The
 * sums of the encoder's d(offsets | logits) over each pyramid level: their total is a bias gradient, times W they are d level_embed
 * (depthaware_transformer.py:232-240).  with_total != 0: out has n_levels + 1 rows, the last one the sum of all the ranges' sums (added in the
 * same launch, in workgroup order).  partials: mono_colsum_levels_blocks(...) * C floats of scratch (0: unusable bounds). */
int mono_colsum_levels_blocks(int batch, long long S, int n_levels, const int *bounds);
int mono_colsum_levels_f32(const float *g, float *out, float *partials, int batch, long long S, int C, int n_levels, const int *bounds,
                           int with_total, void *stream);

/* GroupNorm(32 groups, 256 channels) (+ ReLU when relu != 0) on a channels-last tensor x [B, HW, 256]
 * (reference: nn.GroupNorm(32, hidden_dim) in monodetr.py:68-88 input_proj and depth_predictor.py:27-52).
 * pre_bias [256] or NULL: added to x first -- the bias of the convolution in front (nn.Sequential(Conv2d, GroupNorm)),
 * so that convolution can run bias-free.  stats: f64 [B, 32, 2] scratch, ZERO on entry.  mean_rstd: f32 [B, 32, 2]. */
int mono_groupnorm_nhwc_fwd_f32(const float *x, const float *pre_bias, const float *gamma, const float *beta, float *y,
                                double *stats, float *mean_rstd, int B, int HW, int C, int G, float eps, int relu,
                                void *stream);

/* Workgroups of the backward = rows (of 256 floats) of gbias_partials it needs. */
int mono_groupnorm_blocks(int B, int HW);

/* gx [B, HW, 256].  part: f64 [B, 256, 2], ZERO on entry; on return part[b][c] = {sum gy'*xhat, sum gy'} so that
 * ggamma[c] = sum_b part[b][c][0], gbeta[c] = sum_b part[b][c][1].  y = forward output (ReLU mask), NULL if relu == 0.
 * With pre_bias != NULL: gbias [256] = gradient of pre_bias, gbias_partials = scratch.
 * ggamma_gbeta (nullable): f32 [2][256] = {ggamma, gbeta}, written by the second kernel (no launch of its own). */
int mono_groupnorm_nhwc_bwd_f32(const float *gy, const float *x, const float *pre_bias, const float *y,
                                const float *mean_rstd, const float *gamma, float *gx, double *part, float *gbias,
                                float *gbias_partials, float *ggamma_gbeta, int B, int HW, int C, int G, int relu, void *stream);

/* DDN depth-map loss (depth_predictor/ddn_loss/ddn_loss.py:12-127 + balancer.py + focalloss.py) in one kernel per
 * direction.  logits [B, C = num_bins + 1, H, W] addressed with (batch, channel, pixel) strides in floats (NCHW or
 * channels-last in place); boxes [B, N, 4] xyxy in depth-map pixels, depth [B, N], valid [B, N] (bytes).
 * forward: partial[mono_ddn_loss_blocks(B, H, W)] block sums of weight * pixel loss; loss = sum(partial) / (B H W).
 * backward: grad_logits (same strides) = grad_total[0] * d loss / d logits. */
int mono_ddn_loss_blocks(int B, int H, int W);
int mono_ddn_loss_fwd_f32(const float *logits, const float *boxes, const float *depth, const unsigned char *valid, float *partial,
                          int B, int C, int H, int W, int N, long long sb, long long sc, long long sp, float alpha, float gamma,
                          float fg_weight, float bg_weight, float depth_min, float depth_max, void *stream);
int mono_ddn_loss_bwd_f32(const float *logits, const float *boxes, const float *depth, const unsigned char *valid,
                          const float *grad_total, float *grad_logits, int B, int C, int H, int W, int N, long long sb, long long sc,
                          long long sp, float alpha, float gamma, float fg_weight, float bg_weight, float depth_min, float depth_max,
                          void *stream);

/* Expected depth of the bin distribution (depth_predictor/depth_predictor.py:90-91): out [B, H, W] = sum_c softmax(logits)_c *
 * values[c]; logits [B, C, H, W] with (batch, channel, pixel) strides as above.  backward: grad_logits (same strides) =
 * grad_out * p_c * (values[c] - out). */
int mono_depth_expect_fwd_f32(const float *logits, const float *values, float *out, int B, int C, int H, int W, long long sb,
                              long long sc, long long sp, void *stream);
int mono_depth_expect_bwd_f32(const float *logits, const float *values, const float *expect, const float *grad_out, float *grad_logits,
                              int B, int C, int H, int W, long long sb, long long sc, long long sp, void *stream);

/* Classification side of SetCriterion for all decoder layers (monodetr.py:396-449; sigmoid focal loss :302-330).
 * logits [NL, B, Q, C]; idx [3, NL, K] int64 = (image, query, flat target) of the matched pairs; labels [T] int64; sizes [B]
 * float targets per image.  out [NL, 3] = { focal-loss SUM over (image, query, class), class_error in %, cardinality_error }.
 * alpha < 0 disables the alpha weighting.  backward: grad_logits [NL, B, Q, C] = grad_out[l] * d focal / d logit.
 * Limits: C <= 255, B <= 256, B * Q <= 32768. */
int mono_focal_fwd_f32(const float *logits, const long long *idx, const long long *labels, const float *sizes, float *out, int NL,
                       int B, int Q, int C, int K, float alpha, float gamma, void *stream);
int mono_focal_bwd_f32(const float *logits, const long long *idx, const long long *labels, const float *grad_out, float *grad_logits,
                       int NL, int B, int Q, int C, int K, float alpha, float gamma, void *stream);

/* The Hungarian matcher's cost, restricted to each image's own targets (reference matcher.py:53-88: focal-style class cost, L1
 * of the projected 3-D centre and of (l, r, t, b), generalised IoU, weighted): logits [NL, B, Q, C], boxes [NL, B, Q, 6] (cx, cy, l,
 * r, t, b); labels [T] int64 and tboxes [T, 6] of the batch's concatenated targets; cols [B, N] int64 = flat target of slot j of
 * image b (padding slots repeat a valid target).  out [NL, B, Q, N].  Every operation is rounded like the PyTorch expression's
 * (no contraction), so the assignments are the reference's. */
int mono_match_cost_f32(const float *logits, const float *boxes, const long long *labels, const float *tboxes, const long long *cols,
                        float *out, int NL, int B, int Q, int C, int N, float w_class, float w_3d, float w_bbox, float w_giou,
                        void *stream);

/* The Hungarian matcher's assignments ON THE DEVICE (reference matcher.py:94-103: scipy.optimize.linear_sum_assignment per image and
 * query group, on the host; csrc/lsap.cpp is this repo's host solver with the same contract): cost [NL, B, Q, T] as written by
 * mono_match_cost_f32 (image b's targets in columns [0, sizes[b])); for every layer, image and group g of Q / G queries one
 * rectangular assignment problem, solved by one wavefront with scipy's algorithm, arithmetic (double) and tie-breaks, so the
 * pairs are the reference's.  meta [3, B] int32 = sizes | first | toff (first[b] = sum over earlier images of G min(Q / G, sizes),
 * toff[b] = sum of earlier sizes).  out_idx [3, NL, K] int64 = (image, query, toff + target) per pair in (image, group) order, pairs
 * of a group sorted by query, K = first[B - 1] + G min(Q / G, sizes[B - 1]) -- the flat index tensor lsap_match_flat_f32 returns.
 * status (device int, OR-ed): 1 = a cost is NaN / -inf or a problem is infeasible (scipy raises), 2 = a problem exceeds the
 * kernel's tables.  No host synchronisation: the caller reads `status` when it likes.
 * Limits: Q / G <= 128, max(sizes) <= 128, min(Q / G, n) max(Q / G, n) <= 8192 per problem.  Returns -3 beyond the first limit. */
int mono_lsap_match_flat_f32(const float *cost, int NL, int B, int Q, int T, int G, const int *meta, long long *out_idx, long long K,
                             int *status, void *stream);

/* Per-level tail of MonoDETR's detection heads (monodetr.py:238-263), one launch each way: coords [B, Q, 6] = sigmoid(tmp),
 * depth_ave [B, Q, 2] = ((1 / (sigmoid(depth_reg0) + 1e-6) - 1 + size3d0 / max((coords4 + coords5) img_h, 1) fu
 *                        + bilinear(wdepth [B, H, W]; coords0, coords1 -- F.grid_sample, align_corners, zero padding)) / 3, depth_reg1).
 * backward: g_coords / g_depth_ave may be NULL (no gradient); g_wdepth [B, H, W] must be ZERO on entry (atomics); the sampling
 * location carries no gradient through the depth map (detached in the reference). */
int mono_head_tail_fwd_f32(const float *tmp, const float *size3d, const float *depth_reg, const float *wdepth, const float *fu,
                           const float *img_h, float *coords, float *depth_ave, int B, int Q, int H, int W, const float *ref,
                           int ref_dim, void *stream);
int mono_head_tail_bwd_f32(const float *tmp, const float *size3d, const float *depth_reg, const float *wdepth, const float *fu,
                           const float *img_h, const float *g_coords, const float *g_depth_ave, float *g_tmp, float *g_size3d,
                           float *g_depth_reg, float *g_wdepth, int B, int Q, int H, int W, const float *ref, int ref_dim,
                           void *stream);
/* ref (optional, NULL = none): detached reference boxes [B, Q, ref_dim]; the box logits are then tmp + inverse_sigmoid(ref) on the
 * first ref_dim coordinates (monodetr.py:224-232).
 * The decoder's detached reference refinement (depthaware_transformer.py:602-613): out [n, 6] = sigmoid(tmp [n, 6] +
 * inverse_sigmoid(ref [n, ref_dim]) on the first ref_dim coordinates). */
int mono_refine_reference_f32(const float *tmp, const float *ref, float *out, int n, int ref_dim, void *stream);

#ifdef __cplusplus
}
#endif
#endif
