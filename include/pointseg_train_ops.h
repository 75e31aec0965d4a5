/* pointseg_train_ops.h -- the op-level kernels of the PointSegment TRAINING step in libpointseg_hip.so.
 *
 * include/pointseg.h is the stable surface of the hot path (context, knn_search, the tf_map pyramid, grid subsampling, volume -> cloud,
 * the RandLA-Net forward, the six Network.* ops of RandLANet.py:337-401 and the training step behind ONE call, ps_randla_train_step).
 * This header declares what that one call is built from: forward / backward pairs of the shared-MLP ops, BatchNorm with batch
 * statistics (tf.layers.batch_normalization(training=True), RandLANet.py:115, helper_tf_util.py:167), the fused and recompute forms the
 * native trainer (csrc/trainer.hip) chooses between, deterministic scatter-adds, the weighted cross-entropy (RandLANet.py:62-84) and
 * Adam.  They are exported so that the host-side tape of point-unet_amd/train.py (Trainer(engine="python"), the A/B reference of the
 * native step) and the parity tests can call every kernel on its own; a host that trains through ps_randla_train_step /
 * ps_randla_backward never needs them.  Same conventions as pointseg.h: device pointers, int status codes, ps_last_error().
 */
#ifndef POINTSEG_TRAIN_OPS_H
#define POINTSEG_TRAIN_OPS_H

#include "pointseg.h"

#ifdef __cplusplus
extern "C" {
#endif

/* bfloat16 STORAGE of the LFA branch's [N*K, h] activation rows (BASELINE configs[2], "bf16 MLPs": what the products of that mode
 * round their operands to anyway).  While on != 0 -- and only together with ps_set_train_gemm_bf16 -- these pointers are rows of
 * bfloat16 (row strides in ELEMENTS, rows 16-byte aligned), everything else stays fp32:
 *   ps_op_locse_train_apply / ps_op_locse_train_bwd                         out (written rounded to nearest even) / dz (read)
 *   ps_op_conv_bn_train_sums / _apply / _bwd_sums[2] / _bwd_apply[_w]      x (read), _apply's out (written), dz (read), dx (written)
 *   ps_op_att_pool_train_fwd_split / _bwd_split[_rows]                      fr (read), dfr (written), K = 16, d in {16, 64, 128}
 * The GRADIENT rows of those tensors (dz, dx, dfr) travel in the same format: written rounded to nearest even; an accumulating op
 * (accumulate != 0, or dfr under the trainer's add-in-place form) adds to the stored value and rounds the sum.  So do the gathered
 * half's gradient rows: ps_op_att_pool_train_bwd_split_rows writes dfl_rows as bfloat16 and ps_op_gather_reduce_rows[_ordered] reads
 * `rows` as bfloat16 (sums and dst in fp32).  Every weight gradient and the float-atomic form's dfl stay fp32.  The value a consumer sees is the stored one: the weighted sum of att_pooling and the 8-channel
 * convolution, which keep fp32 arithmetic in the bf16-MLP mode, read rounded inputs.  Off by default; ps_train_options.act_bf16 turns
 * it on inside ps_randla_train_step for the levels whose kernels take it. */
int ps_set_train_act_bf16(ps_context* ctx, int on);

/* ---- training-step ops (device pointers; dense row-major fp32 [rows, channels]) --------------------------------------
 * The reference trains with TF autodiff over the same graph with tf.layers.batch_normalization(training=True)
 * (helper_tf_util.py:167,246; RandLANet.py:115), the class-weighted softmax cross-entropy of RandLANet.py:267-274 and
 * tf.train.AdamOptimizer (RandLANet.py:89).  point-unet_amd/train.py records a tape of these ops. */
/* dW[cin,cout] = x^T . dy (overwritten), db[cout] = column sums of dy (may be NULL) */
int ps_op_linear_wgrad(ps_context* ctx, const float* x, const float* dy, int64_t R, int64_t cin, int64_t cout,
                       float* dW, float* db);
/* y = act(gamma * (x - mean_batch) * rsqrt(var_batch + eps) + beta); saves mean, invstd, var (population variance);
 * scratch2C: 2*C floats */
int ps_op_bn_train_fwd(ps_context* ctx, const float* x, const float* gamma, const float* beta, int64_t R, int64_t C,
                       float eps, int leaky, float* y, float* mean, float* invstd, float* var, float* scratch2C);
int ps_op_bn_train_bwd(ps_context* ctx, const float* dy, const float* x, const float* gamma, const float* beta,
                       const float* mean, const float* invstd, int64_t R, int64_t C, int leaky, float* dx,
                       float* dgamma, float* dbeta);
/* The same two ops split at their per-channel reduction, for BatchNorm statistics shared by the GPUs of a data-parallel job
 * (keeps "8 GPUs x 1 cloud" numerically equal to "1 GPU x 8 clouds"): the caller sums `sums2C` = [sum x | sum x^2]
 * (resp. dbeta = sum g, dgamma = sum g*xhat) over the ranks between the halves and passes the global row count R_total. */
int ps_op_bn_train_sums(ps_context* ctx, const float* x, int64_t R, int64_t C, float* sums2C);
int ps_op_bn_train_apply(ps_context* ctx, const float* x, const float* gamma, const float* beta, const float* sums2C,
                         int64_t R, int64_t R_total, int64_t C, float eps, int leaky, float* y, float* mean, float* invstd,
                         float* var);
int ps_op_bn_train_bwd_sums(ps_context* ctx, const float* dy, const float* x, const float* gamma, const float* beta,
                            const float* mean, const float* invstd, int64_t R, int64_t C, int leaky, float* dgamma,
                            float* dbeta);
int ps_op_bn_train_bwd_apply(ps_context* ctx, const float* dy, const float* x, const float* gamma, const float* beta,
                             const float* mean, const float* invstd, const float* sum_g, const float* sum_gx, int64_t R,
                             int64_t R_total, int64_t C, int leaky, float* dx);
/* backward of gather_neighbour / nearest_interpolation: dpc[b*N + idx[row], :] += drows[row, :] */
int ps_op_scatter_add_rows(ps_context* ctx, const float* drows, const int32_t* idx, int64_t B, int64_t N,
                           int64_t rows_per_cloud, int64_t d, float* dpc);
/* ---- row-strided variants of the ops above (ld* = row stride in floats, >= the channel count).  tf.concat of two
 * [B,N,K,C/2] tensors (RandLANet.py:328,332) and the split of its gradient are copies in the reference; here the two
 * producers write straight into the left / right columns of the concat buffer and the consumers of the gradient read its
 * column blocks in place.  `accumulate` != 0: y += act(x.w + b) (gradient accumulation in the GEMM epilogue instead of a
 * separate add pass). */
int ps_op_gather_neighbour_ex(ps_context* ctx, const float* pc, const int32_t* idx, int64_t B, int64_t N, int64_t M,
                              int64_t K, int64_t d, float* out, int64_t ldo);
int ps_op_conv1x1_ex(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R,
                     int64_t cin, int64_t cout, int leaky, int accumulate, float* y, int64_t ldy);
int ps_op_linear_wgrad_ex(ps_context* ctx, const float* x, int64_t ldx, const float* dy, int64_t lddy, int64_t R,
                          int64_t cin, int64_t cout, float* dW, float* db);
int ps_op_bn_train_fwd_ex(ps_context* ctx, const float* x, const float* gamma, const float* beta, int64_t R, int64_t C,
                          float eps, int leaky, float* y, int64_t ldy, float* mean, float* invstd, float* var,
                          float* scratch2C);
/* ps_op_bn_train_fwd_ex plus the moving-statistics update of the reference's extra_update_ops (RandLANet.py:90,163; momentum 0.99):
 * moving = momentum * moving + (1 - momentum) * batch, done by the kernel that finishes the batch statistics (three launches in all). */
int ps_op_bn_train_fwd_mov(ps_context* ctx, const float* x, const float* gamma, const float* beta, int64_t R, int64_t C,
                           float eps, int leaky, float* y, int64_t ldy, float* mean, float* invstd, float* var,
                           float* scratch2C, float* moving_mean, float* moving_var, float momentum);
int ps_op_bn_train_bwd_ex(ps_context* ctx, const float* dy, int64_t lddy, const float* x, const float* gamma,
                          const float* beta, const float* mean, const float* invstd, int64_t R, int64_t C, int leaky,
                          float* dx, float* dgamma, float* dbeta);
int ps_op_bn_train_apply_ex(ps_context* ctx, const float* x, const float* gamma, const float* beta, const float* sums2C,
                            int64_t R, int64_t R_total, int64_t C, float eps, int leaky, float* y, int64_t ldy,
                            float* mean, float* invstd, float* var);
int ps_op_bn_train_bwd_sums_ex(ps_context* ctx, const float* dy, int64_t lddy, const float* x, const float* gamma,
                               const float* beta, const float* mean, const float* invstd, int64_t R, int64_t C,
                               int leaky, float* dgamma, float* dbeta);
int ps_op_bn_train_bwd_apply_ex(ps_context* ctx, const float* dy, int64_t lddy, const float* x, const float* gamma,
                                const float* beta, const float* mean, const float* invstd, const float* sum_g,
                                const float* sum_gx, int64_t R, int64_t R_total, int64_t C, int leaky, float* dx);
int ps_op_scatter_add_rows_ex(ps_context* ctx, const float* drows, int64_t ldd, const int32_t* idx, int64_t B,
                              int64_t N, int64_t rows_per_cloud, int64_t d, float* dpc);
/* att_pooling core: probs = softmax over K of scores, agg = sum_K fset * probs   (RandLANet.py:396-398).  probs may be NULL (not
 * kept); ps_op_softmax_pool_bwd_scores then forms the softmax again from the scores (same arithmetic as the forward; dscores may alias
 * scores): one [R*K, d] tensor less written and kept per pooling. */
int ps_op_softmax_pool_fwd(ps_context* ctx, const float* fset, const float* scores, int64_t R, int64_t K, int64_t d,
                           float* probs, float* agg);
int ps_op_softmax_pool_bwd(ps_context* ctx, const float* dagg, const float* fset, const float* probs, int64_t R,
                           int64_t K, int64_t d, float* dfset, float* dscores);
int ps_op_softmax_pool_bwd_scores(ps_context* ctx, const float* dagg, const float* fset, const float* scores, int64_t R,
                                  int64_t K, int64_t d, float* dfset, float* dscores);
/* att_pooling's score product + softmax + weighted sum FUSED per point for the training step (csrc/attpool_train.hip):
 * agg[n,c] = sum_k softmax_k(fset . wfc)[n,k,c] * fset[n,k,c]   (RandLANet.py:394-398, wfc = the dense kernel [d,d], no bias).
 * fset rows have stride ld (a column block of a wider buffer is fine).  Neither the scores nor the probabilities are written; the
 * backward recomputes them from fset and returns dfset (row stride lddf, overwritten) and dwfc [d,d] (overwritten; summed in a
 * fixed order: deterministic).  K = 16, d in {16, 32, 64} (ps_op_att_pool_train_supported); follows ps_set_train_gemm_bf16. */
int ps_op_att_pool_train_supported(int64_t K, int64_t d);
/* ... with the mode: d = 128 exists for the bf16-MLP mode only (both weight orientations as bfloat16 in LDS; ps_set_train_gemm_bf16 on) */
int ps_op_att_pool_train_supported_ex(int64_t K, int64_t d, int bf16_mode);
int ps_op_att_pool_train_fwd(ps_context* ctx, const float* fset, int64_t ld, const float* wfc, int64_t R, int64_t K, int64_t d,
                             float* agg);
int ps_op_att_pool_train_bwd(ps_context* ctx, const float* fset, int64_t ld, const float* wfc, const float* dagg, int64_t R,
                             int64_t K, int64_t d, float* dfset, int64_t lddf, float* dwfc);
/* The wide levels (d = 128 / 256 / 512: encoder levels 2-4; csrc/attpool_gemm.hip): the same fused op on the frame of the large split-bf16
 * GEMMs -- a wave owns the 16 neighbour rows of two points, the scores live only in its accumulator registers.  fwd writes agg [R, d];
 * bwd (d = 512: two launches, each over half of the dfset columns) recomputes the scores and returns dfset (row stride lddf; accumulate != 0: added to what the rows hold) and dscores [R*K, d]
 * (row stride ldds) -- the weight gradient is dwfc = fset^T . dscores (ps_op_linear_wgrad_ex).  K = 16, rows 16-byte aligned;
 * follows ps_set_train_gemm_bf16 (one plane of rounded operands instead of the exact three-way split).
 * Replaces: tf.layers.dense + tf.nn.softmax + tf.reduce_sum and their gradients, RandLANet.py:394-398. */
int ps_op_att_pool_gemm_supported(int64_t K, int64_t d);
int ps_op_att_pool_gemm_fwd(ps_context* ctx, const float* fset, int64_t ld, const float* wfc, int64_t R, int64_t K, int64_t d,
                            float* agg);
int ps_op_att_pool_gemm_bwd(ps_context* ctx, const float* fset, int64_t ld, const float* wfc, const float* dagg, int64_t R,
                            int64_t K, int64_t d, float* dfset, int64_t lddf, int accumulate, float* dscores, int64_t ldds);
/* ... and with gather_neighbour and the concat folded in (RandLANet.py:326-333: fset[b,n,k,:] = [ fl[b, idx[b,n,k], :] | fr[b,n,k,:] ], never
 * materialised; d = 128 / 256): fl [B*n_src, d/2] (row stride ldl), idx i32[B, n_q, K] cloud-local rows of fl, fr [B*n_q*K, d/2] (ldr).  bwd: the
 * gathered half's gradient leaves as plain rows dfl_rows [B*n_q*K, d/2] (ld_rows; follow with ps_op_gather_reduce_rows over the inverse
 * index of idx), the fr half goes to dfr (lddr; accumulate != 0: added), dscores [B*n_q*K, d] as above; the weight gradient over the split
 * source is ps_op_linear_wgrad_split (>= 16 384 rows, d a multiple of 128: the split-bf16 weight-gradient kernel with the row gather in
 * its loader). */
int ps_op_att_pool_gemm_fwd_split(ps_context* ctx, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src,
                                  int64_t n_q, const float* fr, int64_t ldr, const float* wfc, int64_t K, int64_t d, float* agg);
int ps_op_att_pool_gemm_bwd_split(ps_context* ctx, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src,
                                  int64_t n_q, const float* fr, int64_t ldr, const float* wfc, const float* dagg, int64_t K, int64_t d,
                                  float* dfl_rows, int64_t ld_rows, float* dfr, int64_t lddr, int accumulate, float* dscores,
                                  int64_t ldds);
/* dW [cin, cout] = X^T . dy for X = [xl[xidx] | xr] (cin = twice the width of each half), rows = B * n_q * K */
int ps_op_linear_wgrad_split(ps_context* ctx, const float* xl, int64_t ldxl, const int32_t* xidx, int64_t B, int64_t n_src, int64_t n_q,
                             int64_t K, const float* xr, int64_t ldxr, const float* dy, int64_t lddy, int64_t cin, int64_t cout,
                             float* dW);
/* The same with gather_neighbour and the concat folded in (RandLANet.py:326-333: fset = concat(gather_neighbour(f, neigh_idx), f_xyz)):
 * fset[b, n, k, :] = [ fl[b, idx[b,n,k], :] | fr[b, n, k, :] ] is never materialised.  fl [B*n_src, d/2] (row stride ldl), idx [B, n_q, K]
 * cloud-local, fr [B*n_q*K, d/2] (row stride ldr).  The backward writes dfr (row stride lddr, overwritten), ADDS the gathered half's
 * gradient into dfl (row stride lddl; float atomics, like ps_op_scatter_add_rows) and writes dwfc (overwritten, deterministic). */
int ps_op_att_pool_train_fwd_split(ps_context* ctx, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src,
                                   int64_t n_q, const float* fr, int64_t ldr, const float* wfc, int64_t K, int64_t d, float* agg);
int ps_op_att_pool_train_bwd_split(ps_context* ctx, const float* fl, int64_t ldl, const int32_t* idx, int64_t B, int64_t n_src,
                                   int64_t n_q, const float* fr, int64_t ldr, const float* wfc, const float* dagg, int64_t K, int64_t d,
                                   float* dfl, int64_t lddl, float* dfr, int64_t lddr, float* dwfc);
/* The LocSE branch of the training step, f_xyz = LeakyReLU(BN_train(relative_pos_encoding(xyz, idx) . w + b))   (RandLANet.py:323-325,
 * 377-386; helper_tf_util.conv2d :115-170), without the [B*N*K, 10] encoding, the product or their gradients in memory: everything is
 * recomputed from xyz [B*N,3] and idx [B,N,K] (csrc/locse_train.hip).  w [10,h], b [h]; h in {8,16,32,64} (ps_op_locse_train_supported).
 *   _sums : sums[0:h] = sum_rows y, sums[h:2h] = sum_rows y^2, y = enc10 . w + b, accumulated and returned in float64 (the variance is a
 *           difference of nearly equal numbers when a channel's mean is large against its spread; the caller forms mean / variance; SyncBN:
 *           all-reduce first)
 *   _apply: out[r, :] = LeakyReLU((y - mean) scale + beta), scale = gamma invstd; row stride ldo
 *   _bwd  : one pass over dz (row stride lddz): sums = S1[h] | S2[h] | XS[h] | A[10,h] | G[10,h] | E[16] with xh = (y - mean) invstd,
 *           g = dz lrelu', S1 = sum g, S2 = sum g xh, XS = sum xh, A = enc10^T g, G = enc10^T xh, E = sum enc10 (23 h + 16 floats); then
 *           dgamma = S2, dbeta = S1, dw = gamma invstd (A - E x S1/M - G . S2/M), M = rows of all ranks.  Deterministic (fixed-order sums). */
int ps_op_locse_train_supported(int64_t K, int64_t h);
int ps_op_locse_train_sums(ps_context* ctx, const float* xyz, const int32_t* idx, int64_t B, int64_t N, int64_t K, const float* w,
                           const float* b, int64_t h, double* sums);
int ps_op_locse_train_apply(ps_context* ctx, const float* xyz, const int32_t* idx, int64_t B, int64_t N, int64_t K, const float* w,
                            const float* b, int64_t h, const float* mean, const float* scale, const float* beta, float* out, int64_t ldo);
int ps_op_locse_train_bwd(ps_context* ctx, const float* xyz, const int32_t* idx, int64_t B, int64_t N, int64_t K, const float* w,
                          const float* b, int64_t h, const float* scale, const float* beta, const float* mean, const float* invstd,
                          const float* dz, int64_t lddz, float* sums);
/* conv2d(C -> C, bias) + batch_normalization(training=True) + LeakyReLU on [R, C] rows without the pre-BatchNorm product or its gradient in
 * memory (LFA mlp2 of building_block, RandLANet.py:331; csrc/smallconv_train.hip): y = x . w + b is recomputed from 16-row tiles of x on
 * the fp32 MFMA wherever it is needed.  w [C,C], C in {8,16,32,64} (ps_op_conv_bn_train_supported); CP = max(C, 16).
 *   _sums     : sums = sum y [CP] | sum y^2 [CP] | sum x [CP] in float64 (the caller forms mean / variance; SyncBN: all-reduce first)
 *   _apply    : out[r, :] = LeakyReLU((y - mean) scale + beta), scale = gamma invstd
 *   _bwd_sums : one pass over dz: S1 [CP] | S2 [CP] | XS [CP] | A [CP,CP] | G [CP,CP] with xh = (y - mean) invstd, g = dz lrelu',
 *               S1 = sum g, S2 = sum g xh, XS = sum xh, A = x^T g, G = x^T xh; then dgamma = S2, dbeta = S1,
 *               dw = gamma invstd (A - (sum x) x S1/M - G . S2/M), M = rows of all ranks
 *   _bwd_apply: dx (+)= (gamma invstd (g - m1 - xh m2)) . w^T with m1 = S1/M, m2 = S2/M
 * Deterministic (per-workgroup partials merged in a fixed order).  With ps_set_train_gemm_bf16 on and C % 16 == 0 the operands of the
 * products (x and w; dy and w^T; x and dy) are rounded to bfloat16 first, fp32 accumulation -- the rule of ps_op_conv1x1_ex. */
int ps_op_conv_bn_train_supported(int64_t C);
int ps_op_conv_bn_train_sums(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C, double* sums);
int ps_op_conv_bn_train_apply(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C,
                              const float* mean, const float* scale, const float* beta, float* out, int64_t ldo);
int ps_op_conv_bn_train_bwd_sums(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C,
                                 const float* mean, const float* invstd, const float* scale, const float* beta, const float* dz,
                                 int64_t lddz, float* sums);
int ps_op_conv_bn_train_bwd_apply(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C,
                                  const float* mean, const float* invstd, const float* scale, const float* beta, const float* m1,
                                  const float* m2, const float* dz, int64_t lddz, int accumulate, float* dx, int64_t lddx);
/* The two backward passes in the form the C++ training step uses: the weight gradient comes out of the apply pass as dW = x^T dy, db = sum dy
 * (dy = gamma invstd (g - S1/M - xh S2/M), the gradient of the pre-BatchNorm product, which never reaches memory), so the sums pass only
 * carries S1 and S2.  C = 8 runs one THREAD per row (csrc/convbn_rows.hip: 32-byte rows in registers, every pass at the HBM rate).
 *   _bwd_sums2   : s12 = S1 [C] | S2 [C] (| C floats of scratch: the buffer holds 3 C floats); dgamma = S2, dbeta = S1; SyncBN: all-reduce 2 C
 *   _bwd_apply_w : dx (+)= dy . w^T, dw [C, C] and db [C] overwritten; s12 = the sums of all ranks, inv_rows = 1 / rows of all ranks */
int ps_op_conv_bn_train_bwd_sums2(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C,
                                  const float* mean, const float* invstd, const float* scale, const float* beta, const float* dz,
                                  int64_t lddz, float* s12);
int ps_op_conv_bn_train_bwd_apply_w(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t C,
                                    const float* mean, const float* invstd, const float* scale, const float* beta, const float* s12,
                                    float inv_rows, const float* dz, int64_t lddz, int accumulate, float* dx, int64_t lddx, float* dw,
                                    float* db);
/* The same recompute scheme for the shared MLPs that WIDEN their rows (csrc/rectconv_train.hip): conv2d(cin -> cout, bias) +
 * batch_normalization(training=True) [+ LeakyReLU when leaky != 0] on [R, cin] rows -- Encoder mlp2 / shortcut (RandLANet.py:312-321, no
 * activation) and fc1 (:145).  (cin, cout) in {(8,32), (16,32), (32,64), (32,128), (64,128)} (ps_op_convbn_train_supported); rows 16-byte
 * aligned, pitches % 4 == 0.  3 cin + 11 cout row passes op by op become 5 cin + 3 cout.
 *   _sums      : sums = sum y [cout] | sum y^2 [cout] in float64 (SyncBN: all-reduce, then mean / variance)
 *   _apply     : out[r, :] = act((y - mean) scale + beta), scale = gamma invstd
 *   _bwd_sums  : s12 = S1 [cout] | S2 [cout] (dbeta | dgamma; SyncBN: all-reduce)
 *   _bwd_apply : dx (+)= dy . w^T (dx may be NULL), dw [cin, cout] and db [cout] overwritten; s12 of all ranks, inv_rows = 1 / rows of all ranks
 * Deterministic; bf16-MLP mode (ps_set_train_gemm_bf16, cin % 16 == 0): operands of the three products rounded to bfloat16 first. */
int ps_op_convbn_train_supported(int64_t cin, int64_t cout);
int ps_op_convbn_train_sums(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin,
                            int64_t cout, double* sums);
int ps_op_convbn_train_apply(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin,
                             int64_t cout, const float* mean, const float* scale, const float* beta, int leaky, float* out, int64_t ldo);
/* _apply_add : out[r, :] = LeakyReLU((y - mean) scale + beta + addend[r, :]) -- the residual sum of dilated_res_block (mlp2's output +
 * the shortcut's, RandLANet.py:306-307) inside the pass that writes the second summand; bit-identical to _apply (leaky = 0) followed by
 * ps_op_add_lrelu.  addend [R, cout] rows (ld_add), 16-byte aligned. */
int ps_op_convbn_train_apply_add(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin,
                                 int64_t cout, const float* mean, const float* scale, const float* beta, const float* addend,
                                 int64_t ld_add, float* out, int64_t ldo);
int ps_op_convbn_train_bwd_sums(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin,
                                int64_t cout, const float* mean, const float* invstd, const float* scale, const float* beta, int leaky,
                                const float* dz, int64_t lddz, float* s12);
int ps_op_convbn_train_bwd_apply(ps_context* ctx, const float* x, int64_t ldx, const float* w, const float* b, int64_t R, int64_t cin,
                                 int64_t cout, const float* mean, const float* invstd, const float* scale, const float* beta, int leaky,
                                 const float* s12, float inv_rows, const float* dz, int64_t lddz, int accumulate, float* dx,
                                 int64_t lddx, float* dw, float* db);
/* ---- deterministic scatter-adds (csrc/invidx.hip).  The backward of tf.batch_gather (gather_neighbour, nearest_interpolation,
 * random_sample: RandLANet.py:345-386) adds gradient rows onto the rows they were gathered from; with float atomics the order of the
 * additions changes from run to run.  ps_op_inverse_index inverts a gather table idx i32[B, rows_per_cloud] (values in [0, N)):
 * offsets i32[B*N + 1], src i32[B*rows_per_cloud] = the flat rows r that read source row j = b*N + idx[r], ASCENDING, for
 * j's segment offsets[j] .. offsets[j+1]; workspace: ps_op_inverse_index_workspace(B*N, B*rows_per_cloud) int32 words, 8-byte aligned
 * (tables of 65 536 rows and more: one stable bucket pass + a sort inside every bucket; smaller ones: count / scan / fill / per-segment sort).  ps_op_gather_reduce_rows then
 * forms dst[j, :] (+)= sum over the segment, in that order, of rows[src, :] -- every backward scatter as a gather-reduction, no atomics. */
int64_t ps_op_inverse_index_workspace(int64_t n_dst, int64_t rows);
int ps_op_inverse_index(ps_context* ctx, const int32_t* idx, int64_t B, int64_t N, int64_t rows_per_cloud, int32_t* offsets,
                        int32_t* src, int32_t* workspace);
int ps_op_gather_reduce_rows(ps_context* ctx, const float* rows, int64_t ldr, const int32_t* offsets, const int32_t* src,
                             int64_t n_dst, int64_t d, float* dst, int64_t ldd, int accumulate);
/* The same sums with the destinations WALKED in the order i32[n_dst / n_cloud, n_cloud] (cloud-local rows: ps_pyramid.order of the
 * destinations' level; NULL = ascending), one contiguous eighth per XCD: the rows of a segment belong to the destination's spatial
 * neighbours, so a spatially coherent walk finds them in the XCD's L2.  Results are bit-identical to ps_op_gather_reduce_rows. */
int ps_op_gather_reduce_rows_ordered(ps_context* ctx, const float* rows, int64_t ldr, const int32_t* offsets, const int32_t* src,
                                     int64_t n_dst, int64_t d, float* dst, int64_t ldd, int accumulate, const int32_t* order,
                                     int64_t n_cloud);
/* ps_op_random_sample_bwd through an inverse index.  pool_idx i32[B, M, K] must be the first M rows per cloud of a table
 * i32[B, N', K] with N' >= M (the pyramid's sub_idx = neigh_idx[:, :M], runBraTS.py:150) and offsets / src the inverse index of THAT
 * table (rows_per_cloud = N'*K, here N' = N): the pooling rows are a prefix of every segment, no second index is built.
 * ties: u8[B*M, d] from ps_op_random_sample_ties (how many of the K rows attain the maximum), or NULL: then they are recounted into
 * share_ws (B*M*d floats) by an extra pass. */
int ps_op_random_sample_bwd_inv(ps_context* ctx, const float* dout, const float* out, const float* feature,
                                const int32_t* pool_idx, const int32_t* offsets, const int32_t* src, int64_t B, int64_t N,
                                int64_t M, int64_t K, int64_t d, const uint8_t* ties, float* share_ws, float* dfeature);
/* ps_op_random_sample that also writes ties u8[B*M, d] (d % 4 == 0, K <= 255): tf.reduce_max's gradient is shared evenly by the rows
 * that attain the maximum, and the forward has all K of them in registers anyway */
int ps_op_random_sample_ties(ps_context* ctx, const float* feature, const int32_t* pool_idx, int64_t B, int64_t N, int64_t M,
                             int64_t K, int64_t d, float* out, uint8_t* ties);
/* ps_op_att_pool_train_bwd_split with the gathered half's gradient written as plain rows dfl_rows f32[B*n_q*K, d/2] (row stride
 * ld_rows) instead of scatter-added: follow it with ps_op_gather_reduce_rows over the inverse index of idx. */
int ps_op_att_pool_train_bwd_split_rows(ps_context* ctx, const float* fl, int64_t ldl, const int32_t* idx, int64_t B,
                                        int64_t n_src, int64_t n_q, const float* fr, int64_t ldr, const float* wfc,
                                        const float* dagg, int64_t K, int64_t d, float* dfl_rows, int64_t ld_rows, float* dfr,
                                        int64_t lddr, float* dwfc);
/* backward of random_sample (max over K); ties share the gradient evenly like tf.reduce_max; dfeature accumulates */
int ps_op_random_sample_bwd(ps_context* ctx, const float* dout, const float* out, const float* feature,
                            const int32_t* pool_idx, int64_t B, int64_t N, int64_t M, int64_t K, int64_t d,
                            float* dfeature);
/* y = LeakyReLU(a + b) (RandLANet.py:321) and its backward ds = dy * act'(y) */
int ps_op_add_lrelu(ps_context* ctx, const float* a, const float* b, int64_t n, float* y);
int ps_op_add_lrelu_bwd(ps_context* ctx, const float* dy, const float* y, int64_t n, float* ds);
int ps_op_axpy(ps_context* ctx, float alpha, const float* x, int64_t n, float* y);
int ps_op_mul(ps_context* ctx, const float* a, const float* b, int64_t n, float* y);
/* class_weights[label] * softmax-CE averaged over the VALID rows (RandLANet.py:62-84, 267-274); a label outside [0, C) marks an
 * ignored point (cfg.ignored_label_inds: dropped before the loss by the reference): zero weight, zero gradient row, not counted in
 * the mean.  *loss is a device float; dlogits may be NULL.  Deterministic (no float atomics). */
int ps_op_weighted_ce(ps_context* ctx, const float* logits, const int32_t* labels, const float* class_weights,
                      int64_t R, int64_t C, float* loss, float* dlogits);
/* tf.train.AdamOptimizer update, step >= 1 */
int ps_op_adam(ps_context* ctx, float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
               float beta2, float eps, int64_t step);
/* tf.nn.dropout: y = x * mask, mask = (u < keep_prob) / keep_prob from a counter-based hash of (element, seed) */
int ps_op_dropout(ps_context* ctx, const float* x, int64_t n, uint32_t seed, float keep_prob, float* y, float* mask);

#ifdef __cplusplus
}
#endif
#endif /* POINTSEG_TRAIN_OPS_H */
