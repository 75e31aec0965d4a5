/* include/pointseg.h -- C ABI of libpointseg_hip.so: the MI355X-native replacement for the native ops and
 * the forward graph on Point-Unet's PointSegment hot path.  Plain pointers and sizes only; no torch / numpy /
 * C++ types cross this boundary.  All citations are relative to the reference repository root.
 *
 * What each entry point replaces
 * ------------------------------
 *   ps_knn_batch / ps_knn_batch_i64    cpp_knn_batch_omp / cpp_knn_batch / cpp_knn / cpp_knn_omp
 *                                      PointSegment/utils/nearest_neighbors/knn_.h:2-17, knn_.cxx:22-135,
 *                                      as bound by knn.pyx:33-109 and DataProcessing.knn_search
 *                                      (PointSegment/helper_tool.py:84-94)
 *   ps_pyramid_build                   tf_map's per-layer loop, PointSegment/runBraTS.py:147-156
 *                                      (== runPancreas.py:131-140)
 *   ps_grid_subsample                  grid_subsampling(), .../cpp_subsampling/grid_subsampling/
 *                                      grid_subsampling.h:84-91, as bound by wrapper.cpp:58-286 and
 *                                      DataProcessing.grid_sub_sampling (helper_tool.py:123-143)
 *   ps_volume_to_cloud                 load_volume's normalisation + convert_pc2ply's voxel -> point extraction,
 *                                      PointSegment/utils/dataPrepareBraTS.py:33-49, 75-89
 *   ps_randla_*                        Network.inference, PointSegment/RandLANet.py:110-152, and the blocks it
 *                                      calls (:314-401) with helper_tf_util.conv2d / conv2d_transpose
 *                                      (PointSegment/helper_tf_util.py:115-250) in inference mode
 *   ps_op_*                            the individual Network.* static methods (RandLANet.py:337-401) for
 *                                      callers that keep the reference's op-by-op graph
 *
 * Conventions
 * -----------
 *   - every function returns 0 on success, a PS_E* code otherwise; ps_last_error() returns a thread-local
 *     message (the reference's KNN entry points are `void` and abort on error, knn_.h:2-27; the grid op raises
 *     RuntimeError with a text, wrapper.cpp:76-229 -- the Python facade re-creates those).
 *   - all buffers are caller-owned.  `device_ptrs != 0`: pointers are device memory valid on the context's
 *     device and the call is asynchronous on the context's stream.  `device_ptrs == 0`: pointers are host
 *     memory; the call copies in/out and returns when the result is in the host buffer (NumPy drop-in mode).
 *   - the context owns a workspace that grows to the high-water mark and is then reused (no hipMalloc on the
 *     hot path after warm-up), one HIP stream (its own or one adopted with ps_set_stream), no global state.
 *   - arrays are dense row-major; indices are int32 on the wire (the reference casts its int64 to int32 at
 *     helper_tool.py:94).  Clouds of a batch are independent; point i of cloud b is row b*N + i.
 */
#ifndef POINTSEG_H
#define POINTSEG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PS_OK 0
#define PS_EINVAL 1  /* bad argument (shape, dtype flag, null pointer)            */
#define PS_EHIP 2    /* a HIP runtime call failed                                 */
#define PS_ENOMEM 3  /* workspace allocation failed                               */
#define PS_ESTATE 4  /* call made in the wrong state (e.g. forward before weights) */

#define PS_MAX_LAYERS 8

typedef struct ps_context ps_context;
typedef struct ps_randla ps_randla;

/* ---- context ------------------------------------------------------------------------------------------ */
int ps_create(int device, ps_context** out);
int ps_destroy(ps_context* ctx);
/* A new context launches on a private non-blocking stream.  ps_set_stream adopts the caller's hipStream_t instead
 * (e.g. torch's current stream, so that the caller's allocator and copies are ordered with the kernels);
 * hip_stream == NULL selects the device's default (null) stream. */
int ps_set_stream(ps_context* ctx, void* hip_stream);
int ps_synchronize(ps_context* ctx);
/* on != 0: ps_pyramid_build stops synchronising with the host; its device-side status words (tree deeper than the
 * traversal stack, builder queue overflow, unbalanced cloud needing the slow build path) are copied to pinned memory
 * and validated by the next ps_synchronize(), which then returns PS_ESTATE.  Lets the host enqueue the forward while
 * the pyramid is still being built.  Default off (every call validates before returning). */
int ps_set_deferred_checks(ps_context* ctx, int on);
/* on != 0: the op-level GEMMs of the training step -- ps_op_conv1x1[_ex] (forward and input-gradient GEMMs) on layers whose
 * channel count is a multiple of 16, and ps_op_linear_wgrad[_ex] -- round both operands to bf16 (round-to-nearest-even) as
 * they are loaded and accumulate in fp32 on v_mfma_f32_16x16x32_bf16 / 16x16x16_bf16: BASELINE configs[2]'s "bf16 MLPs"
 * (SURVEY 8d, config 3).  Activations, gradients, BatchNorm, softmax, loss and Adam stay fp32.  Default off (fp32 MFMA).
 * The fused inference path (ps_randla_forward) is never affected. */
int ps_set_train_gemm_bf16(ps_context* ctx, int on);
/* on != 0 (default): ps_op_conv1x1[_ex] runs its large, matrix-pipe-bound fp32 shapes (>= 16384 rows, cin >= 128 and a multiple of 32,
 * cout a multiple of 128: att_pooling's score products at d_out >= 256 and their input gradients) on v_mfma_f32_32x32x16_bf16 over exact
 * three-way bfloat16 splits of both operands with fp32 accumulation (csrc/gemm_b3.hip) -- fp32-level error at 2.7 x less matrix-pipe
 * time; on == 0: the fp32 MFMA for every shape.  Ignored while ps_set_train_gemm_bf16 is on. */
int ps_set_train_gemm_b3(ps_context* ctx, int on);
/* Matrix instruction of ps_randla_forward's matrix-pipe-bound stages: the fused attentive-pooling kernels at d_out >= 64 (att_pooling,
 * PointSegment/RandLANet.py:388-401, with LocSE and the neighbour gather fused in; csrc/attpool32b.hip) AND the dense layers of at
 * least 0.3 GFLOP -- [mlp2 ; shortcut] of levels 2-4, decoder_0, the first decoder steps (RandLANet.py:130-143, 314-321; csrc/gemm32b.hip).
 * on != 0 (default): v_mfma_f32_32x32x16_bf16 over exact three-way bfloat16 splits of the fp32 operands with fp32 accumulation --
 * fp32-level error, 2.7 x less matrix-pipe time; on == 0: the fp32 MFMA for all of them (csrc/attpool32.hip, csrc/gemm32.hip).  Both
 * meet the same parity bar. */
int ps_set_att_bf16x3(ps_context* ctx, int on);
const char* ps_last_error(void);
/* "pointseg-hip <version> gfx950" */
const char* ps_version(void);
/* Layout revision of the structs a caller fills (ps_pyramid, ps_train_options, ps_randla_config): bumped whenever one of them grows.
 * A host built against an older header would make the library read past its struct -- compare ps_abi_version() with the
 * PS_ABI_VERSION it was compiled with before the first call (point-unet_amd/_lib.py does).  6: ps_pyramid.built, ps_train_options.act_bf16. */
#define PS_ABI_VERSION 6
int ps_abi_version(void);

/* Per-call kernel timing on the context's stream, measured with hipEvents recorded on THAT stream.
 * ps_timing_begin arms it; every subsequent API call accumulates per-stage device time; ps_timing_end
 * synchronises and writes up to `cap` (name, ms, launches) rows.  Used by bench.py for the roofline line. */
typedef struct {
    char name[48];
    double ms;
    int64_t launches;
} ps_timing_row;
int ps_timing_begin(ps_context* ctx);
/* Restrict event recording to ONE stage (NULL or "" = all stages).  Each recorded event costs ~3 us of stream time, so
 * bench.py profiles every stage in a separate pass and keeps only the dominant stage's events in its timed region. */
int ps_timing_select(ps_context* ctx, const char* stage_name);
int ps_timing_end(ps_context* ctx, ps_timing_row* rows, int cap, int* n_rows);

/* ---- KNN ---------------------------------------------------------------------------------------------- */
/* Exact K nearest neighbours, squared L2 in fp32, ascending, equal-distance ties broken exactly as the
 * reference's nanoflann 1.2.3 kd-tree (leaf 10) visits them.  support f32[B,n_support,3],
 * queries f32[B,n_queries,3], out int32[B,n_queries,K].  dim must be 3.  If n_support < K the trailing
 * K-n_support slots of every row are written as 0 (the reference returns np.zeros there, knn.pyx:93). */
int ps_knn_batch(ps_context* ctx, const float* support, const float* queries, int64_t B, int64_t n_support,
                 int64_t n_queries, int64_t dim, int64_t K, int32_t* out_idx, int device_ptrs);
/* Same with the reference's wire type (`long* batch_indices`, knn_.h:15-17). */
int ps_knn_batch_i64(ps_context* ctx, const float* support, const float* queries, int64_t B,
                     int64_t n_support, int64_t n_queries, int64_t dim, int64_t K, int64_t* out_idx,
                     int device_ptrs);

/* ---- index pyramid ------------------------------------------------------------------------------------ */
/* For layer i (N_0 = n0, N_{i+1} = N_i / ratio[i]):
 *   xyz[i]        f32[B,N_i,3]    the first N_i points of every cloud ("random sampling" of a pre-shuffled
 *                                 cloud is a prefix slice, runBraTS.py:149)
 *   neigh_idx[i]  i32[B,N_i,K]    knn(xyz[i], xyz[i], K)
 *   sub_idx[i]    i32[B,N_{i+1},K] = neigh_idx[i][:, :N_{i+1}, :]
 *   interp_idx[i] i32[B,N_i,1]    knn(xyz[i+1], xyz[i], 1)
 * All buffers device memory, caller-allocated with exactly these shapes. */
typedef struct {
    int32_t num_layers;
    int32_t K;
    int64_t B;
    int64_t n[PS_MAX_LAYERS + 1];
    float* xyz[PS_MAX_LAYERS];
    int32_t* neigh_idx[PS_MAX_LAYERS];
    int32_t* sub_idx[PS_MAX_LAYERS];
    int32_t* interp_idx[PS_MAX_LAYERS];
    /* optional, may be NULL per layer: i32[B, n_l], a spatially coherent processing order of the layer's points -- the row of
     * the t-th point in kd-tree leaf order.  ps_pyramid_build fills it when a buffer is given; ps_randla_forward then walks the
     * points of the attentive-pooling kernels in that order, XCD by XCD (neighbour gathers hit the XCD's L2). */
    int32_t* order[PS_MAX_LAYERS];
    /* written by ps_pyramid_build (0 in a caller-filled struct): a stamp over the table pointers and shapes that says "sub_idx[i] IS the
     * first n[i+1] rows per cloud of neigh_idx[i]" -- what the deterministic max-pool backward of the training step relies on.  A caller
     * that rewrites a table of a built pyramid in place, or fills the struct by hand, must leave / set this to 0: the trainer then
     * compares the tables itself on every step. */
    uint64_t built;
} ps_pyramid;
int ps_pyramid_build(ps_context* ctx, const float* xyz0, int64_t B, int64_t n0, int32_t num_layers,
                     const int32_t* ratios, int32_t K, ps_pyramid* pyr);

/* ---- grid subsampling --------------------------------------------------------------------------------- */
/* Voxel-grid barycentre subsampling (points f32[n,3]; optional features f32[n,fdim]; optional classes
 * i32[n,ldim]).  Two-call protocol because M is data dependent: call with out_points == NULL to get *M, then
 * with buffers of M rows.  Rows are emitted in ascending cell-key order (the reference emits unordered_map
 * iteration order, grid_subsampling.cpp:85).  Majority-label ties resolve to the smallest label (the
 * reference's tie order is implementation defined, grid_subsampling.cpp:100-101). Host pointers only. */
int ps_grid_subsample(ps_context* ctx, const float* points, int64_t n, const float* features, int64_t fdim,
                      const int32_t* classes, int64_t ldim, float sampleDl, int64_t* M, float* out_points,
                      float* out_features, int32_t* out_classes);
/* The same on DEVICE memory, in one call (every pointer but M is a device pointer; inputs are read in place, the outputs are written by
 * the reduction kernel itself): `capacity` = rows the output buffers hold (n always suffices); *M = the sub-cloud's size (host word,
 * valid on return: the call synchronises twice, for the bounding box and for M, like the host form).  Lets the dataset preparation of
 * dataPrepareBraTS.py:75-116 run volume -> cloud -> grid -> 1-NN projection without leaving HBM (point-unet_amd/prepare.py). */
int ps_grid_subsample_dev(ps_context* ctx, const float* points, int64_t n, const float* features, int64_t fdim,
                          const int32_t* classes, int64_t ldim, float sampleDl, int64_t capacity, int64_t* M,
                          float* out_points, float* out_features, int32_t* out_classes);

/* ---- volume -> point cloud ------------------------------------------------------------------------------ */
/* First half of the reference's dataset preparation (PointSegment/utils/dataPrepareBraTS.py:33-49 itensity_normalize_one_volume,
 * :75-89 convert_pc2ply): `volumes` f32[4, X, Y, Z] (the four MR modalities, raw intensities), optional `seg` i32[X, Y, Z].
 * Every voxel where any z-scored modality (mean / population std of its voxels > 0, float64) is non-zero becomes a point, in
 * x-major order: xyz f32[n,3] = index / shape, colors f32[n,4], labels i32[n] (may be NULL), xyz_origin i32[n,3] (may be
 * NULL).  Two-call protocol like ps_grid_subsample: xyz == colors == NULL returns the count in *n; the second call takes
 * the row capacity in *n and returns the count.  Host pointers only. */
int ps_volume_to_cloud(ps_context* ctx, const float* volumes, const int32_t* seg, int64_t X, int64_t Y, int64_t Z, int64_t* n,
                       float* xyz, float* colors, int32_t* labels, int32_t* xyz_origin);
/* The same on DEVICE memory, in one call: volumes / seg and the four outputs are device pointers (xyz and colors required), *n holds the
 * row capacity of the outputs on entry (X*Y*Z always suffices) and the number of points on return (host word; one synchronisation). */
int ps_volume_to_cloud_dev(ps_context* ctx, const float* volumes, const int32_t* seg, int64_t X, int64_t Y, int64_t Z, int64_t* n,
                           float* xyz, float* colors, int32_t* labels, int32_t* xyz_origin);

/* ---- RandLA-Net forward ------------------------------------------------------------------------------- */
typedef struct {
    int32_t num_layers;            /* ConfigBraTS.num_layers (helper_tool.py:23)           */
    int32_t k_n;                   /* ConfigBraTS.k_n (helper_tool.py:22); 16 or 32        */
    int32_t num_classes;           /* helper_tool.py:26                                    */
    int32_t in_channels;           /* xyz + modalities: 7 BraTS (runBraTS.py:142), 4 Pancreas */
    int32_t d_out[PS_MAX_LAYERS];  /* helper_tool.py:36                                    */
} ps_randla_config;

int ps_randla_create(ps_context* ctx, const ps_randla_config* cfg, ps_randla** out);
int ps_randla_destroy(ps_randla* net);
/* Number of floats ps_randla_set_weights expects, and the layout (see DESIGN.md "weight blob"):
 * inference-mode BatchNorm folded into each conv's W and b on the host (point-unet_amd/weights.py). */
int64_t ps_randla_weight_count(const ps_randla* net);
int ps_randla_set_weights(ps_randla* net, const float* blob, int64_t count); /* host pointer */
/* features f32[B,N0,in_channels] -> logits f32[B,N0,num_classes]; device pointers. */
int ps_randla_forward(ps_randla* net, const ps_pyramid* pyr, const float* features, float* logits);
/* Debug/parity taps: copies an internal activation to a host buffer after the last forward.
 * which: 0 fc0 [N0,8]; 10+i enc_i [N_i,2d_i]; 20+i pool_i [N_{i+1},2d_i]; 30 decoder_0; 40+j dec_j.
 * Returns PS_EINVAL if count does not match the tensor's size. */
int ps_randla_tap(ps_randla* net, int which, float* host_out, int64_t count);
/* on != 0: forwards also STORE the activations nobody but ps_randla_tap reads -- today the output rows of the last decoder step
 * (tap 40 + num_layers - 1: 23 MB per 180 000-point cloud), which otherwise live only in the registers of the head's layer chain.
 * Off by default. */
int ps_randla_keep_taps(ps_randla* net, int on);

/* ---- op-by-op surface (Network.* static methods, RandLANet.py:337-401); device pointers --------------- */
/* gather_neighbour: pc f32[B,N,d], idx i32[B,M,K] -> out f32[B,M,K,d] */
int ps_op_gather_neighbour(ps_context* ctx, const float* pc, const int32_t* idx, int64_t B, int64_t N,
                           int64_t M, int64_t K, int64_t d, float* out);
/* relative_pos_encoding: xyz f32[B,N,3], idx i32[B,N,K] -> out f32[B,N,K,10] = [dis, rel, centre, nbr] */
int ps_op_relative_pos_encoding(ps_context* ctx, const float* xyz, const int32_t* idx, int64_t B, int64_t N,
                                int64_t K, float* out);
/* random_sample: feature f32[B,N,d], pool_idx i32[B,M,K] -> out f32[B,M,d] = max over K */
int ps_op_random_sample(ps_context* ctx, const float* feature, const int32_t* pool_idx, int64_t B, int64_t N,
                        int64_t M, int64_t K, int64_t d, float* out);
/* nearest_interpolation: feature f32[B,N,d], interp_idx i32[B,M,1] -> out f32[B,M,d] */
int ps_op_nearest_interpolation(ps_context* ctx, const float* feature, const int32_t* interp_idx, int64_t B,
                                int64_t N, int64_t M, int64_t d, float* out);
/* conv2d 1x1 (helper_tf_util.conv2d with BN folded): x f32[R,cin], w f32[cin,cout], b f32[cout] (may be
 * NULL) -> y f32[R,cout]; leaky != 0 applies LeakyReLU(0.2). */
int ps_op_conv1x1(ps_context* ctx, const float* x, const float* w, const float* b, int64_t R, int64_t cin,
                  int64_t cout, int leaky, float* y);
/* att_pooling up to (not including) its trailing conv2d: fset f32[R,K,d], wfc f32[d,d] ->
 * agg f32[R,d] = sum_K fset * softmax_K(fset . wfc) */
int ps_op_att_pool(ps_context* ctx, const float* fset, const float* wfc, int64_t R, int64_t K, int64_t d,
                   float* agg);

/* Widening of binary16 features to fp32 (BASELINE configs[4]: "mixed fp16 features / int32 KNN indices" -- the reference feeds
 * tf.float32 everywhere, runPancreas.py:110-118; a half-precision feature file is an input-format variant of this build).
 * Network.inference accepts float16 feature tensors and calls this in front of fc0.  Device pointers. */
int ps_op_half_to_float(ps_context* ctx, const uint16_t* in, int64_t n, float* out);

/* point -> volume scatter of the class probabilities (PointSegment/testBraTS.py:83-101, 226-231):
 * volume f32[Z, Y, X, C] (the layout after the reference's np.moveaxis(volume, 1, 2)); out[z,y,x,:] = softmax(logits[j])
 * for the LAST point i on voxel xyz_origin[i] = (x,y,z) and the LAST row j with p_idx[j] == i (p_idx NULL = identity),
 * zeros elsewhere.  scratch: total + Z*X*Y int32.  Device pointers. */
int ps_op_probs_to_volume(ps_context* ctx, const float* logits, int64_t n, int64_t C, const int32_t* p_idx,
                          const int32_t* xyz_origin, int64_t total, int64_t Z, int64_t X, int64_t Y, float* volume,
                          int32_t* scratch);

/* The op-level kernels of the training step -- forward / backward pairs of the ops above, BatchNorm in training mode, the fused and
 * recompute forms the native trainer chooses between, the deterministic scatter-adds, loss and Adam -- are declared in
 * pointseg_train_ops.h: the building blocks behind ps_randla_train_step (and of point-unet_amd/train.py's A/B tape), not needed by a
 * host that drives the path through the calls of this header. */

/* ---- the training step behind one call ----------------------------------------------------------------------------------
 * Replaces Network.__init__'s loss / optimizer and Network.train's sess.run([train_op, extra_update_ops, ...])
 * (PointSegment/RandLANet.py:62-90, 162-169, 267-274) over the graph of Network.inference in training mode (:110-152, 314-401;
 * tf.layers.batch_normalization(training=True), helper_tf_util.py:167,246; tf.nn.dropout before the last layer, :553-574).
 * csrc/trainer.hip holds the tape in C++: no Python, no torch in the loop.
 *
 * Memory: parameters, gradients, Adam moments and BatchNorm moving statistics are CALLER-OWNED flat fp32 device buffers
 * (ps_trainer_bind) of ps_trainer_param_count() (first four) and ps_trainer_buffer_count() floats; ps_trainer_layout enumerates
 * the named tensors inside them (the reference's variable names without the "layers/" scope, in graph order: for every layer
 * kernel / weights [cin,cout] ([cout,cin] for the transposed convolutions of the decoder), bias, BatchNorm gamma, beta; buffers:
 * moving_mean, moving_variance per BatchNorm).  Activations and gradients of a step live in a pool the trainer owns (grows to the
 * high-water mark during the first step, then reused; ps_trainer_pool_peak_bytes).
 *
 * Collectives: the library links no communication library.  A data-parallel host passes an all-reduce callback
 * (RCCL: `ncclAllReduce(buf, buf, count, dtype ? ncclDouble : ncclFloat, ncclSum, comm, (hipStream_t)hip_stream)`); the step
 * calls it once for the flat gradient buffer (then divides by world_size) and, with sync_bn != 0, twice per BatchNorm layer for
 * the [sum | sum of squares] / [sum g | sum g*xhat] vectors, so that "N GPUs x 1 cloud" is the same optimisation step as "1 GPU x N clouds"
 * -- exactly so when every rank holds the same number of NON-IGNORED points (always, without ignored labels): each rank's loss is the
 * mean over its own valid points and the ranks' gradients are averaged with equal weights, where the one-GPU batch takes one mean over
 * all valid points; with ignored labels unevenly spread over the ranks the two differ by that weighting.
 *
 * Labels: raw labels in [0, num_classes + num_ignored); the ignored ones are dropped and the rest renumbered like the reference's
 * reducing_list (RandLANet.py:68-81).  A label OUTSIDE that range is treated as ignored (the reference's tf.gather would raise).
 *
 * Pyramids: ps_randla_train_step / ps_randla_backward accept any caller-filled ps_pyramid.  The deterministic max-pool backward walks the
 * neighbour table's inverse index and therefore needs sub_idx[i] to be the first n[i+1] rows per cloud of neigh_idx[i] (what
 * ps_pyramid_build writes, and vouches for in ps_pyramid.built); for any other pyramid the trainer compares the two tables on EVERY step
 * (one synchronising check per level; only a negative answer is remembered, it selects the always-correct float-atomic form). */
typedef struct ps_trainer ps_trainer;
typedef struct {
    float learning_rate;            /* cfg.learning_rate (helper_tool.py:33); Adam beta1 0.9, beta2 0.999, eps 1e-8 (TF defaults) */
    float keep_prob;                /* dropout in front of the last layer, RandLANet.py:148 (0.5) */
    int32_t mlp_bf16;               /* BASELINE configs[2] "bf16 MLPs": the shared-MLP GEMMs round their operands to bf16 (fp32 accumulate);
                                     * the LocSE convolution 10 -> h (position encoding, K = 10) stays fp32 */
    int32_t fused_att;              /* attentive pooling (+ gather / concat / scatter-add) as one kernel per direction where compiled (d <= 64) */
    int32_t fused_locse;            /* the LocSE branch recomputed from coordinates and indices instead of materialised */
    int32_t num_ignored;            /* cfg.ignored_label_inds (RandLANet.py:68-81): labels dropped from the loss, <= 8 */
    int32_t ignored_label_inds[8];
    int32_t deterministic;          /* every scatter-add of the backward pass as a fixed-order gather-reduction over an inverse index built
                                     * once per level and step (csrc/invidx.hip): two runs of a step give bit-identical gradients */
    int32_t fused_convbn;           /* LFA mlp2 (conv h -> h + BatchNorm + LeakyReLU on the [N*K, h] rows, RandLANet.py:331) and the widening shared
                                     * MLPs (Encoder mlp2 / shortcut, fc1: ps_op_convbn_train_supported) with the pre-BatchNorm
                                     * product recomputed instead of stored (h <= 64; in the bf16-MLP mode the operands of its three products are
                                     * rounded like the GEMMs it replaces -- h % 16 == 0 --, the 8-channel layer stays fp32 in both forms) */
    int32_t overlap_wgrad;          /* the weight-gradient products of the backward pass on a second HIP stream of the trainer (they feed nothing
                                     * before the step's one reduction launch): same kernels, same results, bit for bit */
    int32_t act_bf16;               /* with mlp_bf16: the [N*K, h] activation rows of the LFA branch (the LocSE output and LFA mlp2's, RandLANet.py:325,
                                     * 331) are STORED as bfloat16 at the levels whose kernels read them that way (h = 8 / 32 / 64: levels 0-2 of the
                                     * BraTS network) -- what that mode's products round them to anyway; the weighted sum of att_pooling and the
                                     * 8-channel convolution then see the rounded values too -- and so are the rows of their GRADIENTS
                                     * (ps_set_train_act_bf16, pointseg_train_ops.h) */
} ps_train_options;
/* In-place sum over the ranks of `count` elements at device pointer `buf` (dtype 0: float32, 1: float64), ordered on `hip_stream`
 * (the context's stream).  Returns 0 on success. */
typedef int (*ps_allreduce_fn)(void* user, void* buf, int64_t count, int dtype, void* hip_stream);

int ps_trainer_create(ps_context* ctx, const ps_randla_config* cfg, const ps_train_options* opt, ps_trainer** out);
int ps_trainer_destroy(ps_trainer* t);
int64_t ps_trainer_param_count(const ps_trainer* t);   /* trainable floats: 4 992 852 for the BraTS model */
int64_t ps_trainer_buffer_count(const ps_trainer* t);  /* BatchNorm moving statistics */
int ps_trainer_layout_rows(const ps_trainer* t);
/* row-th named tensor: its offset (floats) and shape inside the parameter buffer (is_buffer 0) or the statistics buffer (1) */
int ps_trainer_layout(const ps_trainer* t, int row, char* name, int name_cap, int64_t* offset, int64_t* rows, int64_t* cols,
                      int* is_buffer);
/* device pointers; adam_m / adam_v may be NULL for a host that only calls ps_randla_backward */
int ps_trainer_bind(ps_trainer* t, float* params, float* grads, float* adam_m, float* adam_v, float* bn_buffers);
/* sync_bn: bit 0 = share the BatchNorm statistics over the ranks (two 2*C-float all-reduces per layer and step); PS_COLLECTIVE_AT_WORLD_ONE
 * = call fn even in a world of ONE rank.  Without that bit a one-rank world never calls fn (a one-rank sum is the identity) and keeps the
 * cheaper one-rank BatchNorm kernels: a host that always passes its callback gets the single-GPU step, bit for bit. */
#define PS_COLLECTIVE_AT_WORLD_ONE 2
int ps_trainer_set_collective(ps_trainer* t, ps_allreduce_fn fn, void* user, int world_size, int rank, int sync_bn);
int ps_trainer_set_options(ps_trainer* t, const ps_train_options* opt);  /* everything but the ignored labels */
int ps_trainer_set_step(ps_trainer* t, int64_t step);                    /* optimisation steps taken so far (checkpoint resume) */
int64_t ps_trainer_get_step(const ps_trainer* t);
int64_t ps_trainer_pool_peak_bytes(const ps_trainer* t);                 /* activation + gradient footprint of the last step */
/* on != 0: every step records hipEvents at its section boundaries (fc0, each encoder level, decoder, head; forward and backward
 * separately; loss, gradient all-reduce, Adam) and synchronises at its end; ps_trainer_profile returns the last step's rows
 * (name, device ms; `launches` unused).  Off by default. */
int ps_trainer_set_profile(ps_trainer* t, int on);
int ps_trainer_profile(const ps_trainer* t, ps_timing_row* rows, int cap, int* n_rows);
/* The collectives of the last step: calls into the ps_allreduce_fn callback (the flat-gradient all-reduce + two per BatchNorm layer
 * with sync_bn), bytes handed over, host time spent inside the callback, and -- on profiled steps (ps_trainer_set_profile) -- the
 * device time between an event pair around every call.  A NULL callback, or world_size 1 without PS_COLLECTIVE_AT_WORLD_ONE, means no
 * calls.  Any out pointer may be NULL. */
int ps_trainer_collective_stats(const ps_trainer* t, int64_t* calls, int64_t* bytes, double* host_ms, double* device_ms);
/* Training-mode forward + class-weighted cross-entropy + backward: fills the bound gradient buffer (this rank's gradients, no
 * collective), updates the BatchNorm moving statistics, writes the loss (device float) and optionally the logits
 * f32[B*N0, classes] (NULL: not wanted).  features f32[B,N0,in_channels], labels i32[B,N0], class_weights f32[classes]; device pointers. */
int ps_randla_backward(ps_trainer* t, const ps_pyramid* pyr, const float* features, const int32_t* labels,
                       const float* class_weights, float* loss, float* logits);
/* The whole optimisation step: ps_randla_backward + mean of the gradients over the ranks (if a collective is set) + Adam. */
int ps_randla_train_step(ps_trainer* t, const ps_pyramid* pyr, const float* features, const int32_t* labels,
                         const float* class_weights, float* loss, float* logits);

#ifdef __cplusplus
}
#endif
#endif /* POINTSEG_H */
