/*
 * kpx.h -- C ABI of libkpx_hip.so: the MI355X (gfx950) kernels behind the detector_translator
 * per-frame hot path.
 *
 * The reference (YunjiKim/Unsupervised-Keypoint-Learning-...) has no FFI / custom-op boundary: its
 * arithmetic is TensorFlow-1.12 ops called from Python.  Each entry point below therefore names the
 * reference call site(s) (file:line under /root/reference) whose TF op it replaces.
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes, no torch / C++ types; every pointer is DEVICE memory owned by the caller
 *     (the library never allocates, frees or keeps global state);
 *   - tensors are float32 NHWC; conv kernels HWIO [kh][kw][Cin][Cout]; "ld*" = pixel stride in floats
 *     (>= channel count) so that producers can write into / consumers read from channel slices of a wider
 *     buffer (this is how tf.concat(axis=-1) is realised without a copy);
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default stream);
 *   - return 0 on success, KPX_EINVAL (-1) for a bad argument, or -(hipError_t) if the launch failed;
 *     nothing throws across the ABI; re-entrant, callable from any host thread.  The only process state is a per-device
 *     "large-LDS attribute already set" bit per kernel family (std::atomic, idempotent) and the tuning switches (KPX_* environment
 *     variables), which are parsed ONCE, on first use (csrc/kpx_env.hip; no launch path calls getenv); kpx_reload_env() parses them again.
 */
#ifndef KPX_H
#define KPX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KPX_EINVAL (-1)
#define KPX_ABI_VERSION 3

enum { KPX_ACT_NONE = 0, KPX_ACT_RELU = 1, KPX_ACT_LRELU = 2 /* slope 0.01 */, KPX_ACT_TANH = 3 /* forward only */ };

int kpx_abi_version(void);

/* Parse the KPX_* tuning / debugging switches again (they are otherwise read once, on first use).  For harnesses that flip a switch
 * between launches (bench.py: KPX_NO_WINO=1 to time the direct kernel on the roofline layer).  Call it from the launching thread, between
 * launches.  Returns 0.  (No reference counterpart: TF-1.12 reads its TF_* switches the same way, once.) */
int kpx_reload_env(void);

/* `arith` argument of kpx_conv2d_fwd_f32 / _dgrad_f32 / _wgrad_f32: the arithmetic of the layers those entries run on the bf16 matrix
 * pipe (csrc/conv_gemm3.hip: strided / 4x4 / 1x1 layers and their weight gradients).  KPX_ARITH_F32: every fp32 operand is split exactly
 * into three bf16 terms and six partial products are accumulated in fp32 -- fp32-equivalent, the fp32 configuration (the fp32-MFMA
 * kernels the other shapes run on are unaffected by the argument).  KPX_ARITH_BF16: operands truncated to bf16, one product, fp32
 * accumulation -- the bf16 configuration (BASELINE configs[2]).  Per call: two models of different configurations can share a process.
 * No reference counterpart (TF-1.12 computes models/networks/layers.py:6-9 in the graph's one dtype). */
enum { KPX_ARITH_F32 = 0, KPX_ARITH_BF16 = 1 };

/* ---- convolution: replaces tf.pad + tf.layers.conv2d(padding='same') (models/networks/layers.py:6-9)
 *      and tf.nn.conv2d + bias_add + relu (models/networks/vgg.py:51-54).
 *      y[n,oh,ow,:] = act( sum_{r,q,c} x[n, oh*stride + r - pad_t, ow*stride + q - pad_l, c] * w[r,q,c,:] + bias )
 *      pad_t/pad_l are the TOTAL top/left padding (explicit tf.pad + TF's SAME split, decided by the caller);
 *      bottom/right padding is implied by Ho/Wo.  bias may be NULL.  fp32 MFMA implicit GEMM.
 *      `workspace` (optional, may be NULL): split-K scratch for small-M / long-K layers; size from
 *      kpx_conv2d_fwd_workspace_bytes (0 = not needed).  Without it the layer runs unsplit. */
size_t kpx_conv2d_fwd_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW);
int kpx_conv2d_fwd_f32(const float* x, int N, int Hi, int Wi, int Cin, int ldx,
                       const float* w_hwio, int KH, int KW, const float* bias,
                       float* y, int Ho, int Wo, int Cout, int ldy,
                       int stride, int pad_t, int pad_l, int act, int arith,
                       void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused Winograd F(2x2,3x3) path of the 3x3 stride-1 SAME convolution with PRE-TRANSFORMED filters (fp32 MFMA; same reference
 *      call sites as kpx_conv2d_fwd_f32 for kernel=3, stride=1).  kpx_conv2d_fwd_f32 / _dgrad_f32 take this path by themselves when a
 *      workspace is given (transforming the filter on every call); a caller that keeps U = G g G^T between calls -- constant filters
 *      (VGG19, vgg.py:57-61), or once per optimiser update for trainable ones -- uses these entry points directly:
 *        kpx_wino_u_bytes: size of U for a [3][3][Cin][Cout] filter (either direction);
 *        kpx_wino_filter_transform_f32: one filter; dgrad != 0 transforms the flipped / transposed filter of the data gradient;
 *        kpx_wino_filter_transform_batch_f32: n filters in ONE launch, `descs` = DEVICE array of KpxWinoDesc;
 *        kpx_conv3x3_wino_f32: out[N,H,W,Nn] = act(conv3x3_same(in[N,H,W,K]) + bias) with U for (K gathered, Nn produced) channels.
 *      Shapes: H, W multiples of 16 (or 8x8 images with N a multiple of 4), see kpx_conv3x3_wino_eligible. */
typedef struct KpxWinoDesc { const float* w; float* u; int cin, cout, dgrad, reserved; } KpxWinoDesc;
size_t kpx_wino_u_bytes(int Cin, int Cout);
int kpx_wino_filter_transform_f32(const float* w_hwio, int Cin, int Cout, int dgrad, float* u, void* stream);
int kpx_wino_filter_transform_batch_f32(const void* descs, int n, void* stream);
int kpx_conv3x3_wino_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in);
int kpx_conv3x3_wino_f32(const float* in, int N, int H, int W, int K, int ldin, const float* u, const float* bias,
                         float* out, int Nn, int ldout, int act, void* stream);
/* The same convolution, additionally writing the batch-norm statistics of its OUTPUT (tf.contrib.layers.batch_norm follows every
 * generator conv, models/networks/layers.py:13-14): tile_stats[N * (H/16) * (W/16)][2][Nn] = per 16x16-pixel tile and channel, the
 * sum and the sum of squares of out (fp32, fixed summation order).  kpx_bn_stats_from_tiles_f32 turns a tile range into mean /
 * invstd / moving statistics, replacing kpx_bn_stats_f32's pass over the activation.  H, W multiples of 16. */
size_t kpx_conv3x3_wino_stats_tiles(int N, int H, int W);
int kpx_conv3x3_wino_stats_f32(const float* in, int N, int H, int W, int K, int ldin, const float* u, const float* bias,
                               float* out, int Nn, int ldout, int act, float* tile_stats, void* stream);

/* ---- bf16 STORAGE (the bf16 configuration proper, BASELINE configs[2] "bf16 storage + fp32 accumulate"): the same layers.conv call sites
 *      (layers.py:4-10; networks/__init__.py:13-24,50-62,80-97; vgg.py:20-40) with bf16 activation tensors in HBM.  `in` is bf16
 *      [N,H,W,K] (pixel stride ldin ELEMENTS, a multiple of 8; 16-byte aligned), `out` bf16 [N,H,W,Nn] (out_f32 = 0; Nn, ldout multiples
 *      of 8) or fp32 (out_f32 = 1; multiples of 4).  Filters stay fp32 masters; kpx_conv3x3_bf16s_prepare_f32 (or the batched form: one
 *      launch per optimiser update over a device table of {const float* w; void* wf; int Cin, Cout, dgrad, NB = 4 * ceil(Nn / 128)}
 *      entries, 32 bytes each) writes the fragment-ordered bf16 copy of one direction (kpx_conv3x3_bf16s_weights_bytes(K, Nn) bytes).
 *      out = mask_gate( act( conv3x3_same(in, wf) + bias ) ): `mask` (optional, bf16 [N,H,W,Nn], pixel stride ldmask) zeroes the output
 *      where mask <= 0 (ReLU backward of the tensor a data gradient belongs to); `stats` (optional) receives per-workgroup channel sums
 *      and sums of squares of the fp32 results before the activation, [kpx_conv3x3_bf16s_stats_tiles(...)][2][Nn] floats, for the batch
 *      norm that follows (layers.py:13-14).  Shapes: W a multiple of 32, or W = 16 / 8 (whole images per tile); K a multiple of 32 or
 *      8 / 16; kpx_conv3x3_bf16s_eligible tells. */
size_t kpx_conv3x3_bf16s_weights_bytes(int K, int Nn);
int kpx_conv3x3_bf16s_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in);
int kpx_conv3x3_bf16s_stats_tiles(int N, int H, int W, int K, int Nn);
int kpx_conv3x3_bf16s_prepare_f32(const float* w_hwio, int Cin, int Cout, int dgrad, void* wf, void* stream);
int kpx_conv3x3_bf16s_prepare_batch_f32(const void* table, int ndesc, void* stream);
int kpx_conv3x3_bf16s(const void* in, int N, int H, int W, int K, int ldin, const void* wf, const float* bias,
                      void* out, int Nn, int ldout, int out_f32, int act, const void* mask, int ldmask, float* stats, void* stream);
/* Data gradient towards bn_y = relu(batch_norm(.)) (bf16 [N,H,W,Nn]) with that batch norm's backward sums reduced in the epilogue: out = conv3x3(in, wf)
 * gated by bn_y > 0, stats[tile][2][Nn] = sum(dz), sum(dz * (bn_y - beta)) per workgroup -- kpx_bn_train_bwd_bf16 takes them as tile_stats and skips
 * its reduction pass over (dz, x) (the fp32 configuration's kpx_conv3x3_wino43_bnbwd_stats_f32). */
int kpx_conv3x3_bf16s_bnbwd(const void* in, int N, int H, int W, int K, int ldin, const void* wf, void* out, int Nn, int ldout,
                            const void* bn_y, int ld_bn_y, const float* beta, float* stats, void* stream);

/* dx = d(loss)/dx given dy (gradient of the conv output BEFORE activation).  Writes every element of dx.
 * stride <= 2.  `workspace`: optional split-K scratch, see kpx_conv2d_fwd_f32. */
size_t kpx_conv2d_dgrad_workspace_bytes(int N, int Hi, int Wi, int Cin, int Cout, int KH, int KW, int stride);
int kpx_conv2d_dgrad_f32(const float* dy, int N, int Ho, int Wo, int Cout, int lddy,
                         const float* w_hwio, int KH, int KW,
                         float* dx, int Hi, int Wi, int Cin, int lddx,
                         int stride, int pad_t, int pad_l, int arith,
                         void* workspace, size_t workspace_bytes, void* stream);

/* The same data gradient multiplied by the activation backward of the tensor it belongs to: dx = dgrad(dy) * act_in'(y_in), y_in = the
 * ACTIVATED tensor (same shape as dx, pixel stride ld_y_in), act_in = KPX_ACT_RELU / KPX_ACT_LRELU.  This is the chain
 * conv -> leaky_relu -> conv of models/networks/__init__.py:141-151 (img_discr) walked backwards with the leaky-ReLU backward (tf.gradients
 * of tf.nn.leaky_relu) applied in the epilogue of the data gradient ABOVE it instead of a pass of its own (kpx_act_bwd_f32).  Fused on the
 * implicit-GEMM kernels (and their split-K reduce); the specialised kernels take one extra pass over dx (contiguous tensors). */
int kpx_conv2d_dgrad_act_f32(const float* dy, int N, int Ho, int Wo, int Cout, int lddy,
                             const float* w_hwio, int KH, int KW,
                             float* dx, int Hi, int Wi, int Cin, int lddx,
                             int stride, int pad_t, int pad_l, int arith,
                             const float* y_in, int ld_y_in, int act_in,
                             void* workspace, size_t workspace_bytes, void* stream);

/* dw[r,q,c,k] = sum_{n,oh,ow} x[n, oh*s+r-pad_t, ow*s+q-pad_l, c] * dy[n,oh,ow,k].
 * `workspace` holds split-K partial slabs; query its size with kpx_conv2d_wgrad_workspace_bytes. */
size_t kpx_conv2d_wgrad_workspace_bytes(int N, int Ho, int Wo, int Cin, int Cout, int KH, int KW);
int kpx_conv2d_wgrad_f32(const float* x, int N, int Hi, int Wi, int Cin, int ldx,
                         const float* dy, int Ho, int Wo, int Cout, int lddy,
                         float* dw_hwio, int KH, int KW, int stride, int pad_t, int pad_l, int arith,
                         void* workspace, size_t workspace_bytes, void* stream);

/* dz = dy * act'(y) (y = the activated conv output; dz may alias dy): relu / leaky_relu(0.01) / tanh backward
 * (tf.nn.relu vgg.py:54, tf.nn.leaky_relu networks/__init__.py:145,148, tf.tanh layers.py:28 in stage-2 training). */
int kpx_act_bwd_f32(const float* dy, const float* y, float* dz, size_t n, int act, void* stream);

/* ---- per-channel reductions over P pixels: sum[c] = sum_p x[p,c] (double accumulation).
 *      Used for the conv bias gradient (tf.layers.conv2d use_bias, layers.py:9).
 *      `scratch` (also for kpx_bn_stats_f32 / kpx_bn_bwd_f32) needs kpx_chan_reduce_scratch_bytes(C) bytes. */
size_t kpx_chan_reduce_scratch_bytes(int C);
int kpx_chan_sum_f32(const float* x, size_t P, int C, int ldx, float* sum_out, void* scratch, void* stream);

/* ---- batch norm: replaces tf.contrib.layers.batch_norm(eps=1e-5, center, scale) (layers.py:13-14).
 * train forward, step 1: batch mean / biased variance over P pixels (fp64 accumulation) ->
 *   mean[C], invstd[C] = rsqrt(var+eps); if moving_mean/moving_var != NULL they are updated in place with
 *   decay (TF fused-BN rule: unbiased variance, moving -= (moving-batch)*(1-decay)). */
int kpx_bn_stats_f32(const float* x, size_t P, int C, int ldx, float eps,
                     float* mean, float* invstd, float* var_biased,
                     float* moving_mean, float* moving_var, float decay,
                     void* scratch, void* stream);
/* Data gradient of a 3x3 layer whose INPUT was y = relu(BN(x)): out = dgrad (written as usual) and, per 16x16-pixel tile and channel,
 * tile_stats[tile][2][Nn] = sum(dz), sum(dz * (bn_y - bn_beta)) with dz = out * [bn_y > 0] -- the two reductions of that batch norm's
 * backward (bn_y: its output at the same pixels / channels, pixel stride ld_bn_y).  kpx_bn_bwd_from_tiles_f32 consumes them. */
int kpx_conv3x3_wino_bnbwd_stats_f32(const float* in, int N, int H, int W, int K, int ldin, const float* u,
                                     float* out, int Nn, int ldout, const float* bn_y, int ld_bn_y, const float* bn_beta,
                                     float* tile_stats, void* stream);
/* Fused Winograd F(4x4,3x3): the same contract as kpx_conv3x3_wino_f32 (forward, or data gradient with dgrad-transformed filters) with
 * 36 instead of 144 multiplies per 4x4 outputs; fp32 error ~2-3e-6 rel-L2 (F(2x2,3x3): ~4e-7).  Eligible when H % 16 == 0, W % 32 == 0,
 * K >= 16, Nn > 32 and the usual 16-B alignment; `u` comes from kpx_wino43_filter_transform(_batch)_f32 (kpx_wino43_u_bytes bytes,
 * descriptors as for kpx_wino_filter_transform_batch_f32).  Replaces the same tf.layers.conv2d call sites (reference
 * models/networks/__init__.py:13,22,80-97; models/networks/vgg.py:51). */
int kpx_conv3x3_wino43_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in_ptr);
size_t kpx_wino43_u_bytes(int Cin, int Cout);
int kpx_wino43_filter_transform_f32(const float* w_hwio, int Cin, int Cout, int dgrad, float* u, void* stream);
int kpx_wino43_filter_transform_batch_f32(const void* descs_dev, int n, void* stream);
int kpx_conv3x3_wino43_f32(const float* in, int N, int H, int W, int K, int ldin, const float* u, const float* bias,
                           float* out, int Nn, int ldout, int act, void* stream);
/* The same with the batch-norm statistics of the output from the epilogue, as kpx_conv3x3_wino_stats_f32 but per 4 x 16-pixel strip:
 * tile_stats[strip][2][Nn], kpx_conv3x3_wino43_stats_tiles(N, H, W) strips, consumed by kpx_bn_stats_from_tiles_f32 (tile_pixels 64). */
/* ... and with VGG19's two epilogue options: mask_y (or NULL): the output is zeroed where mask_y <= 0 (the ReLU backward of the tensor this
 * data gradient belongs to); pool_y (or NULL): the 2x2 max-pool of the activated output is written too ([N,H/2,W/2,Nn]).  Nn % 64 == 0. */
int kpx_conv3x3_wino43_ex_f32(const float* in, int N, int H, int W, int K, int ldin, const float* u, const float* bias,
                              float* out, int Nn, int ldout, int act, const float* mask_y, int ld_mask, float* pool_y, int ld_pool, void* stream);
size_t kpx_conv3x3_wino43_stats_tiles(int N, int H, int W);
int kpx_conv3x3_wino43_stats_f32(const float* in, int N, int H, int W, int K, int ldin, const float* u, const float* bias,
                                 float* out, int Nn, int ldout, int act, float* tile_stats, void* stream);
/* The same F(4x4,3x3) layers FP32-EQUIVALENT ON THE BF16 MATRIX PIPE (csrc/conv_wino43b.hip): U pre-split into three exact bf16 terms in
 * fragment order by kpx_wino43b_filter_transform(_batch)_f32 (kpx_wino43b_u_bytes bytes; descriptors as for
 * kpx_wino_filter_transform_batch_f32), V split in registers after the input transform, six v_mfma_f32_32x32x16_bf16 products per block,
 * fp32 accumulate: the fp32 configuration's arithmetic (error vs float64 as the fp32-MFMA kernel's), not the bf16 mode.  Eligible when
 * (H % 16 == 0 and W % 32 == 0, or 16 x 16 images in even number: two to a workgroup, no tile_stats), K >= 16, K % 4 == 0, Nn > 32, ldin >= K,
 * 16-B alignment.  One entry for every form of the launch: tile_stats
 * alone = kpx_conv3x3_wino43_stats_f32's statistics strips (kpx_conv3x3_wino43_stats_tiles); tile_stats + bn_y + bn_beta =
 * kpx_conv3x3_wino43_bnbwd_stats_f32 (no bias / act); mask_y / pool_y = kpx_conv3x3_wino43_ex_f32's options (Nn % 64 == 0); all NULL = the
 * plain convolution.  Replaces the same tf.layers.conv2d call sites (reference models/networks/layers.py:4-10 on
 * models/networks/__init__.py:13,22,80-97; models/networks/vgg.py:51). */
int kpx_conv3x3_wino43b_eligible(int N, int H, int W, int K, int Nn, int ldin, const void* in_ptr);
size_t kpx_wino43b_u_bytes(int Cin, int Cout);
int kpx_wino43b_filter_transform_f32(const float* w_hwio, int Cin, int Cout, int dgrad, void* u, void* stream);
int kpx_wino43b_filter_transform_batch_f32(const void* descs_dev, int n, void* stream);
int kpx_conv3x3_wino43b_f32(const float* in, int N, int H, int W, int K, int ldin, const void* u, const float* bias,
                            float* out, int Nn, int ldout, int act, const float* mask_y, int ld_mask, float* pool_y, int ld_pool,
                            float* tile_stats, const float* bn_y, int ld_bn_y, const float* bn_beta, void* stream);
/* 3x3 stride-1 SAME convolution PRODUCING exactly 16 channels on 16x16x4 MFMA blocks (no cout padding): the forward of the key-point
 * detector's last decoder block (64 -> 16, 16 -> 16 at full resolution, reference models/networks/__init__.py:50-54) and the data gradient
 * of its 16 -> 16 layer.  w_hwio is the layer's own filter: [3][3][K][16] forward, [3][3][16][K] for dgrad (K = gathered channels, a
 * multiple of 16; H, W multiples of 16).  tile_stats (or NULL): per 16x16-pixel tile sums as kpx_conv3x3_wino_stats_f32 writes them. */
int kpx_conv3x3_c16_eligible(int N, int H, int W, int K, int Nn, int ldin, int ldout, const void* in_ptr);
int kpx_conv3x3_c16_f32(const float* in, int N, int H, int W, int K, int ldin, const float* w_hwio, int dgrad, const float* bias,
                        float* out, int ldout, int act, float* tile_stats, void* stream);
/* kpx_bn_bwd_f32 (act must be KPX_ACT_RELU) with the channel reductions taken from those tile sums over tiles [tile0, tile0+ntiles). */
int kpx_bn_bwd_from_tiles_f32(const float* dy, int lddy, const float* x, int ldx, size_t P, int C,
                              const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                              float* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                              const float* tile_stats, size_t tile0, size_t ntiles, void* scratch, void* stream);
/* kpx_bn_stats_f32 from per-tile sums (tile_stats[tile][2][C], `tile_pixels` pixels per tile) over tiles [tile0, tile0 + ntiles):
 * fp64 fixed-order reduction; same outputs and moving-statistics rule. */
int kpx_bn_stats_from_tiles_f32(const float* tile_stats, size_t tile0, size_t ntiles, int tile_pixels, int C, float eps,
                                float* mean, float* invstd, float* var_biased,
                                float* moving_mean, float* moving_var, float decay, void* stream);
/* invstd[c] = rsqrt(moving_var[c]+eps) for inference-mode BN (models/keypoint_model.py:48-50). */
/* Train-mode batch norm of `groups` weight-sharing calls batched along N (the two pose_encoder calls of a pair, reference
 * detector_translator_model.py:166-167, each with batch statistics of its own; layers.py:13-14): x = [groups][P pixels][C], one launch per
 * phase for all groups.  Forward: statistics per group -- from tile_stats (per-tile sums of a convolution epilogue, tiles_per_group tiles per
 * group, or NULL: a reduction pass over x) -- mean / invstd [groups][C] (outputs, kept for the backward), moving statistics updated once per
 * group IN ORDER, then y = act(gamma * (x - mean) * invstd + beta).  Backward: dx, and dgamma / dbeta summed over the groups in order
 * (accumulate != 0: added to what the destinations hold).  Bitwise identical to `groups` calls of kpx_bn_stats(_from_tiles)_f32 +
 * kpx_bn_apply_f32 / of kpx_bn_bwd_f32.  scratch: kpx_bn_train_scratch_bytes(C, groups) bytes. */
size_t kpx_bn_train_scratch_bytes(int C, int groups);
int kpx_bn_train_fwd_f32(const float* x, size_t P, int groups, int C, int ldx, const float* tile_stats, size_t tiles_per_group,
                         float eps, const float* gamma, const float* beta, float* mean, float* invstd,
                         float* moving_mean, float* moving_var, float decay, float* y, int ldy, int act, void* scratch, void* stream);
int kpx_bn_train_bwd_f32(const float* dy, int lddy, const float* x, int ldx, size_t P, int groups, int C,
                         const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                         float* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                         const float* tile_stats, size_t tiles_per_group, void* scratch, void* stream);
/* (backward tile_stats, or NULL: per-tile sum(dz), sum(dz * (y - beta)) written by the epilogue of the data-gradient kernel that produced dy --
 *  kpx_conv3x3_wino43_bnbwd_stats_f32 / kpx_conv3x3_wino_bnbwd_stats_f32, tiles_per_group tiles per group -- instead of the reduction pass over
 *  (dy, x); dy must then already be zero wherever the batch norm's ReLU output is, which those kernels guarantee.) */
int kpx_conv3x3_wino43_bnbwd_stats_f32(const float* in, int N, int H, int W, int K, int ldin, const float* u,
                                       float* out, int Nn, int ldout, const float* bn_y, int ld_bn_y, const float* bn_beta,
                                       float* tile_stats, void* stream);
/* Inference-mode batch norm folded into the convolution in front of it (is_training = False: reference keypoint_model.py:48-50,
 * final_model.py:62,68,95): w_out[r][c] = w[r][c] * s[c], b_out[c] = (bias[c] - moving_mean[c]) * s[c] + beta[c] with
 * s = gamma * rsqrt(moving_var + eps); w = [rows = kh*kw*Cin][C] (HWIO), bias may be NULL.  conv(x, w_out) + b_out followed by ReLU then
 * equals relu(batch_norm(conv(x, w) + bias)) up to fp32 rounding, without the pass over the activation. */
int kpx_bn_fold_conv_f32(const float* w, const float* bias, size_t rows, int C, const float* gamma, const float* beta,
                         const float* moving_mean, const float* moving_var, float eps, float* w_out, float* b_out, void* stream);
int kpx_bn_invstd_f32(const float* var, int C, float eps, float* invstd, void* stream);
/* step 2: y = act((x-mean)*invstd*gamma + beta), act in {NONE, RELU} (tf.nn.relu networks/__init__.py:12...). */
int kpx_bn_apply_f32(const float* x, size_t P, int C, int ldx, const float* mean, const float* invstd,
                     const float* gamma, const float* beta, float* y, int ldy, int act, void* stream);
/* backward of y = act(BN_train(x)): dx, dgamma, dbeta from dy (gradient wrt y) and the saved x, mean, invstd.
 * accumulate != 0: dgamma / dbeta are ADDED to (a second call of a weight-sharing batch norm with its own statistics,
 * models/detector_translator_model.py:166-167). */
int kpx_bn_bwd_f32(const float* dy, int lddy, const float* x, int ldx, size_t P, int C,
                   const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                   float* dx, int lddx, float* dgamma, float* dbeta, int accumulate, void* scratch, void* stream);

/* ---- tf.image.resize_images(x, 2x) legacy bilinear (models/networks/__init__.py:63,98). */
int kpx_resize2x_fwd_f32(const float* x, int N, int H, int W, int C, int ldx, float* y, int ldy, void* stream);
int kpx_resize2x_bwd_f32(const float* dy, int N, int H, int W, int C, int lddy, float* dx, int lddx, void* stream);

/* ---- channel-slice copy: one half of tf.concat(axis=-1) (networks/__init__.py:44, detector_translator_model.py:170). */
int kpx_copy_channels_f32(const float* src, int ldsrc, float* dst, int lddst, size_t P, int C, void* stream);

/* ---- key-point head: model_utils.get_coord twice + tf.stack (utils/model.py:63-70, networks/__init__.py:68-72).
 * logits [B,H,W,K] -> mu [B,K,2] (x,y) in [-1,1]; also the two softmax profiles prob_y [B,H,K], prob_x [B,W,K].
 * scratch: kpx_keypoint_head_scratch_bytes(B,H,W,K). */
size_t kpx_keypoint_head_scratch_bytes(int B, int H, int W, int K);
int kpx_keypoint_head_fwd_f32(const float* logits, int B, int H, int W, int K,
                              float* mu, float* prob_y, float* prob_x, void* scratch, void* stream);
int kpx_keypoint_head_bwd_f32(const float* dmu, const float* mu, const float* prob_y, const float* prob_x,
                              int B, int H, int W, int K, float* dlogits, void* stream);

/* ---- 1x1 head + key-point head in one: layers.conv(x, n_pts, kernel=1) followed by the two get_coord calls
 * (models/networks/__init__.py:54,68-72; utils/model.py:63-70).  The axis means commute with the 1x1 projection, so the logits
 * [B,H,W,K] are never formed.  x [B,H,W,C] contiguous (C % 4 == 0), wk [C,K] (the HWIO 1x1 filter), bias [K] or NULL.
 * Forward also returns the row / column sums of x (xs_y [B,H,C], xs_x [B,W,C]) that the backward needs.
 * Backward: dx [B,H,W,C] (or NULL), dw [C,K] and db [K] (each may be NULL; accumulate != 0 adds into them).
 * scratch: kpx_keypoint_head_proj_scratch_bytes(B,H,W,C,K) for either direction. */
size_t kpx_keypoint_head_proj_scratch_bytes(int B, int H, int W, int C, int K);
/* 1 when the folded operator takes the shape (C % 4 == 0, C <= 256, K <= 64, H, W in [2, 512], its profile rows within 64 KB of LDS);
 * otherwise the caller runs kpx_conv2d_fwd_f32 (1x1) followed by kpx_keypoint_head_fwd_f32 -- the same reference call sites. */
int kpx_keypoint_head_proj_eligible(int B, int H, int W, int C, int K);
int kpx_keypoint_head_proj_fwd_f32(const float* x, const float* wk, const float* bias, int B, int H, int W, int C, int K,
                                   float* mu, float* prob_y, float* prob_x, float* xs_y, float* xs_x, void* scratch, void* stream);
int kpx_keypoint_head_proj_bwd_f32(const float* dmu, const float* mu, const float* prob_y, const float* prob_x,
                                   const float* xs_y, const float* xs_x, const float* wk, int B, int H, int W, int C, int K,
                                   float* dx, float* dw, float* db, int accumulate, void* scratch, void* stream);

/* ---- Gaussian heat-map render: model_utils.get_gaussian_maps (utils/model.py:49-60).
 * mu [B,K,2] (x,y) -> maps [B,H,W,K] written with pixel stride ldy (NHWC directly; no BKHW transpose pass). */
int kpx_gaussian_maps_fwd_f32(const float* mu, int B, int K, int H, int W, double inv_std, float* maps, int ldy, void* stream);
int kpx_gaussian_maps_bwd_f32(const float* dmaps, int lddy, const float* mu, int B, int K, int H, int W, double inv_std,
                              float* dmu, void* stream);

/* ---- translator heads + blend: mask = sigmoid(raw[...,3]); final = im*mask + crude*(1-mask)
 *      (networks/__init__.py:87-89, detector_translator_model.py:174).  raw4 [P,4] = crude(3) ‖ mask logit(1). */
int kpx_head_blend_fwd_f32(const float* im, const float* raw4, size_t P, float* final_out, float* crude_out, float* mask_out, void* stream);
int kpx_head_blend_bwd_f32(const float* dfinal, const float* im, const float* raw4, size_t P, float* draw4, void* stream);

/* ---- VGG input transform: rgb in [-1,1] -> (x+1)/2*255 -> BGR - mean
 *      (detector_translator_model.py:262-263, vgg.py:16-19). */
int kpx_vgg_prep_fwd_f32(const float* rgb, size_t P, float* bgr, void* stream);
int kpx_vgg_prep_bwd_f32(const float* dbgr, size_t P, float* drgb, void* stream);

/* ---- tf.nn.max_pool 2x2 s2 SAME (vgg.py:45-46). */
int kpx_maxpool2_fwd_f32(const float* x, int N, int H, int W, int C, float* y, void* stream);
int kpx_maxpool2_bwd_f32(const float* dy, const float* x, int N, int H, int W, int C, float* dx, void* stream);

/* ---- perceptual L1 term: loss = mean|f[0:half] - f[half:2*half]| (detector_translator_model.py:280-284).
 * fwd writes one float; bwd writes d/d f_pred (second half only) = -sign(gt-pred) * g,
 * g = gscale_host * (gscale_dev ? *gscale_dev : 1).  scratch >= 8 KiB. */
int kpx_l1_pair_fwd_f32(const float* f, size_t half, float* loss_out, void* scratch, void* stream);
int kpx_l1_pair_bwd_f32(const float* f, size_t half, const float* gscale_dev, float gscale_host, float* dpred, void* stream);
/* Gradient arriving at a VGG19 feature y = relu(conv) in one pass: ReLU mask of (max-pool backward of dy_pooled + L1 backward), i.e.
 * kpx_maxpool2_bwd_f32 + kpx_l1_pair_bwd_f32 + their sum + the ReLU backward fused (reference vgg.py:43,45-55 and
 * detector_translator_model.py:274-289).  f = [gt; pred] halves of the feature [2B,H,W,C], C % 4 == 0; dy_pooled [B,ceil(H/2),ceil(W/2),C]
 * or NULL for the last feature; d [B,H,W,C]. */
int kpx_vgg_feat_bwd_f32(const float* f, size_t half, const float* gscale_dev, float gscale_host, const float* dy_pooled,
                         int B, int H, int W, int C, float* d, void* stream);

/* ---- tf.nn.sigmoid_cross_entropy_with_logits + reduce_mean (detector_translator_model.py:249-254,265-267).
 * labels: the first n0 logits get label0, the next n1 get label1 (n1 may be 0).
 * loss_out[3] = { mean(group0) + mean(group1), mean(group0), mean(group1) }.
 * bwd: dlogits = (sigmoid(x) - label) * g / n_group, g = gscale_host * (gscale_dev ? *gscale_dev : 1). */
int kpx_sigmoid_xent_fwd_f32(const float* logits, size_t n0, float label0, size_t n1, float label1, float* loss_out, void* stream);
int kpx_sigmoid_xent_bwd_f32(const float* logits, size_t n0, float label0, size_t n1, float label1,
                             const float* gscale_dev, float gscale_host, float* dlogits, void* stream);

/* ---- tf.train.AdamOptimizer.apply (detector_translator_model.py:198-202) over one flat bucket.
 * alpha = lr*sqrt(1-b2^t)/(1-b1^t) (host); g is pre-scaled by gscale (1/world_size for DP). */
int kpx_adam_tf_flat_f32(float* p, const float* g, float* m, float* v, size_t n,
                         float alpha, float beta1, float beta2, float eps, float gscale, void* stream);
/* The same update with alpha read from DEVICE memory (one float): for launches captured into a HIP graph, whose arguments are frozen at
 * capture time while alpha (learning-rate decay :193-195, bias correction) changes every step; the caller refreshes *alpha_dev before each replay. */
int kpx_adam_tf_flat_dev_alpha_f32(float* p, const float* g, float* m, float* v, size_t n,
                                   const float* alpha_dev, float beta1, float beta2, float eps, float gscale, void* stream);

/* ---- evaluate.py / FinalModel rollout (forward only).  Dense layers run as 1x1 kpx_conv2d_fwd_f32.
 * LSTMCell pointwise part (models/networks/layers.py:17-21): gates [B,4U] = (i,j,f,o) pre-activations. */
int kpx_lstm_pointwise_f32(const float* gates, const float* c_prev, float forget_bias, float* c_out, float* h_out,
                           int B, int U, void* stream);
/* tf.tile over a new time axis (models/final_model.py:58-66,85-87): dst[((b*T+t)*pix+p)*lddst+c] = src[(b*pix+p)*ldsrc+c]. */
int kpx_tile_batch_f32(const float* src, int ldsrc, int B, int T, int pix, int C, float* dst, int lddst, void* stream);
/* blend with the T-times tiled source image + optional clip_by_value(-1,1) of crude and final (final_model.py:95-99). */
int kpx_head_blend_tiled_fwd_f32(const float* im, const float* raw4, size_t P, int HW, int T, int clip,
                                 float* final_out, float* crude_out, float* mask_out, void* stream);

/* ---- utilities */
int kpx_fill_f32(float* p, size_t n, float value, void* stream);
int kpx_axpy_f32(float* y, const float* x, size_t n, float a, void* stream);   /* y += a*x (gradient accumulation) */

/* Stage-2 training (next row, SURVEY 8f-4 last item: models/motion_generator_model.py).
 * LSTMCell backward of one step (layers.py:17-21): gates [B,4U] pre-activations (i,j,f,o), c_prev [B,U] (NULL = zeros),
 * dh [B,U], dc_in [B,U] (NULL = zeros) -> dgates [B,4U], dc_prev [B,U]. */
int kpx_lstm_pointwise_bwd_f32(const float* gates, const float* c_prev, const float* dh, const float* dc_in, float forget_bias,
                               float* dgates, float* dc_prev, int B, int U, void* stream);
/* Whole-sequence LSTM layer with zero initial state (dynamic_rnn / unrolled cells, networks/__init__.py:105-138); the time loop
 * runs inside the library.  x [T,B,In], kernel [In+U,4U], bias [4U]; xin [T,B,In+U], gates [T,B,4U], cs, hs [T,B,U] are written
 * (and are what the backward needs); zeros_bu = B*U zero floats; workspace as for kpx_conv2d_fwd_f32 / _dgrad_f32 of a
 * [B,1,1,In+U] x [1,1,In+U,4U] convolution.  Backward: dhs [T,B,U] -> dgates [T,B,4U] and (if non-NULL) dx [T,B,In];
 * scratch dxin [B,In+U], dh [B,U], dc0 [B,U] (zero-filled by the caller), dc1 [B,U].  The weight gradient is then ONE
 * kpx_conv2d_wgrad_f32 over xin / dgates viewed as [T*B,1,1,.], the bias gradient one kpx_chan_sum_f32 of dgates. */
int kpx_lstm_layer_fwd_f32(const float* x, int T, int B, int In, const float* kernel, const float* bias, int U,
                           float* xin, float* gates, float* cs, float* hs, const float* zeros_bu,
                           void* workspace, size_t workspace_bytes, void* stream);
int kpx_lstm_layer_bwd_f32(const float* dhs, int T, int B, int In, const float* kernel, int U,
                           const float* gates, const float* cs, float* dgates, float* dx,
                           float* dxin, float* dh, float* dc0, float* dc1,
                           void* workspace, size_t workspace_bytes, void* stream);
/* z = mu + stddev*eps and the KL term of motion_generator_model.py:146,291-293 on logit = [mu | stddev] [B,2V]; backward from dz
 * (NULL = zeros) and the KL gradient gkl_dev[0] * gkl_host. */
int kpx_vae_sample_kl_fwd_f32(const float* logit, const float* eps, float* z, float* kl_out, int B, int V, void* stream);
int kpx_vae_sample_kl_bwd_f32(const float* logit, const float* eps, const float* dz, const float* gkl_dev, float gkl_host,
                              float* dlogit, int B, int V, void* stream);

/* Input pipeline (next row): uint8 frames -> float32 in [-1,1] on the device, with the reference's arithmetic
 * dst = float32(src / 255.0) * 2 - 1  (data/image_pair_dataloader.py:163-164 float64 division, tf.data float32 cast,
 * map_fn :64-69 in float32).  Lets the loader ship uint8 over PCIe (a quarter of the bytes). */
int kpx_u8_to_unit_f32(const unsigned char* src, size_t n, float* dst, void* stream);

/* Host utility (no device access): CRC-32C of n bytes continuing from `crc` (0 to start) -- the checksum of TensorFlow V2
 * checkpoint bundles (next row; models/base_model.py:74-91 saves / restores through tf.train.Saver). */
unsigned int kpx_crc32c_host(unsigned int crc, const void* data, size_t n);

/* The strided / 4x4 / 1x1 layers (discriminator, the encoders' stride-2 layers: networks/__init__.py:16-24,141-151) in the bf16 configuration:
 * the gather convolution of kpx_conv2d_fwd_f32 / _dgrad_f32 / _wgrad_f32 with KPX_ARITH_BF16 on bf16 TENSORS (pixel strides in elements, multiples
 * of 8; 16-byte aligned).  y / dx are bf16 (y_f32 / dx_f32 = 0) or fp32; y_in (dgrad, optional): the bf16 activated tensor dx is the gradient
 * of (as kpx_conv2d_dgrad_act_f32).  Workspaces: the fp32 entries' queries.  KPX_EINVAL for shapes the bf16-pipe kernels do not take (Cin, Cout
 * multiples of 8 / 4 and >= 16): convert and use the fp32 entry. */
int kpx_conv2d_fwd_bf16(const void* x, int N, int Hi, int Wi, int Cin, int ldx, const float* w, int KH, int KW, const float* bias,
                        void* y, int y_f32, int Ho, int Wo, int Cout, int ldy, int stride, int pad_t, int pad_l, int act,
                        void* workspace, size_t workspace_bytes, void* stream);
int kpx_conv2d_dgrad_bf16(const void* dy, int N, int Ho, int Wo, int Cout, int lddy, const float* w, int KH, int KW,
                          void* dx, int dx_f32, int Hi, int Wi, int Cin, int lddx, int stride, int pad_t, int pad_l,
                          const void* y_in, int ld_y_in, int act_in, void* workspace, size_t workspace_bytes, void* stream);
int kpx_conv2d_wgrad_bf16(const void* x, int N, int Hi, int Wi, int Cin, int ldx, const void* dy, int Ho, int Wo, int Cout, int lddy,
                          float* dw, int KH, int KW, int stride, int pad_t, int pad_l, void* workspace, size_t workspace_bytes, void* stream);

/* Image-input layers in the bf16 configuration (models/networks/pose_encoder.py conv_1 7x7x3 -> 32, img_discriminator.py conv_0 4x4/s2 3 -> 64,
 * vgg.py:51 conv1_1 3x3x3 -> 64): the image stays fp32, what the layer produces / receives is bf16.  KPX_EINVAL (-1) for shapes the LDS-resident
 * image kernels do not take (the caller converts and uses the fp32 entries).
 *   fwd:   x fp32 [N,Hi,Wi,Cin<=4] contiguous -> y bf16 (pixel stride ldy elements), Cout <= 64
 *   dgrad: dy bf16 (pixel stride lddy, a multiple of 8; Cout a multiple of 16) -> dx fp32 [N,Hi,Wi,Cin] (3x3/s1 and 4x4/s2 filters)
 *   wgrad: x fp32, dy bf16 (pixel stride a multiple of 4) -> dw fp32 HWIO; workspace = kpx_conv2d_wgrad_workspace_bytes */
int kpx_conv_image_fwd_bf16(const float* x, int N, int Hi, int Wi, int Cin, const float* w, int KH, int KW, const float* bias,
                            void* y, int Ho, int Wo, int Cout, int ldy, int stride, int pad_t, int pad_l, int act, void* stream);
int kpx_conv_image_dgrad_bf16(const void* dy, int N, int Ho, int Wo, int Cout, int lddy, const float* w, int KH, int KW,
                              float* dx, int Hi, int Wi, int Cin, int lddx, int stride, int pad_t, int pad_l, void* stream);
int kpx_conv_image_wgrad_bf16(const float* x, int N, int Hi, int Wi, int Cin, int ldx, const void* dy, int Ho, int Wo, int Cout, int lddy,
                              float* dw, int KH, int KW, int stride, int pad_t, int pad_l, void* workspace, size_t workspace_bytes, void* stream);

/* Weight gradient of a 3x3 stride-1 SAME layer in the bf16 configuration (gradient of layers.py:6-9): x bf16 [N,H,W,>=Cin] (pixel stride ldx;
 * the pad channels up to the next multiple of 8 must hold finite values), dy bf16 [N,H,W,>=Cout], dw fp32 HWIO [3,3,Cin,Cout], written.
 * W a multiple of 16; workspace = kpx_conv3x3_wgrad_bf16_workspace_bytes (partial slabs of the pixel splits, summed in fixed order). */
int kpx_conv3x3_wgrad_bf16_eligible(int N, int H, int W, int Cin, int ldx, int Cout, int lddy, const void* x, const void* dy);
size_t kpx_conv3x3_wgrad_bf16_workspace_bytes(int N, int H, int W, int Cin, int Cout);
int kpx_conv3x3_wgrad_bf16(const void* x, int N, int H, int W, int Cin, int ldx, const void* dy, int Cout, int lddy,
                           float* dw, void* workspace, size_t workspace_bytes, void* stream);

/* ---- bf16 STORAGE variants of the streaming kernels (BASELINE configs[2]): bf16 tensors in HBM, fp32 arithmetic, fp64 reductions, fp32
 *      statistics / parameter gradients.  C and the pixel strides (in ELEMENTS) are multiples of 8, pointers 16-byte aligned, unless an
 *      entry says otherwise.  Reference call sites as the fp32 entries of the same name above. */
/* dst[p][0:C] = src[p][0:C] with conversion: kind 0 = f32 -> bf16, 1 = bf16 -> f32, 2 = bf16 -> bf16 (tf.concat channel slices); any C.
 * kind 3 = f32 -> bf16 into a contiguous 8-channel tensor (lddst = 8, C <= 8) whose channels C .. 7 are written as zeros. */
int kpx_cast_channels(const void* src, int ldsrc, void* dst, int lddst, size_t P, int C, int kind, void* stream);
int kpx_chan_sum_bf16(const void* x, size_t P, int C, int ldx, float* sum_out, void* scratch, void* stream);
/* layers.batch_norm, train mode, all weight-sharing groups in one launch per phase (kpx_bn_train_fwd_f32 / _bwd_f32): x bf16, y bf16
 * (y_f32 = 0) or fp32 (y_f32 = 1: the tensor the fp32 key-point head reads); backward: dy bf16 (dy_f32 = 0) or fp32, dx bf16. */
int kpx_bn_train_fwd_bf16(const void* x, size_t P, int groups, int C, int ldx, const float* tile_stats, size_t tiles_per_group,
                          float eps, const float* gamma, const float* beta, float* mean, float* invstd,
                          float* moving_mean, float* moving_var, float decay, void* y, int ldy, int y_f32, int act, void* scratch, void* stream);
int kpx_bn_train_bwd_bf16(const void* dy, int lddy, int dy_f32, const void* x, int ldx, size_t P, int groups, int C,
                          const float* mean, const float* invstd, const float* gamma, const float* beta, int act,
                          void* dx, int lddx, float* dgamma, float* dbeta, int accumulate,
                          const float* tile_stats, size_t tiles_per_group, void* scratch, void* stream);
int kpx_act_bwd_bf16(const void* dy, const void* y, void* dz, size_t n, int act, void* stream);
int kpx_resize2x_fwd_bf16(const void* x, int N, int H, int W, int C, int ldx, void* y, int ldy, void* stream);
int kpx_resize2x_bwd_bf16(const void* dy, int N, int H, int W, int C, int lddy, void* dx, int lddx, void* stream);
int kpx_maxpool2_fwd_bf16(const void* x, int N, int H, int W, int C, void* y, void* stream);
int kpx_vgg_feat_bwd_bf16(const void* f, size_t half, const float* gscale_dev, float gscale_host, const void* dy_pooled,
                          int B, int H, int W, int C, void* d, void* stream);
int kpx_l1_pair_fwd_bf16(const void* f, size_t half, float* loss_out, void* scratch, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* KPX_H */
