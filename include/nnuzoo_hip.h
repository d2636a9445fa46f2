/*
 * nnuzoo_hip.h — C-ABI of libnnuzoo_hip.so, the MI355X (gfx950) kernels behind nnUZoo's per-patch
 * forward/backward hot path.
 *
 * The reference (AI-in-Cardiovascular-Medicine/nnUZoo) has no FFI boundary of its own: it is pure Python and
 * reaches device code through torch.nn / mamba_ssm.  Each entry point below therefore cites the Python call
 * site whose device work it replaces (paths relative to /root/reference).  INTEGRATION.md shows the ctypes stub
 * a maintainer of the reference would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless it is a `desc`/`ksel` table (host memory, read at launch);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); all work is stream-ordered,
 *     re-entrant, and allocation-free (graph-capturable); workspaces/outputs belong to the caller;
 *   - return value: 0 on success, a positive hipError_t from the runtime, or -22 (EINVAL) for a shape /
 *     alignment the kernels do not support.  Nothing is ever silently routed to a CPU path;
 *   - activations are channels-last fp16 ((N, D, H, W, C), "ld" = elements between consecutive voxels, so a
 *     tensor can be a channel-slice of a wider concat buffer); statistics, parameters and parameter
 *     gradients are fp32.
 */
#ifndef NNUZOO_HIP_H
#define NNUZOO_HIP_H

#include <stdint.h>
#include "../nnuzoo_amd/csrc/conv_params.h"

#ifdef __cplusplus
extern "C" {
#endif

/* library / device probe: returns 0 and fills arch ("gfx950...") when a GPU is visible */
int nnz_device_info(char* arch, int arch_len, int* num_cu, long* hbm_bytes);
int nnz_version(void);

/* ---- dense contractions of the PlainConvUNet ("nnUNet") --------------------------------------------------
 * replaces torch.nn.Conv3d / ConvTranspose3d (and Conv2d / ConvTranspose2d: a 2-D layer is the depth-1 volume with
 * kernel (1,kh,kw), stride (1,sh,sw); per-axis kernel sizes 1/3 and strides 1/2) forward + autograd backward of
 * dynamic_network_architectures.PlainConvUNet, instantiated at nnunetv2/utilities/get_network_from_plans.py:27-57
 * and driven by nnUNetTrainer.train_step (nnunetv2/training/nnUNetTrainer/nnUNetTrainer.py:1112-1144). */
int nnz_conv_tap_forward(const void* in_f16, void* out_f16, const void* w_packed_f16, const float* bias,
                         const nnz_conv_desc* desc, void* stream);
/* same, and adds the per-(n, cout) {sum, sumsq} of the fp16 outputs into stats[N][Cout][2] (pre-zeroed by the caller):
 * the InstanceNorm statistics pass (nnz_instnorm_stats) fused into the convolution's epilogue.  Plain forward
 * convolutions only (one tap group, unit output stride, no accumulate); stats == NULL is nnz_conv_tap_forward. */
int nnz_conv_tap_forward_stats(const void* in_f16, void* out_f16, const void* w_packed_f16, const float* bias,
                               const nnz_conv_desc* desc, float* stats, void* stream);
/* Tuning / diagnostics knob of the conv kernels' tile dispatch (per-layer experiments; defaults are the measured best):
 * knob 0 = depth-reuse tap loop for k3 s1 layers with one 32-wide cout block on >= 64^3 grids (default 1),
 * knob 1 = the same loop, one cout block per workgroup, for Cout % 64 == 0 layers (default 1),
 * knob 2 = smallest m-grid edge (cube root of the voxel count) that takes that loop (default 16),
 * knob 3 = workgroup order: cout block fastest (1, default) or m-tile fastest (0),
 * knob 4 = smallest Cin for knob 1 (default 32).  Process-wide. */
int nnz_conv_tuning(int knob, int value);
/* the current value of a knob (0 for an unknown one) */
int nnz_conv_tuning_get(int knob);

int nnz_conv_tap_wgrad(const void* boxed_f16, const void* plain_f16, float* dw /* [T][A][B] */,
                       const nnz_conv_desc* desc, int dw_pre_zeroed, void* stream);
/* two-stage form used by the training schedule: every workgroup stores its [T][32][32] partial block into `workspace`
 * (nnz_conv_tap_wgrad_workspace_floats(desc) floats, no zero-fill, no atomics), then one reduction kernel sums the
 * blocks in a fixed order straight into the torch-layout gradient
 *   grad[a*sa + b*sb + ksel[t]*sk] (+)= dW[t][a][b]          (deterministic; replaces wgrad + unpack)           */
long nnz_conv_tap_wgrad_workspace_floats(const nnz_conv_desc* desc);
int nnz_conv_tap_wgrad_to_grad(const void* boxed_f16, const void* plain_f16, float* workspace, long ws_floats,
                               float* grad, long sa, long sb, long sk, const int* ksel, int accumulate,
                               const nnz_conv_desc* desc, void* stream);
int nnz_pack_conv_weight(const float* src, void* dst_f16, int R, int C, int T, long sr, long sc, long sk,
                         const int* ksel, void* stream);
/* batched packing: a device-resident table of nnz_pack_job_bytes()-sized records (filled on the host with
 * nnz_pack_job_fill, uploaded once) packs every layer's weights in one launch */
int nnz_pack_job_bytes(void);
int nnz_pack_job_fill(void* out_host, const float* src, void* dst_f16, int R, int C, int T, long sr, long sc, long sk,
                      const int* ksel);
int nnz_pack_conv_weights_batched(const void* jobs_device, int njobs, void* stream);
/* Both packed forms of every layer from ONE read of the fp32 parameters (round 3): forward Wf[X/16][Y/32][T][32][16] and
 * data-gradient Wb[Y/16][X/32][T][32][16] (X = the forward convolution's input channels, Y = its output channels; taps
 * contiguous in the parameter; x_inner = 1 for Conv weights (Y, X, k...), 0 for ConvTranspose weights (X, Y, k...)).
 * One workgroup per 32 x 32 channel block; jobs carry the prefix sum of their block counts. */
int nnz_pack_dual_job_bytes(void);
int nnz_pack_dual_job_fill(void* out, const float* src, void* dst_fwd_f16, void* dst_dgrad_f16, int X, int Y, int nk,
                           int x_inner, int first_block, const int* ksel_fwd, const int* ksel_dgrad);
int nnz_pack_dual_batched(const void* jobs_device, int njobs, int total_blocks, int max_nk, void* stream);
int nnz_unpack_conv_wgrad(const float* dw, float* grad, int A, int B, int T, long sa, long sb, long sk,
                          const int* ksel, int accumulate, void* stream);

/* stem Conv3d(1 -> 32, k3, p1) on the fp32 input patch and the 1x1x1 deep-supervision heads */
int nnz_stem_conv_forward(const float* x, const float* w, const float* bias, void* y_f16, int N, int D, int H, int W,
                          int Cout, int ldy, void* stream);
int nnz_stem_conv_wgrad(const float* x, const void* dy_f16, float* dw, int N, int D, int H, int W, int Cout, int lddy,
                        void* stream);
int nnz_seg_head_forward(const void* x_f16, const float* w, const float* bias, void* logits_f16_nc, int N, long V,
                         int C, int K, int ldx, void* stream);
int nnz_seg_head_dgrad(const void* dlogits_f16_nc, const float* w, void* dx_f16, int N, long V, int C, int K, int lddx,
                       int accumulate, void* stream);
int nnz_seg_head_wgrad(const void* x_f16, const void* dlogits_f16_nc, float* dw, float* db, int N, long V, int C,
                       int K, int ldx, void* stream);

/* ---- InstanceNorm(affine, eps) + LeakyReLU(slope) ---------------------------------------------------------
 * replaces nn.InstanceNorm3d + nn.LeakyReLU of every conv block (arch kwargs at
 * nnunetv2/experiment_planning/experiment_planners/default_experiment_planner.py:285-305). */
int nnz_norm_tuning(int knob, int value);   /* knob 0: target workgroups per norm launch (default 2048; A/B runs) */
int nnz_instnorm_stats(const void* x_f16, float* stats /* [N][C][2] sum,sumsq */, int N, long V, int C, int ldx,
                       int stats_pre_zeroed, void* stream);
int nnz_instnorm_lrelu_apply(const void* x_f16, const float* stats, const float* gamma, const float* beta, void* y_f16,
                             int N, long V, int C, int ldx, int ldy, float eps, float slope, void* stream);
int nnz_instnorm_lrelu_bwd_reduce(const void* x_f16, const void* g_f16, const float* stats, const float* gamma,
                                  const float* beta, float* red /* [N][C][2] */, int N, long V, int C, int ldx,
                                  int ldg, float eps, float slope, int red_pre_zeroed, void* stream);
int nnz_instnorm_lrelu_bwd_apply(const void* x_f16, const void* g_f16, const float* stats, const float* red,
                                 const float* gamma, const float* beta, void* dx_f16, int N, long V, int C, int ldx,
                                 int ldg, int lddx, float eps, float slope,
                                 float* dgamma /* [C] = sum_n red[n][c][1], may be NULL */,
                                 float* dbeta /* [C] = sum_n red[n][c][0], NULL iff dgamma is */, void* stream);

/* Deterministic variants (round 3).  fp32 atomicAdd makes the statistics depend on the order the workgroups finish in;
 * these entry points fold their partials in a fixed order per workgroup, add them across workgroups as FIXED-POINT 64-bit
 * integers (integer adds commute: bit-identical results run to run) and let the launch's last workgroup write the float
 * tables - no zero-fill launches, no sumsq/V - mean^2 cancellation in fp32 (moments are taken about a pilot value and
 * re-centred in double).  `acc`: N * C * 2 records of nnz_fxacc_bytes() bytes; `counter`: one 32-bit word; both zero
 * before their first use and left zero by every launch.
 *   nstat[N][C][4] = {mean, rstd, rstd * gamma, beta - mean * rstd * gamma}     (InstanceNorm table of a conv output)
 *   nred [N][C][2] = {mean of g', mean of g' * xhat},  g' = g * lrelu'(y)        (backward reduction) */
int nnz_fxacc_bytes(void);
int nnz_conv_tap_forward_norm(const void* in_f16, void* out_f16, const void* w_packed_f16, const float* bias,
                              const nnz_conv_desc* desc, void* acc, void* counter, const float* gamma,
                              const float* beta, float eps, float* nstat, void* stream);
/* ... and with a caller-provided fp32 workspace: few-tile / long-reduction layers (the <= 8^3 levels) split the reduction
 * over workgroups (split-K), partials in the workspace, folded in split order (deterministic) */
int nnz_conv_tap_forward_ws(const void* in_f16, void* out_f16, const void* w_packed_f16, const float* bias,
                            const nnz_conv_desc* desc, float* workspace, long ws_floats, void* stream);
int nnz_conv_tap_forward_norm_ws(const void* in_f16, void* out_f16, const void* w_packed_f16, const float* bias,
                                 const nnz_conv_desc* desc, void* acc, void* counter, const float* gamma,
                                 const float* beta, float eps, float* nstat, float* workspace, long ws_floats,
                                 void* stream);
/* Data-gradient launch (any dgrad table) whose epilogue also closes the reductions of the InstanceNorm(affine) + LeakyReLU
 * backward of the layer BELOW (the layer whose activation gradient this launch writes): x_raw = that layer's fp16 conv
 * output [N][out voxels][ld_x], nstat = its table.  Writes nred[N][Cout][2] = {mean g', mean g' xhat} and dgamma / dbeta
 * [Cout] (may both be NULL); the caller then runs nnz_instnorm_lrelu_bwd_apply_tab only.  Replaces the reducing launch of
 * nnz_instnorm_lrelu_bwd_tab (two full reads of x and g) - reference op: the autograd backward of
 * nn.InstanceNorm3d + nn.LeakyReLU inside ConvDropoutNormReLU (dynamic_network_architectures, instantiated at
 * nnunetv2/utilities/get_network_from_plans.py:10-42). */
int nnz_conv_tap_dgrad_normred(const void* in_f16, void* out_f16, const void* w_packed_f16, const nnz_conv_desc* desc,
                               const void* x_raw_f16, int ld_x, const float* nstat, float slope, void* acc, void* counter,
                               float* nred, float* dgamma, float* dbeta, void* stream);
/* BatchNorm2d in training mode (REBNCONV): stats[N][C][2] per-sample {sum, sumsq} of the conv epilogue -> bstats[C][2] batch
 * sums (what the apply kernel reads with N = 1) and F.batch_norm's running-estimate update (unbiased variance n / (n - 1),
 * momentum-weighted; running_mean / running_var may both be NULL) in one launch; n = number of values per channel */
int nnz_bn_batch_stats_finish(const float* stats, int N, int C, float n, float momentum, float* bstats, float* running_mean,
                              float* running_var, void* stream);
int nnz_instnorm_lrelu_bwd_apply_tab(const void* x_f16, const void* g_f16, const float* nstat, const float* nred,
                                     void* dx_f16, int N, long V, int C, int ldx, int ldg, int lddx, float slope,
                                     void* stream);
int nnz_stem_conv_wgrad_det(const float* x, const void* dy_f16, float* dw, int N, int D, int H, int W, int Cout, int lddy,
                            void* acc /* >= 864 records */, void* counter, void* stream);
int nnz_seg_head_wgrad_det(const void* x_f16, const void* dlogits_f16, float* dw, float* db, int N, long V, int C, int K,
                           int ldx, void* acc /* >= 8 * (C + 1) records */, void* counter, void* stream);
int nnz_dc_ce_loss_forward_det(const void* logits, int logits_is_f16, const int16_t* target, float* sums /* written */,
                               int B, int C, long V, int ignore_label, void* acc /* B * (3C + 1) records */,
                               void* counter, void* stream);
int nnz_grad_sumsq_nonfinite_det(const float* grads, long n, float* out2 /* written */, void* acc /* 2 records */,
                                 void* counter, void* stream);
int nnz_instnorm_stats_det(const void* x_f16, int N, long V, int C, int ldx, void* acc, void* counter,
                           const float* gamma /* NULL iff nstat is */, const float* beta, float eps,
                           float* nstat /* may be NULL */, float* sums /* [N][C][2] {sum, sumsq}, may be NULL */,
                           void* stream);
int nnz_instnorm_lrelu_apply_tab(const void* x_f16, const float* nstat, void* y_f16, int N, long V, int C, int ldx,
                                 int ldy, float slope, void* stream);
int nnz_instnorm_lrelu_bwd_tab(const void* x_f16, const void* g_f16, const float* nstat, void* acc, void* counter,
                               float* nred, void* dx_f16, int N, long V, int C, int ldx, int ldg, int lddx, float slope,
                               float* dgamma /* [C], may be NULL */, float* dbeta /* NULL iff dgamma is */,
                               void* stream);

/* ---- fused optimizer tail of train_step (nnUNetTrainer.py:1131-1139: grad_scaler.unscale_ -> clip_grad_norm_(12) ->
 * SGD(momentum, nesterov, weight decay) step, skipped when a gradient is not finite) over a flat fp32 gradient arena.
 * nnz_grad_sumsq_nonfinite adds {sum of squares, count of non-finite elements} of grads[0..n) to out2 (caller zeroes).
 * nnz_sgd_nesterov_fused: one workgroup per chunk record (nnz_sgd_chunk_bytes() bytes each, filled on the host by
 * nnz_sgd_chunk_fill: parameter and momentum pointers of <= 16384 elements + their gradient offset in the arena):
 *   g = grad * inv_scale * min(1, max_norm / (sqrt(sumsq) * inv_scale + 1e-6)) + weight_decay * p
 *   buf = momentum * buf + g;  p -= lr * (g + momentum * buf)        (buf starts at 0 == torch's first-step rule)
 * nothing is written when stats2[1] > 0 or the sum of squares is not finite; in the latter case stats2[1] is set to 1
 * so that the caller's found_inf flag reports the skip.  inv_scale_device may be NULL (1). */
int nnz_sgd_chunk_bytes(void);
int nnz_sgd_chunk_fill(void* out_host, float* param, float* momentum, long arena_offset, int n);
int nnz_grad_sumsq_nonfinite(const float* grads, long n, float* out2_zeroed, void* stream);
int nnz_sgd_nesterov_fused(const void* chunks_device, int nchunks, const float* arena, float* stats2,
                           const float* inv_scale_device, float max_norm, float lr, float momentum, float weight_decay,
                           int first_step, void* stream);

/* ---- hipGraph rewriting pass (csrc/graph_tools.hip): replaces every memset node of a captured, not yet instantiated
 * hipGraph_t by a fill-kernel node with the same dependencies / dependents and the same bytes (a captured hipMemsetAsync
 * node replays with a corrupted pattern on ROCm 7.2 once the process has allocated since capture; ATen's multi-block
 * reductions and some library paths emit such nodes).  *n_replaced = nodes rewritten.  nnz_graph_node_census counts the
 * nodes per hipGraphNodeType (diagnostics). */
int nnz_graph_replace_memsets(void* hip_graph, int* n_replaced);
int nnz_graph_node_census(void* hip_graph, int* counts, int ncounts);

/* ---- depthwise 3x3 weight / bias gradient (csrc/depthwise_wgrad.hip), stride 1, padding = dilation ------------------------
 * the depthwise nn.Conv2d layers of the 2-D zoo nets (nnunetv2/nets/ssnd2net.py: GSC and the SSND `convnd`;
 * nets/light_mamba2net.py: get_dwconv_layer), reached through torch's convolution_backward in the reference.
 * in, dy: [B][C][H][W] contiguous, fp16 (is_f16 = 1) or fp32; dw [C][3][3] and db [C] (or NULL) fp32; workspace:
 * nnz_dwconv2d_wgrad_workspace_floats floats.  Deterministic (two-stage sum, no atomics). */
long nnz_dwconv2d_wgrad_workspace_floats(int B, int C, int H, int W);
int nnz_dwconv2d_wgrad(const void* in, const void* dy, int is_f16, float* workspace, float* dw, float* db, int B, int C,
                       int H, int W, int dilation, void* stream);

/* ---- ConvTranspose with kernel = stride (UNetDecoder.transpconvs of PlainConvUNet; csrc/conv_transpose.hip) -----------------
 * out[n][s m + p][:Cout] = bias + in[n][m][:Cin] W[:, :, p]  and its data gradient, channels-last fp16 activations with row
 * strides ldi / ldo (elements, multiples of 8: a tensor may be a channel slice of a wider buffer), W = the fp32 parameter in
 * torch's ConvTranspose layout (Cin, Cout, kd, kh, kw), strides 1 or 2 per axis (2-D: sd = 1, Di = 1).
 * nnz_convT_supported: 1 when the P x Cin x Cout weights fit LDS as MFMA fragments (the full-resolution stages); other
 * shapes return -22 and the caller keeps the tap-table path (nnz_conv_tap_forward with a transposed-conv table). */
int nnz_convT_supported(int Cin, int Cout, int sd, int sh, int sw, int dgrad);
int nnz_convT_forward(const void* in, const float* W, const float* bias, void* out, int N, int Di, int Hi, int Wi, int Cin,
                      int Cout, int sd, int sh, int sw, int ldi, int ldo, void* stream);
int nnz_convT_dgrad(const void* dout, const float* W, void* din, int N, int Di, int Hi, int Wi, int Cin, int Cout, int sd,
                    int sh, int sw, int ldi, int ldo, void* stream);

/* ---- GPU-side input pipeline ----------------------------------------------------------------------------
 * Replaces the voxel-moving part of nnUNetDataLoader.generate_train_batch
 * (/root/reference/nnunetv2/training/dataloading/data_loader.py:180-259): data_all[j] = crop_and_pad_nd(data, bbox, 0),
 * seg_all[j] = crop_and_pad_nd(seg, bbox, -1) (:207-218), plus of the transform chain the MirrorTransform
 * (nnUNetTrainer.py:917-920) and DownsampleSegForDSTransform (:971) - for cases that are RESIDENT in HBM.
 * src: B host-side entries, each a DEVICE pointer to a case volume [C][sd][sh][sw]; shapes, lbs: B x 3 host ints (case
 * extent, lower bbox corner in case coordinates - may be negative or reach beyond the case: padded); flips: B host ints
 * (bit a = mirror the patch along axis a: 0 depth, 1 height, 2 width) or NULL; dst: device [B][C][pd][ph][pw].
 * 2-D data: pd = 1.  nnz_downsample_nearest_i16: torch's interpolate(mode='nearest-exact') index rule, bit-exact. */
int nnz_crop_pad_f32(const void* const* src, const int* shapes, const int* lbs, const int* flips, float* dst, int B, int C,
                     int pd, int ph, int pw, float pad_value, void* stream);
int nnz_crop_pad_i16(const void* const* src, const int* shapes, const int* lbs, const int* flips, short* dst, int B, int C,
                     int pd, int ph, int pw, int pad_value, void* stream);
int nnz_downsample_nearest_i16(const short* src, short* dst, long nc, int id, int ih, int iw, int od, int oh, int ow,
                               void* stream);

/* ---- device-side training augmentations (round 5; csrc/augment.hip) -------------------------------------------------------
 * The interpolating / intensity transforms of nnUNetTrainer.get_training_transforms (nnUNetTrainer.py:825-973: SpatialTransform
 * :845-852, GaussianNoiseTransform :857-863, MultiplicativeBrightnessTransform :872-878, ContrastTransform :879-886,
 * GammaTransform :897-914, RemoveLabelTansform :929-931) on a batch resident in HBM.  The transform classes are
 * batchgeneratorsv2's (absent from the reference tree): arithmetic restated, parity unpinned. */
/* dst[b][c][o] = src[b][c] sampled at M_b (o - centre) + centre + shift_b; mats = B x 12 HOST floats (3 x 4, rows z y x);
 * f32 trilinear / i16 nearest, pad_value outside; D = 1 for 2-D batches; src != dst */
int nnz_aug_affine_f32(const float* src, float* dst, const float* mats, int B, int C, int D, int H, int W, float pad_value,
                       void* stream);
int nnz_aug_affine_i16(const short* src, short* dst, const float* mats, int B, int C, int D, int H, int W, int pad_value,
                       void* stream);
/* stats[bc] = {mean, population std, min, max} of x[bc][0..n); workspace of nnz_aug_stats_workspace_floats(nbc) floats */
long nnz_aug_stats_workspace_floats(int nbc);
int nnz_aug_stats_f32(const float* x, long n, int nbc, float* workspace, float* stats, void* stream);
/* in place over x[nbc][n]; rec = device [nbc][4] {active, p0, p1, -}; op 0 noise (sigma p0), 1 linear p0 v + p1, 2 contrast
 * (factor p0, clamp to [min, max] of stats_a), 3 gamma (exponent p0 on the [min, max] range of stats_a), 4 restore the mean / std
 * of stats_b given the current statistics stats_a */
int nnz_aug_intensity_f32(float* x, long n, int nbc, int op, const float* rec, const float* stats_a, const float* stats_b,
                          int seed, void* stream);
/* GaussianBlurTransform (:864-871), one axis (0 z, 1 y, 2 x) per launch: rec = device [nbc][4] {active, sigma_z, sigma_y, sigma_x};
 * taps within 3 sigma (at most 4), edge voxels repeated; inactive rows are copied; src != dst */
int nnz_aug_blur_axis_f32(const float* src, float* dst, int nbc, int D, int H, int W, int axis, const float* rec, void* stream);
/* SimulateLowResolutionTransform (:887-896): nearest down-sampling to round(size * scale), linear up-sampling back, one pass;
 * rec = device [nbc][4] {active, scale, -, -}; keep_z: 2-D / dummy-2-D batches */
int nnz_aug_lowres_f32(const float* src, float* dst, int nbc, int D, int H, int W, int keep_z, const float* rec, void* stream);
int nnz_aug_relabel_i16(short* x, long n, int from, int to, void* stream);
/* Label-side transforms whose arithmetic the reference defines itself (training/data_augmentation/custom_transforms/):
 *   ConvertSegmentationToRegionsTransform (region_based_training.py:7-39; chain: nnUNetTrainer.py:961-969): out [B][R][n] int16 =
 *     1 where seg[b][seg_channel] is one of region r's labels (labels[begin[r] .. begin[r + 1]), host arrays, <= 64 labels in all);
 *   MoveSegAsOneHotToData (cascade_transforms.py:10-39; chain :932-939): channels c0 .. c0 + K - 1 of data [B][Cd][n] float are
 *     written with (seg[b][seg_channel] == labels[k]); the caller has allocated the widened tensor and copied the image channels;
 *   MaskTransform (masking.py:6-24; chain :921-927): data[b][c][seg[b][mask_channel] < 0] = value for the channels in the bit set. */
int nnz_aug_seg_to_regions_i16(const short* seg, short* out, int B, int Cs, int seg_channel, long n, const int* begin,
                               const int* labels, int R, void* stream);
int nnz_aug_seg_onehot_to_data_f32(const short* seg, float* data, int B, int Cs, int seg_channel, int Cd, int c0, long n,
                                   const int* labels, int K, void* stream);
int nnz_aug_mask_outside_f32(float* data, const short* seg, int B, int Cd, int Cs, int mask_channel, long n,
                             long channels, float value, void* stream);

/* ---- x_proj of the cross-scan SS2D block on channel-major fp32 activations (the einsum of SS2D.forward_core,
 * /root/reference/nnunetv2/nets/m2net.py:179-184, in the two-source formulation of nnz_ss2d_scan_*):
 *   forward     P[s][b][c][l]   = sum_d W[s][c][d] x2[s][b][d][l]                       c < C2 <= 80, Di % 32 == 0
 *   backward_x  dx2[s][b][d][l] = sum_c W[s][c][d] dP[s][b][c][l] + du[b][s][d][l] + du[b][s+2][d][l]
 *   backward_w  dW[s][c][d]    += sum_{b,l} dP[s][b][c][l] x2[s][b][d][l]     (fp32, caller zeroes; L % 64 == 0 and
 *                                                                              ceil8(C2)/8 * Di/8 <= 256, else -22) */
/* cp: 0 = W (dW) is [2][C2][Di]; > 0 = W (dW) is the module's own x_proj_weight layout [4][cp][Di], C2 = 2 cp (direction
 * k = s + 2 j holds rows [j cp, (j + 1) cp) of source s) - no stacked copy of the weight / un-stacking copy of its gradient */
int nnz_ss2d_xproj_forward(const float* x2, const float* W, float* P, int B, int Di, int C2, long L, int cp, void* stream);
int nnz_ss2d_xproj_backward_x(const float* dP, const float* W, const float* du, float* dx2, int B, int Di, int C2,
                              long L, int cp, void* stream);
int nnz_ss2d_xproj_backward_w(const float* dP, const float* x2, float* dW, int B, int Di, int C2, long L, int cp,
                              void* stream);
/* Two-stage (bit-reproducible) forms of the weight gradients whose token range is spread over many workgroups: every workgroup
 * stores its partial block into `workspace` (nnz_*_workspace_floats) and a second kernel folds the blocks in a fixed order;
 * the gradient is WRITTEN (no pre-zeroing).  The plain entry points add with fp32 atomics (arrival order in the low bits). */
long nnz_ss2d_xproj_backward_w_workspace_floats(int B, int Di, int C2, long L);
int nnz_ss2d_xproj_backward_w_ws(const float* dP, const float* x2, float* dW, float* workspace, long ws_floats, int B,
                                 int Di, int C2, long L, int cp, void* stream);

/* ---- token-major Linear layers with many tokens and few features (VSS / SSND in_proj, out_proj, patch merge / expand:
 * /root/reference/nnunetv2/nets/m2net.py:97,103,258,300) under the autocast step: fp16 activations, fp32 master weight
 * (converted while it is staged into LDS), fp32 accumulate, fp16 result.
 * nnz_token_linear_forward: out[T][Mo] = in[T][Kr] A^T + bias with A = W[Mo][Kr] (transposed = 0: the forward) or
 * A = W^T, W[Kr][Mo] (transposed = 1: the input gradient dX = dY W).  Kr in {16, 32, 64, 128, 256}; Mo % 8 == 0, <= 256;
 * nnz_token_linear_supported(Kr, Mo) tells whether a shape is served (else -22 here; callers use the library GEMM).
 * nnz_token_linear_wgrad: dW[N][K] += sum_t dy[t][n] x[t][k], db[n] += sum_t dy[t][n] (fp32, caller zeroes; db may be
 * NULL); N, K multiples of 8 with (N/8)(K/8) <= 256. */
int nnz_token_linear_forward(const void* in_f16, const float* W, const float* bias, void* out_f16, long T, int Kr, int Mo,
                             int transposed, void* stream);
int nnz_token_linear_supported(int Kr, int Mo);
int nnz_token_linear_wgrad(const void* dy_f16, const void* x_f16, float* dW, float* db, long T, int N, int K,
                           void* stream);
/* two-stage (bit-reproducible) form: dW / db are written from per-workgroup partial blocks folded in a fixed order */
long nnz_token_linear_wgrad_workspace_floats(long T, int N, int K);
int nnz_token_linear_wgrad_ws(const void* dy_f16, const void* x_f16, float* dW, float* db, float* workspace,
                              long ws_floats, long T, int N, int K, void* stream);

/* ---- fp32 token-major Linear layers on v_mfma_f32_32x32x2_f32 (csrc/dense32.hip, round 3) -----------------------------
 * The Swin / ViT trainers run without autocast (nnunetv2/training/nnUNetTrainer/nnUNetTrainerSwT2Net.py:112-130): qkv /
 * proj (nets/swt2net.py:584-619), Mlp (:496-515), patch merging / expanding, skip fusions (:843-868) are exact fp32
 * F.linear calls there.  x [T][K], W [N][K] (torch layout), y [T][N]; K, N multiples of 4.
 *   forward: y = x W^T + bias;  gelu != 0: y keeps the pre-activation and y_act = GELU(y) (erf form, nn.GELU default)
 *   dgrad:   dx = dy W  (times GELU'(h) when h [T][K] is given)
 *   wgrad:   dW = dy^T x, db = column sums of dy (db may be NULL); token splits fold in a fixed order (deterministic) */
int nnz_dense32_forward(const float* x, const float* W, const float* bias, float* y, float* y_act, long T, int K, int N,
                        int gelu, void* stream);
int nnz_dense32_dgrad(const float* dy, const float* W, const float* h, float* dx, long T, int K, int N, void* stream);
long nnz_dense32_wgrad_workspace_floats(long T, int K, int N);
int nnz_dense32_wgrad(const float* dy, const float* x, float* dW, float* db, float* workspace, long T, int K, int N,
                      void* stream);
/* ALL weight gradients of a backward pass in one launch (+ one fold launch): the ~670 Linear layers of a Swin / ViT step are
 * 18-30 us problems each; they do not sit on the data-gradient chain, so the caller queues them and runs them together.
 * Same split rule and arithmetic as nnz_dense32_wgrad (bit-identical results).  Protocol: nnz_dense32_group_plan per
 * problem -> its workgroups, fold blocks (0: written directly) and workspace floats; jobs are laid out back to back
 * (wg_begin / blk_begin = running sums); nnz_dense32_group_fill writes one record per job into HOST tables
 * (nnz_dense32_group_record_bytes(0) bytes per weight-gradient record, (1) per fold record; fold records only for jobs with
 * fold blocks, numbered in job order); the caller builds the int32 maps workgroup -> job and fold block -> fold job, copies
 * the four arrays to the device and calls nnz_dense32_group_launch. */
int nnz_dense32_group_record_bytes(int which);
int nnz_dense32_group_class(long T, int K, int N);   /* 0: 64 x 64 tiles, 1: 128 x 128; one launch per class */
int nnz_dense32_group_plan(long T, int K, int N, int* wgs, int* fold_blocks, long* ws_floats);
int nnz_dense32_group_fill(void* job_host, void* fold_host, const float* dy, const float* x, float* dW, float* db,
                           float* workspace, long T, int K, int N, int wg_begin, int blk_begin);
int nnz_dense32_group_launch(const void* jobs_dev, const int* wg_job_dev, int total_wgs, const void* fold_dev,
                             const int* blk_job_dev, int total_blks, int tile_class, void* stream);

/* online-Dice statistics of the validation step (nnUNetTrainer.validation_step, nnUNetTrainer.py:1185-1226 +
 * get_tp_fp_fn_tn, training/loss/dice.py:122-180, label-map targets): argmax over classes (first maximum on ties)
 * against the int16 label map in one read; counts_u64[c] = {tp, fp, fn} exact (zeroed by the call). */
int nnz_argmax_tp_fp_fn(const void* logits_nc, int logits_is_f16, const int16_t* target, void* counts_u64 /* [C][3] */,
                        int B, int C, long V, int ignore_label /* -32768: none */, void* stream);

/* ---- 1-D Mamba block pieces around the scan (nets/seg_mamba/mamba_simple.py:190-357, mamba_inner_ref in
 * nets/seg_mamba/selective_scan_interface.py:640-674): causal depthwise conv1d (width W <= 8, padding W-1, truncated to
 * L) + SiLU on fp32 (B, D, L) rows, and the z gate out = y * z * sigmoid(z) that selective_scan_fn(..., z=z) applies.
 * Backward of the conv recomputes the pre-activation; dw [D][W] and dbias [D] are zeroed by the call. */
int nnz_causal_conv1d_silu_forward(const float* x, const float* w /* [D][W] */, const float* bias, float* y, int B, int D,
                                   int L, int W, void* stream);
int nnz_causal_conv1d_silu_backward(const float* x, const float* w, const float* bias, const float* dy, float* dx,
                                    float* dw, float* dbias, int B, int D, int L, int W, void* stream);
int nnz_silu_gate_forward(const float* y, const float* z, float* out, long n, void* stream);
int nnz_silu_gate_backward(const float* dout, const float* y, const float* z, float* dy, float* dz, long n, void* stream);

/* ---- sliding-window inference accumulation ---------------------------------------------------------------------
 * replaces the tensor arithmetic of nnUNetPredictor._internal_maybe_mirror_and_predict
 * (nnunetv2/inference/predict_from_raw_data.py:549-564: `prediction += torch.flip(...)`, `prediction /= n`) and of
 * _internal_predict_sliding_window_return_logits (:566-643: `prediction *= gaussian`, `predicted_logits[sl] +=`,
 * `n_predictions[sl[1:]] += gaussian`, `predicted_logits /= n_predictions`, isinf check) on half tensors, rounding to
 * fp16 at the same points (bit-identical to the reference loop for equal network outputs).
 * preds [M][K][td][th][tw]: the network outputs for the tile and its mirrored copies (flip_bits[m]: bit a = the input of
 * prediction m was flipped along volume axis a; flip_bits[0] must be 0), in the reference's accumulation order;
 * tile_dims / image_dims / offset are host int[3] (2-D tiles: depth 1); gaussian may be NULL (weight 1). */
int nnz_sliding_window_accumulate(const void* preds_f16, int M, const int* flip_bits, const void* gaussian_f16,
                                  void* logits_f16 /* [K][D][H][W] */, void* npred_f16 /* [D][H][W] */, int K,
                                  const int* tile_dims, const int* image_dims, const int* offset, void* stream);
int nnz_sliding_window_finalize(void* logits_f16, const void* npred_f16, int K, long V, int* inf_flag_device,
                                void* stream);

/* ---- SS2D cross-scan: the four scan directions of the SS2D block (m2net.py:170-206) over ONE input --------------------
 * Direction k = s + 2j reads source s (0: row-major tokens, 1: column-major tokens = transposed image), reversed in time
 * for j = 1, by index arithmetic inside the scan kernels - no stacked / flipped copies, no materialised delta.
 *   x2  [2][B][Dg][L]      both sources of the block input (nnz_ss2d_prepare)
 *   P   [2][B][2 Cp][L]    projections [W_s ; W_{s+2}] x_s, Cp = R + 32: rows j*Cp.. = R dt rows, 16 B rows, 16 C rows
 *   Wdt [4 Dg][R]          dt_projs_weight; delta = softplus(Wdt . dt + delta_bias) formed in the kernel (R <= 8)
 *   y, du [B][4 Dg][L]     per direction, in the source's token order (un-reversed)
 *   dy2 [2][B][Dg][L]      gradient seen by both directions of a source (nnz_ss2d_split); dP like P; dWdt like Wdt
 * a_is_log: A holds A_log [4 Dg][16]; the kernels use A = -exp(A_log) and return dA_log = dA * A (m2net.py:196).
 * chunk_state / grad_state / workspace: nnz_ss2d_scan_state_floats / _grad_state_floats / _workspace_floats(B, Dg, L)
 * floats.  chunk_state holds the forward's checkpoints (state entering every chunk and, for the channels-on-lanes
 * kernels of csrc/ss2d_scan_rl.hpp, every 16-step sub-block) and must reach the backward unchanged.
 * nnz_scan_tuning(knob, value): 0 = channels-on-lanes kernels on/off (default on; they need L % 16 == 0 and Dg = 32 or a
 * multiple of 64, other shapes take the time-on-lanes kernels), 1 = forced chunk length in 16-step sub-blocks (4/8/16),
 * 4 = (round 5, default 1) where several workgroups share a (batch, direction) group's dB / dC / d dt rows - channel chunks of a
 * wide group - each writes a slab of its own behind P / S in `workspace` and a fold launch sums the slabs in a fixed order:
 * the backward is bit-identical from run to run (0: the fp32 atomics of rounds 1-4).
 * nnz_ss2d_merge: out (B, H, W, Dg) = y0 + y2 + (y1 + y3)^T;  nnz_ss2d_merge_dx: dx (B, Dg, H, W) = du0 + du2 + dx2[0]
 * + (du1 + du3 + dx2[1])^T in x's type (du may be NULL). */
int nnz_ss2d_prepare(const void* x, int x_is_f16, float* x2, int Bt, int D, int H, int W, void* stream);
int nnz_ss2d_merge(const float* y, float* out_tokens, int Bt, int D, int H, int W, void* stream);
int nnz_ss2d_split(const float* dout_tokens, float* dy2, int Bt, int D, int H, int W, void* stream);
int nnz_ss2d_merge_dx(const float* du, const float* dx2, void* dx, int dx_is_f16, int Bt, int D, int H, int W,
                      void* stream);
/* depthwise 3x3 conv (padding 1) + SiLU of the SS2D block (m2net.py:214) reading the token-major half of the in_proj
 * output in place (rows x_row_stride elements apart, f16 or f32) and writing both scan sources x2 [2][B][D][L] (f32);
 * backward from the gradient of x2: dx_tokens [B][H][W][D] dense in x's type, dweight [D][9], dbias [D] (may be NULL). */
int nnz_ss2d_dwconv_silu_forward(const void* x_tokens, int x_is_f16, long x_row_stride, const float* weight,
                                 const float* bias, float* x2, int Bt, int D, int H, int W, void* stream);
int nnz_ss2d_dwconv_silu_backward(const void* x_tokens, int x_is_f16, long x_row_stride, const float* weight,
                                  const float* bias, const float* dx2, void* dx_tokens, float* dweight, float* dbias,
                                  int Bt, int D, int H, int W, void* stream);
/* two-stage (bit-reproducible) weight / bias gradient: per-tile partial rows in `workspace`, fixed-order fold */
long nnz_ss2d_dwconv_silu_backward_workspace_floats(int Bt, int D, int H, int W);
int nnz_ss2d_dwconv_silu_backward_ws(const void* x_tokens, int x_is_f16, long x_row_stride, const float* weight,
                                     const float* bias, const float* dx2, void* dx_tokens, float* dweight, float* dbias,
                                     float* workspace, long ws_floats, int Bt, int D, int H, int W, void* stream);
long nnz_ss2d_scan_state_floats(int Bt, int Dg, int L);
long nnz_ss2d_scan_grad_state_floats(int Bt, int Dg, int L);
long nnz_ss2d_scan_workspace_floats(int Bt, int Dg, int L);
int nnz_scan_tuning(int knob, int value);
int nnz_scan_tuning_get(int knob);   /* knob 3: number of channels-on-lanes launches so far */
int nnz_ss2d_scan_forward(const float* x2, const float* P, const float* Wdt, const float* A, const float* D,
                          const float* delta_bias, float* y, float* chunk_state, float* workspace, int Bt, int Dg, int R,
                          int L, int delta_softplus, int a_is_log, void* stream);
int nnz_ss2d_scan_backward(const float* x2, const float* P, const float* Wdt, const float* A, const float* D,
                           const float* delta_bias, const float* dy2, const float* chunk_state, float* grad_state,
                           float* workspace, float* du, float* dP, float* dWdt, float* dA, float* dD, float* dbias,
                           int Bt, int Dg, int R, int L, int delta_softplus, int a_is_log, void* stream);

/* ---- residual add with stochastic depth: out = input + x * mask[b] * scale  (VSSBlock.forward, m2net.py:530, with timm's
 * DropPath: mask drawn per sample by the caller, scale = 1 / keep_prob).  input / x / out: [B][per_sample] f16 or f32
 * (per_sample % 4 == 0), mask [B] f16 or f32 or NULL.  Backward: dx = dout * mask[b] * scale (d input = dout). */
int nnz_residual_droppath_forward(const void* input, int input_is_f16, const void* x, int x_is_f16, const void* mask,
                                  int mask_is_f16, float scale, void* out, int out_is_f16, int B, long per_sample,
                                  void* stream);
int nnz_residual_droppath_backward(const void* dout, int dout_is_f16, const void* mask, int mask_is_f16, float scale,
                                   void* dx, int dx_is_f16, int B, long per_sample, void* stream);
/* the same pair with the mask made inside: rand = B fp32 uniform draws, mask[b] = floor(rand[b] + keep) - the reference's
 * Swin DropPath (swt2net.py:379-388: keep_prob + torch.rand(..), floor_(), x.div(keep_prob) * mask) without its add and
 * floor launches; keep > 0 */
int nnz_residual_droppath_rand_forward(const void* input, int input_is_f16, const void* x, int x_is_f16, const float* rand,
                                       float keep, float scale, void* out, int out_is_f16, int B, long per_sample,
                                       void* stream);
int nnz_residual_droppath_rand_backward(const void* dout, int dout_is_f16, const float* rand, float keep, float scale,
                                        void* dx, int dx_is_f16, int B, long per_sample, void* stream);

/* ---- depthwise-separable conv -> BatchNorm -> ReLU unit of SwT2Net's RSU4F stages, fp32 channels-last (csrc/sepconv32.hip).
 * Replaces, for /root/reference/nnunetv2/nets/swt2net.py:17-31 REBNCONV (get_dwconv_layer -> nn.BatchNorm2d -> ReLU) inside RSU4F
 * (:873-905), the cuDNN / MIOpen calls of the reference's fp32 step; the pointwise 1x1 between them is nnz_dense32_forward_fused.
 * x / y [B][H][W][C] fp32, C % 4 == 0; w [C][3][3] (torch's [C, 1, 3, 3]); flip = 1: the input gradient (x = dy, y = dx). */
int nnz_dw3x3_nhwc_f32(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int flip,
                       void* stream);
long nnz_dw3x3_nhwc_wgrad_workspace_floats(int B, int H, int W, int C);
/* dw [C][3][3] is written (no zero fill); per-range partials + fixed-order fold: deterministic */
int nnz_dw3x3_nhwc_wgrad_f32(const float* x, const float* dy, float* workspace, float* dw, int B, int H, int W, int C,
                             void* stream);
/* y = relu(batch_norm(x)) on [T][C] fp32: training != 0 - batch statistics (two-pass), running estimates updated in place with
 * torch.nn.BatchNorm2d's rule (momentum, unbiased variance); 0 - the running estimates.  mean / rstd [C] are written. */
int nnz_bn_relu_nhwc_forward_f32(const float* x, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                 float* mean, float* rstd, float* y, long T, int C, int training, float momentum, float eps,
                                 void* stream);
/* backward of the training-mode unit: dx [T][C], dgamma / dbeta [C] written; fixed-order reductions */
int nnz_bn_relu_nhwc_backward_f32(const float* x, const float* dy, const float* gamma, const float* beta, const float* mean,
                                  const float* rstd, float* dx, float* dgamma, float* dbeta, long T, int C, void* stream);

/* weight gradient of a pointwise convolution with few channels over many tokens: dW [N][K] = dy^T x, dy [T][N], x [T][K] fp32,
 * N, K multiples of 4 and <= 64 (the full-resolution stems / heads of the Swin U-net stages, swt2net.py:757, 795: a streaming
 * reduction, VALU 4 x 4 register blocks from LDS-staged token chunks).  dW is written; per-range partials + fixed-order folds. */
long nnz_pw_wgrad_small_workspace_floats(long T, int N, int K);
int nnz_pw_wgrad_small_f32(const float* dy, const float* x, float* workspace, float* dW, long T, int N, int K, void* stream);
/* the same for fp32 or IEEE-half token rows (is_f16: the 1x1 patch embeddings / stage outputs of the fp16-autocast X^2-Nets,
 * m2net.py:486-512 `PatchEmbed2D.proj`, :431-473 `seg_layers`) and, db != NULL, the bias gradient db [N] = sum_t dy[t] from the same
 * pass; workspace: nnz_pw_wgrad_small_workspace_floats_b(T, N, K, db != NULL) floats.  dW / db are written; fixed-order folds. */
long nnz_pw_wgrad_small_workspace_floats_b(long T, int N, int K, int with_bias);
int nnz_pw_wgrad_small(const void* dy, const void* x, int is_f16, float* workspace, float* dW, float* db, long T, int N, int K,
                       void* stream);

/* 1x1 convolution to N <= 8 output channels in fp32: the side heads (`side1 .. side6`, C -> classes) and the fuse convolution
 * (`outconv`, 6 classes -> classes) of SwT2Net, /root/reference/nnunetv2/nets/swt2net.py:1021-1028, 1130-1141.  y / dy [B][N][P]
 * (NCHW), x addressed as x[b * xsb + p * xsp + k * xsk] (token-major stage output or NCHW concatenation), w [N][K], N * K <= 8192.
 * wgrad: dwb [N][K + 1] = {dW | db} written, per-range partials + fixed-order fold (workspace:
 * nnz_head1x1_wgrad_workspace_floats). */
int nnz_head1x1_forward_f32(const float* x, const float* w, const float* bias, float* y, int B, int N, int K, long P, long xsb,
                            long xsp, long xsk, void* stream);
int nnz_head1x1_dgrad_f32(const float* dy, const float* w, float* dx, int B, int N, int K, long P, long xsb, long xsp, long xsk,
                          void* stream);
long nnz_head1x1_wgrad_workspace_floats(int B, int N, int K, long P);
int nnz_head1x1_wgrad_f32(const float* x, const float* dy, float* workspace, float* dwb, int B, int N, int K, long P, long xsb,
                          long xsp, long xsk, void* stream);
/* the same three with fp16 activations (x, y, dy, dx: IEEE half), fp32 weights / bias / sums / dwb: the fuse convolution of the
 * fp16-autocast X^2-Nets (`outconv`, /root/reference/nnunetv2/nets/m2net.py:881, 948-950), whose operands autocast rounds to fp16 */
int nnz_head1x1_forward_f16(const void* x, const float* w, const float* bias, void* y, int B, int N, int K, long P, long xsb,
                            long xsp, long xsk, void* stream);
int nnz_head1x1_dgrad_f16(const void* dy, const float* w, void* dx, int B, int N, int K, long P, long xsb, long xsp, long xsk,
                          void* stream);
int nnz_head1x1_wgrad_f16(const void* x, const void* dy, float* workspace, float* dwb, int B, int N, int K, long P, long xsb,
                          long xsp, long xsk, void* stream);

/* ---- bilinear up-sampling of the side outputs and its adjoint (`_upsample_like`, /root/reference/nnunetv2/nets/m2net.py:33-36 =
 * F.interpolate(src, size, mode='bilinear'), align_corners=False; called at :948-950 and in u2net.py / swt2net.py alike).
 * src [B][h][w] -> dst [B][H][W], B = samples x channels, fp32 or IEEE half (is_f16), fp32 arithmetic, ATen's source-index rule.
 * backward: din [B][h][w] is WRITTEN with the adjoint as a fixed-order gather (no atomics, no library GEMM). */
int nnz_bilinear_up_forward(const void* src, void* dst, int is_f16, int B, int h, int w, int H, int W, void* stream);
int nnz_bilinear_up_backward(const void* dout, void* din, int is_f16, int B, int h, int w, int H, int W, void* stream);

/* ---- top / left zero padding of a channels-last fp32 map to the window multiple and the crop back (SwinTransformerBlock.forward,
 * swt2net.py:643-645 F.pad(x, (0, 0, ws - W % ws, 0, ws - H % ws, 0)) and :660 x[:, -H:, -W:, :]; each is the other's
 * backward).  small [B][H][W][C], big [B][H + py][W + px][C], C % 4 == 0, B * (H + py) <= 65535; one launch each. */
int nnz_pad_top_left(const float* small, float* big, int B, int H, int W, int C, int py, int px, void* stream);
int nnz_crop_top_left(const float* big, float* small, int B, int H, int W, int C, int py, int px, void* stream);

/* ---- LayerNorm over the last dimension of token-major tensors (nn.LayerNorm in the VSS / Swin blocks: m2net.py:101,521,
 * ssnd2net.py, swt2net.py:630-660).  x: [rows][C] f16 or f32, C % 4 == 0, C <= 2048; y, dy: f32 (what autocast gives);
 * dx has x's type; gamma / beta / dgamma / dbeta may be NULL (elementwise_affine = False).  mean / rstd: [rows] f32. */
int nnz_layer_norm_forward(const void* x, int x_is_f16, const float* gamma, const float* beta, void* y, int y_is_f16,
                           float* mean, float* rstd, float* zero_2c, long rows, int C, float eps, void* stream);
int nnz_layer_norm_backward(const void* x, int x_is_f16, const float* gamma, const float* mean, const float* rstd,
                            const void* dy, int dy_is_f16, void* dx, float* dgamma, float* dbeta, int pre_zeroed,
                            long rows, int C, void* stream);
/* y = LayerNorm(x) * silu(z): the gated output norm of the SS2D block (m2net.py:220 `self.out_norm(y) * F.silu(z)`).
 * z: f16 or f32, rows z_row_stride elements apart (z is one half of the in_proj output); dz: [rows][C] in z's type.
 * y_is_f16 (all four): write y as f16 when its only consumer is an autocast Linear (the cast that consumer would apply
 * rounds the same fp32 value once - identical numbers, one pass and one launch less); dy then arrives as f16.
 * zero_2c (forward, may be NULL): the [2][C] buffer the matching backward accumulates dgamma / dbeta into; the forward
 * launch zeroes it so that the backward (pre_zeroed = 1) needs no zeroing launch. */
int nnz_layer_norm_gate_forward(const void* x, int x_is_f16, const float* gamma, const float* beta, const void* z,
                                int z_is_f16, long z_row_stride, void* y, int y_is_f16, float* mean, float* rstd,
                                float* zero_2c, long rows, int C, float eps, void* stream);
int nnz_layer_norm_gate_backward(const void* x, int x_is_f16, const float* gamma, const float* beta, const void* z,
                                 int z_is_f16, long z_row_stride, const float* mean, const float* rstd, const void* dy,
                                 int dy_is_f16, void* dx, void* dz, float* dgamma, float* dbeta, int pre_zeroed,
                                 long rows, int C, void* stream);

/* ---- fused soft-Dice + cross-entropy statistics on NC(D)HW logits --------------------------------------------
 * replaces softmax + one-hot + reductions + CE of DC_and_CE_loss (nnunetv2/training/loss/compound_losses.py:31-56,
 * dice.py:72-119, robust_ce_loss.py:12-16).  sums[b] = {intersect[C], sum_pred[C], sum_gt[C], ce_sum};
 * coef[b] = {dL/dintersect[C], dL/dsum_pred[C], dL/dce_sum}.  target: int16 class ids [B][V].
 * ignore_label: voxels carrying this label are left out of every sum and get zero gradient (the loss_mask of the Dice
 * terms and the CE's ignore_index, compound_losses.py:38-52); pass -32768 for none. */
int nnz_dc_ce_loss_forward(const void* logits, int logits_is_f16, const int16_t* target, float* sums, int B, int C,
                           long V, int ignore_label, void* stream);
int nnz_dc_ce_loss_backward(const void* logits, int logits_is_f16, const int16_t* target, const float* coef,
                            void* dlogits, int B, int C, long V, int ignore_label, void* stream);
/* value and gradient coefficients of DC_and_CE_loss from the sums, on the device (one launch per deep-supervision
 * output instead of ~25 element-wise ones): dice = -mean((2I + smooth) / clip(G + P + smooth, 1e-8)) over (b, c) or,
 * with batch_dice, over c of the batch sums (dice.py:105-119), classes 1.. only unless do_bg; ce = sum(ce_sum) / n with
 * n = B*V, or the number of non-ignored voxels clamped to >= 1 when use_valid_count (compound_losses.py:44-52).
 * loss_accum[0] += ds_weight * (weight_ce * ce + weight_dice * dice)   (deep_supervision.py:30: sum of weighted losses)
 * coef[b] = ds_weight * {dL/dI[C], dL/dP[C], dL/dce_sum}.  _backward_scaled multiplies coef by the device scalar
 * gmul_device[0] (the upstream gradient, i.e. the GradScaler's loss scale) when it is not NULL. */
int nnz_dc_ce_loss_finalize(const float* sums, float* loss_accum, float* coef, int B, int C, long V, int batch_dice,
                            int do_bg, float smooth, float weight_ce, float weight_dice, float ds_weight,
                            int use_valid_count, void* stream);
int nnz_dc_ce_loss_backward_scaled(const void* logits, int logits_is_f16, const int16_t* target, const float* coef,
                                   const float* gmul_device, void* dlogits, int B, int C, long V, int ignore_label,
                                   void* stream);

/* region-based training (label_manager.has_regions): sigmoid soft-Dice (do_bg) + BCE-with-logits statistics of
 * DC_and_BCE_loss (compound_losses.py:59-109) on one-hot region targets [B][Ct][V] (int16 0/1; Ct = C, or C+1 with the
 * ignore mask in the last channel).  sums[b] = {intersect[C], sum_pred[C], sum_gt[C], bce_sum, mask_sum};
 * coef[b] = {dL/dintersect[C], dL/dsum_pred[C], dL/dbce_sum}.  nnz_region_tp_fp_fn: validation statistics with the
 * prediction sigmoid(z) > 0.5 (nnUNetTrainer.validation_step, nnUNetTrainer.py:1188-1216), counts_u64[c] = {tp, fp, fn}. */
int nnz_dc_bce_loss_forward(const void* logits, int logits_is_f16, const int16_t* target_regions, float* sums, int B,
                            int C, int Ct, long V, void* stream);
int nnz_dc_bce_loss_backward(const void* logits, int logits_is_f16, const int16_t* target_regions, const float* coef,
                             void* dlogits, int B, int C, int Ct, long V, void* stream);
int nnz_region_tp_fp_fn(const void* logits, int logits_is_f16, const int16_t* target_regions, void* counts_u64, int B,
                        int C, int Ct, long V, void* stream);

/* ---- selective scan (Mamba S6), fp32, N = 16, B/C of shape (B, K, N, L), z = None ----------------------------
 * replaces mamba_ssm's selective_scan_cuda.fwd/bwd behind selective_scan_fn as called at
 * nnunetv2/nets/m2net.py:193-199 and ssnd2net.py:271-277 (definition: selective_scan_ref,
 * nnunetv2/nets/seg_mamba/selective_scan_interface.py:86-152).  u, delta, y: (B, K*Dg, L); A: (K*Dg, N).
 * chunk_state (forward checkpoints, kept for backward): nnz_selective_scan_state_floats floats; grad_state:
 * nnz_selective_scan_grad_state_floats floats; workspace: nnz_selective_scan_workspace_floats floats.  Large calls
 * (L % 64 == 0, Dg = 32 or a multiple of 64, at least nnz_scan_tuning knob 2 row-steps) run on the channels-on-lanes
 * kernels of csrc/ss2d_scan_rl.hpp like the cross-scan below; the rest on the time-on-lanes kernels. */
long nnz_selective_scan_workspace_floats(int Bt, int KD, int L);
long nnz_selective_scan_state_floats(int Bt, int KD, int L);
long nnz_selective_scan_grad_state_floats(int Bt, int KD, int L);
int nnz_selective_scan_forward(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                               const float* D, const float* delta_bias, float* y, float* chunk_state,
                               float* workspace, int Bt, int K, int Dg, int N, int L, int delta_softplus,
                               void* stream);
int nnz_selective_scan_backward(const float* u, const float* delta, const float* A, const float* Bm, const float* Cm,
                                const float* D, const float* delta_bias, const float* dy, const float* chunk_state,
                                float* grad_state, float* workspace, float* du, float* ddelta, float* dA, float* dB,
                                float* dC, float* dD, float* dbias, int Bt, int K, int Dg, int N, int L,
                                int delta_softplus, void* stream);

/* ---- Swin window attention core (7x7 windows, fp32, head_dim <= 32) ------------------------------------------
 * replaces roll + window partition + softmax(q*scale k^T + bias (+ shift mask)) v + un-partition + roll back of
 * WindowAttention.forward (nnunetv2/nets/swt2net.py:584-619 == nets/swt.py:346-383).  qkv: (B, H, W, 3C) output of
 * the qkv Linear, channel order (3, heads, head_dim); bias_table: the (169, heads) relative_position_bias_table
 * parameter, bias_index: relative_position_index as int32 (49 x 49); out: (B, H, W, C) input of proj.
 * H, W multiples of 7; shift = 0 or 3. */
int nnz_window_attention_forward(const float* qkv, const float* bias_table, const int* bias_index, float* out, int B,
                                 int H, int W, int C, int heads, int shift, float scale, void* stream);
/* backward: dbias_table is WRITTEN (deterministic: fixed-order sums per table entry, fixed-point adds across workgroups);
 * acc = heads * 170 zeroed records of nnz_fxacc_bytes() bytes (169 table entries + one ticket record per head), left zero;
 * counter is unused (may be NULL).
 * bias_index must have the reference's displacement layout index[i][j] = (yi - yj + 6) * 13 + (xi - xj + 6). */
int nnz_window_attention_backward(const float* qkv, const float* bias_table, const int* bias_index, const float* dout,
                                  float* dqkv, float* dbias_table, void* acc, void* counter, int B, int H, int W, int C,
                                  int heads, int shift, float scale, void* stream);

/* LayerNorm backward with deterministic dgamma / dbeta (fixed-point cross-workgroup sums, csrc/common.hpp): acc = 2 * C
 * zeroed records of nnz_fxacc_bytes() bytes, counter = one zeroed 32-bit word, both left zero; dgamma / dbeta are written */
int nnz_layer_norm_backward_det(const void* x, int x_is_f16, const float* gamma, const float* mean, const float* rstd,
                                const void* dy, int dy_is_f16, void* dx, float* dgamma, float* dbeta, void* acc,
                                void* counter, long rows, int C, void* stream);
int nnz_layer_norm_gate_backward_det(const void* x, int x_is_f16, const float* gamma, const float* beta, const void* z,
                                     int z_is_f16, long z_row_stride, const float* mean, const float* rstd,
                                     const void* dy, int dy_is_f16, void* dx, void* dz, float* dgamma, float* dbeta,
                                     void* acc, void* counter, long rows, int C, void* stream);
/* nnz_layer_norm_backward_det with dx = dres + (LayerNorm backward): dres [rows][C], of the type of x / dx, is the gradient
 * arriving on the residual stream x that the block also normalised (x + f(LayerNorm(x)), swt2net.py:646-659, m2net.py:530);
 * NULL = the plain backward. */
int nnz_layer_norm_backward_det_res(const void* x, int x_is_f16, const float* gamma, const float* mean, const float* rstd,
                                    const void* dy, int dy_is_f16, const void* dres, void* dx, float* dgamma,
                                    float* dbeta, void* acc, void* counter, long rows, int C, void* stream);

/* ---- global (ViT) multi-head self-attention core, fp32 MFMA, flash-style (csrc/global_attention.hip, round 3) -----------
 * replaces monai's SABlock einsum / softmax / einsum between its qkv and out_proj Linears (bound by the reference at
 * nnunetv2/nets/unetr2net.py:10,1414-1428).  qkv [B][L][3][H][D], out [B][L][H*D], lse [B][H][L]; D even, <= 32.
 * The backward writes all of dqkv and uses no atomics (keys-side pass: dK, dV; queries-side pass: dQ). */
int nnz_global_attention_forward(const float* qkv, float* out, float* lse, int B, int L, int H, int D, float scale,
                                 void* stream);
int nnz_global_attention_backward(const float* qkv, const float* out, const float* lse, const float* dout, float* dqkv,
                                  int B, int L, int H, int D, float scale, void* stream);

/* ---- consumer-side InstanceNorm + LeakyReLU (round 4): "conv + norm + act fused" --------------------------------------------
 * The reference op sequence of every PlainConvUNet block is Conv -> InstanceNorm(affine) -> LeakyReLU
 * (nnunetv2/experiment_planning/experiment_planners/default_experiment_planner.py:285-305).  The statistics need the whole
 * volume, so the producer cannot normalise its own tiles; instead every CONSUMER of a block's activation takes the block's RAW
 * fp16 conv output plus its table nstat[N][C][4] = {mean, rstd, scale, shift} (written by the producer launch's last workgroup,
 * nnz_conv_tap_forward_norm*) and applies y = lrelu(x * scale + shift) while it stages its operand - fp32 FMA, LeakyReLU, ONE
 * rounding to fp16: bit for bit what nnz_instnorm_lrelu_apply_tab would have written.  The activated tensor never exists in HBM
 * (one write and one read of every activation less per forward pass, the `act` buffers gone).  Zero padding stays zero.
 *   tab == NULL selects the plain behaviour where the comment says so.  c0: channels [0, c0) of the operand are used as they are
 *   (a cat buffer's transposed-conv half); tab then holds C - c0 channels per sample. */
int nnz_conv_tap_forward_innorm(const void* in_raw_f16, void* out_f16, const void* w_packed_f16, const float* bias,
                                const nnz_conv_desc* desc, const float* in_tab, int in_c0 /* % 16 == 0 */, float in_slope,
                                void* acc, void* counter, const float* gamma, const float* beta, float eps,
                                float* nstat /* acc .. nstat: all NULL = no statistics of the output */,
                                float* workspace /* may be NULL */, long ws_floats, void* stream);
/* weight gradient with raw operands: boxed = the layer input of a convolution, plain = the lower-resolution activation of a
 * transposed convolution; either table may be NULL (operand used as is); c0 % 32 == 0; N <= 64 */
int nnz_conv_tap_wgrad_to_grad_innorm(const void* boxed_f16, const void* plain_f16, float* workspace, long ws_floats, float* grad,
                                      long sa, long sb, long sk, const int* ksel, int accumulate, const nnz_conv_desc* desc,
                                      const float* boxed_tab, int boxed_c0, float boxed_slope, const float* plain_tab,
                                      int plain_c0, float plain_slope, void* stream);
/* 1x1 segmentation head and its weight gradient on the raw output of the stage's last block (in_tab NULL = plain entry points) */
int nnz_seg_head_forward_innorm(const void* x_raw_f16, const float* in_tab, float in_slope, const float* w, const float* bias,
                                void* logits_f16_nc, int N, long V, int C, int K, int ldx, void* stream);
int nnz_seg_head_wgrad_innorm(const void* x_raw_f16, const float* in_tab, float in_slope, const void* dlogits_f16, float* dw,
                              float* db, int N, long V, int C, int K, int ldx, void* acc, void* counter, void* stream);
/* kernel = stride transposed convolution (full-resolution stages) on the raw output of the block below */
int nnz_convT_forward_innorm(const void* in_raw_f16, const float* in_tab, float in_slope, const float* W, const float* bias,
                             void* out, int N, int Di, int Hi, int Wi, int Cin, int Cout, int sd, int sh, int sw, int ldi, int ldo,
                             void* stream);

/* ---- fused AdamW tail of the zoo trainers (round 4): GradScaler.unscale_ + clip_grad_norm_ + AdamW step (the X^2-Net plugins'
 * optimizer, nnunetv2/training/nnUNetTrainer/nnUNetTrainerM2Net.py:58-65 behind nnUNetTrainer.py:1131-1139) as TWO launches over
 * a device table of chunks {float* param; const float* grad; float* exp_avg; float* exp_avg_sq; int n; int owner}
 * (nnz_adam_chunk_bytes() bytes each, n <= 16384), replacing ~260 multi-tensor launches per step.  stats2 (2 floats, written) =
 * {sum of squares of the unscaled gradients g * inv_scale, > 0 when the step was skipped (non-finite gradients)}; acc / counter: 3 zeroed
 * fixed-point records (nnz_fxacc_bytes() each) + one zeroed 32-bit word, left zero; steps[nsteps]: the parameters' fp32 step
 * counters (chunk.owner indexes them; bias corrections are per parameter), advanced by one when the step is applied; inv_scale_device: 1 / loss scale on the device or NULL; max_norm <= 0
 * disables clipping.  Deterministic (fixed-point gradient norm), no host synchronisation. */
int nnz_adam_chunk_bytes(void);
int nnz_adamw_fused(const void* chunks_device, int nchunks, float* stats2, void* acc, void* counter,
                    const float* inv_scale_device, float max_norm, float lr, double beta1, double beta2 /* doubles: 1 - beta
                    is formed in double like torch does */, float eps, float weight_decay, float* steps, int nsteps,
                    void* stream);

/* ---- round 5: the Swin block as five forward and seven backward launches (SwinTransformerBlock.forward,
 * nnunetv2/nets/swt2net.py:622-661: F.pad -> norm1 -> WindowAttention (qkv, core, proj) -> DropPath + residual -> norm2 -> Mlp ->
 * DropPath + residual -> crop; the block used to be 11 + 11 launches).  Host side: nnuzoo_amd/swin_block.py.
 *   forward:  [pad gather + LayerNorm + qkv Linear] -> window attention (out on the unpadded grid) -> [proj Linear + DropPath +
 *             residual] -> [LayerNorm + fc1 + GELU] -> [fc2 + DropPath + residual]
 *   backward: [fc2 dgrad x DropPath x GELU'] -> fc1 dgrad -> [LayerNorm backward + skip] -> [proj dgrad x DropPath] -> window
 *             attention backward (dout on the unpadded grid) -> qkv dgrad -> [LayerNorm backward + skip + crop]; the weight
 *             gradients (DropPath folded into their dy operand) and the LayerNorm dgamma / dbeta folds ride in the pass's ONE
 *             grouped launch.
 * nnz_dense32_forward_fused: y = epilogue(LN?(x) W^T + bias), T = rows of y.
 *   ln_mean != NULL: LayerNorm prologue over K (eps ln_eps, affine ln_gamma / ln_beta or NULL): writes ln_mean / ln_rstd [T] and,
 *     when ln_y != NULL, the normalised rows [T][K]; pad_h > 0 (LayerNorm mode only): x is [B][pad_h][pad_w][K] and row r of y is
 *     a token of the top / left padded grid (pad_h + pad_y) x (pad_w + pad_x); padded tokens are zero rows (normalised: beta);
 *   res != NULL: y = res + s (x W^T + bias), res laid out like y; dp_rand != NULL: s = floor(dp_keep + dp_rand[row / dp_rps]) /
 *     dp_keep over dp_nb samples (the reference's DropPath, swt2net.py:379-388), NULL: s = 1; gelu as in nnz_dense32_forward;
 *   workspace: nnz_dense32_splitk_workspace_floats(T, K, N) floats or NULL.  Skinny products (< 192 tiles of 64 x 64, contraction
 *     >= 256: the 8^2 ... 16^2 token levels of the Swin U-nets) are cut along the contraction into <= 16 ranges whose partials a
 *     second launch folds in range order (bit-identical run to run) before the epilogue.
 * nnz_dense32_dgrad_fused: dx = s (dy W) [* GELU'(h)], s as above; workspace: ..._floats(T, N, K) (contraction N, K columns). */
long nnz_dense32_splitk_workspace_floats(long T, int contraction, int out_cols);
int nnz_dense32_forward_fused(const float* x, const float* W, const float* bias, float* y, float* y_act, long T, int K, int N,
                              int gelu, const float* ln_gamma, const float* ln_beta, float ln_eps, float* ln_mean,
                              float* ln_rstd, float* ln_y, int pad_h, int pad_w, int pad_y, int pad_x, const float* res,
                              const float* dp_rand, float dp_keep, int dp_rps, int dp_nb, float* workspace, void* stream);
int nnz_dense32_dgrad_fused(const float* dy, const float* W, const float* h, float* dx, long T, int K, int N,
                            const float* dp_rand, float dp_keep, int dp_rps, int dp_nb, float* workspace, void* stream);
/* The two products for fp16 activations (round 5): x, y, dy, dx are _Float16, W / bias fp32, fp32 accumulation - the token
 * Linears of the autocast nets that the fp16 token kernel does not take (F.linear under torch.autocast, nnUNetTrainer.py:1128-1139;
 * m2net.py:230-262 at the 8^2 .. 32^2 levels), without cast launches.  workspace as for the _fused entry points. */
int nnz_dense32_forward_h16(const void* x, const float* W, const float* bias, void* y, long T, int K, int N, float* workspace,
                            void* stream);
int nnz_dense32_dgrad_h16(const void* dy, const float* W, void* dx, long T, int K, int N, float* workspace, void* stream);
/* grouped weight-gradient record with dy and x as _Float16 (see nnz_dense32_group_fill) */
int nnz_dense32_group_fill_h16(void* job_host, void* fold_host, const void* dy, const void* x, float* dW, float* db,
                               float* workspace, long T, int K, int N, int wg_begin, int blk_begin);
/* grouped weight gradient record with the DropPath scale on its dy operand (per token: s of sample token / dp_rps); and a
 * fold-only record: dst[i] = sum_{q < parts} part[q * n + i] in part order (fold blocks: (n + 255) / 256) */
int nnz_dense32_group_fill_scaled(void* job_host, void* fold_host, const float* dy, const float* x, float* dW, float* db,
                                  float* workspace, long T, int K, int N, int wg_begin, int blk_begin, const float* dp_rand,
                                  float dp_keep, int dp_rps, int dp_nb);
int nnz_dense32_group_fill_fold(void* fold_host, const float* part, float* dst, long n, int parts, int blk_begin);
/* fold records alone (for grouped launches of other kernel families) */
int nnz_group_fold_launch(const void* fold_dev, const int* blk_job_dev, int total_blks, void* stream);
/* ---- grouped weight gradients of the fp16 token Linears (csrc/token_linear.hip, round 5): every nnz_token_linear_wgrad problem of
 * a backward pass (in_proj / out_proj / patch merge / expand of the VSS blocks, m2net.py:97,103,258,300: ~160 per M2Net step) in
 * ONE launch over a job table + ONE fold launch.  Protocol as nnz_dense32_group_*: _plan gives workgroups, dynamic LDS bytes and
 * workspace floats (workgroups x (N K + N): each workgroup writes its partial dW | db block); _fill writes a HOST record of
 * _record_bytes() bytes; the launch takes the LARGEST LDS size of its jobs; the partial blocks are summed in workgroup order by
 * fold records (nnz_dense32_group_fill_fold, n = N K + N, parts = workgroups) and nnz_group_fold_launch: no zero fills, no float
 * atomics - bit-identical from run to run. */
int nnz_token_linear_wgrad_group_record_bytes(void);
int nnz_token_linear_wgrad_group_plan(long T, int N, int K, int* wgs, int* lds_bytes, long* ws_floats);
int nnz_token_linear_wgrad_group_fill(void* job_host, const void* dy_f16, const void* x_f16, float* workspace, long T, int N, int K,
                                      int wg_begin);
int nnz_token_linear_wgrad_group_launch(const void* jobs_dev, const int* wg_job_dev, int total_wgs, int max_lds_bytes,
                                        void* stream);
/* the same for the SS2D x_proj weight gradients (csrc/ss2d_xproj.hip, ~66 per M2Net step): workgroups = 2 sources x token ranges,
 * workspace = one partial [2][C2][Di] matrix per token range; fold records with n = 2 C2 Di, parts = workgroups / 2 */
int nnz_ss2d_xproj_backward_w_group_record_bytes(void);
int nnz_ss2d_xproj_backward_w_group_plan(int B, int Di, int C2, long L, int* wgs, int* lds_bytes, long* ws_floats);
int nnz_ss2d_xproj_backward_w_group_fill(void* job_host, const float* dP, const float* x2, float* workspace, int B, int Di, int C2,
                                         long L, int cp, int wg_begin);
int nnz_ss2d_xproj_backward_w_group_launch(const void* jobs_dev, const int* wg_job_dev, int total_wgs, int max_lds_bytes,
                                           void* stream);
/* window attention with qkv / dqkv on the block's top / left padded grid H x W and out / dout on the unpadded grid
 * (H - py) x (W - px): rows of padded tokens are never written (forward) and read as zeros (backward) */
int nnz_window_attention_forward_pad(const float* qkv, const float* bias_table, const int* bias_index, float* out, int B,
                                     int H, int W, int C, int heads, int shift, float scale, int py, int px, void* stream);
int nnz_window_attention_backward_pad(const float* qkv, const float* bias_table, const int* bias_index, const float* dout,
                                      float* dqkv, float* dbias_table, void* acc, void* counter, int B, int H, int W, int C,
                                      int heads, int shift, float scale, int py, int px, void* stream);
/* the backward with the bias-table gradient left as per-workgroup shares dpart[nnz_window_attention_backward_parts(B, H, W,
 * heads)][169][heads] for a fold record (n = 169 * heads): no fixed-point adds, no last-workgroup pass in the launch */
long nnz_window_attention_backward_parts(int B, int H, int W, int heads);
int nnz_window_attention_backward_partial(const float* qkv, const float* bias_table, const int* bias_index, const float* dout,
                                          float* dqkv, float* dpart, int B, int H, int W, int C, int heads, int shift,
                                          float scale, int py, int px, void* stream);
/* LayerNorm backward (fp32) with dx = dres + ..., dgamma | dbeta as per-workgroup partials part[nnz_layer_norm_backward_parts(rows,
 * C)][2 C] for a fold record, and - pad_h > 0 - rows on the top / left padded grid (dy, mean, rstd) with x / dres / dx on the
 * unpadded [B][pad_h][pad_w] grid (padded tokens: x = 0, they still count in dgamma / dbeta, their dx is dropped) */
long nnz_layer_norm_backward_parts(long rows, int C);
int nnz_layer_norm_backward_partial(const float* x, const float* gamma, const float* mean, const float* rstd, const float* dy,
                                    const float* dres, float* dx, float* part, long rows, int C, int pad_h, int pad_w,
                                    int pad_y, int pad_x, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* NNUZOO_HIP_H */
