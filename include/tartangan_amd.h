/*
 * tartangan_amd.h -- C ABI of libtartangan_amd.so (gfx950 / MI355X).
 *
 * The drop-in boundary for tartangan's SA-GAN / SA-GAN-IQN G+D training step
 * (SURVEY.md §8b).  The reference has no native code: every entry point below
 * replaces the ATen op(s) that the cited reference line dispatches through
 * torch.nn / torch.nn.functional.  Paths are relative to
 * /root/reference/tartangan.
 *
 * Conventions
 *   - all tensors fp32, contiguous, NCHW (the reference's layout); raw device
 *     pointers; the caller owns every byte including workspaces;
 *   - enqueue-only on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream); no allocation, no synchronisation, graph-capture safe;
 *   - return 0 on success, a negative TG_E* code for a bad argument or an
 *     unsupported shape, a positive value = hipError_t from the launch;
 *   - re-entrant; no mutable global state.
 */
#ifndef TARTANGAN_AMD_H
#define TARTANGAN_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TG_OK 0
#define TG_EINVAL (-1)      /* null pointer / non-positive dimension */
#define TG_EUNSUPPORTED (-2) /* shape outside what the kernels implement */
#define TG_EWORKSPACE (-3)  /* workspace too small */

/* library / build identification */
int tg_version(void);
const char* tg_arch(void);

/* ---------------------------------------------------------------- convolution
 * nn.Conv2d(k=3,pad=1) / nn.Conv2d(k=1): models/blocks/generator.py:41,44,52,124
 * models/blocks/discriminator.py:17,63,66,78  models/blocks/attention.py:14-17
 * ks in {1,3}, stride 1, zero pad ks/2.  w is OIHW [Cout][Cin][ks][ks].        */
int tg_conv2d_fwd(const float* x, const float* w, const float* bias /*nullable*/,
                  const float* residual /*nullable: y = conv + bias + residual (x + h, generator.py:62)*/,
                  float* y, int B, int Cin, int Cout, int H, int W, int ks, void* stream);
/* The generator block's `h + up(project(x))` (generator.py:52-62: x = F.interpolate(x, scale_factor=2) first, then
 * convs(x) + project_input(x)): the 1x1 projection commutes with nearest-neighbour upsampling, so the shortcut stays at
 * HALF the resolution and the last 3x3 conv adds it upsampled on the fly: y = conv3x3(x) + bias + up2x(residual_lo),
 * residual_lo is (B, Cout, H/2, W/2); H, W even.  No high-resolution copy of the shortcut is ever written. */
int tg_conv2d_fwd_up2res(const float* x, const float* w, const float* bias /*nullable*/, const float* residual_lo,
                         float* y, int B, int Cin, int Cout, int H, int W, void* stream);
/* conv3x3(nearest_up2x(a)) + bias [+ residual] without the upsampled tensor (generator.py:52-58: the first 3x3
 * conv of every generator block reads an upsampled activation).  Output pixel (2i+dy, 2j+dx) only sees a 2x2 block
 * of source pixels, so the layer is four 2x2-tap convolutions at the LOW resolution with row/column-summed filters:
 *   tg_upconv3x3_weights  w [Cout][Cin][3][3] -> wp [4 phases (dy,dx)][Cout][Cin][2][2]
 *   tg_upconv3x3_fwd      a (B,Cin,H,W), wp -> y (B,Cout,2H,2W); residual (nullable) has y's shape
 * 2.25x fewer multiply-adds than the reference's formulation, same result up to fp32 rounding of the summed taps. */
int tg_upconv3x3_weights(const float* w, float* wp, int Cout, int Cin, void* stream);
int tg_upconv3x3_fwd(const float* a, const float* wp, const float* bias /*nullable*/, const float* residual /*nullable*/,
                     float* y, int B, int Cin, int Cout, int H, int W, void* stream);
/* gradient of tg_upconv3x3_fwd w.r.t. its low-resolution input (what autograd computes as pool2x2sum(dgrad3x3(gy))):
 * a 4x4-tap stride-2 convolution over gy (B,Cout,2H,2W) with w4t [Cin][Cout][4][4] from tg_upconv3x3_weights_t.
 * Only where the low-resolution plane gives >= 256 workgroups (tg_upconv3x3_dgrad_supported); callers fall back to
 * tg_conv2d_dgrad + tg_pool2 otherwise.                                                                    */
int tg_upconv3x3_weights_t(const float* w, float* w4t, int Cout, int Cin, void* stream);
/* both layouts in one launch (a training step needs both) */
int tg_upconv3x3_weights_pair(const float* w, float* wp, float* w4t, int Cout, int Cin, void* stream);
int tg_upconv3x3_dgrad_supported(int B, int Cin, int Cout, int H, int W);
int tg_upconv3x3_dgrad(const float* gy, const float* w4t, float* ga, int B, int Cin, int Cout, int H, int W, void* stream);
/* AvgPool2d(2)(conv3x3(x) + bias) [+ residual] as ONE 4x4-tap stride-2 convolution (discriminator.py:60-66: the second
 * 3x3 conv of every discriminator block is followed by the block's average pool): 16 products per OUTPUT pixel and
 * channel pair instead of 36, no full-resolution conv output, no pool pass.  The pair is 0.25 x the transpose of the
 * up-conv above with the flipped, transposed filter, so the same two kernels serve with their roles swapped:
 *   tg_poolconv3x3_weights  w [Cout][Cin][3][3] -> w4 [Cout][Cin][4][4] (forward) and wp [4][Cin][Cout][2][2] (dgrad)
 *   tg_poolconv3x3_fwd      x (B,Cin,2H,2W), w4 -> y (B,Cout,H,W); bias / residual (shape of y) nullable
 *   tg_poolconv3x3_dgrad    gy (B,Cout,H,W), wp -> gx (B,Cin,2H,2W)
 * Supported where H x W >= 16 x 16 gives both kernels >= 256 workgroups (tg_poolconv3x3_supported); the weight
 * gradient is tg_conv2d_wgrad(x, 0.25 * up2x(gy)).                                                          */
int tg_poolconv3x3_weights(const float* w, float* w4, float* wp, int Cout, int Cin, void* stream);
int tg_poolconv3x3_supported(int B, int Cin, int Cout, int H, int W);
int tg_poolconv3x3_fwd(const float* x, const float* w4, const float* bias /*nullable*/, const float* residual /*nullable*/,
                       float* y, int B, int Cin, int Cout, int H, int W, void* stream);
int tg_poolconv3x3_dgrad(const float* gy, const float* wp, float* gx, int B, int Cin, int Cout, int H, int W, void* stream);
/* Weight gradients of the two stride-2 forms (convolution_backward grad_weight of the 3x3 filter, deterministic):
 * both are T[o][i][u][v] = sum lo[o][r][c] * hi[i][2r-1+u][2c-1+v] folded back onto the 3x3 taps -- 16 products per
 * low-resolution pixel instead of 36 on the materialised high-resolution pair.  (B, ., H, W) is the LOW-resolution
 * plane; where the plane is smaller than 16 x 16 they return TG_EUNSUPPORTED and the caller uses tg_conv2d_wgrad.
 * gbias (nullable, Cout floats): the bias gradient sum_{b,h,w} gy, taken from the operands these kernels stage anyway
 * (convolution_backward's grad_bias; `accumulate` applies to it as to gw).                                          */
size_t tg_poolconv3x3_wgrad_workspace(int B, int Cin, int Cout, int H, int W);
int tg_poolconv3x3_wgrad(const float* x /*(B,Cin,2H,2W)*/, const float* gy /*(B,Cout,H,W)*/, float* gw, float* workspace,
                         size_t workspace_bytes, int B, int Cin, int Cout, int H, int W, int accumulate, float* gbias /*nullable*/,
                         void* stream);
size_t tg_upconv3x3_wgrad_workspace(int B, int Cin, int Cout, int H, int W);
int tg_upconv3x3_wgrad(const float* a /*(B,Cin,H,W)*/, const float* gy /*(B,Cout,2H,2W)*/, float* gw, float* workspace,
                       size_t workspace_bytes, int B, int Cin, int Cout, int H, int W, int accumulate, float* gbias /*nullable*/,
                       void* stream);
/* gx = d/dx: correlation of gy with the transposed, spatially flipped filter
 * (what autograd's convolution_backward computes for grad_input)              */
int tg_conv2d_dgrad(const float* gy, const float* w, float* gx,
                    int B, int Cin, int Cout, int H, int W, int ks, void* stream);
/* gw[co][ci][kh][kw] = sum_{b,h,w} gy[b,co,h,w] * x[b,ci,h+kh-p,w+kw-p]
 * (convolution_backward grad_weight) and, when gbias != NULL, gbias[co] = sum_{b,h,w} gy (grad_bias) from
 * the same pass; deterministic two-stage reduction.  accumulate != 0: add into gw / gbias instead of
 * overwriting (gradient accumulation straight into the flat .grad bucket).                              */
size_t tg_conv2d_wgrad_workspace(int B, int Cin, int Cout, int H, int W, int ks);
int tg_conv2d_wgrad(const float* x, const float* gy, float* gw, float* gbias /*nullable*/,
                    float* workspace, size_t workspace_bytes,
                    int B, int Cin, int Cout, int H, int W, int ks, int accumulate, void* stream);
/* SelfAttention2d's theta / phi / g projections (attention.py:15-17, 22-26: three bias-free 1x1 convolutions of the SAME input)
 * as ONE pass each way: x is read once for the three outputs, the three output gradients give the input gradient in one pass
 * (no three-way add) and the three filter gradients in one pass.  w / gw: the three filters back to back,
 * [c0 + c1 + c2][Cin] (they are adjacent in the parameter bucket); y_k / gy_k: (B, c_k, H, W) each.  Bandwidth-bound
 * streaming kernels (no LDS); need H*W % 16 == 0, at most 64 channels either side, 16-byte aligned tensors. */
int tg_conv1x1_multi_supported(int c0, int c1, int c2, int B, int Cin, int H, int W);
int tg_conv1x1_multi_fwd(const float* x, const float* w, float* y0, float* y1, float* y2, int c0, int c1, int c2,
                         int B, int Cin, int H, int W, void* stream);
int tg_conv1x1_multi_dgrad(const float* gy0, const float* gy1, const float* gy2, const float* w, float* gx,
                           int c0, int c1, int c2, int B, int Cin, int H, int W, void* stream);
size_t tg_conv1x1_multi_wgrad_workspace(int c0, int c1, int c2, int B, int Cin, int H, int W);
int tg_conv1x1_multi_wgrad(const float* x, const float* gy0, const float* gy1, const float* gy2, float* gw,
                           float* workspace, size_t workspace_bytes, int c0, int c1, int c2, int B, int Cin, int H, int W,
                           int accumulate /* gw += */, void* stream);
/* The same in two steps, so that ONE launch can finish the reduction for many layers (a whole backward pass):
 *   tg_conv2d_wgrad_partials     stage 1 only: per-workgroup partial sums into `workspace` (which must then stay
 *                                untouched until the batch reduce has run);
 *   tg_conv2d_wgrad_reduce_batch stage 2 for n_items layers.  `items` is a HOST array of
 *                                n_items x TG_WGRAD_ITEM_FIELDS 64-bit words:
 *                                  [0] workspace (device address)   [1] gw (device address)
 *                                  [2] gbias (device address or 0; nonzero needs want_bias != 0 in stage 1)
 *                                  [3..8] B, Cin, Cout, H, W, ks     [9] accumulate
 * Results are bit-identical to tg_conv2d_wgrad (same partials, same summation order).  Within one call every
 * gw / gbias may appear at most once (items are reduced concurrently); further contributions to the same
 * gradient go into a later call.                                                                           */
typedef int64_t tg_host_i64;
#define TG_WGRAD_ITEM_FIELDS 10
int tg_conv2d_wgrad_partials(const float* x, const float* gy, float* workspace, size_t workspace_bytes,
                             int B, int Cin, int Cout, int H, int W, int ks, int want_bias, void* stream);
int tg_conv2d_wgrad_reduce_batch(const tg_host_i64* items /*host*/, int n_items, void* stream);
/* tg_poolconv3x3_weights for n_items layers in one launch (a discriminator's stride-2 layers all need their derived filters at
 * the start of a pass).  `items`: HOST array of n_items x TG_FORM_ITEM_FIELDS words: [0] w  [1] w4  [2] wp (device addresses)
 * [3] Cout  [4] Cin.  Same arithmetic per layer as the single call. */
#define TG_FORM_ITEM_FIELDS 5
int tg_poolconv3x3_weights_batch(const tg_host_i64* items /*host*/, int n_items, void* stream);
/* The stride-2 weight gradients in two steps, like tg_conv2d_wgrad_partials / _reduce_batch: stage 1 per layer into a workspace of its
 * own (tg_*_wgrad_workspace bytes, untouched until the reduce has run), then ONE launch that sums, folds onto the 3x3 taps and
 * finishes the bias gradients of all recorded layers.  Items as for tg_conv2d_wgrad_reduce_batch with [8] = 0 (pooled conv: stage 1
 * by tg_poolconv3x3_wgrad_partials) or 1 (up-conv: tg_upconv3x3_wgrad_partials) in the place of ks; (B, ., H, W) the LOW-resolution
 * plane; [2] gbias nonzero needs want_bias != 0 in stage 1.  Bit-identical to the one-call forms.                               */
/* The discriminator's from-RGB 1x1 convolution composed into the 3x3 convolution that follows it (discriminator.py:11-22 + :60-61,
 * nothing in between): wc [Cout][Cimg + 1][3][3] = sum_m [w1 | b1][m][c] * w3[co][m][tap] (the from-RGB bias on an all-ones input channel),
 * and its backward (gw1 [C][Cimg], gb1 [C], gw3 [Cout][C][3][3]; accumulate: += into all three).  One launch each.          */
int tg_rgb_compose_fwd(const float* w1, const float* b1, const float* w3, float* wc, int Cout, int C, int Cimg, void* stream);
int tg_rgb_compose_bwd(const float* gwc, const float* w1, const float* b1, const float* w3, float* gw1, float* gb1, float* gw3, int Cout,
                       int C, int Cimg, int accumulate, void* stream);
int tg_poolconv3x3_wgrad_partials(const float* x, const float* gy, float* workspace, size_t workspace_bytes, int B, int Cin, int Cout,
                                  int H, int W, int want_bias, void* stream);
int tg_upconv3x3_wgrad_partials(const float* a, const float* gy, float* workspace, size_t workspace_bytes, int B, int Cin, int Cout,
                                int H, int W, int want_bias, void* stream);
int tg_s2_wgrad_reduce_batch(const tg_host_i64* items /*host*/, int n_items, void* stream);
/* out[c] (+)= sum_{b,p} x[b][c][p]   (linear bias grad); workspace: tg_bn_workspace(B,C,HW) bytes */
int tg_channel_sum(const float* x, float* out, float* workspace, int B, int C, int HW, int accumulate,
                   void* stream);
/* out[b][c][p] = v[c] (transpose of tg_channel_sum) */
int tg_channel_bcast(const float* v, float* out, int B, int C, int HW, void* stream);

/* ---------------------------------------------------------------- GEMM
 * nn.Linear (generator.py:70-72, discriminator.py:137-139,158-160, iqn.py:33-35)
 * and torch.bmm (attention.py:32,34).  Row-major; C[b] = op(A[b]) op(B[b]) (+ bias[n]) + beta*C[b].
 * op(A) is MxK, op(B) is KxN; lda/ldb/ldc are row strides of the stored arrays. */
int tg_gemm(const float* A, const float* Bm, float* C, const float* bias_n /*nullable*/,
            int M, int N, int K, int lda, int ldb, int ldc, int transA, int transB,
            int batch, int64_t strideA, int64_t strideB, int64_t strideC, float beta, void* stream);

/* ---------------------------------------------------------------- BatchNorm2d (+LeakyReLU)
 * nn.BatchNorm2d train mode followed by nn.LeakyReLU(0.2): generator.py:38-44,
 * discriminator.py:60-66,133-136,153-156.  slope = 1 gives plain BatchNorm.
 * Workspace: tg_bn_workspace(B,C,HW) bytes.                                       */
size_t tg_bn_workspace(int B, int C, int HW);
/* batch mean / 1/sqrt(biased var + eps); optional running-stat update
 * (running_var uses the unbiased variance, momentum as nn.BatchNorm2d); *num_batches_tracked += 1.
 * replicate >= 1: x stands for a tensor in which every element occurs `replicate` times (the generator normalises
 * nearest-upsampled activations, generator.py:52-57: same mean and biased variance, so the statistics are taken at the
 * low resolution; only the unbiased correction n/(n-1) of running_var needs the true count n = replicate*B*HW).     */
int tg_bn_train_stats(const float* x, float* mean, float* invstd,
                      float* running_mean /*nullable*/, float* running_var /*nullable*/,
                      int64_t* num_batches_tracked /*nullable*/,
                      float momentum, float eps, float* workspace, int B, int C, int HW, int replicate, void* stream);
/* training forward in one call: tg_bn_train_stats followed by tg_bn_act_fwd (one kernel when a channel
 * has <= 16384 elements)                                                                              */
int tg_bn_train_fwd(const float* x, float* mean, float* invstd,
                    float* running_mean /*nullable*/, float* running_var /*nullable*/,
                    int64_t* num_batches_tracked /*nullable*/, const float* gamma, const float* beta,
                    float slope, float momentum, float eps, float* z, float* workspace,
                    int B, int C, int HW, int replicate, void* stream);
/* eval mode: mean = running_mean, invstd = 1/sqrt(running_var + eps) */
int tg_bn_eval_stats(const float* running_mean, const float* running_var, float* mean, float* invstd,
                     float eps, int C, void* stream);
/* z = lrelu(gamma * (x - mean) * invstd + beta, slope) */
int tg_bn_act_fwd(const float* x, const float* mean, const float* invstd, const float* gamma,
                  const float* beta, float slope, float* z, int B, int C, int HW, void* stream);
/* first backward (native_batch_norm_backward o leaky_relu_backward):
 * gyh = gz * lrelu'(y); ggamma = sum gyh*xhat; gbeta = sum gyh;
 * training: gx = gamma*invstd*(gyh - mean(gyh) - xhat*mean(gyh*xhat)); eval: gx = gamma*invstd*gyh */
int tg_bn_act_bwd(const float* gz, const float* x, const float* mean, const float* invstd,
                  const float* gamma, const float* beta, float slope, int training,
                  float* gx /*nullable*/, float* ggamma, float* gbeta, float* workspace,
                  int B, int C, int HW, int accumulate /* ggamma, gbeta += */,
                  const float* gx_add /*nullable: gx = ... + gx_add, a second gradient of x folded into the pass*/,
                  void* stream);
/* ---- grouped forms: ONE tensor of groups*B images normalised per (group, channel).  trainers/cnn.py:122-123 run the
 * discriminator on the real and on the fake batch as two forwards with the same weights; here both halves travel through
 * the network as one tensor of 2B images (convolutions are per image), and BatchNorm keeps the statistics of each half:
 * group g = images [g*B, (g+1)*B).  mean / invstd are [groups][C]; the running statistics are updated once per group, in
 * group order (exactly what the two forwards leave behind), *num_batches_tracked += groups; ggamma / gbeta are the sums
 * over all groups.  Workspace: tg_bn_workspace(B, groups*C, HW) bytes.  groups = 1 is tg_bn_train_fwd / tg_bn_act_bwd.
 * gx_add (nullable) holds the first add_groups*B images only (the R1 second-order gradient exists for the real half). */
int tg_bn_train_fwd_groups(const float* x, float* mean, float* invstd,
                           float* running_mean /*nullable*/, float* running_var /*nullable*/,
                           int64_t* num_batches_tracked /*nullable*/, const float* gamma, const float* beta,
                           float slope, float momentum, float eps, float* z, float* workspace,
                           int groups, int B, int C, int HW, int replicate, void* stream);
int tg_bn_act_bwd_groups(const float* gz, const float* x, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float slope, int training,
                         float* gx /*nullable*/, float* ggamma, float* gbeta, float* workspace,
                         int groups, int B, int C, int HW, int accumulate,
                         const float* gx_add /*nullable*/, int add_groups, void* stream);
/* second backward of the training-mode map (gz, x, gamma) -> (gx, ggamma, gbeta)
 * (NativeBatchNormBackwardBackward0; R1 penalty path, models/losses.py:23-26).
 * v = adjoint of gx, vgamma/vbeta = adjoints of ggamma/gbeta (nullable = 0).   */
int tg_bn_act_dbwd(const float* v, const float* vgamma /*nullable*/, const float* vbeta /*nullable*/,
                   const float* gz, const float* x, const float* mean, const float* invstd,
                   const float* gamma, const float* beta, float slope,
                   float* adj_gz, float* adj_x, float* adj_gamma, float* workspace,
                   int B, int C, int HW, int accumulate /* adj_gamma += */, void* stream);

/* ---- BatchNorm under data parallelism (SyncBN; SURVEY.md 8e collective 3).  The reference normalises over the whole
 * batch (trainers/cnn.py:122-123 evaluates D on all B images); when the batch is sharded over ranks the same numbers
 * need global statistics.  Each pass is split in three: *_local writes this rank's per-channel sums (float64,
 * [C][K] channel-major: K = 3 statistics (mean, mean^2, biased var), K = 2 first backward, K = 5 second backward), the
 * CALLER all-reduces them (SUM; torch.distributed -- RCCL on device buffers), *_finish completes the pass with the
 * global sums.  count_global = world * B * HW (equal shards).  Parameter gradients (ggamma, gbeta, adj_gamma) come out as
 * this rank's share, to be averaged by the gradient all-reduce like every other parameter gradient.
 * Workspace: tg_bn_workspace(B, C, HW) bytes; the *_finish call of the backward passes must get the same workspace
 * pointer contents-wise only in that it is scratch (nothing is carried from *_local).                              */
int tg_bn_sync_stats_local(const float* x, double* sums /*[C][3]*/, float* workspace, int B, int C, int HW,
                           void* stream);
int tg_bn_sync_stats_finish(const double* sums /*[C][3], all-reduced*/, int world, float* mean, float* invstd,
                            float* running_mean /*nullable*/, float* running_var /*nullable*/,
                            int64_t* num_batches_tracked /*nullable*/, float momentum, float eps,
                            int64_t count_global, int replicate, int C, void* stream);
int tg_bn_sync_bwd_local(const float* gz, const float* x, const float* mean, const float* invstd,
                         const float* gamma, const float* beta, float slope, double* sums /*[C][2]*/,
                         float* workspace, int B, int C, int HW, void* stream);
int tg_bn_sync_bwd_finish(const float* gz, const float* x, const float* mean, const float* invstd,
                          const float* gamma, const float* beta, float slope,
                          const double* local_sums /*[C][2]*/, const double* global_sums /*[C][2], all-reduced*/,
                          int64_t count_global, float* gx /*nullable*/, float* ggamma, float* gbeta,
                          float* workspace, int B, int C, int HW, int accumulate,
                          const float* gx_add /*nullable*/, void* stream);
int tg_bn_sync_dbwd_local(const float* v, const float* gz, const float* x, const float* mean,
                          const float* invstd, const float* gamma, const float* beta, float slope,
                          double* sums /*[C][5]*/, float* workspace, int B, int C, int HW, void* stream);
int tg_bn_sync_dbwd_finish(const float* v, const float* gz, const float* x, const float* mean,
                           const float* invstd, const float* gamma, const float* beta, float slope,
                           const double* global_sums /*[C][5], all-reduced*/, int64_t count_global, int world,
                           float* adj_gz, float* adj_x, float* adj_gamma, float* workspace,
                           int B, int C, int HW, int accumulate /* adj_gamma += */, void* stream);

/* ---------------------------------------------------------------- resampling
 * F.interpolate(scale_factor=2,'nearest') generator.py:58 and nn.AvgPool2d(2)
 * discriminator.py:67 are transposes of each other up to a factor:
 *   up2x : y[2h+i][2w+j] = alpha * x[h][w]       pool2: y[h][w] = alpha * sum_{i,j} x[2h+i][2w+j]
 * (nearest fwd = up2x(1), bwd = pool2(1); avgpool fwd = pool2(.25), bwd = up2x(.25)).
 * H, W are the dims of the SMALL side for up2x (output 2H x 2W) and of the LARGE
 * (input) side for pool2; both must be even for pool2.                          */
int tg_up2x(const float* x, float* y, float alpha, int BC, int H, int W, void* stream);
int tg_pool2(const float* x, const float* residual /*nullable: y = residual + pooled (discriminator.py:95)*/,
             float* y, float alpha, int BC, int H, int W, void* stream);
/* F.interpolate(scale_factor=0.5, mode='bilinear', align_corners=True) discriminator.py:55-57
 * input HxW -> output floor(H/2) x floor(W/2); _bwd is its transpose (scatter-free gather form), optionally
 * added to `residual` (the gradient that reached the same tensor through the block's main path) */
int tg_bilinear_half_fwd(const float* x, float* y, int BC, int H, int W, void* stream);
int tg_bilinear_half_bwd(const float* gy, const float* residual /*nullable, shape of gx*/, float* gx,
                         int BC, int H, int W, void* stream);
/* Input side of the FID / Inception-score pipeline (8f-2).  out (B, C, OH, OW) = bilinear resize (align_corners=True;
 * F.interpolate(x, size=(299, 299), mode='bilinear', align_corners=True), inception_utils.py:49-50) of x (B, C, H, W) after
 * `stages` applications of  x <- ((x + 1) / 2 - mean[c]) / std[c]  -- one in WrapInception.forward (inception_utils.py:44-47), a
 * second, identical one in accumulate_inception_activations' transform (:254-258), which the reference applies to the
 * generator's samples before WrapInception sees them.  stages = 0: resize only; OH == H and OW == W: normalise only. */
int tg_inception_preprocess(const float* x, const float* mean /*[C]*/, const float* stdv /*[C]*/, float* out,
                            int B, int C, int H, int W, int OH, int OW, int stages, void* stream);

/* F.max_pool2d(x,[2,2]) attention.py:25-26; idx = argmax position 0..3 inside the window
 * (first maximum in row-major window order, like ATen)                          */
int tg_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int BC, int H, int W, void* stream);
/* gx[window(o)[idx[o]]] = gy[o], zeros elsewhere (H,W = dims of gx) */
int tg_maxpool2_bwd(const float* gy, const uint8_t* idx, float* gx, int BC, int H, int W, void* stream);
/* y[o] = x[window(o)[idx[o]]] (transpose of _bwd; H,W = dims of x) */
int tg_maxpool2_gather(const float* x, const uint8_t* idx, float* y, int BC, int H, int W, void* stream);

/* ---------------------------------------------------------------- row ops
 * out[r] = alpha * sum_c x[r][c]  (torch.sum(feats,[2,3]) discriminator.py:143,165) */
int tg_row_sum(const float* x, float* out, float alpha, int rows, int cols, void* stream);
/* out[r][c] = alpha * v[r] */
int tg_row_bcast(const float* v, float* out, float alpha, int rows, int cols, void* stream);
/* x.repeat(reps,1) (iqn.py:93): out[q*rows + r][c] = alpha * x[r][c]            */
int tg_repeat_rows(const float* x, float* out, float alpha, int rows, int cols, int reps, void* stream);
/* out[r][c] = alpha * sum_q x[q*rows + r][c]  (p_target_tau.mean(0), discriminator.py:174-175) */
int tg_sum_reps(const float* x, float* out, float alpha, int rows, int cols, int reps, void* stream);
/* the same over `groups` independent row blocks back to back (x: groups x rows rows; the repeated side: groups x reps x rows): the
 * real and the fake batch of a discriminator step through the IQN head as one tensor, each half repeated / averaged on its own */
int tg_repeat_rows_groups(const float* x, float* out, float alpha, int rows, int cols, int reps, int groups, void* stream);
int tg_sum_reps_groups(const float* x, float* out, float alpha, int rows, int cols, int reps, int groups, void* stream);

/* ---------------------------------------------------------------- elementwise */
int tg_add(const float* a, const float* b, float* out, int64_t n, void* stream);
/* out = ((a + b) + c) + d, c / d nullable: the gradient fan-in of a tensor with several consumers (attention.py:22-35: x
 * feeds theta, phi, g and the residual) in one pass */
int tg_add4(const float* a, const float* b, const float* c /*nullable*/, const float* d /*nullable*/, float* out,
            int64_t n, void* stream);
int tg_mul(const float* a, const float* b, float* out, int64_t n, void* stream);
int tg_scale(const float* x, float alpha, float* out, int64_t n, void* stream);
/* out = (alpha * *s) * x, s a device scalar */
int tg_scale_dev(const float* s, float alpha, const float* x, float* out, int64_t n, void* stream);
/* out = *s * a + b  (gamma * o + x, attention.py:35) */
int tg_scale_add_dev(const float* s, const float* a, const float* b, float* out, int64_t n, void* stream);
/* *out = alpha * sum_i a[i]*b[i]  (deterministic two-stage; workspace tg_reduce_workspace(n)) */
size_t tg_reduce_workspace(int64_t n);
int tg_dot(const float* a, const float* b, float alpha, float* out, float* workspace, int64_t n,
           int accumulate, void* stream);
/* out = g * (x >= 0 ? 1 : slope)  (LeakyReLU fwd when g == x; leaky_relu_backward otherwise) */
int tg_lrelu_bwd(const float* g, const float* x, float slope, float* out, int64_t n, void* stream);
int tg_tanh_fwd(const float* x, float* y, int64_t n, void* stream);
/* out = g * (1 - y*y) */
int tg_tanh_bwd(const float* g, const float* y, float* out, int64_t n, void* stream);
/* nn.ELU / nn.SELU (trainers/cnn.py:42-44 --activation elu|selu): y = x > 0 ? scale*x : scale*alpha*(exp(x)-1) */
int tg_elu_fwd(const float* x, float alpha, float scale, float* y, int64_t n, void* stream);
/* order 1: out = g * f'(x) (elu_backward); order 2: out = g * f''(x) (elu_double_backward, the R1 real branch) */
int tg_elu_bwd(const float* g, const float* x, float alpha, float scale, int order, float* out, int64_t n,
               void* stream);
int tg_fill(float* x, float value, int64_t n, void* stream);
/* dst (B, Cd, HW) <- src (B, Cs, HW): channels [0, min(Cs, Cd)) copied, channels >= Cs set to `fill`.  Cd = Cs + 1, fill 1:
 * the all-ones channel that carries the from-RGB bias when discriminator.py:11-22 (1x1) is composed into the first 3x3
 * (blocks/discriminator.py forward_from_rgb; replaces torch.cat + ones); Cd < Cs: its transpose (drop the channel). */
int tg_copy_channels(const float* src, float* dst, int B, int Cs, int Cd, int HW, float fill, void* stream);

/* ---------------------------------------------------------------- softmax (attention.py:32) */
int tg_softmax_fwd(const float* s, float* y, int rows, int cols, void* stream);
/* gs = y * (gy - sum(gy*y)) */
int tg_softmax_bwd(const float* gy, const float* y, float* gs, int rows, int cols, void* stream);
/* adjoint of y in gs = softmax_bwd(gy, y) given v = adjoint of gs:
 * out = v*gy - v*sum(gy*y) - gy*sum(v*y)                                        */
int tg_softmax_dbwd(const float* v, const float* gy, const float* y, float* out, int rows, int cols, void* stream);

/* ---------------------------------------------------------------- fused attention core (attention.py:27-34)
 * o = g softmax_m(theta^T phi)^T without materialising the (N x M) map.
 * theta (B,D,N), phi (B,D,M), g (B,DV,M), o (B,DV,N), lse (B,N) = log sum_m exp(score).
 * Compiled for (D,DV) in {(1,4),(2,8),(4,16),(8,32),(16,64)} (C = 8..128); others: TG_EUNSUPPORTED
 * (the host then composes tg_gemm / tg_softmax_*).  _bwd is the first-order backward;
 * workspace: tg_attn_bwd_workspace() bytes.  Second order (the R1 penalty differentiates _bwd): the
 * host forms the (N x M) maps with tg_gemm and calls tg_attn_dbwd_rows for everything row-wise.       */
int tg_attn_supported(int D, int DV);
size_t tg_attn_bwd_workspace(int B, int D, int DV, int N, int M);
int tg_attn_fwd(const float* theta, const float* phi, const float* g, float* o, float* lse,
                int B, int D, int DV, int N, int M, void* stream);
int tg_attn_bwd(const float* go, const float* theta, const float* phi, const float* g, const float* o,
                const float* lse, float* dtheta, float* dphi, float* dg, float* workspace,
                int B, int D, int DV, int N, int M, void* stream);
/* Row-wise middle of the derivative of tg_attn_bwd.  With a, b, c the adjoints of (dtheta, dphi, dg), on entry
 *   s = theta^T phi,  gp = go^T g,  u = a^T phi + theta^T b,  v = go^T c      (rows = B*N, cols = M, lse as tg_attn_fwd)
 * and on return, in place (P = exp(s - lse); delta, eps, zeta = row sums of P gp, P u, P dP):
 *   s <- P,  gp <- gS = P (gp - delta),  u <- dgP = P (u - eps),  v <- dS = P (dP - zeta),  dP = v + u (gp - delta) - gp eps
 * from which  d theta = phi dS^T + b gS^T,  d phi = theta dS + a gS,  d go = g dgP^T + c P^T,  d g = go dgP.          */
int tg_attn_dbwd_rows(float* s, const float* lse, float* gp, float* u, float* v, int rows, int cols, void* stream);
/* The same derivative in one kernel, no (N x M) map in memory, for key counts a wave can keep a whole row of in
 * registers (M <= 256 at D + DV = 20; fewer for the wider heads): a (B,D,N), b (B,D,M), c (B,DV,M) are the adjoints of
 * tg_attn_bwd's (dtheta, dphi, dg); d_go (B,DV,N), d_theta (B,D,N), d_phi (B,D,M), d_g (B,DV,M) the adjoints of its
 * (go, theta, phi, g).  workspace: tg_attn_dbwd_workspace() bytes.  Otherwise TG_EUNSUPPORTED (compose as above). */
int tg_attn_dbwd_supported(int D, int DV, int M);
size_t tg_attn_dbwd_workspace(int B, int D, int DV, int N, int M);
int tg_attn_dbwd(const float* go, const float* theta, const float* phi, const float* g, const float* lse,
                 const float* a, const float* b, const float* c, float* d_go, float* d_theta, float* d_phi, float* d_g,
                 float* workspace, int B, int D, int DV, int N, int M, void* stream);

/* ---------------------------------------------------------------- IQN head (models/iqn.py)
 * out[i][j] = cos((taus[i] * pi) * range[j])   iqn.py:41-45 (fp32, this evaluation order) */
int tg_iqn_cos_embed(const float* taus, const float* range, float* out, int n, int dims, void* stream);
/* iqn.py:111-130: preds (Q*B), row = q*B + b; target (B); taus (Q*B).
 * *loss = sum_q,b |tau - 1[err<0]| * huber_k(err) / B ; dpreds = d loss / d preds       */
int tg_iqn_loss(const float* preds, const float* target, const float* taus, float k,
                float* loss, float* dpreds, float* workspace, int Q, int B, void* stream);
/* `groups` evaluations back to back (preds / taus rows g*Q*B + q*B + b, target rows g*B + b): *loss = the SUM of their losses
 * (trainers/iqn.py:118-120: loss_real + loss_fake), dpreds per evaluation as above */
int tg_iqn_loss_groups(const float* preds, const float* target, const float* taus, float k,
                       float* loss, float* dpreds, float* workspace, int Q, int B, int groups, void* stream);

/* ---------------------------------------------------------------- losses
 * nn.BCEWithLogitsLoss (mean) trainers/cnn.py:88,131,147; dlogits = (sigmoid(x)-t)/n    */
int tg_bce_logits(const float* logits, const float* targets, float* loss, float* dlogits,
                  float* workspace, int n, void* stream);
/* *out = alpha * sum x^2  (R1: grad.pow(2).view(B,-1).sum(1).mean(), losses.py:27-29)   */
int tg_sumsq(const float* x, float alpha, float* out, float* workspace, int64_t n, void* stream);

/* ---------------------------------------------------------------- spectral norm (north_star a15)
 * The reference never applies spectral norm (prep4web.py:24,33-51 only strips torch.nn.utils.spectral_norm
 * from a checkpoint); the hook is every block's conv_factory (discriminator.py:28,52, generator.py:34,117).
 * Semantics pinned against torch.nn.utils.spectral_norm:  n_iter x { v = normalize(W^T u, eps);
 * u = normalize(W v, eps) } in place, then *sigma = u^T W v (sigma nullable).  W: rows x cols row-major.    */
int tg_sn_power_iter(const float* W, float* u, float* v, float* sigma /*nullable*/,
                     int rows, int cols, int n_iter, float eps, void* stream);
/* out = 1 / x */
int tg_recip(const float* x, float* out, int64_t n, void* stream);

/* ---------------------------------------------------------------- optimiser / EMA
 * torch.optim.Adam (trainers/cnn.py:84-85), single flat tensor.
 * hyper (device, 6 floats): [lr/bias_correction1, sqrt(bias_correction2), beta1, beta2,
 * 1-beta1, 1-beta2] (host doubles rounded to fp32) so the captured graph is step-independent:
 *   m = lerp(m, g, 1-beta1); v = beta2*v + (1-beta2)*g*g;
 *   p -= hyper[0] * m / (sqrt(v)/hyper[1] + eps)                                        */
int tg_adam_step(float* p, const float* g, float* m, float* v, const float* hyper, float eps,
                 int64_t n, void* stream);
/* update_target_generator (trainers/cnn.py:158-165): t += (p - t) * lr */
int tg_ema(float* t, const float* p, float lr, int64_t n, void* stream);

/* ---------------------------------------------------------------- FID / Inception-score math
 * inception_utils.py:97-124 torch_cov, :129-144 sqrt_newton_schulz, :205-235 torch_calculate_frechet_distance,
 * :239-246 calculate_inception_score -- everything after the (network-less) pooled features / softmax outputs.   */
/* C (M x N) = alpha * op(A) B + diag * I on the fp32 matrix cores, row-major, 128 x 128 tiles staged by LDS-DMA.
 * transA: A is (K x M) and A^T is used (the covariance product X_c^T X_c).  The Newton-Schulz step
 * T = 0.5 (3 I - Z Y) is one call with alpha = -0.5, diag = 1.5.  Needs lda, ldb, K % 4 == 0 and 16-byte aligned
 * A, B (tg_gemm_big_supported); otherwise TG_EUNSUPPORTED (use tg_gemm).                                          */
int tg_gemm_big_supported(int M, int N, int K, int lda, int ldb, int transA);
int tg_gemm_big(const float* A, const float* Bm, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                int transA, float alpha, float diag, void* stream);
/* X[n][d] -= mean[d]  (torch_cov centres its argument in place, inception_utils.py:120) */
int tg_center_rows(float* X, const float* mean, int N, int D, void* stream);
/* out[0] = trace(A), A (D x D) with leading dimension ld */
int tg_trace(const float* A, float* out, int D, int ld, void* stream);
/* rows[n] = sum_c p[n][c] * (log p[n][c] - log mean[c])  (the KL term of calculate_inception_score) */
int tg_is_kl_rows(const float* p, const float* mean, float* rows, int N, int Cn, void* stream);

/* ---------------------------------------------------------------- input pipeline
 * ImageBytesDataset (image_bytes_dataset.py:12-49) + ToPILImage -> RandomCrop(size) -> ToTensor -> Normalize(.5, .5)
 * (trainers/trainer.py:69-78) for a whole batch: archive = device-resident uint8 (n_images, H, W, channels) exactly
 * as np.savez_compressed stores it, index[b] = image of batch row b (device int64), crop_y / crop_x = top-left corner
 * of the size x size crop per row (device int32, nullable = 0: image size == crop size, where RandomCrop draws nothing).
 * out (B, channels, size, size) fp32 = ((u8 / 255) - 0.5) / 0.5 in ToTensor's / Normalize's operation order.
 * The caller guarantees 0 <= index < n_images and crop + size <= H, W.                                            */
int tg_image_bytes_batch(const uint8_t* archive, const int64_t* index, const int* crop_y /*nullable*/,
                         const int* crop_x /*nullable*/, float* out, int B, int64_t n_images, int H, int W,
                         int channels, int size, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* TARTANGAN_AMD_H */
