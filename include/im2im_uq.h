/* im2im_uq.h -- C ABI of libim2im_uq.so, the MI355X (gfx950) kernel library behind the
 * im2im-uq hot path (quantile-regression UNet training + RCPS calibration).
 *
 * The reference (aangelopoulos/im2im-uq) has no native/FFI boundary: its device arithmetic is
 * whatever torch ops its Python reaches.  Each entry point below therefore cites the reference
 * Python call site (file:line under the reference root) whose torch/scipy arithmetic it replaces.
 *
 * Conventions
 *   - plain C: raw device pointers, sizes, a hipStream_t passed as void*; no torch types.
 *   - return 0 on success, <0 on error (IM2IM_ERR_*); im2im_last_error() gives the message
 *     (thread-local).  Entry points never allocate, never synchronise, and launch on `stream`.
 *   - device buffers are caller-owned; workspaces have explicit size queries.
 *   - all device tensors are dense, row-major in the index order written in the comment.
 */
#ifndef IM2IM_UQ_H
#define IM2IM_UQ_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define IM2IM_ABI_VERSION 3
#define IM2IM_BN_COUNTERS 64        /* ints in a `counters` array of the BatchNorm entry points ([ABI 3], see im2im_bn_finalize) */
#define IM2IM_OK 0
#define IM2IM_ERR_INVALID (-1)     /* bad argument / unsupported shape */
#define IM2IM_ERR_HIP (-2)         /* a HIP runtime call or launch failed */
#define IM2IM_ERR_UNSUPPORTED (-3)

typedef void* im2im_stream_t;      /* hipStream_t */

/* element type of activation tensors ("compute dtype").  The reference runs fp32 everywhere
 * (core/scripts/train.py:149); IM2IM_F32 is the tight-parity mode (exact-fp32 MFMA), IM2IM_BF16 the
 * throughput mode (bf16 storage + bf16 MFMA inputs, fp32 accumulate). */
#define IM2IM_F32 0
#define IM2IM_BF16 1
/* operand type of the fp8 forward convolution (im2im_conv_fwd_fp8): OCP e4m3fn on the block-scaled MFMA; activations stay
 * IM2IM_BF16 in memory.  Not a storage dtype: the entry points that take `dtype` accept IM2IM_F32 / IM2IM_BF16 only. */
#define IM2IM_FP8 2

int im2im_abi_version(void);
const char* im2im_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * RCPS calibration scoring (SURVEY K11+K12).  Replaces, for ALL lambdas in one pass over HBM,
 *   quantile_regression_nested_sets_from_output   core/models/finallayers/quantile_layer.py:39-42
 *   ModelWithUncertainty.nested_sets_from_output  core/models/add_uncertainty.py:33-38
 *   fraction_missed_loss                          core/calibration/calibrate_model.py:76-80
 * as looped by get_rcps_losses_from_outputs (calibrate_model.py:21-29) inside the lambda scan
 * (calibrate_model.py:134-136) and by get_loss_table (core/scripts/eval.py:116-125).
 *
 *   out3    [N][K][P] fp32   model output, P = C*H*W; its planes and how they become the nested set are given by `form`:
 *                            IM2IM_SETS_QUANTILE (0)  K = 3 (lower, prediction, upper): quantile_layer.py:34-44 and
 *                                                     quantile_l1_layer.py:34-44 (clamp to pred -+ 1e-6, then
 *                                                     pred -+ lam * width)
 *                            IM2IM_SETS_SCALE (1)     K = 2 (prediction, magnitude): pred -+ lam * magnitude,
 *                                                     residual_magnitude(_l1)_layer.py:27-36
 *                            IM2IM_SETS_SQRT (2)      K = 2 (mean, variance): mean -+ lam * sqrt(variance),
 *                                                     gaussian_layer.py:25-34
 *                            IM2IM_SETS_SOFTMAX (3)   K = 3 (lower quantile, prediction, upper quantile) as written by
 *                                                     im2im_softmax_sets_summary: pred - lam*relu(pred - lq),
 *                                                     pred + lam*relu(uq - pred), softmax_layer.py:50-51
 *                            all followed by ModelWithUncertainty's +-1e-6 floor (add_uncertainty.py:35-36)
 *   label   [N][P]    fp32
 *   lam     [L]       fp32   device; ascending grid of the lambdas the edges are evaluated at
 *                            (the caller passes lambdas - dlambda for calibrate_model, Q1)
 *   hist_ws           int32  workspace of im2im_rcps_workspace_bytes(N, P, L) bytes (no initialisation needed)
 *   table   [N][L]    fp32   table[n][j] = (#pixels of image n missed at lam[j]) / P, bit-identical
 *                            to the reference's fp32 mean of 0/1 indicators
 *   counts  [N][L]    int32  optional (may be NULL): the integer miss counts
 * Edge arithmetic is fp32 with separate multiply and add (no FMA), as on the reference CPU path.
 */
int64_t im2im_rcps_workspace_bytes(int64_t N, int64_t P, int32_t L);
#define IM2IM_SETS_QUANTILE 0
#define IM2IM_SETS_SCALE 1
#define IM2IM_SETS_SQRT 2
#define IM2IM_SETS_SOFTMAX 3
int im2im_rcps_loss_table(const float* out3, const float* label, int64_t N, int64_t P,
                          const float* lam, int32_t L, int32_t form, int32_t* hist_ws, float* table,
                          int32_t* counts, im2im_stream_t stream);

/* Spatial miscoverage counts at one lambda; replaces the accumulation at
 * core/calibration/calibrate_model.py:47,55.
 *   map [C][HW] int32, zeroed by the call: map[c][i] = #images n with label > upper or label < lower
 */
int im2im_rcps_miscoverage(const float* out3, const float* label, int64_t N, int32_t C, int64_t HW,
                           float lam, int32_t form, int32_t* map, im2im_stream_t stream);

/* Elementwise nested sets at one lambda (a10): same two call sites as above, materialising
 * (lower_edge, upper_edge) [N][P] for the given `form`.  floor != 0: with ModelWithUncertainty's +-1e-6 floor
 * (add_uncertainty.py:35-36); floor == 0: the final layer's own *_nested_sets_from_output.  If clamp_inplace != 0
 * (quantile form), out3's lower/upper planes are overwritten with the clamped values like the reference does
 * (quantile_layer.py:39-40, Q5). */
int im2im_nested_sets(float* out3, int64_t N, int64_t P, float lam, int32_t form, float* lower_edge,
                      float* upper_edge, int32_t clamp_inplace, int32_t floor, im2im_stream_t stream);

/* Per-image miss fraction for already-materialised edges; replaces fraction_missed_loss
 * core/calibration/calibrate_model.py:76-80.  lower/upper/label [N][P] fp32 -> loss [N] fp32
 * = fp32(#pixels with lower > label or upper < label) / fp32(P). */
int im2im_fraction_missed(const float* lower_edge, const float* upper_edge, const float* label,
                          int64_t N, int64_t P, float* loss, im2im_stream_t stream);

/* Hoeffding-Bentkus upper confidence bound, host float64; replaces HB_mu_plus
 * core/calibration/bounds.py:17-29 (scipy binom.cdf + brentq), including its "return 1.0 on any
 * solver exception" path (muhat == 0 -> NaN, Q3). */
double im2im_hb_mu_plus(double muhat, int64_t n, double delta, int32_t maxiters);

/* `count` bounds at once, spread over host threads; replaces the per-lambda list comprehension of
 * evaluate_from_loss_table, core/calibration/calibrate_model.py:69 (driven 100 x num_lambdas times by
 * experiments/fastmri_test/plot.py:126-139).  muhat are the float32 empirical risks as the reference holds them:
 * floor(n * muhat) is taken on the fp32 product like np.floor(n * muhat) on a 0-dim fp32 tensor.  out [count] float64. */
int im2im_hb_mu_plus_batch(const float* muhat, int64_t count, int64_t n, double delta, int32_t maxiters, double* out);

/* The descending lambda scan that turns a loss table into lambda-hat (SURVEY K14, section 8(b) `rcps_scan`); replaces the
 * loop of calibrate_model, core/calibration/calibrate_model.py:130-144, with its quirks: column j of the table holds the
 * losses at lambdas[j] - dlambda (Q1), columns left of the stop are never visited (Q2), Rhat == 0 gives RhatPlus = 1.0
 * (Q3), the stop test is `Rhat >= alpha or RhatPlus > alpha` and, when nothing stops the scan, lhat =
 * lambdas[L-1] + dlambda - 1e-9 evaluated in fp32 (Q4).  Host function (the table is on the host; <= L scalar solves).
 *   table     fp32, element (image i, lambda j) at table[i*row_stride + j*col_stride]
 *   lambdas   [L] fp32 ascending grid (torch.linspace(minimum_lambda, maximum_lambda, num_lambdas))
 *   outputs   *stop_index: first visited column; *stopped: 0 when the scan ran off the grid; *lhat; *visited: columns
 *             evaluated; rhat [L] fp32 / rhat_plus [L] float64 (either may be NULL): the visited columns' values.
 * Rhat is the correctly rounded fp32 mean of a column (float64 accumulation) unless rhat_in (NULL, or [L] with NaN = "not
 * given") supplies the caller's own mean for that column (torch's `losses.mean()`, :135, whose last bits depend on the host). */
int im2im_rcps_scan(const float* table, int64_t N, int32_t L, int64_t row_stride, int64_t col_stride,
                    const float* lambdas, double alpha, double delta, int32_t maxiters, int32_t* stop_index,
                    int32_t* stopped, float* lhat, int32_t* visited, float* rhat, double* rhat_plus, const float* rhat_in);

/* Run-time switch for within-process A/B measurements of kernel variants (tools/, bench.py); no reference counterpart.
 * "wgrad_co128" / "wgrad_tile16": 0 switches the 128-output-channel / 256-pixel-tile forms of the weight gradient off;
 * "wgrad_roll": 0 = the bf16 and fp8 3x3 weight gradients on conv_wgrad_pipe_kernel / conv_wgrad_fp8_kernel instead of the roll
 *   kernels (default 1; same bits either way); "wgrad_fp8_co128": 0 = the fp8 weight gradient's 64-output-channel form everywhere;
 * "conv_splitk": 0 = im2im_conv_fwd_split_ws never splits, n = it aims at n * 256 workgroups (default 3);
 * "bn_fused_small": n = BatchNorm statistics / backward sums of <= n partial rows in one small-grid launch (0 = never, 1 = the default
 *   of 256 rows); "bn_onelaunch": [ABI 3] 1 = above that row count ONE launch when `counters` are given (default 0: the device-scope fence the ticket
 *   needs flushes the XCD's L2 in every block and the step is 3-10 % slower with it, profiles/r06_ab_experiments.txt);
 * "pool_bwd_blocks": workgroups of im2im_bn_relu_pool_bwd, 1..6144 (anything else = the default 2048);
 * "conv_roll": only in libraries built with IM2IM_BUILD_EXPERIMENTAL=1 (csrc/conv_roll.hip); elsewhere any value but 0 is an error.
 * ("wgrad_wgs" became the target_wgs argument of im2im_conv_wgrad_split / im2im_conv_wgrad_fp8 in ABI 3.) */
int im2im_set_option(const char* key, int32_t value);

/* ---------------------------------------------------------------------------------------------
 * Convolution by MFMA implicit GEMM.  Activations are NHWC, element type `dtype`.
 * Replaces nn.Conv2d 3x3 pad 1 (core/models/trunks/unet_parts.py:16,19) and 1x1 (:90) and their
 * autograd backward.
 *
 * im2im_pack_conv_weight: w [Co][Ci][taps] fp32 (torch parameter layout, taps = kh*3+kw) ->
 *   wf [Co][taps][Ci]            operand of the forward conv
 *   wd [Ci][taps reversed][Co]   operand of dgrad (= forward conv of dz with flipped taps); may be NULL
 * The packed buffers are opaque operands of the conv entry points (same element counts as above).  For bf16 3x3 weights
 * with Co % 32 == 0 and Ci % 32 == 0 the storage order is fragment-major -- [row block of 32][tap][32-channel chunk]
 * [16-channel k-step][lane 64][8] -- so that a wave reads one MFMA operand fragment with a single coalesced 1 KiB load
 * (csrc/conv_common.h wfrag_index); everything else is stored in the logical order.
 */
int im2im_pack_conv_weight(const float* w, int32_t Co, int32_t Ci, int32_t taps, int32_t dtype,
                           void* wf, void* wd, im2im_stream_t stream);

/* The same packing for n tensors in one launch (a training step re-packs every conv weight after the optimizer changed
 * them; 18 separate launches were 6 % of a small-batch step's launches).  w[i] [Co[i]][Ci[i]][taps[i]] fp32 ->
 * wf[i] [Co][taps][Ci], wd[i] [Ci][taps reversed][Co] (wd[i] may be NULL), all in `dtype`.  Host arrays of n entries. */
int im2im_pack_conv_weights_multi(int32_t n_tensors, const float* const* w, const int32_t* Co, const int32_t* Ci,
                                  const int32_t* taps, int32_t dtype, void* const* wf, void* const* wd,
                                  im2im_stream_t stream);

/* y[b,h,w,co] = epi( sum_{tap,ci} x[b,h+kh-1,w+kw-1,ci] * wf[co][tap][ci] + bias[co] )
 *   x [B][H][W][Ci], y [B][H][W][Co] (dtype), Ci % 32 == 0, Co % 32 == 0, taps in {9,1}
 *   bias fp32 [Co] or NULL;  scale/shift fp32 [Co] or both NULL: v = v*scale + shift (folded eval-mode
 *   BatchNorm, unet_parts.py:17,20);  relu != 0: v = max(v,0) (unet_parts.py:18,21)
 *   stats (may be NULL): [rows][3][Co] fp32 per-tile (mean, M2 = sum of squared deviations from that mean, count)
 *   of the STORED values over valid pixels, rows = im2im_conv_stats_rows(B,H,W,Co): train-mode BatchNorm
 *   statistics without re-reading y, in the cancellation-free form im2im_bn_finalize merges pairwise in fp64
 *   (torch's CPU BatchNorm accumulates in double).  dgrad uses the same entry point with x = dz and wf = wd.
 *   center (may be NULL): fp32 [Co] subtracted from the stored output.  The bf16 train path passes the layer's
 *   running_mean so the pre-BatchNorm tensor is stored roughly zero-mean (BatchNorm is shift invariant; bf16 rounding
 *   is then relative to the spread of a channel instead of its offset); im2im_bn_finalize(centered = 1) adds it back
 *   into the running-mean update.
 *   in_scale_shift (may be NULL): fp32 [2][Ci].  When given, x holds the PRE-BatchNorm output z of the producing
 *   layer and every consumer applies a = max(z*scale + shift, 0) while staging its operand ("lazy" BatchNorm+ReLU,
 *   unet_parts.py:17-18): the normalised activation is never written to HBM.  Same option on im2im_conv_wgrad
 *   (x_scale_shift), im2im_maxpool2_* (in_scale_shift) and im2im_upsample2x_concat_fwd (deep/skip_scale_shift). */
int64_t im2im_conv_stats_rows(int32_t B, int32_t H, int32_t W, int32_t Co);
int im2im_conv_fwd(const void* x, const float* in_scale_shift, const void* wf, const float* bias,
                   const float* center, const float* scale, const float* shift, void* y, float* stats, int32_t B,
                   int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t relu,
                   int32_t dtype, im2im_stream_t stream);

/* Channel-split operands: the Up block's torch.cat([x2, x1], dim=1) (unet_parts.py:68) is never materialised.
 *   x_hi (may be NULL): the convolution's input channels [Ci_lo, Ci) are read from x_hi and [0, Ci_lo) from x; both
 *     tensors are [B][H][W][Ci_lo] (Ci == 2*Ci_lo: skip and upsampled halves of the UNet have equal width), each with
 *     its own optional lazy coefficients (in_scale_shift [2][Ci_lo], in_scale_shift_hi [2][Ci_lo]).
 *   y_hi (may be NULL): output channels [Co_lo, Co) are written to y_hi [B][H][W][Co-Co_lo], [0, Co_lo) to
 *     y [B][H][W][Co_lo] (Co_lo and Co-Co_lo multiples of 64): the data-gradient of such a convolution lands
 *     directly in d(skip) and d(upsampled) without a concatenated gradient tensor.  stats/scale/shift must be NULL.
 *   With x_hi == y_hi == NULL this is im2im_conv_fwd. */
int im2im_conv_fwd_split(const void* x, const float* in_scale_shift, const void* x_hi,
                         const float* in_scale_shift_hi, int32_t Ci_lo, const void* wf, const float* bias,
                         const float* center, const float* scale, const float* shift, void* y, void* y_hi,
                         int32_t Co_lo, float* stats, int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                         int32_t taps, int32_t relu, int32_t dtype, im2im_stream_t stream);

/* The same convolution with a caller-provided workspace, which lets UNDER-FILLED launches split their reduction (split-K):
 * a strong-scaled data-parallel job leaves ~10 images per GPU (core/scripts/train.py:112-115 with
 * experiments/fastmri_test/config.yml:44-45, batch 78 over 8 devices), and at the 40x40 / 20x20 levels the convolution is
 * then 180-250 workgroups for 256 CUs, each reducing K = 9*Ci = 4,608-9,216 alone.  With a workspace of at least
 * im2im_conv_splitk_workspace_bytes(...) bytes (0 = this shape is never split) such a launch is cut into 2-8 input-channel
 * ranges whose fp32 partial sums go to the workspace and are added in a fixed order by a second kernel that also applies
 * the bias, rounds, stores y (/ y_hi) and takes the BatchNorm statistics (same `stats` rows as the one-kernel path;
 * deterministic; scale/shift == NULL only).  workspace == NULL: exactly im2im_conv_fwd_split. */
int64_t im2im_conv_splitk_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t taps);
int im2im_conv_fwd_split_ws(const void* x, const float* in_scale_shift, const void* x_hi,
                            const float* in_scale_shift_hi, int32_t Ci_lo, const void* wf, const float* bias,
                            const float* center, const float* scale, const float* shift, void* y, void* y_hi,
                            int32_t Co_lo, float* stats, int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                            int32_t taps, int32_t relu, int32_t dtype, void* workspace, int64_t workspace_bytes,
                            im2im_stream_t stream);

/* [r4] Eval-mode conv3x3 + folded BatchNorm + ReLU of a block whose output ALSO feeds MaxPool2d(2) (DoubleConv followed by
 * Down.maxpool_conv[0], unet_parts.py:33-36, in model.eval()): y as im2im_conv_fwd_split with scale / shift / relu = 1, and
 * pool_y [B][H/2][W/2][Co] = MaxPool2d(2)(y), taken from the epilogue's LDS tile -- same values as im2im_maxpool2_fwd on y, without
 * re-reading y.  H and W even.  x_hi / Ci_lo as in im2im_conv_fwd_split (NULL / Ci for a single input). */
int im2im_conv_fwd_eval_pool(const void* x, const void* x_hi, int32_t Ci_lo, const void* wf, const float* scale,
                             const float* shift, void* y, void* pool_y, int32_t B, int32_t H, int32_t W, int32_t Ci,
                             int32_t Co, int32_t dtype, im2im_stream_t stream);

/* [r4] Eval-mode LAST block of the trunk: conv3x3 (Ci -> 64) + folded BatchNorm + ReLU whose result is consumed only by OutConv's
 * 1x1 convolution (unet.py:45-46, unet_parts.py:87-94) -- f_out [B][H][W][C1] = conv1x1(relu(affine(conv3x3(x))), w1) + b1, the
 * 1x1 evaluated on the epilogue's LDS tile in the order im2im_conv_fwd(taps = 1) takes it (same bits as the two kernels); the
 * 64-channel tensor is neither written nor read back.  w1 [C1][64] = im2im_pack_conv_weight's wf of the 1x1 weights, b1 [C1]|NULL,
 * C1 == 32 (the reference trunk's n_channels_middle); x_hi / Ci_lo as in im2im_conv_fwd_split. */
int im2im_conv_fwd_eval_tail(const void* x, const void* x_hi, int32_t Ci_lo, const void* wf, const float* scale,
                             const float* shift, const void* w1, const float* b1, void* f_out, int32_t B, int32_t H,
                             int32_t W, int32_t Ci, int32_t C1, int32_t dtype, im2im_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * fp8 forward convolution (BASELINE configs[4] "fp8 MFMA conv path"): 3x3 pad-1 conv whose operands are OCP e4m3 on
 * v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64, ~2x the bf16 MFMA rate), fp32 accumulate, bf16 in/out.  Same contract as
 * im2im_conv_fwd_split with dtype IM2IM_BF16 and taps = 9 (the nn.Conv2d forward of unet_parts.py:16,19), except:
 *   wq / wscale come from im2im_pack_conv_weight_fp8: wq [Co][9][Ci] e4m3 bytes = w / wscale[co], wscale[co] the power
 *     of two that maps the channel's max |w| into (128, 256]; an opaque operand -- for Co % 32 == 0 and Ci % 64 == 0 the
 *     storage order is fragment-major (csrc/conv_fp8.hip wfrag8_index), read straight from L2 into registers by the kernel;
 *   x (bf16, optionally the producer's pre-BatchNorm z with lazy coefficients) is scaled by 2^4, clamped to +-448 and
 *     converted to e4m3 while it is staged into LDS (the 2^4 is undone by the instruction's block scale);
 *   Ci % 64 == 0 (Ci_lo % 64 == 0 when split), Co % 64 == 0; no split output, no `center`.
 * stats rows: im2im_conv_fp8_stats_rows(B, H, W). */
int64_t im2im_conv_fp8_stats_rows(int32_t B, int32_t H, int32_t W);
int im2im_pack_conv_weight_fp8(const float* w, int32_t Co, int32_t Ci, int32_t taps, void* wq, float* wscale,
                               im2im_stream_t stream);
int im2im_conv_fwd_fp8(const void* x, const float* in_scale_shift, const void* x_hi, const float* in_scale_shift_hi,
                       int32_t Ci_lo, const void* wq, const float* wscale, const float* bias, const float* scale,
                       const float* shift, void* y, float* stats, int32_t B, int32_t H, int32_t W, int32_t Ci,
                       int32_t Co, int32_t relu, im2im_stream_t stream);

/* fp8 data-gradient of the same convolution (the autograd backward of unet_parts.py:16,19 with respect to its input):
 * dx[b,h,w,cx] = sum_{tap,cz} dz[b,h+..,w+..,cz] * w[cz][cx][flipped tap], on the same kernel with the gradient operand in
 * OCP e5m2 (range over precision) under a per-tensor power-of-two scale, weights e4m3 with one scale per cx.
 *   wq_d / wscale_d from im2im_pack_conv_weight_fp8_dgrad(w [Co][Ci][taps] fp32): wq_d [Ci][taps reversed][Co], wscale_d [Ci];
 *   dz [B][H][W][Cz] bf16 (Cz = the layer's Co), dx [B][H][W][Cx] bf16 (Cx = the layer's Ci), or split into
 *     dx [..][Cx_lo] and dx_hi [..][Cx - Cx_lo] (both multiples of 64) like im2im_conv_fwd_split's y / y_hi;
 *   delayed scaling: *amax_prev = max |dz| of this tensor at the previous step (the caller seeds it once); the kernel
 *     scales dz so that amax_prev lands in [2^13, 2^14) of e5m2's 57344, accumulates this launch's max |dz| into
 *     *amax_now (atomic max; must be 0 on entry) and zeroes *amax_next for the launch after -- a caller rotates three
 *     floats.  amax_now / amax_next may be NULL (fixed scale from amax_prev).
 * Cz % 64 == 0, Cx % 64 == 0.  The weight gradient stays bf16 (im2im_conv_wgrad). */
int im2im_pack_conv_weight_fp8_dgrad(const float* w, int32_t Co, int32_t Ci, int32_t taps, void* wq_d, float* wscale_d,
                                     im2im_stream_t stream);
/* [r4] The fp8 operands of many 3x3 conv weights in one launch (per kind): n_tensors host arrays of device pointers / sizes as for
 * im2im_pack_conv_weights_multi; dgrad == 0: wq [Co][9][Ci] + wscale [Co] of im2im_pack_conv_weight_fp8, dgrad != 0: wq_d
 * [Ci][9 reversed][Co] + wscale_d [Ci] of im2im_pack_conv_weight_fp8_dgrad (same bits as the single-tensor entries). */
int im2im_pack_conv_weights_fp8_multi(int32_t n_tensors, const float* const* w, const int32_t* Co, const int32_t* Ci,
                                      void* const* wq, float* const* wscale, int32_t dgrad, im2im_stream_t stream);
int im2im_conv_dgrad_fp8(const void* dz, const void* wq_d, const float* wscale_d, void* dx, void* dx_hi, int32_t Cx_lo,
                         const float* amax_prev, float* amax_now, float* amax_next, int32_t B, int32_t H, int32_t W,
                         int32_t Cz, int32_t Cx, im2im_stream_t stream);

/* Data-gradient of a convolution whose input was a lazy BatchNorm+ReLU activation (unet_parts.py:16-21 chained): dx is
 * the gradient da of that activation, and the epilogue that writes it also reads the producer's pre-BN output bn_z
 * [B][H][W][Co] and accumulates the two BatchNorm-backward sums per channel, sum(g) and sum(g*xhat) with
 * g = da*[z*scale+shift > 0], xhat = (z-mean)*invstd, into bn_partial [im2im_conv_stats_rows(B,H,W,Co)][2][Co]
 * (one row per tile, no atomics).  im2im_bn_relu_bwd_from_partial then finishes the BatchNorm+ReLU backward without the
 * separate reduction pass over da and z.  dz is the incoming gradient [B][H][W][Ci], wd the data-gradient operand from
 * im2im_pack_conv_weight. */
int im2im_conv_dgrad_bn(const void* dz, const void* wd, void* dx, const void* bn_z, const float* bn_scale_shift,
                        const float* bn_mean_invstd, float* bn_partial, int32_t B, int32_t H, int32_t W, int32_t Ci,
                        int32_t Co, int32_t taps, int32_t dtype, im2im_stream_t stream);

/* dw[co][ci][tap] (fp32, torch layout) = sum_{b,h,w} dz[b,h,w,co] * x[b,h+kh-1,w+kw-1,ci]
 *   Ci % 32 == 0, Co % 32 == 0.  workspace: im2im_conv_wgrad_workspace_bytes(...) bytes (split-K slabs,
 *   reduced deterministically). */
int64_t im2im_conv_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t taps);
int im2im_conv_wgrad(const void* x, const float* x_scale_shift, const void* dz, float* dw, void* workspace,
                     int64_t workspace_bytes, int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                     int32_t taps, int32_t dtype, im2im_stream_t stream);
/* weight gradient of a convolution with channel-split input (see im2im_conv_fwd_split); Ci_lo % 64 == 0.
 * [ABI 3] target_wgs: workgroups a bf16 3x3 launch aims at (split-K slabs = target / channel blocks; 0 = 256, one per CU).  A
 *   launch that shares the chip with another stream's kernels is faster for the STEP at half width (the Python host passes 128
 *   from its weight-gradient stream): it was a process-global option ("wgrad_wgs") until ABI 2, which two threads could race on.
 *   The split count sets the order of the fp32 partial sums: results are bit-identical for one target, not across targets.
 * [ABI 3] dw == NULL with nsplit != NULL: the split-K slabs are left in `workspace` ([*nsplit][Co][taps][Ci] fp32, *nsplit a host
 *   int written before the call returns) for im2im_wgrad_reduce_multi, which finishes MANY layers' gradients in one launch. */
int im2im_conv_wgrad_split(const void* x, const float* x_scale_shift, const void* x_hi,
                           const float* x_scale_shift_hi, int32_t Ci_lo, const void* dz, float* dw, void* workspace,
                           int64_t workspace_bytes, int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                           int32_t taps, int32_t dtype, int32_t target_wgs, int32_t* nsplit, im2im_stream_t stream);
/* [ABI 3] dw_i [Co_i][Ci_i][taps_i] = sum of the nsplit_i slabs at slabs_i (what im2im_conv_wgrad_split / _fp8 left there), for
 * n_tensors layers in ONE launch (<= 32 per launch, more are chunked): the same per-output summation order as the single-layer
 * reduction, so the same bits.  Host arrays of device pointers / sizes, as im2im_pack_conv_weights_multi. */
int im2im_wgrad_reduce_multi(int32_t n_tensors, const float* const* slabs, const int32_t* nsplit, const int32_t* Co,
                             const int32_t* Ci, const int32_t* taps, float* const* dw, im2im_stream_t stream);

/* [r4] fp8 weight gradient (BASELINE configs[4] "fp8 MFMA conv path"; the dW half of autograd's backward of nn.Conv2d 3x3,
 * unet_parts.py:16,19 under loss.backward(), train.py:159): dz as OCP e5m2 under the tensor's delayed power-of-two scale
 * (amax_prev: max |dz| of the previous step, the slot im2im_conv_dgrad_fp8 maintains), the layer input as e4m3 x 2^4 -- the
 * operand the fp8 forward staged, lazy BatchNorm+ReLU included -- on v_mfma_scale_f32_32x32x64_f8f6f4 with K = 64 pixels,
 * fp32 accumulation, deterministic split-K reduction into dw [Co][Ci][3][3] fp32.  x, dz bf16; Ci % 64 == 0, Co % 64 == 0;
 * workspace as for im2im_conv_wgrad (im2im_conv_wgrad_workspace_bytes with taps = 9). */
int im2im_conv_wgrad_fp8(const void* x, const float* x_scale_shift, const void* x_hi, const float* x_scale_shift_hi,
                         int32_t Ci_lo, const void* dz, const float* amax_prev, float* dw, void* workspace,
                         int64_t workspace_bytes, int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                         int32_t target_wgs, int32_t* nsplit, im2im_stream_t stream);


/* Shared scratch for the deterministic two-stage "sum over pixels" reductions: bytes needed to reduce
 * K columns (used by bn_finalize; other entry points have their own *_workspace_bytes). */
int64_t im2im_reduce_workspace_bytes(int64_t K);

/* ---------------------------------------------------------------------------------------------
 * BatchNorm2d + ReLU: nn.BatchNorm2d / nn.ReLU at core/models/trunks/unet_parts.py:17-18,20-21
 * (torch defaults eps = 1e-5, momentum = 0.1), tensors [M = B*H*W][C] in `dtype`.
 *
 * im2im_bn_finalize: per-tile partial moments from the conv epilogue (partial [R][3][C]: mean, M2, count) -> batch
 *   mean and biased variance (pairwise merge in fp64, never E[z^2]-E[z]^2); writes mean_invstd [2][C],
 *   scale_shift [2][C] (scale = gamma*invstd, shift = beta - mean*scale) and, if non-NULL, the running statistics
 *   (unbiased variance, as torch).  ws: im2im_reduce_workspace_bytes(3*C) bytes.  num_batches_tracked (device int64
 *   scalar or NULL) is incremented by one: nn.BatchNorm2d's bookkeeping in the same launch.
 *   [ABI 3] counters (device, IM2IM_BN_COUNTERS zero-initialised int32, or NULL): with counters the two reduction stages of this
 *   entry point -- and of the backward sums inside im2im_bn_relu_bwd / _phase 2 / _from_partial / im2im_bn_relu_pool_bwd -- run as
 *   ONE launch whatever the number of partial rows: the block that finishes last (a ticket from a device-scope atomic) merges the
 *   split rows in the two-launch form's order, so the results are the same bits, and leaves the counters zero again.  The caller
 *   owns the counters, zeroes them once, and hands one array to ONE stream at a time.  NULL = the two-launch form.  The one-launch
 *   form is additionally gated by im2im_set_option("bn_onelaunch", 1) -- off by default, it measured slower (see there).
 * im2im_bn_fold_eval: eval-mode fold into the conv epilogue: scale = gamma/sqrt(rv+eps),
 *   shift = beta + (conv_bias - rm)*scale.
 * im2im_bn_relu_apply: a = max(z*scale + shift, 0).
 * im2im_bn_relu_bwd: given da, the saved pre-BN z and the forward's scale_shift / mean_invstd:
 *   dgamma = sum g*xhat, dbeta = sum g, dz = scale*(g - dbeta/M - xhat*dgamma/M), g = da*[a > 0].
 */
int im2im_bn_finalize(const float* partial, int64_t R, int32_t C, int64_t count, const float* gamma,
                      const float* beta, float* running_mean, float* running_var, float momentum,
                      float eps, int32_t centered, float* mean_invstd, float* scale_shift, void* ws,
                      int32_t* counters, int64_t* num_batches_tracked, im2im_stream_t stream);
int im2im_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, const float* conv_bias, float eps, int32_t C,
                       float* scale_shift, im2im_stream_t stream);
int im2im_bn_relu_apply(const void* z, const float* scale_shift, void* a, int64_t M, int32_t C,
                        int32_t dtype, im2im_stream_t stream);
int64_t im2im_bn_bwd_workspace_bytes(int64_t M, int32_t C);
int im2im_bn_relu_bwd(const void* da, const void* z, const float* scale_shift, const float* mean_invstd,
                      void* dz, float* dgamma, float* dbeta, int64_t M, int32_t C, int32_t dtype,
                      void* ws, int64_t ws_bytes, int32_t* counters, im2im_stream_t stream);
/* im2im_bn_relu_bwd cut into phases over row ranges, for a caller that pipelines it against the data-gradient kernels
 * working on the other half of the batch (nn_ops.BnReluLazy; no reference counterpart -- torch's batch_norm backward is
 * one call).  phase 1: partial sums of rows [row0,row1), row0 a multiple of im2im_bn_bwd_rows_per_block(M);
 * phase 2: dgamma, dbeta and the coefficients from all partial rows; phase 4: dz rows [row0,row1).  Same ws, same block
 * decomposition and therefore bit-identical results to im2im_bn_relu_bwd. */
int64_t im2im_bn_bwd_rows_per_block(int64_t M);
int im2im_bn_relu_bwd_phase(const void* da, const void* z, const float* scale_shift, const float* mean_invstd,
                            void* dz, float* dgamma, float* dbeta, int64_t M, int32_t C, int32_t dtype,
                            void* ws, int64_t ws_bytes, int32_t phase, int64_t row0, int64_t row1,
                            int32_t* counters, im2im_stream_t stream);
/* im2im_bn_relu_bwd with the reduction already done by im2im_conv_dgrad_bn: partial [R][2][C].
 * ws: im2im_reduce_workspace_bytes(2*C) + 2*C*4 bytes. */
int im2im_bn_relu_bwd_from_partial(const void* da, const void* z, const float* scale_shift, const float* mean_invstd,
                                   const float* partial, int64_t R, void* dz, float* dgamma, float* dbeta, int64_t M,
                                   int32_t C, int32_t dtype, void* ws, int64_t ws_bytes, int32_t* counters,
                                   im2im_stream_t stream);

/* BatchNorm+ReLU backward of a skip-connection layer fused with the backward of the MaxPool2d(2) that consumes the
 * same activation (unet.py:35-38; unet_parts.py:34 / 17-18,20-21): the activation's gradient
 *   g = da (gradient from the Up block, [B][H][W][C], may be NULL) + scatter_to_window_argmax(dpool [B][H/2][W/2][C])
 * is formed on the fly (first maximum wins, as torch; same storage-type rounding as the separate max-pool backward
 * + add would produce) and never written to HBM.  z [B][H][W][C] pre-BN, dz out; C/(16 B / elem) a power of two.
 * Outputs as im2im_bn_relu_bwd. */
int64_t im2im_bn_relu_pool_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C);
int im2im_bn_relu_pool_bwd(const void* da, const void* dpool, const void* z, const float* scale_shift,
                           const float* mean_invstd, void* dz, float* dgamma, float* dbeta, int32_t B,
                           int32_t H, int32_t W, int32_t C, int32_t dtype, void* ws, int64_t ws_bytes,
                           int32_t* counters, im2im_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * GroupNorm(+ReLU) -- named by BASELINE.json:north_star; NOT in the reference (its DoubleConv uses BatchNorm2d,
 * core/models/trunks/unet_parts.py:17,20).  Selectable extra behind DoubleConv(norm="group"), never the
 * default; oracle = torch.nn.GroupNorm on the CPU.  Statistics are per image and channel group (biased variance).
 *
 *   im2im_conv_fwd_per_image: im2im_conv_fwd with ONE image per tile (16x16 pixel tiles at every extent), so the rows
 *     of `stats` [B * im2im_conv_tiles_per_image(H,W)][3][Co] belonging to image b are contiguous; the lazy input
 *     coefficients are per image: in_scale_shift_per_image [B][2][Ci] (or NULL).  +bias epilogue only.
 *   im2im_groupnorm_stats: the same (mean, M2, n) partial rows for a tensor z [B][HW][C] this library's conv did not
 *     produce: partial [im2im_groupnorm_stats_rows(B,HW)][3][C], rows of an image contiguous.
 *   im2im_groupnorm_finalize: rows -> mean_rstd [B][2][C] (the group's mean / rstd replicated per channel) and
 *     scale_shift [B][2][C] (scale = gamma*rstd, shift = beta - mean*scale): consumers apply max(z*scale+shift, 0).
 *   im2im_affine_relu_apply_per_image: a = max(z*scale + shift, 0) with those per-image coefficients.
 *   im2im_groupnorm_relu_bwd: da [B][HW][C] -> dz, dgamma [C], dbeta [C]:
 *     g = da*[z*scale+shift > 0]; dz = rstd*(gamma*g - mean_g(gamma*g) - xhat*mean_g(gamma*g*xhat)), means over the
 *     group's channels and the image's pixels.  Deterministic (no atomics).  C % 8 == 0, C <= 1024, C % G == 0. */
int64_t im2im_conv_tiles_per_image(int32_t H, int32_t W);
int im2im_conv_fwd_per_image(const void* x, const float* in_scale_shift_per_image, const void* wf, const float* bias,
                             void* y, float* stats, int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                             int32_t taps, int32_t dtype, im2im_stream_t stream);
int64_t im2im_groupnorm_stats_rows(int32_t B, int64_t HW);
int im2im_groupnorm_stats(const void* z, int32_t B, int64_t HW, int32_t C, int32_t dtype, float* partial,
                          im2im_stream_t stream);
int im2im_groupnorm_finalize(const float* partial, int32_t B, int32_t rows_per_image, int32_t C, int32_t G,
                             const float* gamma, const float* beta, float eps, float* mean_rstd,
                             float* scale_shift, im2im_stream_t stream);
int im2im_affine_relu_apply_per_image(const void* z, const float* scale_shift, void* a, int32_t B, int64_t HW,
                                      int32_t C, int32_t dtype, im2im_stream_t stream);
int64_t im2im_groupnorm_relu_bwd_workspace_bytes(int32_t B, int64_t HW, int32_t C);
int im2im_groupnorm_relu_bwd(const void* da, const void* z, const float* scale_shift, const float* mean_rstd,
                             const float* gamma, void* dz, float* dgamma, float* dbeta, int32_t B, int64_t HW,
                             int32_t C, int32_t G, int32_t dtype, void* ws, int64_t ws_bytes,
                             im2im_stream_t stream);

/* Elementwise pieces of the SURVEY 8(f) layers.
 *   im2im_head_activation_fwd/bwd: the activation on one plane of a final layer's packed output out [B][K][P] fp32 (plane
 *     at element offset plane_offset inside each image of img_stride elements): kind 0 = ReLU (GaussianRegressionLayer's
 *     variance, finallayers/gaussian_layer.py:15-17), 1 = abs (ResidualMagnitude*Layer, residual_magnitude_layer.py:15-17).
 *     fwd rewrites the plane in place and keeps the pre-activation values in pre [B][P]; bwd multiplies the plane's
 *     gradient in place by [pre > 0] or sign(pre).
 *   im2im_depth_space2: the pixel shuffle of nn.ConvTranspose2d(k=2, s=2) evaluated as a 1x1 convolution to 4*C channels
 *     (Up with bilinear=False, unet_parts.py:53): to_space != 0: in [B][h][w][(a, b, c)] -> out [B][2h][2w][c]; else the
 *     inverse (its backward).  C * element size must be a multiple of 16 bytes. */
int im2im_head_activation_fwd(float* out, float* pre, int64_t B, int64_t P, int64_t img_stride, int64_t plane_offset,
                              int32_t kind, im2im_stream_t stream);
int im2im_head_activation_bwd(float* dout, const float* pre, int64_t B, int64_t P, int64_t img_stride,
                              int64_t plane_offset, int32_t kind, im2im_stream_t stream);
int im2im_depth_space2(const void* in, void* out, int64_t B, int32_t h, int32_t w, int32_t C, int32_t to_space,
                       int32_t dtype, im2im_stream_t stream);

/* out[c] = sum_m x[m][c] (bias gradient of the 1x1 out conv, unet_parts.py:90). */
int64_t im2im_colsum_workspace_bytes(int64_t M, int32_t C);
int im2im_colsum(const void* x, float* out, int64_t M, int32_t C, int32_t dtype, void* ws,
                 int64_t ws_bytes, im2im_stream_t stream);

/* MaxPool2d(2) (unet_parts.py:34), NHWC; ties keep the first maximum in (h,w) order. */
int im2im_maxpool2_fwd(const void* x, const float* in_scale_shift, void* y, int32_t B, int32_t H, int32_t W,
                       int32_t C, int32_t dtype, im2im_stream_t stream);
int im2im_maxpool2_bwd(const void* x, const float* in_scale_shift, const void* dy, void* dx, int32_t B,
                       int32_t H, int32_t W, int32_t C, int32_t dtype, im2im_stream_t stream);

/* Up-block input (unet_parts.py:58-68): out = cat([skip, zero_pad(bilinear_x2_align_corners(deep))])
 * on the channel axis, NHWC.  deep [B][h][w][Cd], skip [B][H][W][Cs], out [B][H][W][Cs+Cd], H >= 2h, W >= 2w.
 * bwd: dskip = dout[..., :Cs]; ddeep = transpose of the interpolation (gathered, no atomics).
 * Cs == 0 (skip / skip_scale_shift / dskip NULL): plain upsample + pad, out/dout [B][H][W][Cd] -- used with
 * im2im_conv_fwd_split, which reads the skip half in place. */
int im2im_upsample2x_concat_fwd(const void* deep, const float* deep_scale_shift, const void* skip,
                                const float* skip_scale_shift, void* out, int32_t B, int32_t h, int32_t w,
                                int32_t Cd, int32_t H, int32_t W, int32_t Cs, int32_t dtype,
                                im2im_stream_t stream);
int im2im_upsample2x_concat_bwd(const void* dout, void* ddeep, void* dskip, int32_t B, int32_t h,
                                int32_t w, int32_t Cd, int32_t H, int32_t W, int32_t Cs, int32_t dtype,
                                im2im_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Small-channel direct 3x3 convolutions (pad 1), 16x16-pixel tiles, weights fp32.
 * s2l: in fp32 NCHW [B][CS][H][W] (CS <= 8) -> out NHWC [B][H][W][CL] (CL in {32,64}).
 *   w [CS][9][CL] (tap index reversed when flip != 0); bias [CL]|NULL; scale_shift [2][CL]|NULL; relu;
 *   stats [im2im_smallconv_tiles][3][CL]|NULL as for im2im_conv_fwd.  Uses: first UNet conv
 *   (unet_parts.py:16, Cin = n_in) and the data gradient of the quantile heads.
 * l2s: in NHWC [B][H][W][CL] -> out fp32 NCHW [B][CS][H][W]; w [CS][9][CL]; bias [CS]|NULL.  Use: the three
 *   quantile heads written directly as [B,3,C,H,W] (finallayers/quantile_layer.py:15-17,20).
 * wgrad: dw from S fp32 NCHW [B][CS][H][W] and L NHWC [B][H][W][CL]:
 *   l_major != 0: dw[(l*CS+s)*9+tap] = sum_px L[px][l]*S[s][px+tap]   (first conv: S = input, L = dz)
 *   l_major == 0: dw[(s*CL+l)*9+tap] = sum_px S[s][px]*L[px+tap][l]   (heads: S = dout, L = features)
 *   dbias [CS]|NULL = sum_px S[s][px].
 */
int64_t im2im_smallconv_tiles(int32_t B, int32_t H, int32_t W);
int im2im_smallconv_s2l_fwd(const float* in, const float* w, const float* bias, const float* center,
                            const float* scale_shift, void* out, float* stats, int32_t B, int32_t H, int32_t W, int32_t CS,
                            int32_t CL, int32_t relu, int32_t flip, int32_t dtype, im2im_stream_t stream);
int im2im_smallconv_l2s_fwd(const void* in, const float* w, const float* bias, float* out, int32_t B,
                            int32_t H, int32_t W, int32_t CL, int32_t CS, int32_t dtype,
                            im2im_stream_t stream);

int64_t im2im_smallconv_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t CS, int32_t CL);
int im2im_smallconv_wgrad(const float* S, const void* L, float* dw, float* dbias, int32_t B, int32_t H,
                          int32_t W, int32_t CS, int32_t CL, int32_t l_major, int32_t dtype, void* ws,
                          int64_t ws_bytes, im2im_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Fused quantile loss: quantile_regression_loss_fn (finallayers/quantile_layer.py:23-32) with
 * PinballLoss (losses/pinball.py:12-26) and nn.MSELoss, all mean-reduced:
 *   loss = w_lo*pinball_{q_lo}(lo, y) + w_hi*pinball_{q_hi}(hi, y) + w_mse*mean((mid - y)^2)
 * lo/mid/hi: fp32 planes, element (n, i) at ptr[n*img_stride + i], i < P; target [N][P].
 * bwd writes d_lo/d_mid/d_hi (each may be NULL) at [n*d_stride + i], scaled by the device scalar *grad_out. */
int64_t im2im_quantile_loss_workspace_bytes(void);
int im2im_quantile_loss_fwd(const float* lo, const float* mid, const float* hi, const float* target,
                            int64_t N, int64_t P, int64_t img_stride, float q_lo, float q_hi, float w_lo,
                            float w_hi, float w_mse, float* loss, void* ws, im2im_stream_t stream);
int im2im_quantile_loss_bwd(const float* lo, const float* mid, const float* hi, const float* target,
                            int64_t N, int64_t P, int64_t img_stride, float q_lo, float q_hi, float w_lo,
                            float w_hi, float w_mse, const float* grad_out, float* d_lo, float* d_mid,
                            float* d_hi, int64_t d_stride, im2im_stream_t stream);

/* The training losses of the other final layers on the same fused reduction, all mean-reduced;
 * a, b, c are the output planes addressed like lo/mid/hi above (c NULL for two-plane layers), `kind`:
 *   IM2IM_LOSS_QUANTILE (0)     w0*pinball_{q_lo}(a) + w1*pinball_{q_hi}(c) + w2*MSE(b)      quantile_layer.py:23-32
 *   IM2IM_LOSS_QUANTILE_L1 (1)  ... + w2*L1(b)                                            quantile_l1_layer.py:23-32
 *   IM2IM_LOSS_GAUSSIAN (2)     nn.GaussianNLLLoss(a = mean, y, b = var), eps 1e-6: mean(0.5*(log v + (a-y)^2/v)),
 *                               v = max(var, eps) with the gradient passed straight to var     gaussian_layer.py:19-23
 *   IM2IM_LOSS_RESIDUAL (3)     MSE(a, y) + MSE(b, |y - a|)                                residual_magnitude_layer.py:19-25
 *   IM2IM_LOSS_RESIDUAL_L1 (4)  L1(a, y) + MSE(b, |y - a|)                              residual_magnitude_l1_layer.py:19-25
 *   IM2IM_LOSS_INN (5)          MSE(b, y) + mean(relu(y - c)^2 + relu(a - y)^2 + beta*|c - a|), beta passed as q_lo
 *                                                                                         inn_layer.py:22-28, losses/inn.py:12-21
 * Workspace: im2im_quantile_loss_workspace_bytes(). */
#define IM2IM_LOSS_QUANTILE 0
#define IM2IM_LOSS_QUANTILE_L1 1
#define IM2IM_LOSS_GAUSSIAN 2
#define IM2IM_LOSS_RESIDUAL 3
#define IM2IM_LOSS_RESIDUAL_L1 4
#define IM2IM_LOSS_INN 5
int im2im_uq_loss_fwd(int32_t kind, const float* a, const float* b, const float* c, const float* target, int64_t N,
                      int64_t P, int64_t img_stride, float q_lo, float q_hi, float w0, float w1, float w2,
                      float* loss, void* ws, im2im_stream_t stream);
int im2im_uq_loss_bwd(int32_t kind, const float* a, const float* b, const float* c, const float* target, int64_t N,
                      int64_t P, int64_t img_stride, float q_lo, float q_hi, float w0, float w1, float w2,
                      const float* grad_out, float* d_a, float* d_b, float* d_c, int64_t d_stride,
                      im2im_stream_t stream);

/* Softmax final layer (finallayers/softmax_layer.py, n_channels_out = 1): logits [M = B*H*W][stride] in `dtype`, NHWC
 * pixels x class channels padded to `stride` (a multiple of 8, <= 64), the first K (= num_softmax <= 64) valid.
 *   im2im_softmax_ce_fwd/bwd: softmax_loss_fn (:15-25) = nn.CrossEntropyLoss (mean over pixels) against
 *     torch.bucketize(target, bounds, right=False) folded to K-1; bounds [K] fp32 on the device (the caller passes
 *     torch.linspace(0, 1, K) so class edges are torch's).  bwd writes d(logits) [M][stride] in `dtype` (padding = 0),
 *     scaled by the device scalar *grad_out.  ws: im2im_quantile_loss_workspace_bytes().
 *   im2im_softmax_sets_summary: the lambda-independent part of softmax_nested_sets_from_output (:33-47): softmax,
 *     cumulative sum, lower/upper quantile bins at 0.05/0.95, argmax prediction, collapse guards, [0,1] clamp ->
 *     out3 [N][3][P] fp32 planes (lq, pred, uq), the input of the calibration kernels with form IM2IM_SETS_SOFTMAX. */
int im2im_softmax_ce_fwd(const void* logits, const float* target, const float* bounds, int64_t M, int32_t K,
                         int32_t stride, int32_t dtype, float* loss, void* ws, im2im_stream_t stream);
int im2im_softmax_ce_bwd(const void* logits, const float* target, const float* bounds, int64_t M, int32_t K,
                         int32_t stride, int32_t dtype, const float* grad_out, void* dlogits, im2im_stream_t stream);
int im2im_softmax_sets_summary(const void* logits, int64_t N, int64_t P, int32_t K, int32_t stride, int32_t dtype,
                               float* out3, im2im_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * fastMRI input pipeline: glue kernels around two exact-fp32 MFMA GEMMs (im2im_conv_fwd, taps = 1,
 * dtype IM2IM_F32) that evaluate the centred orthonormal inverse 2-D DFT restricted to the cropped output.  Replaces, on
 * the device, apply_mask core/datasets/fastmri/transforms.py:53-85, ifft2c_new fftc.py:87-110, complex_center_crop
 * transforms.py:130-152, complex_abs math_util.py:56-70, center_crop transforms.py:108-127 and the normalisation of
 * FastMRIDataset.__getitem__ FastMRIDataset.py:147-160.  All tensors fp32.
 *   im2im_fastmri_mask_pack: kspace [B][R][C][2], mask [B][C] (mask_stride = C) or one shared [C] (mask_stride = 0)
 *     -> out [rows_out][Kp]: row (b, r) = the slice's k-space row * mask + 0.0, columns >= 2C and rows >= B*R zero
 *     (GEMM padding: Kp % 32 == 0, rows_out % 16 == 0 are the caller's business).
 *   im2im_complex_transpose: in [B][R][ld_in floats] (X complex used) -> out [B][X][ld_out floats] (R complex, rest 0).
 *   im2im_fastmri_abs_normalize: in [B][X][ld_in floats] (Y complex used, x-major) -> out [B][Y][X] =
 *     (sqrt(re^2 + im^2) - sub) / div with the reference's rounding (no fma, true division).
 *   im2im_center_crop_affine: out [B][H][W] = (centre crop of in [B][Hin][Win] - sub) / div. */
int im2im_fastmri_mask_pack(const float* kspace, const float* mask, int64_t mask_stride, float* out, int64_t B,
                            int32_t R, int32_t C, int64_t rows_out, int32_t Kp, im2im_stream_t stream);
int im2im_complex_transpose(const float* in, float* out, int32_t B, int32_t R, int32_t X, int32_t ld_in,
                            int32_t ld_out, im2im_stream_t stream);
int im2im_fastmri_abs_normalize(const float* in, float* out, int32_t B, int32_t X, int32_t Y, int32_t ld_in,
                                float sub, float div, im2im_stream_t stream);
int im2im_center_crop_affine(const float* in, float* out, int64_t B, int32_t Hin, int32_t Win, int32_t H,
                             int32_t W, float sub, float div, im2im_stream_t stream);

/* Multi-tensor Adam: optim.Adam(net.parameters(), lr) at core/scripts/train.py:120 with torch
 * defaults (betas, eps, no weight decay, no amsgrad).  Host arrays of n_tensors device pointers / sizes;
 * `step` is the 1-based step count used for bias correction. */
int im2im_adam_step(int32_t n_tensors, float* const* params, const float* const* grads,
                    float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes, float lr,
                    float beta1, float beta2, float eps, int64_t step, im2im_stream_t stream);
/* The same update with the step count kept ON THE DEVICE (step_dev: int64, the number of steps taken so far; incremented by
 * the call; coef_dev: 2 floats of scratch owned by the caller for the lifetime of the launches): nothing about the launch
 * depends on a host-side counter, so it can be captured in a HIP graph and replayed -- torch.optim.Adam(capturable=True). */
int im2im_adam_step_dev(int32_t n_tensors, float* const* params, const float* const* grads,
                        float* const* exp_avg, float* const* exp_avg_sq, const int64_t* sizes, float lr,
                        float beta1, float beta2, float eps, int64_t* step_dev, float* coef_dev,
                        im2im_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* IM2IM_UQ_H */
