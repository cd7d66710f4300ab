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

#define IM2IM_ABI_VERSION 1
#define IM2IM_OK 0
#define IM2IM_ERR_INVALID (-1)     /* bad argument / unsupported shape */
#define IM2IM_ERR_HIP (-2)         /* a HIP runtime call or launch failed */
#define IM2IM_ERR_UNSUPPORTED (-3)

typedef void* im2im_stream_t;      /* hipStream_t */

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
 *   out3    [N][3][P] fp32   model output (lower, prediction, upper planes), P = C*H*W
 *   label   [N][P]    fp32
 *   lam     [L]       fp32   device; ascending grid of the lambdas the edges are evaluated at
 *                            (the caller passes lambdas - dlambda for calibrate_model, Q1)
 *   hist_ws [N][L+1]  int32  workspace (zeroed by the call)
 *   table   [N][L]    fp32   table[n][j] = (#pixels of image n missed at lam[j]) / P, bit-identical
 *                            to the reference's fp32 mean of 0/1 indicators
 *   counts  [N][L]    int32  optional (may be NULL): the integer miss counts
 * Edge arithmetic is fp32 with separate multiply and add (no FMA), as on the reference CPU path.
 */
int im2im_rcps_loss_table(const float* out3, const float* label, int64_t N, int64_t P,
                          const float* lam, int32_t L, int32_t* hist_ws, float* table,
                          int32_t* counts, im2im_stream_t stream);

/* Spatial miscoverage counts at one lambda; replaces the accumulation at
 * core/calibration/calibrate_model.py:47,55.
 *   map [C][HW] int32, zeroed by the call: map[c][i] = #images n with label > upper or label < lower
 */
int im2im_rcps_miscoverage(const float* out3, const float* label, int64_t N, int32_t C, int64_t HW,
                           float lam, int32_t* map, im2im_stream_t stream);

/* Elementwise nested sets at one lambda (a10): same two call sites as above, materialising
 * (lower_edge, upper_edge) [N][P].  If clamp_inplace != 0, out3's lower/upper planes are
 * overwritten with the clamped values like the reference does (quantile_layer.py:39-40, Q5). */
int im2im_nested_sets(float* out3, int64_t N, int64_t P, float lam, float* lower_edge,
                      float* upper_edge, int32_t clamp_inplace, im2im_stream_t stream);

/* Per-image miss fraction for already-materialised edges; replaces fraction_missed_loss
 * core/calibration/calibrate_model.py:76-80.  lower/upper/label [N][P] fp32 -> loss [N] fp32
 * = fp32(#pixels with lower > label or upper < label) / fp32(P). */
int im2im_fraction_missed(const float* lower_edge, const float* upper_edge, const float* label,
                          int64_t N, int64_t P, float* loss, im2im_stream_t stream);

/* Hoeffding-Bentkus upper confidence bound, host float64; replaces HB_mu_plus
 * core/calibration/bounds.py:17-29 (scipy binom.cdf + brentq), including its "return 1.0 on any
 * solver exception" path (muhat == 0 -> NaN, Q3). */
double im2im_hb_mu_plus(double muhat, int64_t n, double delta, int32_t maxiters);

#ifdef __cplusplus
}
#endif
#endif /* IM2IM_UQ_H */
