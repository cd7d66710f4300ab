// RCPS calibration scoring kernels for gfx950: HBM-bound.
//
// One pass over (lower, prediction, upper, label) = 16 B/pixel produces the per-image miss
// fraction for EVERY lambda of the grid, instead of the reference's one pass per lambda
// (core/calibration/calibrate_model.py:134-136 -> :21-29 -> add_uncertainty.py:33-38 ->
// quantile_layer.py:39-42 -> calibrate_model.py:76-80).
//
// Per pixel the miss indicator  (lower_edge(lam) > y) | (upper_edge(lam) < y)  is a
// non-increasing step function of lam (fp32 multiply and add are monotone), so it is fully
// described by j* = the number of grid points at which the pixel is missed.  j* is found by an
// analytic estimate followed by an exact walk that evaluates the reference's own fp32 expression
// (separate v_mul_f32 / v_add_f32, never FMA-contracted: SURVEY Q9), then histogrammed per image
// in LDS; a suffix sum of the histogram is the miss count per lambda.  Counts are integers, so
// the result is independent of summation order and bit-identical to the reference.
//
// Compile with -ffp-contract=off (build.py does); the edge expressions additionally use
// __fmul_rn/__fadd_rn/__fsub_rn so no later flag change can fuse them.
#include "common.h"

namespace {

constexpr int HIST_THREADS = 256;
constexpr int HIST_BLOCKS_PER_CU = 8;   // __launch_bounds__(256, 8): <= 64 VGPRs -> 8 workgroups (32 waves) resident per CU
constexpr int MAX_L = 8192;  // LDS: (L+1) int32 histogram + L fp32 grid  <= 64 KiB

// equal contiguous ranges of the pixel stream, one per workgroup; the grid is exactly what is resident at once
// (8 workgroups of 256 threads on each of the 256 CUs; measured best of 2..32 per CU), so there is no second,
// partial wave of workgroups
inline void rcps_partition(int64_t N, int64_t P, int L, int64_t* grid, int64_t* per, int64_t* maxseg, int64_t* units_per_img) {
  (void)L;
  const bool vec = (P & 3) == 0;
  const int64_t upi = vec ? P / 4 : P;
  const int64_t units = N * upi;
  int64_t g = 256 * HIST_BLOCKS_PER_CU;
  const int64_t min_units = (int64_t)HIST_THREADS * 2;        // not finer than 2 units per thread
  if (g > im2im::cdiv(units, min_units)) g = im2im::cdiv(units, min_units);
  if (g < 1) g = 1;
  *per = im2im::cdiv(units, g);
  *grid = im2im::cdiv(units, *per);
  *maxseg = (*per - 1) / upi + 2;                             // images a range of `per` units can touch
  *units_per_img = upi;
}

typedef float f32x4 __attribute__((ext_vector_type(4)));

// How a final layer's output planes turn into the half-widths (d_lo, d_up) of the nested set
//   lower_edge(lam) = min(p - lam*d_lo, p - 1e-6),  upper_edge(lam) = max(lam*d_up + p, p + 1e-6)   (add_uncertainty.py:33-38)
// FORM 0  planes (l, p, u): d_lo = p - min(l, p-1e-6), d_up = max(u, p+1e-6) - p     quantile(_l1)_layer.py:39-42
// FORM 1  planes (p, s):    d_lo = d_up = s                                          residual_magnitude(_l1)_layer.py:33-34
// FORM 2  planes (p, v):    d_lo = d_up = sqrt(v)  (correctly rounded, as torch)      gaussian_layer.py:31-32
// FORM 3  planes (lq, p, uq): d_lo = relu(p - lq), d_up = relu(uq - p)               softmax_layer.py:50-51 (lq, p, uq are
//         the lambda-independent quantile summary of the softmax output, im2im_softmax_sets_summary)
// torch.maximum / torch.minimum / relu propagate NaN (fmaxf / fminf return the other operand): a NaN bound must stay NaN so
// that, as in the reference, every comparison with it is false and the pixel is not counted as missed
__device__ __forceinline__ float nan_max(float a, float b) { return (a != a) ? a : fmaxf(a, b); }
__device__ __forceinline__ float nan_min(float a, float b) { return (a != a) ? a : fminf(a, b); }

template <int FORM> struct SetForm {
  static constexpr int PLANES = (FORM == 0 || FORM == 3) ? 3 : 2;
  static constexpr int PRED = (FORM == 0 || FORM == 3) ? 1 : 0;
  // a = plane 0, b = plane 1, c = plane 2 (FORM 0 only)
  static __device__ __forceinline__ void widths(float a, float b, float c, float& p, float& d_lo, float& d_up) {
    if constexpr (FORM == 0) {
      p = b;
      d_lo = __fsub_rn(p, nan_min(a, __fsub_rn(p, 1e-6f)));
      d_up = __fsub_rn(nan_max(c, __fadd_rn(p, 1e-6f)), p);
    } else if constexpr (FORM == 1) {
      p = a; d_lo = b; d_up = b;
    } else if constexpr (FORM == 3) {
      p = b;
      d_lo = nan_max(__fsub_rn(p, a), 0.f);
      d_up = nan_max(__fsub_rn(c, p), 0.f);
    } else {
      p = a; d_lo = sqrtf(b); d_up = d_lo;                    // correctly rounded (hipcc default -fhip-fp32-correctly-rounded-divide-sqrt; NOT __fsqrt_rn, which maps to the 1-ulp native instruction)
    }
  }
};

// Number of grid points at which the pixel is missed.  Simplifications that keep the result bit-identical to
// evaluating miss_at() at every grid point:
//   * only one side can miss: y > p -> only (upper_edge < y); y < p -> only (lower_edge > y); y == p -> never;
//   * max(v, pp) < y  <=>  v < y and pp < y (pp < y does not depend on lambda); likewise for the lower side;
//   * p - lam*d > y  <=>  lam*d + (-p) < -y  (IEEE negation is exact), so both sides share one form
//         miss(lam)  <=>  fl(fl(lam * d) + P) < Y .
// (for the symmetric forms lam*d may be negative at the shifted first grid point; the argument is unchanged: the floor
//  keeps lower_edge <= p and upper_edge >= p, and fp32 multiply/add stay monotone in lam for d >= 0)
template <int FORM>
__device__ __forceinline__ int critical_index(float a, float b, float c, float y, const float* s_lam, int L,
                                              float g0, float inv_dg) {
  float p, d_lo, d_up;
  SetForm<FORM>::widths(a, b, c, p, d_lo, d_up);
  const float pm = __fsub_rn(p, 1e-6f), pp = __fadd_rn(p, 1e-6f);
  const bool up_side = y > p;
  const float d = up_side ? d_up : d_lo;
  const float P = up_side ? p : -p;
  const float Y = up_side ? y : -y;
  const bool can_miss = up_side ? (pp < y) : (pm > y);                                  // add_uncertainty.py:35-36 floor
  if (!can_miss) return 0;
  // estimate: smallest lam with lam*d >= Y - P, then an exact walk on the reference's fp32 expression
  float t = __fdividef(Y - P, d);
  t = (t == t) ? t : 0.f;
  int j = (int)fminf(fmaxf(ceilf((t - g0) * inv_dg), 0.f), (float)L);
  while (j > 0 && !(__fadd_rn(__fmul_rn(s_lam[j - 1], d), P) < Y)) --j;
  while (j < L && (__fadd_rn(__fmul_rn(s_lam[j], d), P) < Y)) ++j;
  return j;
}

// Persistent, perfectly balanced partition: the N*P pixel stream is cut into gridDim.x equal contiguous ranges
// (units of 4 pixels when P % 4 == 0), so every workgroup moves the same number of bytes and there is no
// tail wave.  A range crosses at most `maxseg` images; at each image boundary the LDS histogram is flushed to the
// (workgroup, segment) row of `partial` with plain stores; the suffix kernel adds the few rows that cover an image.
template <bool VEC, int FORM>
__global__ __launch_bounds__(HIST_THREADS, HIST_BLOCKS_PER_CU) void rcps_hist_kernel(
    const float* __restrict__ out3, const float* __restrict__ label, int64_t N, int64_t P,
    const float* __restrict__ lam, int L, int* __restrict__ partial, int maxseg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_hist = reinterpret_cast<int*>(smem);                 // [L+1]
  float* s_lam = reinterpret_cast<float*>(smem + sizeof(int) * (size_t)(((L + 1) + 3) & ~3));
  for (int i = threadIdx.x; i < L; i += HIST_THREADS) s_lam[i] = lam[i];
  for (int i = threadIdx.x; i <= L; i += HIST_THREADS) s_hist[i] = 0;
  __syncthreads();
  const float g0 = s_lam[0];
  const float span = s_lam[L - 1] - s_lam[0];
  const float inv_dg = (L > 1 && span > 0.f) ? (float)(L - 1) / span : 0.f;

  constexpr int U = VEC ? 4 : 1;                              // pixels per unit
  const int64_t units_per_img = P / U;
  const int64_t total = N * units_per_img;
  const int64_t per = (total + gridDim.x - 1) / gridDim.x;
  int64_t u = (int64_t)blockIdx.x * per;
  const int64_t u_end = min(u + per, total);
  const int64_t n_first = u / units_per_img;
  while (u < u_end) {
    const int64_t n = u / units_per_img;
    const int64_t seg_end = min(u_end, (n + 1) * units_per_img);   // stay inside image n
    constexpr int K = SetForm<FORM>::PLANES;
    const float* lo_p = out3 + (n * K + 0) * P;                 // planes 0, 1 and (FORM 0) 2 of image n
    const float* pr_p = out3 + (n * K + 1) * P;
    const float* up_p = out3 + (n * K + (K - 1)) * P;
    const float* y_p = label + n * P;
    const int64_t base = n * units_per_img;
    int n_full = 0;                                           // pixels missed at every grid point
    for (int64_t v = u + threadIdx.x; v < seg_end; v += HIST_THREADS) {
      const int64_t i = (v - base) * U;
      if constexpr (VEC) {
        // every byte is read exactly once: non-temporal loads (measured +15 % over default-policy loads)
        const f32x4 l4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(lo_p + i));
        const f32x4 p4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(pr_p + i));
        const f32x4 u4 = (K == 3) ? __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(up_p + i)) : p4;
        const f32x4 y4 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(y_p + i));
        const float lv[4] = {l4[0], l4[1], l4[2], l4[3]}, pv[4] = {p4[0], p4[1], p4[2], p4[3]};
        const float uv[4] = {u4[0], u4[1], u4[2], u4[3]}, yv[4] = {y4[0], y4[1], y4[2], y4[3]};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int j = critical_index<FORM>(lv[k], pv[k], uv[k], yv[k], s_lam, L, g0, inv_dg);
          if (j == L) ++n_full;
          else if (j > 0) atomicAdd(&s_hist[j], 1);
        }
      } else {
        const int j = critical_index<FORM>(lo_p[i], pr_p[i], up_p[i], y_p[i], s_lam, L, g0, inv_dg);
        if (j == L) ++n_full;
        else if (j > 0) atomicAdd(&s_hist[j], 1);
      }
    }
    // the all-miss bin is the hot one: reduce it in registers across the wave, one LDS atomic per wave
    for (int off = 32; off > 0; off >>= 1) n_full += __shfl_down(n_full, off, 64);
    if ((threadIdx.x & 63) == 0 && n_full) atomicAdd(&s_hist[L], n_full);
    __syncthreads();
    // this (workgroup, image-segment)'s own row: plain stores, no atomics, no zero-initialisation needed
    int* row = partial + ((int64_t)blockIdx.x * maxseg + (n - n_first)) * (int64_t)(L + 1);
    for (int i = threadIdx.x; i <= L; i += HIST_THREADS) { row[i] = s_hist[i]; s_hist[i] = 0; }
    __syncthreads();
    u = seg_end;
  }
}

// one block per image: hist = sum of the partial rows of the workgroups whose range intersects the image;
// counts[col] = sum_{j > col} hist[j];  table = fp32(count) / fp32(P)
__global__ __launch_bounds__(256) void rcps_suffix_kernel(const int* __restrict__ partial, int maxseg, int64_t units_per_img,
                                                           int64_t per, int64_t total, int L, float Pf,
                                                           float* __restrict__ table, int* __restrict__ counts) {
  extern __shared__ int s_h[];                                // [L+1] then [256]
  int* s_part = s_h + (L + 1);
  const int64_t n = blockIdx.x;
  const int64_t b_lo = (n * units_per_img) / per;
  const int64_t b_hi = min(((n + 1) * units_per_img - 1) / per, (total - 1) / per);
  for (int i = threadIdx.x; i <= L; i += 256) {
    int acc = 0;
    for (int64_t b = b_lo; b <= b_hi; ++b) {
      const int64_t n_first = (b * per) / units_per_img;
      acc += partial[(b * maxseg + (n - n_first)) * (int64_t)(L + 1) + i];
    }
    s_h[i] = acc;
  }
  __syncthreads();
  const int per_t = (L + 255) / 256;
  const int c0 = threadIdx.x * per_t, c1 = min(c0 + per_t, L);
  int local = 0;
  for (int c = c0; c < c1; ++c) local += s_h[c + 1];
  s_part[threadIdx.x] = local;
  __syncthreads();
  int above = 0;                                              // sum of hist over columns owned by higher threads
  for (int t = threadIdx.x + 1; t < 256; ++t) above += s_part[t];
  int run = above;
  for (int c = c1 - 1; c >= c0; --c) {
    run += s_h[c + 1];
    table[n * (int64_t)L + c] = (float)run / Pf;
    if (counts) counts[n * (int64_t)L + c] = run;
  }
}

// Spatial miscoverage counts at one lambda.  grid = (ceil(HW/ (256*4)), C, NSPLIT)
template <int FORM>
__global__ __launch_bounds__(256) void rcps_miscoverage_kernel(
    const float* __restrict__ out3, const float* __restrict__ label, int64_t N, int C, int64_t HW, float lam,
    int* __restrict__ map) {
  const int c = blockIdx.y;
  const int64_t P = (int64_t)C * HW;
  const int64_t n_per = (N + gridDim.z - 1) / gridDim.z;
  const int64_t n0 = blockIdx.z * n_per, n1 = min(n0 + n_per, N);
  const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (i0 >= HW) return;
  int acc[4] = {0, 0, 0, 0};
  const bool vec_ok = ((HW & 3) == 0);
  for (int64_t n = n0; n < n1; ++n) {
    constexpr int K = SetForm<FORM>::PLANES;
    const float* lo_p = out3 + (n * K + 0) * P + (int64_t)c * HW;
    const float* pr_p = out3 + (n * K + 1) * P + (int64_t)c * HW;
    const float* up_p = out3 + (n * K + (K - 1)) * P + (int64_t)c * HW;
    const float* y_p = label + n * P + (int64_t)c * HW;
    float lv[4], pv[4], uv[4], yv[4];
    if (vec_ok) {
      const float4 l4 = *reinterpret_cast<const float4*>(lo_p + i0), p4 = *reinterpret_cast<const float4*>(pr_p + i0);
      const float4 u4 = *reinterpret_cast<const float4*>(up_p + i0), y4 = *reinterpret_cast<const float4*>(y_p + i0);
      lv[0] = l4.x; lv[1] = l4.y; lv[2] = l4.z; lv[3] = l4.w; pv[0] = p4.x; pv[1] = p4.y; pv[2] = p4.z; pv[3] = p4.w;
      uv[0] = u4.x; uv[1] = u4.y; uv[2] = u4.z; uv[3] = u4.w; yv[0] = y4.x; yv[1] = y4.y; yv[2] = y4.z; yv[3] = y4.w;
    } else {
      for (int k = 0; k < 4; ++k) {
        const bool in = i0 + k < HW;
        lv[k] = in ? lo_p[i0 + k] : 0.f; pv[k] = in ? pr_p[i0 + k] : 0.f;
        uv[k] = in ? up_p[i0 + k] : 0.f; yv[k] = in ? y_p[i0 + k] : 0.f;
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float p, d_lo, d_up;
      SetForm<FORM>::widths(lv[k], pv[k], uv[k], p, d_lo, d_up);
      const float pm = __fsub_rn(p, 1e-6f), pp = __fadd_rn(p, 1e-6f);
      const float up = nan_max(__fadd_rn(__fmul_rn(lam, d_up), p), pp);
      const float lo = nan_min(__fsub_rn(p, __fmul_rn(lam, d_lo)), pm);
      acc[k] += (int)(yv[k] > up) + (int)(yv[k] < lo);        // calibrate_model.py:47
    }
  }
  for (int k = 0; k < 4; ++k)
    if (i0 + k < HW && acc[k]) atomicAdd(&map[(int64_t)c * HW + i0 + k], acc[k]);
}

// lower/upper edges at one lambda.  floor != 0: with the +-1e-6 floor of ModelWithUncertainty.nested_sets_from_output
// (add_uncertainty.py:35-36); floor == 0: the final layer's own *_nested_sets_from_output.  clamp_inplace (FORM 0):
// write the clamped lower/upper planes back like the reference's in-place assignment (quantile_layer.py:39-40, Q5).
template <int FORM>
__global__ __launch_bounds__(256) void nested_sets_kernel(float* __restrict__ out3, int64_t N, int64_t P, float lam,
                                                           float* __restrict__ lower_edge, float* __restrict__ upper_edge,
                                                           int clamp_inplace, int floor) {
  constexpr int K = SetForm<FORM>::PLANES;
  const int64_t total = N * P;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / P, q = i - n * P;
    float* a_p = out3 + (n * K + 0) * P + q;
    float* b_p = out3 + (n * K + 1) * P + q;
    float* c_p = out3 + (n * K + (K - 1)) * P + q;
    float p, d_lo, d_up;
    SetForm<FORM>::widths(*a_p, *b_p, *c_p, p, d_lo, d_up);
    const float pm = __fsub_rn(p, 1e-6f), pp = __fadd_rn(p, 1e-6f);
    if (FORM == 0 && clamp_inplace) { *a_p = nan_min(*a_p, pm); *c_p = nan_max(*c_p, pp); }
    float up = __fadd_rn(__fmul_rn(lam, d_up), p);
    // FORM 0: p - lam*d_lo (quantile_layer.py:42); FORM 1/2: (-lam)*d + p (gaussian_layer.py:32) -- the same fp32 value
    float lo = __fsub_rn(p, __fmul_rn(lam, d_lo));
    if (floor) { up = nan_max(up, pp); lo = nan_min(lo, pm); }
    upper_edge[i] = up;
    lower_edge[i] = lo;
  }
}

// one block per image
__global__ __launch_bounds__(256) void fraction_missed_kernel(const float* __restrict__ lower, const float* __restrict__ upper,
                                                               const float* __restrict__ label, int64_t P,
                                                               float* __restrict__ loss) {
  __shared__ int s_cnt[4];
  const int64_t n = blockIdx.x;
  int cnt = 0;
  for (int64_t i = threadIdx.x; i < P; i += 256) {
    const float y = label[n * P + i];
    cnt += (int)((lower[n * P + i] > y) | (upper[n * P + i] < y));   // sum clamped to 1, calibrate_model.py:77-78
  }
  for (int off = 32; off > 0; off >>= 1) cnt += __shfl_down(cnt, off, 64);
  if ((threadIdx.x & 63) == 0) s_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  if (threadIdx.x == 0) loss[n] = (float)(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3]) / (float)P;
}

}  // namespace

extern "C" int im2im_fraction_missed(const float* lower_edge, const float* upper_edge, const float* label,
                                     int64_t N, int64_t P, float* loss, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(lower_edge && upper_edge && label && loss && N >= 0 && P > 0 && P < (1 << 24));
  if (N == 0) return IM2IM_OK;
  hipLaunchKernelGGL(fraction_missed_kernel, dim3((unsigned)N), dim3(256), 0, stream, lower_edge, upper_edge, label, P, loss);
  return im2im::check_launch("fraction_missed_kernel");
}

namespace {
template <int FORM>
void launch_hist(bool vec, int64_t grid, size_t smem, hipStream_t stream, const float* out3, const float* label, int64_t N,
                 int64_t P, const float* lam, int L, int* hist_ws, int maxseg) {
  if (vec) hipLaunchKernelGGL((rcps_hist_kernel<true, FORM>), dim3((unsigned)grid), dim3(HIST_THREADS), smem, stream, out3, label, N, P, lam, L, hist_ws, maxseg);
  else hipLaunchKernelGGL((rcps_hist_kernel<false, FORM>), dim3((unsigned)grid), dim3(HIST_THREADS), smem, stream, out3, label, N, P, lam, L, hist_ws, maxseg);
}
}  // namespace

extern "C" int im2im_rcps_loss_table(const float* out3, const float* label, int64_t N, int64_t P,
                                     const float* lam, int32_t L, int32_t form, int32_t* hist_ws, float* table,
                                     int32_t* counts, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(form >= 0 && form <= 3);
  IM2IM_REQUIRE(N >= 0 && P > 0 && L >= 1 && L <= MAX_L);
  IM2IM_REQUIRE(P < (1 << 24));                               // fp32(count) exact, as in the reference's fp32 mean
  if (N == 0) return IM2IM_OK;                                // an empty shard: nothing to read or write (pointers may be null)
  IM2IM_REQUIRE(out3 && label && lam && hist_ws && table);
  const size_t smem = sizeof(int) * (size_t)(((L + 1) + 3) & ~3) + sizeof(float) * (size_t)L;
  int64_t grid, per, maxseg, units_per_img;
  rcps_partition(N, P, L, &grid, &per, &maxseg, &units_per_img);
  const bool vec = (P & 3) == 0;
  if (form == 0) launch_hist<0>(vec, grid, smem, stream, out3, label, N, P, lam, (int)L, hist_ws, (int)maxseg);
  else if (form == 1) launch_hist<1>(vec, grid, smem, stream, out3, label, N, P, lam, (int)L, hist_ws, (int)maxseg);
  else if (form == 2) launch_hist<2>(vec, grid, smem, stream, out3, label, N, P, lam, (int)L, hist_ws, (int)maxseg);
  else launch_hist<3>(vec, grid, smem, stream, out3, label, N, P, lam, (int)L, hist_ws, (int)maxseg);
  if (int rc = im2im::check_launch("rcps_hist_kernel")) return rc;
  hipLaunchKernelGGL(rcps_suffix_kernel, dim3((unsigned)N), dim3(256), sizeof(int) * (size_t)(L + 1 + 256), stream, hist_ws, (int)maxseg,
                     units_per_img, per, N * units_per_img, (int)L, (float)P, table, counts);
  return im2im::check_launch("rcps_suffix_kernel");
}

extern "C" int64_t im2im_rcps_workspace_bytes(int64_t N, int64_t P, int32_t L) {
  if (N <= 0 || P <= 0 || L < 1) return 0;
  int64_t grid, per, maxseg, upi;
  rcps_partition(N, P, L, &grid, &per, &maxseg, &upi);
  return grid * maxseg * (int64_t)(L + 1) * (int64_t)sizeof(int32_t);
}

extern "C" int im2im_rcps_miscoverage(const float* out3, const float* label, int64_t N, int32_t C, int64_t HW,
                                      float lam, int32_t form, int32_t* map, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(map);
  IM2IM_REQUIRE(form >= 0 && form <= 3);
  IM2IM_REQUIRE(N >= 0 && C >= 1 && C <= 65535 && HW > 0);
  IM2IM_HIP(hipMemsetAsync(map, 0, sizeof(int32_t) * (size_t)C * HW, stream));
  if (N == 0) return IM2IM_OK;
  IM2IM_REQUIRE(out3 && label);
  const int64_t bx = im2im::cdiv(HW, 256 * 4);
  int64_t nz = im2im::cdiv(4096, bx * C);
  if (nz > N) nz = N;
  if (nz > 1024) nz = 1024;
  if (nz < 1) nz = 1;
  const dim3 mgrid((unsigned)bx, (unsigned)C, (unsigned)nz);
  if (form == 0) hipLaunchKernelGGL(rcps_miscoverage_kernel<0>, mgrid, dim3(256), 0, stream, out3, label, N, (int)C, HW, lam, map);
  else if (form == 1) hipLaunchKernelGGL(rcps_miscoverage_kernel<1>, mgrid, dim3(256), 0, stream, out3, label, N, (int)C, HW, lam, map);
  else if (form == 2) hipLaunchKernelGGL(rcps_miscoverage_kernel<2>, mgrid, dim3(256), 0, stream, out3, label, N, (int)C, HW, lam, map);
  else hipLaunchKernelGGL(rcps_miscoverage_kernel<3>, mgrid, dim3(256), 0, stream, out3, label, N, (int)C, HW, lam, map);
  return im2im::check_launch("rcps_miscoverage_kernel");
}

extern "C" int im2im_nested_sets(float* out3, int64_t N, int64_t P, float lam, int32_t form, float* lower_edge,
                                 float* upper_edge, int32_t clamp_inplace, int32_t floor, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(out3 && lower_edge && upper_edge && N >= 0 && P > 0);
  IM2IM_REQUIRE(form >= 0 && form <= 3);
  if (N == 0) return IM2IM_OK;
  int64_t blocks = im2im::cdiv(N * P, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (form == 0) hipLaunchKernelGGL(nested_sets_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, stream, out3, N, P, lam, lower_edge, upper_edge, (int)clamp_inplace, (int)floor);
  else if (form == 1) hipLaunchKernelGGL(nested_sets_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, stream, out3, N, P, lam, lower_edge, upper_edge, 0, (int)floor);
  else if (form == 2) hipLaunchKernelGGL(nested_sets_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, stream, out3, N, P, lam, lower_edge, upper_edge, 0, (int)floor);
  else hipLaunchKernelGGL(nested_sets_kernel<3>, dim3((unsigned)blocks), dim3(256), 0, stream, out3, N, P, lam, lower_edge, upper_edge, 0, (int)floor);
  return im2im::check_launch("nested_sets_kernel");
}
