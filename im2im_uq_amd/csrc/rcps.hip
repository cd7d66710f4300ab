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

constexpr int HIST_THREADS = 512;
constexpr int MAX_L = 8192;  // LDS: (L+1) int32 histogram + L fp32 grid  <= 64 KiB

struct Pix { float lo_d, up_d, p, pm, pp; };

__device__ __forceinline__ bool miss_at(float lam, float l_d, float u_d, float p, float pm, float pp, float y) {
  // quantile_layer.py:41-42 then add_uncertainty.py:35-36
  float up = fmaxf(__fadd_rn(__fmul_rn(lam, u_d), p), pp);
  float lo = fminf(__fsub_rn(p, __fmul_rn(lam, l_d)), pm);
  return (lo > y) | (up < y);                                // calibrate_model.py:77-78
}

__device__ __forceinline__ int critical_index(float l, float p, float u, float y, const float* s_lam, int L,
                                              float g0, float inv_dg) {
  const float pm = __fsub_rn(p, 1e-6f), pp = __fadd_rn(p, 1e-6f);
  l = fminf(l, pm);                                          // quantile_layer.py:39
  u = fmaxf(u, pp);                                          // quantile_layer.py:40
  const float u_d = __fsub_rn(u, p), l_d = __fsub_rn(p, l);
  // estimate: smallest lam with lam*d >= |y-p|
  const float r = fabsf(y - p);
  const float d = (y > p) ? u_d : l_d;
  float t = __fdividef(r, d);
  t = (t == t) ? t : 0.f;
  float jf = ceilf((t - g0) * inv_dg);
  int j = (int)fminf(fmaxf(jf, 0.f), (float)L);
  // exact walk on the reference's fp32 expression (usually 1-2 evaluations)
  while (j > 0 && !miss_at(s_lam[j - 1], l_d, u_d, p, pm, pp, y)) --j;
  while (j < L && miss_at(s_lam[j], l_d, u_d, p, pm, pp, y)) ++j;
  return j;
}

// grid = (S, N): block (s, n) scans pixels [s*chunk, (s+1)*chunk) of image n.
__global__ __launch_bounds__(HIST_THREADS) void rcps_hist_kernel(
    const float* __restrict__ out3, const float* __restrict__ label, int64_t P, int64_t chunk,
    const float* __restrict__ lam, int L, int* __restrict__ hist) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  int* s_hist = reinterpret_cast<int*>(smem);                 // [L+1]
  float* s_lam = reinterpret_cast<float*>(smem + sizeof(int) * (size_t)(((L + 1) + 3) & ~3));
  for (int i = threadIdx.x; i <= L; i += HIST_THREADS) s_hist[i] = 0;
  for (int i = threadIdx.x; i < L; i += HIST_THREADS) s_lam[i] = lam[i];
  __syncthreads();
  const float g0 = s_lam[0];
  const float span = s_lam[L - 1] - s_lam[0];
  const float inv_dg = (L > 1 && span > 0.f) ? (float)(L - 1) / span : 0.f;

  const int64_t n = blockIdx.y;
  const float* lo_p = out3 + (n * 3 + 0) * P;
  const float* pr_p = out3 + (n * 3 + 1) * P;
  const float* up_p = out3 + (n * 3 + 2) * P;
  const float* y_p = label + n * P;
  const int64_t begin = (int64_t)blockIdx.x * chunk;
  const int64_t end = min(begin + chunk, P);
  int n_full = 0;                                             // pixels missed at every grid point

  const bool vec_ok = ((P & 3) == 0);                         // plane bases stay 16-B aligned
  if (vec_ok) {
    for (int64_t i = begin + (int64_t)threadIdx.x * 4; i < end; i += (int64_t)HIST_THREADS * 4) {
      const float4 l4 = *reinterpret_cast<const float4*>(lo_p + i);
      const float4 p4 = *reinterpret_cast<const float4*>(pr_p + i);
      const float4 u4 = *reinterpret_cast<const float4*>(up_p + i);
      const float4 y4 = *reinterpret_cast<const float4*>(y_p + i);
      const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, pv[4] = {p4.x, p4.y, p4.z, p4.w};
      const float uv[4] = {u4.x, u4.y, u4.z, u4.w}, yv[4] = {y4.x, y4.y, y4.z, y4.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int j = critical_index(lv[k], pv[k], uv[k], yv[k], s_lam, L, g0, inv_dg);
        if (j == L) ++n_full;
        else if (j > 0) atomicAdd(&s_hist[j], 1);
      }
    }
  } else {
    for (int64_t i = begin + threadIdx.x; i < end; i += HIST_THREADS) {
      const int j = critical_index(lo_p[i], pr_p[i], up_p[i], y_p[i], s_lam, L, g0, inv_dg);
      if (j == L) ++n_full;
      else if (j > 0) atomicAdd(&s_hist[j], 1);
    }
  }
  // the all-miss bin is the hot one: reduce it in registers across the wave, one LDS atomic per wave
  for (int off = 32; off > 0; off >>= 1) n_full += __shfl_down(n_full, off, 64);
  if ((threadIdx.x & 63) == 0 && n_full) atomicAdd(&s_hist[L], n_full);
  __syncthreads();
  int* g_hist = hist + n * (int64_t)(L + 1);
  for (int i = threadIdx.x + 1; i <= L; i += HIST_THREADS) {
    const int v = s_hist[i];
    if (v) atomicAdd(&g_hist[i], v);
  }
}

// one block per image: counts[col] = sum_{j > col} hist[j];  table = fp32(count) / fp32(P)
__global__ __launch_bounds__(256) void rcps_suffix_kernel(const int* __restrict__ hist, int L, float Pf,
                                                           float* __restrict__ table, int* __restrict__ counts) {
  __shared__ int s_part[256];
  const int64_t n = blockIdx.x;
  const int* h = hist + n * (int64_t)(L + 1);
  const int per = (L + 255) / 256;
  // thread t owns columns [t*per, (t+1)*per); column c needs hist[c+1 .. L]
  const int c0 = threadIdx.x * per, c1 = min(c0 + per, L);
  int local = 0;
  for (int c = c0; c < c1; ++c) local += h[c + 1];
  s_part[threadIdx.x] = local;
  __syncthreads();
  int above = 0;                                              // sum of hist over columns owned by higher threads
  for (int t = threadIdx.x + 1; t < 256; ++t) above += s_part[t];
  int run = above;
  for (int c = c1 - 1; c >= c0; --c) {
    run += h[c + 1];
    table[n * (int64_t)L + c] = (float)run / Pf;
    if (counts) counts[n * (int64_t)L + c] = run;
  }
}

// Spatial miscoverage counts at one lambda.  grid = (ceil(HW/ (256*4)), C, NSPLIT)
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
    const float* lo_p = out3 + (n * 3 + 0) * P + (int64_t)c * HW;
    const float* pr_p = out3 + (n * 3 + 1) * P + (int64_t)c * HW;
    const float* up_p = out3 + (n * 3 + 2) * P + (int64_t)c * HW;
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
      const float p = pv[k];
      const float pm = __fsub_rn(p, 1e-6f), pp = __fadd_rn(p, 1e-6f);
      const float l = fminf(lv[k], pm), u = fmaxf(uv[k], pp);
      const float up = fmaxf(__fadd_rn(__fmul_rn(lam, __fsub_rn(u, p)), p), pp);
      const float lo = fminf(__fsub_rn(p, __fmul_rn(lam, __fsub_rn(p, l))), pm);
      acc[k] += (int)(yv[k] > up) + (int)(yv[k] < lo);        // calibrate_model.py:47
    }
  }
  for (int k = 0; k < 4; ++k)
    if (i0 + k < HW && acc[k]) atomicAdd(&map[(int64_t)c * HW + i0 + k], acc[k]);
}

__global__ __launch_bounds__(256) void nested_sets_kernel(float* __restrict__ out3, int64_t N, int64_t P, float lam,
                                                           float* __restrict__ lower_edge, float* __restrict__ upper_edge,
                                                           int clamp_inplace) {
  const int64_t total = N * P;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / P, q = i - n * P;
    float* lo_p = out3 + (n * 3 + 0) * P + q;
    const float p = out3[(n * 3 + 1) * P + q];
    float* up_p = out3 + (n * 3 + 2) * P + q;
    const float pm = __fsub_rn(p, 1e-6f), pp = __fadd_rn(p, 1e-6f);
    const float l = fminf(*lo_p, pm), u = fmaxf(*up_p, pp);
    if (clamp_inplace) { *lo_p = l; *up_p = u; }
    upper_edge[i] = fmaxf(__fadd_rn(__fmul_rn(lam, __fsub_rn(u, p)), p), pp);
    lower_edge[i] = fminf(__fsub_rn(p, __fmul_rn(lam, __fsub_rn(p, l))), pm);
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

extern "C" int im2im_rcps_loss_table(const float* out3, const float* label, int64_t N, int64_t P,
                                     const float* lam, int32_t L, int32_t* hist_ws, float* table,
                                     int32_t* counts, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(out3 && label && lam && hist_ws && table);
  IM2IM_REQUIRE(N >= 0 && P > 0 && L >= 1 && L <= MAX_L);
  IM2IM_REQUIRE(P < (1 << 24));                               // fp32(count) exact, as in the reference's fp32 mean
  IM2IM_REQUIRE(N <= 65535 * 1024);
  if (N == 0) return IM2IM_OK;
  IM2IM_HIP(hipMemsetAsync(hist_ws, 0, sizeof(int32_t) * (size_t)N * (L + 1), stream));
  // split each image into S chunks so that small N still fills 256 CUs (chunk multiple of 4 px)
  int64_t S = 1;
  if (N < 2048) S = im2im::cdiv(2048, N);
  const int64_t min_chunk = (int64_t)HIST_THREADS * 4 * 2;
  if (S > im2im::cdiv(P, min_chunk)) S = im2im::cdiv(P, min_chunk);
  if (S < 1) S = 1;
  int64_t chunk = im2im::cdiv(im2im::cdiv(P, S), 4) * 4;
  S = im2im::cdiv(P, chunk);
  const size_t smem = sizeof(int) * (size_t)(((L + 1) + 3) & ~3) + sizeof(float) * (size_t)L;
  // y-dim of the grid is limited to 65535: tile N
  for (int64_t n0 = 0; n0 < N; n0 += 65535) {
    const int64_t nb = (N - n0 < 65535) ? (N - n0) : 65535;
    hipLaunchKernelGGL(rcps_hist_kernel, dim3((unsigned)S, (unsigned)nb), dim3(HIST_THREADS), smem, stream,
                       out3 + n0 * 3 * P, label + n0 * P, P, chunk, lam, (int)L, hist_ws + n0 * (L + 1));
    if (int rc = im2im::check_launch("rcps_hist_kernel")) return rc;
  }
  hipLaunchKernelGGL(rcps_suffix_kernel, dim3((unsigned)N), dim3(256), 0, stream, hist_ws, (int)L, (float)P, table, counts);
  return im2im::check_launch("rcps_suffix_kernel");
}

extern "C" int im2im_rcps_miscoverage(const float* out3, const float* label, int64_t N, int32_t C, int64_t HW,
                                      float lam, int32_t* map, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(out3 && label && map);
  IM2IM_REQUIRE(N >= 0 && C >= 1 && C <= 65535 && HW > 0);
  IM2IM_HIP(hipMemsetAsync(map, 0, sizeof(int32_t) * (size_t)C * HW, stream));
  if (N == 0) return IM2IM_OK;
  const int64_t bx = im2im::cdiv(HW, 256 * 4);
  int64_t nz = im2im::cdiv(4096, bx * C);
  if (nz > N) nz = N;
  if (nz > 1024) nz = 1024;
  if (nz < 1) nz = 1;
  hipLaunchKernelGGL(rcps_miscoverage_kernel, dim3((unsigned)bx, (unsigned)C, (unsigned)nz), dim3(256), 0, stream,
                     out3, label, N, (int)C, HW, lam, map);
  return im2im::check_launch("rcps_miscoverage_kernel");
}

extern "C" int im2im_nested_sets(float* out3, int64_t N, int64_t P, float lam, float* lower_edge,
                                 float* upper_edge, int32_t clamp_inplace, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(out3 && lower_edge && upper_edge && N >= 0 && P > 0);
  if (N == 0) return IM2IM_OK;
  int64_t blocks = im2im::cdiv(N * P, 256);
  if (blocks > 256 * 16) blocks = 256 * 16;
  hipLaunchKernelGGL(nested_sets_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, out3, N, P, lam, lower_edge,
                     upper_edge, (int)clamp_inplace);
  return im2im::check_launch("nested_sets_kernel");
}
