// HBM-bound side passes of the UNet train/eval step on gfx950, all NHWC:
// BatchNorm(+ReLU) forward/backward, 2x2 max-pool, bilinear x2 upsample + pad + concat, fused
// quantile (pinball x2 + MSE) loss, multi-tensor Adam.  Every kernel moves 16 B per lane per access
// (8 bf16 or 4 fp32 channels) and does its arithmetic in fp32.
#include "common.h"
#include "dtypes.h"
#include "reduce.h"

namespace {
using namespace im2im;

// ------------------------------------------------------------------------------------------------
// BatchNorm2d, train mode (core/models/trunks/unet_parts.py:17,20; torch defaults eps=1e-5, momentum=0.1)
// Batch statistics arrive as per-tile partials (mean, M2 = sum of squared deviations from that mean, n) written by the
// conv epilogues and are merged with the pairwise update of Chan et al. in fp64 -- never as E[z^2] - E[z]^2, whose
// cancellation costs mean^2/var digits (torch's CPU path accumulates these sums in double; this keeps parity with it).
struct Moments { double n, mean, m2; };
__device__ __forceinline__ Moments merge_moments(const Moments& a, const Moments& b) {
  const double n = a.n + b.n;
  if (n <= 0.0) return Moments{0.0, 0.0, 0.0};
  const double d = b.mean - a.mean;
  return Moments{n, (a.n * a.mean + b.n * b.mean) / n, a.m2 + b.m2 + d * d * (a.n * b.n / n)};
}

// stage 1: partial[R][3][C] fp32 (mean, M2, n) -> tmp[S][3][C] fp64 (n, mean, M2) per split of rows.  Two passes over the
// split's rows (the second one hits L2): N and the split mean first, then M2 = sum M2_t + n_t (mean_t - mean)^2 -- plain
// fp64 sums in a fixed order, no division inside the loops.  Block = 64 channels x 16 row lanes.
__global__ __launch_bounds__(1024) void bn_stats_stage1_kernel(const float* __restrict__ partial, int64_t R, int C,
                                                                int64_t rows_per_split, double* __restrict__ tmp) {
  __shared__ double sh[2][16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool on = c < C;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = (r0 + rows_per_split < R) ? r0 + rows_per_split : R;
  double n = 0.0, a = 0.0;
  if (on)
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      const float* row = partial + r * 3 * C;
      const double nt = (double)row[2 * C + c];
      n += nt; a += nt * (double)row[c];
    }
  sh[0][rl][cl] = n; sh[1][rl][cl] = a;
  __syncthreads();
  double N = 0.0, A = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { N += sh[0][i][cl]; A += sh[1][i][cl]; }
  const double mean = N > 0.0 ? A / N : 0.0;
  double q = 0.0;
  if (on)
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      const float* row = partial + r * 3 * C;
      const double d = (double)row[c] - mean;
      q += (double)row[C + c] + (double)row[2 * C + c] * d * d;
    }
  __syncthreads();
  sh[0][rl][cl] = q;
  __syncthreads();
  if (rl == 0 && on) {
    double Q = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) Q += sh[0][i][cl];
    double* o = tmp + (int64_t)blockIdx.y * 3 * C;
    o[c] = N; o[C + c] = mean; o[2 * C + c] = Q;
  }
}

// one channel's batch moments -> mean_invstd[2][C], scale_shift[2][C] (scale = gamma*invstd, shift = beta - mean*scale),
//   running stats (unbiased variance, as torch).
__device__ __forceinline__ void bn_finalize_channel(const Moments& m, int c, int C, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, float* __restrict__ running_mean,
                                                    float* __restrict__ running_var, float momentum, float eps, int centered,
                                                    float* __restrict__ mean_invstd, float* __restrict__ scale_shift) {
  const double mean = m.mean;
  const double var = m.n > 0.0 ? m.m2 / m.n : 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  mean_invstd[c] = (float)mean;
  mean_invstd[C + c] = invstd;
  const float sc = gamma[c] * invstd;
  scale_shift[c] = sc;
  scale_shift[C + c] = beta[c] - (float)mean * sc;
  if (running_mean) {
    const double unbiased = m.n > 1.0 ? m.m2 / (m.n - 1.0) : var;
    // centered: the statistics describe z - running_mean(old); the true batch mean adds it back
    const float true_mean = (float)mean + (centered ? running_mean[c] : 0.f);
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * true_mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
  }
}

// stage 2: tmp[S][3][C] -> one wave per channel merges the splits, lane 0 finishes (bn_finalize_channel).
__global__ __launch_bounds__(256) void bn_finalize_kernel(const double* __restrict__ tmp, int S, int C, double count,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float* __restrict__ running_mean, float* __restrict__ running_var,
                                                           float momentum, float eps, int centered,
                                                           float* __restrict__ mean_invstd, float* __restrict__ scale_shift,
                                                           long long* __restrict__ num_batches_tracked) {
  // one wave per channel: lane i holds split-row i (S <= 64); the xor butterfly applies the (symmetric) merge, so every
  // lane ends with the same, order-fixed result
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  // BatchNorm2d's `num_batches_tracked += 1` (torch does it in the module's forward; 18 one-element add launches per step)
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  if (c >= C) return;
  Moments m{0.0, 0.0, 0.0};
  if (lane < S) { const double* row = tmp + (size_t)lane * 3 * C; m = Moments{row[c], row[C + c], row[2 * C + c]}; }
  for (int off = 32; off > 0; off >>= 1) {
    const Moments o{__shfl_xor(m.n, off, 64), __shfl_xor(m.mean, off, 64), __shfl_xor(m.m2, off, 64)};
    m = merge_moments(m, o);
  }
  if (lane != 0) return;
  (void)count;                                             // == m.n (kept in the ABI for the caller's bookkeeping)
  bn_finalize_channel(m, c, C, gamma, beta, running_mean, running_var, momentum, eps, centered, mean_invstd, scale_shift);
}

// [r6] stage 1 and stage 2 in ONE launch at every size.  Grid and stage-1 arithmetic of bn_stats_stage1_kernel; a block that has
// written its split row takes a ticket from its channel group's counter (device-scope atomic behind a release fence), and the
// block that draws the last ticket -- every other split row of the group is then visible to it -- merges the S rows with
// bn_finalize_kernel's butterfly (one wave per channel, lane i = split row i: the same order, the same bits) and puts the counter
// back to zero for the next launch on this stream.  counters: >= 16 zero-initialised ints owned by the caller, used by one stream
// at a time.  One ~5 us launch and one dependent-launch gap less per BatchNorm layer and pass.
__device__ __forceinline__ bool last_block_of_group(int32_t* counter, int nblocks) {
  __shared__ int s_last;
  __threadfence();                                          // this block's tmp row is visible device-wide before the ticket
  __syncthreads();
  if (threadIdx.x == 0) s_last = (atomicAdd(counter, 1) == nblocks - 1);
  __syncthreads();
  if (s_last) __threadfence();                              // ... and the other blocks' rows to this one after it
  return s_last != 0;
}

__global__ __launch_bounds__(1024) void bn_stats_onelaunch_kernel(const float* __restrict__ partial, int64_t R, int C, int64_t rows_per_split,
                                                                   double* __restrict__ tmp, int32_t* __restrict__ counters,
                                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                                                   float momentum, float eps, int centered,
                                                                   float* __restrict__ mean_invstd, float* __restrict__ scale_shift,
                                                                   long long* __restrict__ num_batches_tracked) {
  __shared__ double sh[2][16][64];
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  const bool on = c < C;
  const int S = gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = (r0 + rows_per_split < R) ? r0 + rows_per_split : R;
  double n = 0.0, a = 0.0;
  if (on)
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      const float* row = partial + r * 3 * C;
      const double nt = (double)row[2 * C + c];
      n += nt; a += nt * (double)row[c];
    }
  sh[0][rl][cl] = n; sh[1][rl][cl] = a;
  __syncthreads();
  double N = 0.0, A = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { N += sh[0][i][cl]; A += sh[1][i][cl]; }
  const double mean = N > 0.0 ? A / N : 0.0;
  double q = 0.0;
  if (on)
    for (int64_t r = r0 + rl; r < r1; r += 16) {
      const float* row = partial + r * 3 * C;
      const double d = (double)row[c] - mean;
      q += (double)row[C + c] + (double)row[2 * C + c] * d * d;
    }
  __syncthreads();
  sh[0][rl][cl] = q;
  __syncthreads();
  if (rl == 0 && on) {
    double Q = 0.0;
#pragma unroll
    for (int i = 0; i < 16; ++i) Q += sh[0][i][cl];
    double* o = tmp + (int64_t)blockIdx.y * 3 * C;
    o[c] = N; o[C + c] = mean; o[2 * C + c] = Q;
  }
  if (!last_block_of_group(counters + blockIdx.x, S)) return;
  // stage 2 for this group's 64 channels: 16 waves x 4 channels, bn_finalize_kernel's merge
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const volatile double* vt = tmp;                          // written by other blocks during this launch
#pragma unroll 1
  for (int j = 0; j < 4; ++j) {
    const int cc = blockIdx.x * 64 + wv * 4 + j;
    if (cc >= C) break;
    Moments m{0.0, 0.0, 0.0};
    if (lane < S) { const volatile double* row = vt + (size_t)lane * 3 * C; m = Moments{row[cc], row[C + cc], row[2 * C + cc]}; }
    for (int off = 32; off > 0; off >>= 1) {
      const Moments o{__shfl_xor(m.n, off, 64), __shfl_xor(m.mean, off, 64), __shfl_xor(m.m2, off, 64)};
      m = merge_moments(m, o);
    }
    if (lane == 0)
      bn_finalize_channel(m, cc, C, gamma, beta, running_mean, running_var, momentum, eps, centered, mean_invstd, scale_shift);
  }
  if (threadIdx.x == 0) counters[blockIdx.x] = 0;
}

// [r4] few partial rows (R <= BN_FUSED_MAX_ROWS = 256: every layer of the 32x32 config, the deep levels of a small batch): stage 1 over
// ALL rows and stage 2 in one launch, block = 16 channels x 64 row lanes -- one ~5 us launch less per BatchNorm layer where the
// step is a chain of such launches.
constexpr int BN_FUSED_MAX_ROWS = 256;
__global__ __launch_bounds__(1024) void bn_stats_fused_kernel(const float* __restrict__ partial, int64_t R, int C,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               float momentum, float eps, int centered,
                                                               float* __restrict__ mean_invstd, float* __restrict__ scale_shift,
                                                               long long* __restrict__ num_batches_tracked) {
  __shared__ double sh[2][64][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const bool on = c < C;
  if (num_batches_tracked && blockIdx.x == 0 && threadIdx.x == 0) *num_batches_tracked += 1;
  double n = 0.0, a = 0.0;
  if (on)
    for (int64_t r = rl; r < R; r += 64) {
      const float* row = partial + r * 3 * C;
      const double nt = (double)row[2 * C + c];
      n += nt; a += nt * (double)row[c];
    }
  sh[0][rl][cl] = n; sh[1][rl][cl] = a;
  __syncthreads();
  double N = 0.0, A = 0.0;
#pragma unroll 8
  for (int i = 0; i < 64; ++i) { N += sh[0][i][cl]; A += sh[1][i][cl]; }
  const double mean = N > 0.0 ? A / N : 0.0;
  double q = 0.0;
  if (on)
    for (int64_t r = rl; r < R; r += 64) {
      const float* row = partial + r * 3 * C;
      const double d = (double)row[c] - mean;
      q += (double)row[C + c] + (double)row[2 * C + c] * d * d;
    }
  __syncthreads();
  sh[0][rl][cl] = q;
  __syncthreads();
  if (rl == 0 && on) {
    double Q = 0.0;
#pragma unroll 8
    for (int i = 0; i < 64; ++i) Q += sh[0][i][cl];
    bn_finalize_channel(Moments{N, mean, Q}, c, C, gamma, beta, running_mean, running_var, momentum, eps, centered, mean_invstd, scale_shift);
  }
}

// eval-mode fold: scale = gamma/sqrt(rv+eps), shift = beta + (conv_bias - rm)*scale
__global__ __launch_bounds__(256) void bn_fold_eval_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const float* __restrict__ rm, const float* __restrict__ rv,
                                                            const float* __restrict__ conv_bias, float eps, int C,
                                                            float* __restrict__ scale_shift) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] / sqrtf(rv[c] + eps);
  scale_shift[c] = sc;
  scale_shift[C + c] = beta[c] + ((conv_bias ? conv_bias[c] : 0.f) - rm[c]) * sc;
}

// a = relu(z*scale + shift), [M][C].  When C/N divides 256 every thread keeps ONE channel vector for the whole
// grid-stride loop (no per-element integer division, scale/shift live in registers).
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_apply_kernel(const T* __restrict__ z, const float* __restrict__ scale_shift,
                                                             T* __restrict__ a, int64_t nvec, int C) {
  constexpr int N = Vec16<T>::N;
  const int vpr = C / N;
  const bool fixed = (256 % vpr) == 0;
  int c0 = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)vpr) * N;
  float sc[N], sh[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { sc[k] = scale_shift[c0 + k]; sh[k] = scale_shift[C + c0 + k]; }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * 256) {
    if (!fixed) {
      c0 = (int)(i % vpr) * N;
#pragma unroll
      for (int k = 0; k < N; ++k) { sc[k] = scale_shift[c0 + k]; sh[k] = scale_shift[C + c0 + k]; }
    }
    float v[N];
    Vec16<T>::load(z + i * N, v);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
    Vec16<T>::store(a + i * N, v);
  }
}

// backward pass 1: per-block partial sums of g = da*[z*scale+shift > 0] and g*xhat.  partial[blk][2][C]
// A thread keeps one channel vector (256 % (C/N) == 0) and strides rows; per-thread sums are combined through
// LDS in a fixed order, so the result is deterministic.
#ifndef IM2IM_STREAM_U
#define IM2IM_STREAM_U 4
#endif
#ifndef IM2IM_NT_LOADS
#define IM2IM_NT_LOADS 1
#endif
// 16-byte load of a tensor the streaming BatchNorm-backward kernels read once per pass
template <typename T> __device__ __forceinline__ uint4 stream_ld(const T* p) {
#if IM2IM_NT_LOADS
  return __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(p)));
#else
  return *reinterpret_cast<const uint4*>(p);
#endif
}
constexpr int STREAM_U = IM2IM_STREAM_U;      // vector loads a thread of the streaming BatchNorm-backward kernels keeps in flight (per operand)

template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_reduce_kernel(const T* __restrict__ da, const T* __restrict__ z,
                                                                  const float* __restrict__ scale_shift,
                                                                  const float* __restrict__ mean_invstd, int64_t M, int C,
                                                                  int64_t rows_per_block, float* __restrict__ partial) {
  constexpr int N = Vec16<T>::N;
  __shared__ float s_part[256][2 * N + 1];
  const int vpr = C / N;                                   // vectors per row (divides 256)
  // grid.y > 1 (GroupNorm backward): one grid row per image, M = rows of ONE image, coefficients per image
  da += (size_t)blockIdx.y * M * C; z += (size_t)blockIdx.y * M * C;
  scale_shift += (size_t)blockIdx.y * 2 * C; mean_invstd += (size_t)blockIdx.y * 2 * C;
  partial += (size_t)blockIdx.y * gridDim.x * 2 * C;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  const int64_t v0 = r0 * vpr, v1 = r1 * vpr;
  const int cv = threadIdx.x % vpr;
  const int c0 = cv * N;
  float sc[N], sh[N], mu[N], is[N], s1[N], s2[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    sc[k] = scale_shift[c0 + k]; sh[k] = scale_shift[C + c0 + k];
    mu[k] = mean_invstd[c0 + k]; is[k] = mean_invstd[C + c0 + k];
    s1[k] = 0.f; s2[k] = 0.f;
  }
  if (threadIdx.x < (256 / vpr) * vpr) {
    auto one = [&](const uint4& rg, const uint4& rz) {
      float g[N], zz[N];
      Vec16<T>::load(reinterpret_cast<const T*>(&rg), g);
      Vec16<T>::load(reinterpret_cast<const T*>(&rz), zz);
#pragma unroll
      for (int k = 0; k < N; ++k) {
        const float gg = (zz[k] * sc[k] + sh[k] > 0.f) ? g[k] : 0.f;
        s1[k] += gg;
        s2[k] += gg * ((zz[k] - mu[k]) * is[k]);
      }
    };
    const int64_t stride = (256 / vpr) * vpr;
    int64_t i = v0 + threadIdx.x;
    for (; i + (STREAM_U - 1) * stride < v1; i += STREAM_U * stride) {      // several pairs in flight, summed in the same order
      uint4 rg[STREAM_U], rz[STREAM_U];
#pragma unroll
      for (int u = 0; u < STREAM_U; ++u) {
        rg[u] = stream_ld(da + (i + u * stride) * N);
        rz[u] = stream_ld(z + (i + u * stride) * N);
      }
#pragma unroll
      for (int u = 0; u < STREAM_U; ++u) one(rg[u], rz[u]);
    }
    for (; i < v1; i += stride) one(stream_ld(da + i * N), stream_ld(z + i * N));
  }
#pragma unroll
  for (int k = 0; k < N; ++k) { s_part[threadIdx.x][k] = s1[k]; s_part[threadIdx.x][N + k] = s2[k]; }
  __syncthreads();
  const int groups = 256 / vpr;
  for (int j = threadIdx.x; j < 2 * C; j += 256) {
    const int which = j / C, c = j % C;
    float acc = 0.f;
    for (int g = 0; g < groups; ++g) acc += s_part[g * vpr + c / N][which * N + c % N];
    partial[(size_t)blockIdx.x * 2 * C + j] = acc;
  }
}

// stage 2: dgamma = sum g*xhat, dbeta = sum g; coef[3][C] = {scale, dbeta/M, dgamma/M}
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const double* __restrict__ tmp, int S, int C, double count,
                                                               float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               float* __restrict__ coef) {
  const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (c >= C) return;
  double s1 = (lane < S) ? tmp[((size_t)lane * 2 + 0) * C + c] : 0.0;
  double s2 = (lane < S) ? tmp[((size_t)lane * 2 + 1) * C + c] : 0.0;
  for (int off = 32; off > 0; off >>= 1) { s1 += __shfl_xor(s1, off, 64); s2 += __shfl_xor(s2, off, 64); }
  if (lane != 0) return;
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  coef[c] = (float)(s1 / count);
  coef[C + c] = (float)(s2 / count);
}

// [r6] reduce_rows_stage1_kernel + bn_bwd_finalize_kernel in ONE launch at every size (see bn_stats_onelaunch_kernel): grid (ceil(2C / 64), S);
// the last block of a column group sums the S rows of its 64 columns with bn_bwd_finalize_kernel's butterfly (same bits).  Column k of
// [2][C]: k < C = sum g -> dbeta, coef[k]; k >= C = sum g*xhat -> dgamma, coef[k].
__global__ __launch_bounds__(256) void bn_bwd_sums_onelaunch_kernel(const float* __restrict__ partial, int64_t R, int C, int64_t rows_per_split,
                                                                     double* __restrict__ tmp, int32_t* __restrict__ counters, double count,
                                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ coef) {
  __shared__ double s_acc[4][64];
  const int64_t K = 2 * (int64_t)C;
  const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t k = (int64_t)blockIdx.x * 64 + cl;
  const int S = gridDim.y;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = (r0 + rows_per_split < R) ? r0 + rows_per_split : R;
  double acc = 0.0;
  if (k < K)
    for (int64_t r = r0 + rl; r < r1; r += 4) acc += (double)partial[r * K + k];
  s_acc[rl][cl] = acc;
  __syncthreads();
  if (rl == 0 && k < K) tmp[(int64_t)blockIdx.y * K + k] = s_acc[0][cl] + s_acc[1][cl] + s_acc[2][cl] + s_acc[3][cl];
  if (!last_block_of_group(counters + blockIdx.x, S)) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const volatile double* vt = tmp;
#pragma unroll 1
  for (int j = 0; j < 16; ++j) {
    const int64_t kk = (int64_t)blockIdx.x * 64 + wv * 16 + j;
    if (kk >= K) break;
    double v = (lane < S) ? vt[(size_t)lane * K + kk] : 0.0;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    if (lane == 0) {
      if (kk < C) dbeta[kk] = (float)v; else dgamma[kk - C] = (float)v;
      coef[kk] = (float)(v / count);
    }
  }
  if (threadIdx.x == 0) counters[blockIdx.x] = 0;
}

// [r4] the same for few partial rows (R <= BN_FUSED_MAX_ROWS): both stages in one launch, block = 16 channels x 64 row lanes
__global__ __launch_bounds__(1024) void bn_bwd_sums_fused_kernel(const float* __restrict__ partial, int64_t R, int C, double count,
                                                                  float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                  float* __restrict__ coef) {
  __shared__ double sh[2][64][16];
  const int cl = threadIdx.x & 15, rl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double a1 = 0.0, a2 = 0.0;
  if (c < C)
    for (int64_t r = rl; r < R; r += 64) { a1 += (double)partial[r * 2 * C + c]; a2 += (double)partial[r * 2 * C + C + c]; }
  sh[0][rl][cl] = a1; sh[1][rl][cl] = a2;
  __syncthreads();
  if (rl != 0 || c >= C) return;
  double s1 = 0.0, s2 = 0.0;
#pragma unroll 8
  for (int i = 0; i < 64; ++i) { s1 += sh[0][i][cl]; s2 += sh[1][i][cl]; }
  dbeta[c] = (float)s1;
  dgamma[c] = (float)s2;
  coef[c] = (float)(s1 / count);
  coef[C + c] = (float)(s2 / count);
}

// backward pass 2: dz = scale * (g - dbeta/M - xhat * dgamma/M); per-thread channel vector as in bn_relu_apply
template <typename T>
__global__ __launch_bounds__(256) void bn_relu_bwd_apply_kernel(const T* __restrict__ da, const T* __restrict__ z,
                                                                 const float* __restrict__ scale_shift,
                                                                 const float* __restrict__ mean_invstd,
                                                                 const float* __restrict__ coef, T* __restrict__ dz,
                                                                 int64_t nvec, int C, int keep) {
  constexpr int N = Vec16<T>::N;
  const int vpr = C / N;
  const bool fixed = (256 % vpr) == 0;
  int c0 = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)vpr) * N;
  float sc[N], sh[N], mu[N], is[N], k1[N], k2[N];
  auto load_coef = [&]() {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      sc[k] = scale_shift[c0 + k]; sh[k] = scale_shift[C + c0 + k];
      mu[k] = mean_invstd[c0 + k]; is[k] = mean_invstd[C + c0 + k];
      k1[k] = coef[c0 + k]; k2[k] = coef[C + c0 + k];
    }
  };
  load_coef();
  auto one = [&](const uint4& rg, const uint4& rz, int64_t i) {
    float g[N], zz[N], o[N];
    Vec16<T>::load(reinterpret_cast<const T*>(&rg), g);
    Vec16<T>::load(reinterpret_cast<const T*>(&rz), zz);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float gg = (zz[k] * sc[k] + sh[k] > 0.f) ? g[k] : 0.f;
      const float xhat = (zz[k] - mu[k]) * is[k];
      o[k] = sc[k] * (gg - k1[k] - xhat * k2[k]);
    }
    // keep [r5, A/B]: a dz small enough for the Infinity Cache is written with ordinary stores so that its two consumers (the
    // data-gradient and the weight gradient of the same layer, next in line) may find it there; large ones are streamed
    if (keep) Vec16<T>::store(dz + i * N, o); else Vec16<T>::store_nt(dz + i * N, o);
  };
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (fixed) {
    // STREAM_U vector pairs in flight per thread: beside the weight-gradient kernels of the second stream this kernel gets
    // only a few wave slots per CU, and what it then moves is (bytes in flight per wave) / latency -- one pair per
    // iteration ran 2.3x slower there than alone, although alone (full occupancy) it was at the HBM rate either way
    for (; i + (STREAM_U - 1) * stride < nvec; i += STREAM_U * stride) {
      uint4 rg[STREAM_U], rz[STREAM_U];
#pragma unroll
      for (int u = 0; u < STREAM_U; ++u) {
        rg[u] = stream_ld(da + (i + u * stride) * N);
        rz[u] = stream_ld(z + (i + u * stride) * N);
      }
#pragma unroll
      for (int u = 0; u < STREAM_U; ++u) one(rg[u], rz[u], i + u * stride);
    }
  }
  for (; i < nvec; i += stride) {
    if (!fixed) { c0 = (int)(i % vpr) * N; load_coef(); }
    one(stream_ld(da + i * N), stream_ld(z + i * N), i);
  }
}


// ------------------------------------------------------------------------------------------------
// GroupNorm(+ReLU).  Not in the reference (its DoubleConv uses BatchNorm2d, unet_parts.py:17,20): the
// north-star-named, selectable extra (DoubleConv(norm="group")); oracle = torch.nn.GroupNorm on the CPU.
// Same shape as the BatchNorm path above, with the statistics taken per IMAGE and channel group:
//   * the producing conv's epilogue already emits per-tile per-channel (mean, M2, n) rows, and with one image per tile
//     (im2im_conv_fwd_per_image) the rows of image b are [b*tpi, (b+1)*tpi): gn_finalize merges them (and the group's
//     channels) in fp64 and emits per-(image, channel) scale = gamma*rstd, shift = beta - mean*scale, so consumers apply
//     max(z*scale+shift, 0) exactly like lazy BatchNorm, with coefficients indexed by image;
//   * backward: per-image partial sums of g and g*xhat per channel (the BatchNorm reduce kernel, one grid row per image),
//     a per-(image, group) finalize, and one apply pass  dz = rstd*(gamma*g - mean_g(gamma*g) - xhat*mean_g(gamma*g*xhat)).
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int tpi, int C, int G,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           float eps, float* __restrict__ mean_rstd,
                                                           float* __restrict__ scale_shift) {
  __shared__ double sh[3][256];
  const int g = blockIdx.x, b = blockIdx.y, cpg = C / G;
  Moments m{0.0, 0.0, 0.0};
  for (int e = threadIdx.x; e < tpi * cpg; e += 256) {
    const int r = e / cpg, c = g * cpg + e % cpg;
    const float* row = partial + ((size_t)b * tpi + r) * 3 * C;
    m = merge_moments(m, Moments{(double)row[2 * C + c], (double)row[c], (double)row[C + c]});
  }
  sh[0][threadIdx.x] = m.n; sh[1][threadIdx.x] = m.mean; sh[2][threadIdx.x] = m.m2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {                   // fixed-order tree: deterministic
    if ((int)threadIdx.x < off) {
      const Moments a{sh[0][threadIdx.x], sh[1][threadIdx.x], sh[2][threadIdx.x]};
      const Moments o{sh[0][threadIdx.x + off], sh[1][threadIdx.x + off], sh[2][threadIdx.x + off]};
      const Moments r = merge_moments(a, o);
      sh[0][threadIdx.x] = r.n; sh[1][threadIdx.x] = r.mean; sh[2][threadIdx.x] = r.m2;
    }
    __syncthreads();
  }
  const double mean = sh[1][0];
  const double var = sh[0][0] > 0.0 ? sh[2][0] / sh[0][0] : 0.0;            // biased, as torch.nn.GroupNorm
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  for (int j = threadIdx.x; j < cpg; j += 256) {
    const int c = g * cpg + j;
    float* mr = mean_rstd + (size_t)b * 2 * C;
    float* ss = scale_shift + (size_t)b * 2 * C;
    mr[c] = (float)mean; mr[C + c] = rstd;
    const float sc = gamma[c] * rstd;
    ss[c] = sc; ss[C + c] = beta[c] - (float)mean * sc;
  }
}

// per-channel (mean, M2, n) partial rows of an arbitrary [B][HW][C] tensor, in the conv epilogue's format, for a
// GroupNorm over a tensor this library's conv did not produce.  grid (blocks per image, B); partial[b*nblk + blk][3][C].
template <typename T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const T* __restrict__ z, int64_t HW, int C, int64_t rows_per_block,
                                                        float* __restrict__ partial) {
  constexpr int N = Vec16<T>::N;
  __shared__ float s_part[256][3 * N + 1];
  const int vpr = C / N;
  const int b = blockIdx.y;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < HW) ? r0 + rows_per_block : HW;
  const T* zb = z + (size_t)b * HW * C;
  const int active = (256 / vpr) * vpr;
  float K[N], s[N], q[N], cnt = 0.f;
#pragma unroll
  for (int k = 0; k < N; ++k) { K[k] = 0.f; s[k] = 0.f; q[k] = 0.f; }
  if ((int)threadIdx.x < active) {
    bool first = true;
    for (int64_t i = r0 * vpr + threadIdx.x; i < r1 * vpr; i += active) {
      float v[N];
      Vec16<T>::load(zb + i * N, v);
      if (first) {
#pragma unroll
        for (int k = 0; k < N; ++k) K[k] = v[k];              // shift by a sample: no E[z^2]-E[z]^2 cancellation
        first = false;
      }
#pragma unroll
      for (int k = 0; k < N; ++k) { const float d = v[k] - K[k]; s[k] += d; q[k] += d * d; }
      cnt += 1.f;
    }
  }
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
#pragma unroll
  for (int k = 0; k < N; ++k) {
    s_part[threadIdx.x][k] = K[k] + s[k] * inv;               // mean
    s_part[threadIdx.x][N + k] = fmaxf(q[k] - s[k] * s[k] * inv, 0.f);   // M2
    s_part[threadIdx.x][2 * N + k] = cnt;
  }
  __syncthreads();
  const int groups = 256 / vpr;
  for (int c = threadIdx.x; c < C; c += 256) {
    float n = 0.f, m = 0.f, qq = 0.f;
    for (int gidx = 0; gidx < groups; ++gidx) {
      const float* pr = s_part[gidx * vpr + c / N];
      const float n2 = pr[2 * N + c % N], m2 = pr[c % N], q2 = pr[N + c % N];
      const float nn = n + n2;
      const float iv = nn > 0.f ? 1.f / nn : 0.f;
      const float d = m2 - m;
      qq = qq + q2 + d * d * (n * n2 * iv);
      m = (n * m + n2 * m2) * iv;
      n = nn;
    }
    float* row = partial + ((size_t)b * gridDim.x + blockIdx.x) * 3 * C;
    row[c] = m; row[C + c] = qq; row[2 * C + c] = n;
  }
}

// a = relu(z*scale + shift) with per-image coefficients ss[B][2][C]; grid.y = image
template <typename T>
__global__ __launch_bounds__(256) void affine_relu_apply_img_kernel(const T* __restrict__ z, const float* __restrict__ ss,
                                                                     T* __restrict__ a, int64_t nvec_img, int C) {
  constexpr int N = Vec16<T>::N;
  const int vpr = C / N;
  const int b = blockIdx.y;
  const float* ssb = ss + (size_t)b * 2 * C;
  const T* zb = z + (size_t)b * nvec_img * N;
  T* ab = a + (size_t)b * nvec_img * N;
  const bool fixed = (256 % vpr) == 0;
  int c0 = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)vpr) * N;
  float sc[N], sh[N];
#pragma unroll
  for (int k = 0; k < N; ++k) { sc[k] = ssb[c0 + k]; sh[k] = ssb[C + c0 + k]; }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec_img; i += (int64_t)gridDim.x * 256) {
    if (!fixed) {
      c0 = (int)(i % vpr) * N;
#pragma unroll
      for (int k = 0; k < N; ++k) { sc[k] = ssb[c0 + k]; sh[k] = ssb[C + c0 + k]; }
    }
    float v[N];
    Vec16<T>::load(zb + i * N, v);
#pragma unroll
    for (int k = 0; k < N; ++k) v[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
    Vec16<T>::store(ab + i * N, v);
  }
}

// partial[B][nblk][2][C] (sum g, sum g*xhat per channel) -> per (image, group): k1 = sum_c gamma*S1 / (cpg*HW),
// k2 = sum_c gamma*S2 / (cpg*HW) written per channel into coef[B][2][C]; per-image channel sums into sums[B][2][C]
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const float* __restrict__ partial, int nblk, int C, int G,
                                                               double count, const float* __restrict__ gamma,
                                                               float* __restrict__ coef, float* __restrict__ sums) {
  __shared__ double sh[2][256];
  const int g = blockIdx.x, b = blockIdx.y, cpg = C / G;
  double a1 = 0.0, a2 = 0.0;
  for (int j = threadIdx.x; j < cpg; j += 256) {
    const int c = g * cpg + j;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < nblk; ++k) {
      const float* row = partial + ((size_t)b * nblk + k) * 2 * C;
      s1 += (double)row[c]; s2 += (double)row[C + c];
    }
    sums[(size_t)b * 2 * C + c] = (float)s1;
    sums[(size_t)b * 2 * C + C + c] = (float)s2;
    a1 += (double)gamma[c] * s1; a2 += (double)gamma[c] * s2;
  }
  sh[0][threadIdx.x] = a1; sh[1][threadIdx.x] = a2;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) { sh[0][threadIdx.x] += sh[0][threadIdx.x + off]; sh[1][threadIdx.x] += sh[1][threadIdx.x + off]; }
    __syncthreads();
  }
  const float k1 = (float)(sh[0][0] / count), k2 = (float)(sh[1][0] / count);
  for (int j = threadIdx.x; j < cpg; j += 256) {
    const int c = g * cpg + j;
    coef[(size_t)b * 2 * C + c] = k1;
    coef[(size_t)b * 2 * C + C + c] = k2;
  }
}

// dgamma[c] = sum_b S2[b][c], dbeta[c] = sum_b S1[b][c]
__global__ __launch_bounds__(256) void gn_param_grad_kernel(const float* __restrict__ sums, int B, int C,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int b = 0; b < B; ++b) { s1 += (double)sums[(size_t)b * 2 * C + c]; s2 += (double)sums[(size_t)b * 2 * C + C + c]; }
  dbeta[c] = (float)s1; dgamma[c] = (float)s2;
}

// dz = rstd * (gamma*g - k1 - xhat*k2), g = da*[z*scale+shift > 0]; per-image coefficients, grid.y = image
template <typename T>
__global__ __launch_bounds__(256) void gn_relu_bwd_apply_kernel(const T* __restrict__ da, const T* __restrict__ z,
                                                                 const float* __restrict__ ss, const float* __restrict__ mr,
                                                                 const float* __restrict__ coef, const float* __restrict__ gamma,
                                                                 T* __restrict__ dz, int64_t nvec_img, int C) {
  constexpr int N = Vec16<T>::N;
  const int vpr = C / N;
  const int b = blockIdx.y;
  const size_t boff = (size_t)b * 2 * C, toff = (size_t)b * nvec_img * N;
  const bool fixed = (256 % vpr) == 0;
  int c0 = (int)((blockIdx.x * 256u + threadIdx.x) % (unsigned)vpr) * N;
  float sc[N], sh[N], mu[N], is[N], k1[N], k2[N], gm[N];
  auto load_coef = [&]() {
#pragma unroll
    for (int k = 0; k < N; ++k) {
      sc[k] = ss[boff + c0 + k]; sh[k] = ss[boff + C + c0 + k];
      mu[k] = mr[boff + c0 + k]; is[k] = mr[boff + C + c0 + k];
      k1[k] = coef[boff + c0 + k]; k2[k] = coef[boff + C + c0 + k];
      gm[k] = gamma[c0 + k];
    }
  };
  load_coef();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nvec_img; i += (int64_t)gridDim.x * 256) {
    if (!fixed) { c0 = (int)(i % vpr) * N; load_coef(); }
    float g[N], zz[N], o[N];
    Vec16<T>::load(da + toff + i * N, g);
    Vec16<T>::load(z + toff + i * N, zz);
#pragma unroll
    for (int k = 0; k < N; ++k) {
      const float gg = (zz[k] * sc[k] + sh[k] > 0.f) ? g[k] : 0.f;
      const float xhat = (zz[k] - mu[k]) * is[k];
      o[k] = is[k] * (gm[k] * gg - k1[k] - xhat * k2[k]);
    }
    Vec16<T>::store(dz + toff + i * N, o);
  }
}

// ------------------------------------------------------------------------------------------------
// Small elementwise pieces of the (f)-row layers that used to be torch ops:
//   * head activation: ReLU on the Gaussian layer's variance plane (finallayers/gaussian_layer.py:15-17), abs on the
//     residual-magnitude plane (residual_magnitude_layer.py:15-17).  out [B][K][P] fp32, plane `k` of every image is
//     rewritten in place, its pre-activation values are kept in pre [B][P] for the backward mask / sign.
//   * depth <-> space of the learned upsampling (nn.ConvTranspose2d(k=2, s=2), unet_parts.py:53, as a 1x1 conv to
//     4*Co channels): y4 [B][h][w][(a, b, co)] <-> y [B][2h][2w][co], 16-byte vectors.
__global__ __launch_bounds__(256) void head_act_fwd_kernel(float* __restrict__ out, float* __restrict__ pre, int64_t B, int64_t P,
                                                            int64_t img_stride, int64_t plane_off, int kind) {
  const int64_t total = B * P;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / P, p = i - b * P;
    float* o = out + b * img_stride + plane_off + p;
    const float v = *o;
    pre[i] = v;
    *o = kind == 0 ? (v < 0.f ? 0.f : v) : fabsf(v);          // torch's relu: NaN stays NaN (fmaxf would turn it into 0)
  }
}
__global__ __launch_bounds__(256) void head_act_bwd_kernel(float* __restrict__ dout, const float* __restrict__ pre, int64_t B, int64_t P,
                                                            int64_t img_stride, int64_t plane_off, int kind) {
  const int64_t total = B * P;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t b = i / P, p = i - b * P;
    float* g = dout + b * img_stride + plane_off + p;
    const float v = pre[i];
    // torch: relu' = [v > 0]; abs' = sign(v) (0 at 0)
    const float m = kind == 0 ? (v > 0.f ? 1.f : 0.f) : (v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f));
    *g = *g * m;
  }
}
// to_space != 0: in [B][h][w][4*C] -> out [B][2h][2w][C];  else the inverse
__global__ __launch_bounds__(256) void depth_space2_kernel(const uint4* __restrict__ in, uint4* __restrict__ out, int64_t B, int h, int w,
                                                            int vpc, int to_space) {
  const int64_t total = B * h * w * 4 * vpc;                 // 16-byte vectors
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    // i indexes the depth layout: (b, y, x, a, bb, cv)
    int64_t t = i;
    const int cv = (int)(t % vpc); t /= vpc;
    const int bb = (int)(t % 2); t /= 2;
    const int aa = (int)(t % 2); t /= 2;
    const int x = (int)(t % w); t /= w;
    const int y = (int)(t % h);
    const int64_t b = t / h;
    const int64_t j = (((b * 2 * h + 2 * y + aa) * 2 * w) + 2 * x + bb) * vpc + cv;
    if (to_space) out[j] = in[i]; else out[i] = in[j];
  }
}

// ------------------------------------------------------------------------------------------------
// column sums of an [M][C] tensor (bias gradient of the 1x1 out conv): partial[blk][C]; fixed-order, deterministic
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, int64_t M, int C, int64_t rows_per_block,
                                                              float* __restrict__ partial) {
  constexpr int N = Vec16<T>::N;
  __shared__ float s_part[256][N + 1];
  const int vpr = C / N;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = (r0 + rows_per_block < M) ? r0 + rows_per_block : M;
  float s1[N];
#pragma unroll
  for (int k = 0; k < N; ++k) s1[k] = 0.f;
  if (threadIdx.x < (256 / vpr) * vpr) {
    for (int64_t i = r0 * vpr + threadIdx.x; i < r1 * vpr; i += (256 / vpr) * vpr) {
      float v[N];
      Vec16<T>::load(x + i * N, v);
#pragma unroll
      for (int k = 0; k < N; ++k) s1[k] += v[k];
    }
  }
#pragma unroll
  for (int k = 0; k < N; ++k) s_part[threadIdx.x][k] = s1[k];
  __syncthreads();
  const int groups = 256 / vpr, cvt = threadIdx.x % vpr;
  (void)cvt;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (int g = 0; g < groups; ++g) acc += s_part[g * vpr + c / N][c % N];
    partial[(size_t)blockIdx.x * C + c] = acc;
  }
}

__global__ __launch_bounds__(256) void sum_final_kernel(const double* __restrict__ tmp, int S, int64_t K, float* __restrict__ out) {
  const int64_t k = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  double s = 0.0;
  for (int i = 0; i < S; ++i) s += tmp[(size_t)i * K + k];
  out[k] = (float)s;
}

// lazy BatchNorm+ReLU of a producer whose normalised output was never materialised: v <- max(v*scale+shift, 0)
template <int N>
struct LazySS {                       // N scale and N shift values of one channel vector, fetched with 16-byte loads
  float sc[N], sh[N];
  bool on;
  __device__ __forceinline__ LazySS(const float* __restrict__ ss, int C, int c0) : on(ss != nullptr) {
    if (on) {
#pragma unroll
      for (int k = 0; k < N; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(ss + c0 + k);
        const float4 b = *reinterpret_cast<const float4*>(ss + C + c0 + k);
        sc[k] = a.x; sc[k + 1] = a.y; sc[k + 2] = a.z; sc[k + 3] = a.w;
        sh[k] = b.x; sh[k + 1] = b.y; sh[k + 2] = b.z; sh[k + 3] = b.w;
      }
    }
  }
  __device__ __forceinline__ void apply(float (&v)[N]) const {
    if (on) {
#pragma unroll
      for (int k = 0; k < N; ++k) v[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
    }
  }
};
// values as the consumer sees them after the storage round-trip (bf16 mode: the lazily computed activation is rounded
// exactly like the materialised one would have been)
template <typename T, int N>
__device__ __forceinline__ void round_store_type(float (&v)[N]) {
#pragma unroll
  for (int k = 0; k < N; ++k) v[k] = to_float(from_float<T>(v[k]));
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(2) (unet_parts.py:34), NHWC.  Ties keep the first maximum in (h, w) scan order, as torch.
// Index helper for the NHWC side kernels: blockIdx.y walks image rows (b, y), threads walk the row's 16-byte
// vectors; vector index -> (x, channel vector) by shift/mask when C/N is a power of two (32-bit division otherwise).
struct RowVec {
  int vpr, vshift;
  __device__ __forceinline__ void split(int idx, int& x, int& cv) const {
    if (vshift >= 0) { x = idx >> vshift; cv = idx & (vpr - 1); }
    else { x = (int)((unsigned)idx / (unsigned)vpr); cv = idx - x * vpr; }
  }
};
inline RowVec make_rowvec(int vpr) {
  int sh = -1;
  if ((vpr & (vpr - 1)) == 0) { sh = 0; while ((1 << sh) < vpr) ++sh; }
  return RowVec{vpr, sh};
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, const float* __restrict__ ss,
                                                            T* __restrict__ y, int B, int H, int W, int C, RowVec rv) {
  constexpr int N = Vec16<T>::N;
  const int Ho = H / 2, Wo = W / 2;
  const int rowvecs = Wo * rv.vpr;
  for (int row = blockIdx.y; row < B * Ho; row += gridDim.y) {
    const int b = row / Ho, yo = row - b * Ho;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rowvecs; idx += gridDim.x * 256) {
      int xo, cv;
      rv.split(idx, xo, cv);
      const T* p = x + ((((int64_t)b * H + 2 * yo) * W + 2 * xo) * (int64_t)C) + cv * N;
      float v00[N], v01[N], v10[N], v11[N], o[N];
      Vec16<T>::load(p, v00);
      Vec16<T>::load(p + C, v01);
      Vec16<T>::load(p + (int64_t)W * C, v10);
      Vec16<T>::load(p + (int64_t)W * C + C, v11);
      const LazySS<N> lz(ss, C, cv * N);
      lz.apply(v00); lz.apply(v01); lz.apply(v10); lz.apply(v11);
#pragma unroll
      for (int k = 0; k < N; ++k) o[k] = fmaxf(fmaxf(v00[k], v01[k]), fmaxf(v10[k], v11[k]));
      Vec16<T>::store_nt(y + (((int64_t)row * Wo + xo) * C) + cv * N, o);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ x, const float* __restrict__ ss,
                                                            const T* __restrict__ dy, T* __restrict__ dx, int B, int H, int W,
                                                            int C, RowVec rv) {
  constexpr int N = Vec16<T>::N;
  const int Ho = H / 2, Wo = W / 2;
  // one thread per INPUT 2x2 window position incl. the dropped odd row/col (gets zeros)
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2;
  const int rowvecs = Wc * rv.vpr;
  for (int row = blockIdx.y; row < B * Hc; row += gridDim.y) {
    const int b = row / Hc, yo = row - b * Hc;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rowvecs; idx += gridDim.x * 256) {
      int xo, cv;
      rv.split(idx, xo, cv);
      const int64_t base = ((((int64_t)b * H + 2 * yo) * W + 2 * xo) * (int64_t)C) + cv * N;
      float z[N];
#pragma unroll
      for (int k = 0; k < N; ++k) z[k] = 0.f;
      if (yo < Ho && xo < Wo) {
        float v[4][N], g[N], o[4][N];
        Vec16<T>::load(x + base, v[0]);
        Vec16<T>::load(x + base + C, v[1]);
        Vec16<T>::load(x + base + (int64_t)W * C, v[2]);
        Vec16<T>::load(x + base + (int64_t)W * C + C, v[3]);
        if (ss) {
          const LazySS<N> lz(ss, C, cv * N);
#pragma unroll
          for (int q = 0; q < 4; ++q) { lz.apply(v[q]); round_store_type<T, N>(v[q]); }
        }
        Vec16<T>::load(dy + ((((int64_t)b * Ho + yo) * Wo + xo) * (int64_t)C) + cv * N, g);
#pragma unroll
        for (int k = 0; k < N; ++k) {
          int am = 0; float m = v[0][k];
          if (v[1][k] > m) { m = v[1][k]; am = 1; }
          if (v[2][k] > m) { m = v[2][k]; am = 2; }
          if (v[3][k] > m) { m = v[3][k]; am = 3; }
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q][k] = (am == q) ? g[k] : 0.f;
        }
        Vec16<T>::store_nt(dx + base, o[0]);
        Vec16<T>::store_nt(dx + base + C, o[1]);
        Vec16<T>::store_nt(dx + base + (int64_t)W * C, o[2]);
        Vec16<T>::store_nt(dx + base + (int64_t)W * C + C, o[3]);
      } else {
        // odd tail: positions not covered by any window
        for (int dyy = 0; dyy < 2; ++dyy)
          for (int dxx = 0; dxx < 2; ++dxx) {
            const int yy = 2 * yo + dyy, xx = 2 * xo + dxx;
            if (yy < H && xx < W) Vec16<T>::store_nt(dx + ((((int64_t)b * H + yy) * W + xx) * (int64_t)C) + cv * N, z);
          }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// BatchNorm+ReLU backward of a skip-connection layer, fused with the backward of the MaxPool2d that consumes the same
// activation (unet.py:35-38: x1..x4 feed both Down.maxpool and Up.cat).  The incoming gradient of the activation is
//   g = da (from the Up block's conv, may be null) + scatter_to_argmax(dpool)
// and is never written to HBM: each thread owns one 2x2 window x one channel vector, recomputes the window's
// activations from z (lazy BatchNorm+ReLU, rounded to the storage type exactly as maxpool2_fwd saw them), finds the
// first maximum (torch's tie rule), forms g in the storage type's rounding (what the separate add kernel would have
// stored) and goes straight on with the BatchNorm backward.  APPLY = false: partial sums of g*mask and g*mask*xhat
// (partial[block][2][C], fixed-order LDS combine => deterministic); APPLY = true: dz.
// Windows include the odd last row/column (those pixels are not pooled and receive da only).
// FULL [r5]: H and W even and da present (every skip layer of the UNet at its benchmarked sizes): all four pixels of every window
// exist and are pooled, so the loads, the arg-max and the stores are one straight line -- no exec branches around the nine loads
// (with them the compiler cannot count its vmcnt waits and drains the queue at every merge point).  Same arithmetic.
template <typename T, bool APPLY, bool FULL>
__global__ __launch_bounds__(256) void bn_relu_pool_bwd_kernel(const T* __restrict__ da, const T* __restrict__ dpool,
                                                                const T* __restrict__ z, const float* __restrict__ scale_shift,
                                                                const float* __restrict__ mean_invstd,
                                                                const float* __restrict__ coef, T* __restrict__ dz,
                                                                float* __restrict__ partial, int B, int H, int W, int C,
                                                                RowVec rv) {
  constexpr int N = Vec16<T>::N;
  __shared__ float s_part[APPLY ? 1 : 256][2 * N + 1];
  const int Ho = H / 2, Wo = W / 2, Hc = (H + 1) / 2, Wc = (W + 1) / 2;
  const int vpr = rv.vpr;                                  // power of two, divides 256: a thread keeps one channel vector
  const int cv = threadIdx.x & (vpr - 1);
  const int c0 = cv * N;
  float sc[N], sh[N], mu[N], is[N], k1[N], k2[N], s1[N], s2[N];
#pragma unroll
  for (int k = 0; k < N; ++k) {
    sc[k] = scale_shift[c0 + k]; sh[k] = scale_shift[C + c0 + k];
    mu[k] = mean_invstd[c0 + k]; is[k] = mean_invstd[C + c0 + k];
    k1[k] = APPLY ? coef[c0 + k] : 0.f; k2[k] = APPLY ? coef[C + c0 + k] : 0.f;
    s1[k] = 0.f; s2[k] = 0.f;
  }
  const int rowvecs = Wc * vpr;
  for (int row = blockIdx.y; row < B * Hc; row += gridDim.y) {
    const int b = row / Hc, yo = row - b * Hc;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rowvecs; idx += gridDim.x * 256) {
      const int xo = idx >> rv.vshift;
      const bool pooled = FULL || (yo < Ho && xo < Wo);
      float zz[4][N], g[4][N];
      bool ok[4];
      int64_t off[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int yy = 2 * yo + (q >> 1), xx = 2 * xo + (q & 1);
        ok[q] = FULL || (yy < H && xx < W);
        off[q] = ((((int64_t)b * H + yy) * W + xx) * (int64_t)C) + c0;
        if constexpr (FULL) {
          Vec16<T>::load(z + off[q], zz[q]);
          Vec16<T>::load(da + off[q], g[q]);
        } else {
          if (ok[q]) {
            Vec16<T>::load(z + off[q], zz[q]);
            if (da) Vec16<T>::load(da + off[q], g[q]);
          }
          if (!ok[q] || !da) {
#pragma unroll
            for (int k = 0; k < N; ++k) g[q][k] = 0.f;
          }
        }
      }
      if (pooled) {
        float gp[N], a[4][N];
        Vec16<T>::load(dpool + ((((int64_t)b * Ho + yo) * Wo + xo) * (int64_t)C) + c0, gp);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
          for (int k = 0; k < N; ++k) a[q][k] = fmaxf(zz[q][k] * sc[k] + sh[k], 0.f);
          round_store_type<T, N>(a[q]);
        }
#pragma unroll
        for (int k = 0; k < N; ++k) {
          int am = 0; float m = a[0][k];
          if (a[1][k] > m) { m = a[1][k]; am = 1; }
          if (a[2][k] > m) { m = a[2][k]; am = 2; }
          if (a[3][k] > m) { m = a[3][k]; am = 3; }
#pragma unroll
          for (int q = 0; q < 4; ++q) g[q][k] += (am == q) ? gp[k] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) round_store_type<T, N>(g[q]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (!ok[q]) continue;
        float o[N];
#pragma unroll
        for (int k = 0; k < N; ++k) {
          const float gg = (zz[q][k] * sc[k] + sh[k] > 0.f) ? g[q][k] : 0.f;
          const float xhat = (zz[q][k] - mu[k]) * is[k];
          if constexpr (APPLY) o[k] = sc[k] * (gg - k1[k] - xhat * k2[k]);
          else { s1[k] += gg; s2[k] += gg * xhat; }
        }
        if constexpr (APPLY) Vec16<T>::store_nt(dz + off[q], o);
      }
    }
  }
  if constexpr (!APPLY) {
#pragma unroll
    for (int k = 0; k < N; ++k) { s_part[threadIdx.x][k] = s1[k]; s_part[threadIdx.x][N + k] = s2[k]; }
    __syncthreads();
    const int groups = 256 / vpr;
    const size_t blk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int j = threadIdx.x; j < 2 * C; j += 256) {
      const int which = j / C, c = j % C;
      float acc = 0.f;
      for (int gi = 0; gi < groups; ++gi) acc += s_part[gi * vpr + c / N][which * N + c % N];
      partial[blk * 2 * C + j] = acc;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Up block input (unet_parts.py:58-68): cat([skip, pad(upsample_bilinear_x2_align_corners(deep))], C)
//   deep [B][h][w][Cd], skip [B][H][W][Cs] -> out [B][H][W][Cs+Cd]; pad offsets (H-2h)/2, (W-2w)/2.
__device__ __forceinline__ void bilinear_src(int dst, int in_size, int out_size, int& i0, int& i1, float& l0, float& l1) {
  // torch area_pixel_compute_source_index, align_corners=True: src = dst * (in-1)/(out-1)
  const float scale = out_size > 1 ? (float)(in_size - 1) / (float)(out_size - 1) : 0.f;
  const float src = scale * (float)dst;
  i0 = (int)src;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = src - (float)i0;
  l0 = 1.f - l1;
}

template <typename T>
IM2IM_NO_PACKED_FP32 __global__ __launch_bounds__(256) void upcat_fwd_kernel(const T* __restrict__ deep, const float* __restrict__ deep_ss,
                                                         const T* __restrict__ skip, const float* __restrict__ skip_ss,
                                                         T* __restrict__ out, int B, int h, int w, int Cd, int H, int W,
                                                         int Cs, RowVec rvs, RowVec rvd) {
  constexpr int N = Vec16<T>::N;
  const int Ct = Cs + Cd;
  const int py = (H - 2 * h) / 2, px = (W - 2 * w) / 2;
  // blockIdx.z == 0: the skip half (copy / lazy activation); == 1: the upsampled half.  Waves are uniform in work.
  // Cs == 0 (no skip half, grid.z == 1): every block upsamples.
  if (blockIdx.z == 0 && Cs > 0) {
    const int rowvecs = W * rvs.vpr;
    for (int row = blockIdx.y; row < B * H; row += gridDim.y) {
      for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rowvecs; idx += gridDim.x * 256) {
        int x, cv;
        rvs.split(idx, x, cv);
        const int64_t pix = (int64_t)row * W + x;
        const T* sp = skip + pix * Cs + cv * N;
        T* op = out + pix * Ct + cv * N;
        if (skip_ss) {
          float v[N];
          Vec16<T>::load(sp, v);
          const LazySS<N> lz(skip_ss, Cs, cv * N);
          lz.apply(v);
          Vec16<T>::store_nt(op, v);
        } else {
          *reinterpret_cast<uint4*>(op) = *reinterpret_cast<const uint4*>(sp);
        }
      }
    }
    return;
  }
  const int rowvecs = W * rvd.vpr;
  for (int row = blockIdx.y; row < B * H; row += gridDim.y) {
    const int b = row / H, y = row - b * H;
    const int uy = y - py;
    const bool row_in = uy >= 0 && uy < 2 * h;
    int y0 = 0, y1 = 0; float ly0 = 0.f, ly1 = 0.f;
    if (row_in) bilinear_src(uy, h, 2 * h, y0, y1, ly0, ly1);
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rowvecs; idx += gridDim.x * 256) {
      int x, cv;
      rvd.split(idx, x, cv);
      T* op = out + ((int64_t)row * W + x) * Ct + Cs + cv * N;
      float o[N];
      const int ux = x - px;
      if (!row_in || ux < 0 || ux >= 2 * w) {
#pragma unroll
        for (int k = 0; k < N; ++k) o[k] = 0.f;
      } else {
        int x0, x1; float lx0, lx1;
        bilinear_src(ux, w, 2 * w, x0, x1, lx0, lx1);
        const int c = cv * N;
        float v00[N], v01[N], v10[N], v11[N];
        Vec16<T>::load(deep + ((((int64_t)b * h + y0) * w + x0) * (int64_t)Cd) + c, v00);
        Vec16<T>::load(deep + ((((int64_t)b * h + y0) * w + x1) * (int64_t)Cd) + c, v01);
        Vec16<T>::load(deep + ((((int64_t)b * h + y1) * w + x0) * (int64_t)Cd) + c, v10);
        Vec16<T>::load(deep + ((((int64_t)b * h + y1) * w + x1) * (int64_t)Cd) + c, v11);
        if (deep_ss) {
          const LazySS<N> lz(deep_ss, Cd, c);
          lz.apply(v00); lz.apply(v01); lz.apply(v10); lz.apply(v11);
          round_store_type<T, N>(v00); round_store_type<T, N>(v01); round_store_type<T, N>(v10); round_store_type<T, N>(v11);
        }
#pragma unroll
        for (int k = 0; k < N; ++k) o[k] = ly0 * (lx0 * v00[k] + lx1 * v01[k]) + ly1 * (lx0 * v10[k] + lx1 * v11[k]);
      }
      Vec16<T>::store_nt(op, o);
    }
  }
}

// backward: dskip = dout[..., :Cs] (copy), ddeep gathered (deterministic, no atomics).
// grid.y walks the H skip rows then the h deep rows of every image.
template <typename T>
__global__ __launch_bounds__(256) void upcat_bwd_kernel(const T* __restrict__ dout, T* __restrict__ ddeep,
                                                         T* __restrict__ dskip, int B, int h, int w, int Cd, int H, int W,
                                                         int Cs, RowVec rvs, RowVec rvd, int deep_rows) {
  constexpr int N = Vec16<T>::N;
  const int Ct = Cs + Cd;
  const int py = (H - 2 * h) / 2, px = (W - 2 * w) / 2;
  const int rows_skip = Cs > 0 ? B * H : 0, rows_deep = deep_rows ? B * h : 0;     // deep_rows == 0: up2x_bwd_tiled_kernel does ddeep
  for (int row = blockIdx.y; row < rows_skip + rows_deep; row += gridDim.y) {
    if (row < rows_skip) {
      const int rowvecs = W * rvs.vpr;
      for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rowvecs; idx += gridDim.x * 256) {
        int x, cv;
        rvs.split(idx, x, cv);
        const int64_t pix = (int64_t)row * W + x;
        *reinterpret_cast<uint4*>(dskip + pix * Cs + cv * N) = *reinterpret_cast<const uint4*>(dout + pix * Ct + cv * N);
      }
      continue;
    }
    const int r = row - rows_skip;
    const int b = r / h, yi = r - b * h;
    // output rows whose source index pair can contain yi: uy in [2*yi-2, 2*yi+3]
    float wy[6]; int ys[6];
#pragma unroll
    for (int t = 0; t < 6; ++t) {
      const int uy = 2 * yi - 2 + t;
      wy[t] = 0.f; ys[t] = -1;
      if (uy >= 0 && uy <= 2 * h - 1) {
        int y0, y1; float ly0, ly1;
        bilinear_src(uy, h, 2 * h, y0, y1, ly0, ly1);
        wy[t] = (y0 == yi ? ly0 : 0.f) + (y1 == yi ? ly1 : 0.f);
        const int y = uy + py;
        if (wy[t] != 0.f && y >= 0 && y < H) ys[t] = y;
      }
    }
    const int rowvecs = w * rvd.vpr;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < rowvecs; idx += gridDim.x * 256) {
      int xi, cv;
      rvd.split(idx, xi, cv);
      float acc[N];
#pragma unroll
      for (int k = 0; k < N; ++k) acc[k] = 0.f;
#pragma unroll
      for (int u = 0; u < 6; ++u) {
        const int ux = 2 * xi - 2 + u;
        if (ux < 0 || ux > 2 * w - 1) continue;
        int x0, x1; float lx0, lx1;
        bilinear_src(ux, w, 2 * w, x0, x1, lx0, lx1);
        const float wx = (x0 == xi ? lx0 : 0.f) + (x1 == xi ? lx1 : 0.f);
        const int x = ux + px;
        if (wx == 0.f || x < 0 || x >= W) continue;
#pragma unroll
        for (int t = 0; t < 6; ++t) {
          if (ys[t] < 0) continue;
          float g[N];
          Vec16<T>::load(dout + ((((int64_t)b * H + ys[t]) * W + x) * (int64_t)Ct) + Cs + cv * N, g);
          const float ww = wy[t] * wx;
#pragma unroll
          for (int k = 0; k < N; ++k) acc[k] += ww * g[k];
        }
      }
      Vec16<T>::store_nt(ddeep + (((int64_t)r * w + xi) * (int64_t)Cd) + cv * N, acc);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Tiled forms of the two kernels above for the path the training step takes (no skip half in the output: the consumer
// conv reads skip and upsampled tensor as two operands).  The row-walking kernels spend their time on arithmetic, not on
// HBM: the forward applies the lazy BatchNorm+ReLU to four source vectors per output vector (each source pixel is
// transformed ~4 times over) and the backward evaluates 36 candidate taps with their index arithmetic per output vector
// (1.1 + 1.0 ms per step against a 0.45 + 0.45 ms traffic floor).  Here a workgroup owns a tile x 64 channels:
//   forward : 4 x 32 output pixels; the <= 4 x 18 source pixels they interpolate between are transformed ONCE into LDS
//             (fp32, already rounded to the storage type like the materialised activation), then every output vector is
//             four LDS reads and the same arithmetic as upcat_fwd_kernel -- bit-identical results;
//   backward: 4 x 16 source pixels; separable: first the vertical sums V[yi][ux] = sum_uy wy * dout[uy][ux] over the
//             <= 6 candidate rows into LDS, then the horizontal sums over the <= 6 candidate columns (weights in small
//             LDS tables built once per tile; same candidate sets and weights as upcat_bwd_kernel, different order of
//             the fp32 additions).
#ifndef IM2IM_UPB_TR
#define IM2IM_UPB_TR 2
#endif
#ifndef IM2IM_UPF_TR
#define IM2IM_UPF_TR 8
#endif
#ifndef IM2IM_UPF_TC
#define IM2IM_UPF_TC 16
#endif
constexpr int UPF_TR = IM2IM_UPF_TR, UPF_TC = IM2IM_UPF_TC, UPF_SR = UPF_TR / 2 + 2, UPF_SC = UPF_TC / 2 + 2, UP_CC = 64;
constexpr int UPB_TR = IM2IM_UPB_TR, UPB_TC = 16, UPB_HC = 2 * UPB_TC + 4;       // hi-res columns a tile's sums can touch (ux in [2*x0-2, 2*x0+2*TC+1])

template <typename T>
IM2IM_NO_PACKED_FP32 __global__ __launch_bounds__(256) void up2x_fwd_tiled_kernel(const T* __restrict__ deep, const float* __restrict__ deep_ss,
                                                              T* __restrict__ out, int B, int h, int w, int Cd, int H, int W,
                                                              int Ct, int c_off, int tilesY, int tilesX) {
  // out: [B][H][W][Ct], the upsampled channels at [c_off, c_off + Cd) (Ct = Cd, c_off = 0 without a skip half)
  constexpr int N = Vec16<T>::N;
  constexpr int VPC = UP_CC / N;                      // channel vectors per pixel of the chunk
  constexpr int PITCH = UP_CC + 8;                    // floats per staged source pixel (+32 B: neighbours start 8 banks apart)
  __shared__ __attribute__((aligned(16))) float s_src[UPF_SR * UPF_SC * PITCH];
  __shared__ int s_y0[UPF_TR], s_y1[UPF_TR], s_x0[UPF_TC], s_x1[UPF_TC];
  __shared__ float s_ly[UPF_TR][2], s_lx[UPF_TC][2];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int chunk = t % (Cd / UP_CC); t /= (Cd / UP_CC);
  const int tx_id = t % tilesX; t /= tilesX;
  const int ty_id = t % tilesY;
  const int b = t / tilesY;
  const int yt = ty_id * UPF_TR, xt = tx_id * UPF_TC;
  const int py = (H - 2 * h) / 2, px = (W - 2 * w) / 2;
  const int c_base = chunk * UP_CC;
  // first source row / column any valid output of the tile reads
  const int uy_first = max(yt - py, 0), ux_first = max(xt - px, 0);
  int ylo, xlo, dummy_i; float dummy_f0, dummy_f1;
  bilinear_src(min(uy_first, 2 * h - 1), h, 2 * h, ylo, dummy_i, dummy_f0, dummy_f1);
  bilinear_src(min(ux_first, 2 * w - 1), w, 2 * w, xlo, dummy_i, dummy_f0, dummy_f1);
  if (tid < UPF_TR) {
    const int uy = yt + tid - py;
    int y0 = 0, y1 = 0; float l0 = 0.f, l1 = 0.f;
    const bool in = uy >= 0 && uy < 2 * h && yt + tid < H;
    if (in) bilinear_src(uy, h, 2 * h, y0, y1, l0, l1);
    s_y0[tid] = in ? y0 - ylo : -1; s_y1[tid] = y1 - ylo; s_ly[tid][0] = l0; s_ly[tid][1] = l1;
  } else if (tid >= 64 && tid < 64 + UPF_TC) {
    const int i = tid - 64;
    const int ux = xt + i - px;
    int x0 = 0, x1 = 0; float l0 = 0.f, l1 = 0.f;
    const bool in = ux >= 0 && ux < 2 * w && xt + i < W;
    if (in) bilinear_src(ux, w, 2 * w, x0, x1, l0, l1);
    s_x0[i] = in ? x0 - xlo : -1; s_x1[i] = x1 - xlo; s_lx[i][0] = l0; s_lx[i][1] = l1;
  }
  // stage the source pixels: lazy BatchNorm+ReLU and the rounding to the storage type happen once per element
  {
    const int cv = tid % VPC;
    const LazySS<N> lz(deep_ss, Cd, c_base + cv * N);
#pragma unroll
    for (int i = tid; i < UPF_SR * UPF_SC * VPC; i += 256) {
      const int sp = i / VPC;
      const int sy = ylo + sp / UPF_SC, sx = xlo + sp % UPF_SC;
      float v[N];
      if (sy < h && sx < w) {
        Vec16<T>::load(deep + ((((int64_t)b * h + sy) * w + sx) * (int64_t)Cd) + c_base + cv * N, v);
        if (deep_ss) { lz.apply(v); round_store_type<T, N>(v); }
      } else {
#pragma unroll
        for (int k = 0; k < N; ++k) v[k] = 0.f;
      }
      float* d = s_src + sp * PITCH + cv * N;
#pragma unroll
      for (int k = 0; k < N; k += 4) *reinterpret_cast<float4*>(d + k) = make_float4(v[k], v[k + 1], v[k + 2], v[k + 3]);
    }
  }
  __syncthreads();
  for (int i = tid; i < UPF_TR * UPF_TC * VPC; i += 256) {
    const int cv = i % VPC, p = i / VPC;
    const int r = p / UPF_TC, c = p % UPF_TC;
    const int y = yt + r, x = xt + c;
    if (y >= H || x >= W) continue;
    float o[N];
    const int y0 = s_y0[r], x0 = s_x0[c];
    if (y0 < 0 || x0 < 0) {
#pragma unroll
      for (int k = 0; k < N; ++k) o[k] = 0.f;
    } else {
      const int y1 = s_y1[r], x1 = s_x1[c];
      const float ly0 = s_ly[r][0], ly1 = s_ly[r][1], lx0 = s_lx[c][0], lx1 = s_lx[c][1];
      const float* p00 = s_src + (y0 * UPF_SC + x0) * PITCH + cv * N;
      const float* p01 = s_src + (y0 * UPF_SC + x1) * PITCH + cv * N;
      const float* p10 = s_src + (y1 * UPF_SC + x0) * PITCH + cv * N;
      const float* p11 = s_src + (y1 * UPF_SC + x1) * PITCH + cv * N;
#pragma unroll
      for (int k = 0; k < N; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(p00 + k), bq = *reinterpret_cast<const float4*>(p01 + k);
        const float4 cq = *reinterpret_cast<const float4*>(p10 + k), dq = *reinterpret_cast<const float4*>(p11 + k);
        o[k] = ly0 * (lx0 * a.x + lx1 * bq.x) + ly1 * (lx0 * cq.x + lx1 * dq.x);
        o[k + 1] = ly0 * (lx0 * a.y + lx1 * bq.y) + ly1 * (lx0 * cq.y + lx1 * dq.y);
        o[k + 2] = ly0 * (lx0 * a.z + lx1 * bq.z) + ly1 * (lx0 * cq.z + lx1 * dq.z);
        o[k + 3] = ly0 * (lx0 * a.w + lx1 * bq.w) + ly1 * (lx0 * cq.w + lx1 * dq.w);
      }
    }
    Vec16<T>::store_nt(out + ((((int64_t)b * H + y) * W + x) * (int64_t)Ct) + c_off + c_base + cv * N, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void up2x_bwd_tiled_kernel(const T* __restrict__ dout, T* __restrict__ ddeep, int B, int h, int w,
                                                              int Cd, int H, int W, int Ct, int c_off, int tilesY, int tilesX) {
  constexpr int N = Vec16<T>::N;
  constexpr int VPC = UP_CC / N;
  constexpr int PITCH = UP_CC + 8;
  __shared__ __attribute__((aligned(16))) float s_v[UPB_TR * UPB_HC * PITCH];       // vertical sums, [yi][hi-res column]
  __shared__ float s_wy[UPB_TR][6], s_wx[UPB_TC][6];
  __shared__ int s_ys[UPB_TR][6], s_ny[UPB_TR];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int chunk = t % (Cd / UP_CC); t /= (Cd / UP_CC);
  const int tx_id = t % tilesX; t /= tilesX;
  const int ty_id = t % tilesY;
  const int b = t / tilesY;
  const int y_t = ty_id * UPB_TR, x_t = tx_id * UPB_TC;
  const int py = (H - 2 * h) / 2, px = (W - 2 * w) / 2;
  const int c_base = chunk * UP_CC;
  const int ux_base = 2 * x_t - 2;                     // hi-res column of s_v[.][0]
  // weight tables: candidate t of source index i is hi-res index 2*i-2+t (the candidate set of upcat_bwd_kernel)
  if (tid < UPB_TR) {
    // the (at most 6, normally 4) hi-res rows with weight for source row yi, compacted to the front in ascending order
    const int r = tid, yi = y_t + r;
    int n = 0;
    for (int tt = 0; tt < 6; ++tt) {
      const int uy = 2 * yi - 2 + tt;
      float wv = 0.f; int ys = -1;
      if (yi < h && uy >= 0 && uy <= 2 * h - 1) {
        int y0, y1; float l0, l1;
        bilinear_src(uy, h, 2 * h, y0, y1, l0, l1);
        wv = (y0 == yi ? l0 : 0.f) + (y1 == yi ? l1 : 0.f);
        const int y = uy + py;
        if (wv != 0.f && y >= 0 && y < H) ys = y;
      }
      if (ys >= 0) { s_wy[r][n] = wv; s_ys[r][n] = ys; ++n; }
    }
    s_ny[r] = n;
    for (; n < 6; ++n) { s_wy[r][n] = 0.f; s_ys[r][n] = 0; }
  } else if (tid >= 64 && tid < 64 + UPB_TC * 6) {
    const int i = tid - 64;
    const int c = i / 6, u = i % 6;
    const int xi = x_t + c, ux = 2 * xi - 2 + u;
    float wv = 0.f;
    if (xi < w && ux >= 0 && ux <= 2 * w - 1) {
      int x0, x1; float l0, l1;
      bilinear_src(ux, w, 2 * w, x0, x1, l0, l1);
      wv = (x0 == xi ? l0 : 0.f) + (x1 == xi ? l1 : 0.f);
      const int x = ux + px;
      if (x < 0 || x >= W) wv = 0.f;
    }
    s_wx[c][u] = wv;
  }
  __syncthreads();
  // vertical pass: s_v[r][j] = sum_t wy[r][t] * dout[ys[r][t]][ux_base + j + px]
  for (int i = tid; i < UPB_TR * UPB_HC * VPC; i += 256) {
    const int cv = i % VPC, q = i / VPC;
    const int r = q / UPB_HC, j = q % UPB_HC;
    const int x = ux_base + j + px;
    float acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = 0.f;
    if (x >= 0 && x < W && ux_base + j >= 0 && ux_base + j <= 2 * w - 1) {
      // all six candidate rows are fetched unconditionally (a row without weight reads row 0 and counts for nothing):
      // the loads issue back to back instead of each waiting behind a branch
      // the first four table entries are fetched unconditionally (an entry without weight reads row 0 and counts for
      // nothing): the loads issue back to back instead of each waiting behind a branch.  A fifth / sixth row only has
      // weight where float rounding of the source index puts one (weights ~1e-7).
      uint4 raw[4];
#pragma unroll
      for (int tt = 0; tt < 4; ++tt)
        raw[tt] = *reinterpret_cast<const uint4*>(dout + ((((int64_t)b * H + s_ys[r][tt]) * W + x) * (int64_t)Ct) + c_off + c_base + cv * N);
#pragma unroll
      for (int tt = 0; tt < 4; ++tt) {
        const float wv = s_wy[r][tt];
        float g[N];
        Vec16<T>::load(reinterpret_cast<const T*>(&raw[tt]), g);
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] += wv * g[k];
      }
      for (int tt = 4; tt < s_ny[r]; ++tt) {
        const float wv = s_wy[r][tt];
        float g[N];
        Vec16<T>::load(dout + ((((int64_t)b * H + s_ys[r][tt]) * W + x) * (int64_t)Ct) + c_off + c_base + cv * N, g);
#pragma unroll
        for (int k = 0; k < N; ++k) acc[k] += wv * g[k];
      }
    }
    float* d = s_v + q * PITCH + cv * N;
#pragma unroll
    for (int k = 0; k < N; k += 4) *reinterpret_cast<float4*>(d + k) = make_float4(acc[k], acc[k + 1], acc[k + 2], acc[k + 3]);
  }
  __syncthreads();
  // horizontal pass
  for (int i = tid; i < UPB_TR * UPB_TC * VPC; i += 256) {
    const int cv = i % VPC, q = i / VPC;
    const int r = q / UPB_TC, c = q % UPB_TC;
    const int yi = y_t + r, xi = x_t + c;
    if (yi >= h || xi >= w) continue;
    float acc[N];
#pragma unroll
    for (int k = 0; k < N; ++k) acc[k] = 0.f;
#pragma unroll
    for (int u = 0; u < 6; ++u) {
      const float wv = s_wx[c][u];
      if (wv == 0.f) continue;
      const float* sp = s_v + (r * UPB_HC + 2 * c + u) * PITCH + cv * N;      // hi-res column 2*xi-2+u = ux_base + 2*c + u
#pragma unroll
      for (int k = 0; k < N; k += 4) {
        const float4 a = *reinterpret_cast<const float4*>(sp + k);
        acc[k] += wv * a.x; acc[k + 1] += wv * a.y; acc[k + 2] += wv * a.z; acc[k + 3] += wv * a.w;
      }
    }
    Vec16<T>::store_nt(ddeep + ((((int64_t)b * h + yi) * w + xi) * (int64_t)Cd) + c_base + cv * N, acc);
  }
}

// ------------------------------------------------------------------------------------------------
// fused quantile loss (quantile_layer.py:23-32, pinball.py:12-26): three plane pointers + image stride
// so PinballLoss alone can reuse it.  sums[3] (fp64 via partials) then
//   loss = w_lo * S_lo/P + w_hi * S_hi/P + w_mse * S_mse/P
struct LossArgs {
  const float* lo; const float* mid; const float* hi; const float* y;   // planes a, b, c (hi may be null) and the target
  int64_t N, P, img_stride;      // planes: ptr + n*img_stride + i ; y: n*P + i
  float q_lo, q_hi;
  int kind;                      // IM2IM_LOSS_*
};
__device__ __forceinline__ float pinball_term(float e, float q) {         // losses/pinball.py:17-24
  return (e < 0.f) ? q * fabsf(e) : (e > 0.f ? (1.f - q) * fabsf(e) : 0.f);
}
__device__ __forceinline__ float sign0(float v) { return v > 0.f ? 1.f : (v < 0.f ? -1.f : 0.f); }
constexpr float GNLL_EPS = 1e-6f;                                          // torch.nn.GaussianNLLLoss default

// per-element terms t[0..2] of the three mean-reduced sums (unused ones stay 0)
__device__ __forceinline__ void loss_terms(const LossArgs& a, int64_t n, int64_t p, float y, float (&t)[3]) {
  const float va = a.lo[n * a.img_stride + p];
  const float vb = a.mid[n * a.img_stride + p];
  t[0] = t[1] = t[2] = 0.f;
  switch (a.kind) {
    case IM2IM_LOSS_QUANTILE:
    case IM2IM_LOSS_QUANTILE_L1: {
      const float vc = a.hi[n * a.img_stride + p];
      const float em = vb - y;
      t[0] = pinball_term(va - y, a.q_lo);
      t[1] = pinball_term(vc - y, a.q_hi);
      t[2] = a.kind == IM2IM_LOSS_QUANTILE ? em * em : fabsf(em);
      break;
    }
    case IM2IM_LOSS_GAUSSIAN: {
      const float v = fmaxf(vb, GNLL_EPS), e = va - y;
      t[0] = 0.5f * (logf(v) + e * e / v);
      break;
    }
    case IM2IM_LOSS_INN: {                                                  // a = lower, b = prediction, c = upper; beta = q_lo
      const float vc = a.hi[n * a.img_stride + p];
      const float em = vb - y, over = fmaxf(y - vc, 0.f), under = fmaxf(va - y, 0.f);
      t[0] = em * em;
      t[1] = over * over + under * under + a.q_lo * fabsf(vc - va);
      break;
    }
    default: {                                                              // RESIDUAL, RESIDUAL_L1
      const float e = va - y, r = vb - fabsf(y - va);
      t[0] = a.kind == IM2IM_LOSS_RESIDUAL ? e * e : fabsf(e);
      t[1] = r * r;
    }
  }
}
__global__ __launch_bounds__(256) void qloss_partial_kernel(LossArgs a, float* __restrict__ partial) {
  __shared__ float s_red[3][4];
  const int64_t total = a.N * a.P;
  float s_lo = 0.f, s_hi = 0.f, s_m = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / a.P, p = i - n * a.P;
    float t[3];
    loss_terms(a, n, p, a.y[i], t);
    s_lo += t[0]; s_hi += t[1]; s_m += t[2];
  }
  for (int off = 32; off > 0; off >>= 1) {
    s_lo += __shfl_down(s_lo, off, 64); s_hi += __shfl_down(s_hi, off, 64); s_m += __shfl_down(s_m, off, 64);
  }
  if ((threadIdx.x & 63) == 0) { s_red[0][threadIdx.x >> 6] = s_lo; s_red[1][threadIdx.x >> 6] = s_hi; s_red[2][threadIdx.x >> 6] = s_m; }
  __syncthreads();
  if (threadIdx.x < 3)
    partial[(size_t)blockIdx.x * 3 + threadIdx.x] = s_red[threadIdx.x][0] + s_red[threadIdx.x][1] + s_red[threadIdx.x][2] + s_red[threadIdx.x][3];
}
__global__ void qloss_final_kernel(const double* __restrict__ tmp, int S, double count, float w_lo, float w_hi, float w_mse,
                                   float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s[3] = {0, 0, 0};
  for (int i = 0; i < S; ++i) for (int k = 0; k < 3; ++k) s[k] += tmp[(size_t)i * 3 + k];
  // each term is an fp32 mean in the reference; combine in fp32 in the same order
  const float l0 = (float)(s[0] / count), l1 = (float)(s[1] / count), l2 = (float)(s[2] / count);
  loss[0] = w_lo * l0 + w_hi * l1 + w_mse * l2;
}
// d(pred) planes, fp32 ; gscale points at the upstream scalar gradient on the device
__global__ __launch_bounds__(256) void qloss_bwd_kernel(LossArgs a, const float* __restrict__ gscale, float w_lo, float w_hi,
                                                         float w_mse, float* __restrict__ d_lo, float* __restrict__ d_mid,
                                                         float* __restrict__ d_hi, int64_t d_stride) {
  const int64_t total = a.N * a.P;
  const float g = gscale[0] / (float)total;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t n = i / a.P, p = i - n * a.P;
    const float y = a.y[i];
    const float va = a.lo[n * a.img_stride + p];
    const float vb = a.mid[n * a.img_stride + p];
    float ga = 0.f, gb = 0.f, gc = 0.f;
    switch (a.kind) {
      case IM2IM_LOSS_QUANTILE:
      case IM2IM_LOSS_QUANTILE_L1: {
        const float el = va - y, eh = a.hi[n * a.img_stride + p] - y, em = vb - y;
        ga = w_lo * (el < 0.f ? -a.q_lo : (el > 0.f ? 1.f - a.q_lo : 0.f));
        gc = w_hi * (eh < 0.f ? -a.q_hi : (eh > 0.f ? 1.f - a.q_hi : 0.f));
        gb = w_mse * (a.kind == IM2IM_LOSS_QUANTILE ? 2.f * em : sign0(em));
        break;
      }
      case IM2IM_LOSS_INN: {
        const float vc = a.hi[n * a.img_stride + p];
        const float s = sign0(vc - va);
        ga = w_hi * (2.f * fmaxf(va - y, 0.f) - a.q_lo * s);
        gc = w_hi * (-2.f * fmaxf(y - vc, 0.f) + a.q_lo * s);
        gb = w_lo * (2.f * (vb - y));
        break;
      }
      case IM2IM_LOSS_GAUSSIAN: {
        // d/dmean = (mean - y)/v ; d/dvar = 0.5*(1/v - (mean-y)^2/v^2), v = clamped variance (the clamp itself is applied
        // under no_grad in torch, so its gradient goes to var unchanged)
        const float v = fmaxf(vb, GNLL_EPS), e = va - y;
        ga = w_lo * (e / v);
        gb = w_lo * (0.5f * (1.f / v - e * e / (v * v)));
        break;
      }
      default: {
        // loss = term0(a, y) + (b - |y - a|)^2 : the second term also depends on a through |y - a|
        const float e = va - y, r = vb - fabsf(y - va);
        ga = w_lo * (a.kind == IM2IM_LOSS_RESIDUAL ? 2.f * e : sign0(e)) + w_hi * (2.f * r * sign0(y - va));
        gb = w_hi * (2.f * r);
      }
    }
    if (d_lo) d_lo[n * d_stride + p] = g * ga;
    if (d_mid) d_mid[n * d_stride + p] = g * gb;
    if (d_hi) d_hi[n * d_stride + p] = g * gc;
  }
}

// ------------------------------------------------------------------------------------------------
// Softmax final layer (finallayers/softmax_layer.py): logits [M][stride] T (NHWC pixels x padded class channels, the
// first K valid), one thread per pixel, all arithmetic fp32 in the reference's order (max, exp, sequential sum, divide).
constexpr int SOFTMAX_MAX_K = 64;
template <typename T>
__device__ __forceinline__ void load_logits(const T* __restrict__ row, int K, float (&z)[SOFTMAX_MAX_K]) {
  constexpr int N = Vec16<T>::N;
#pragma unroll
  for (int k = 0; k < SOFTMAX_MAX_K; k += N) {
    float v[N];
    if (k < K) Vec16<T>::load(row + k, v);
#pragma unroll
    for (int j = 0; j < N; ++j) z[k + j] = (k + j < K) ? v[j] : -INFINITY;
  }
}
// torch.bucketize(v, bounds, right=False): first index i with bounds[i] >= v; indices >= K fold to K-1 (:21-22)
__device__ __forceinline__ int bucket_of(float v, const float* __restrict__ bounds, int K) {
  int idx = 0;
  for (int k = 0; k < K; ++k) idx += (bounds[k] < v) ? 1 : 0;
  return idx >= K ? K - 1 : idx;
}
// nn.CrossEntropyLoss (mean over pixels): partial[blk] = sum over the block's pixels of -(z_t - max - log sum exp)
template <typename T>
__global__ __launch_bounds__(256) void softmax_ce_partial_kernel(const T* __restrict__ logits, const float* __restrict__ target,
                                                                  const float* __restrict__ bounds, int64_t M, int K, int stride,
                                                                  float* __restrict__ partial) {
  __shared__ float s_red[4];
  float acc = 0.f;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    float z[SOFTMAX_MAX_K];
    load_logits(logits + m * stride, K, z);
    float mx = z[0];
#pragma unroll
    for (int k = 1; k < SOFTMAX_MAX_K; ++k) mx = fmaxf(mx, z[k]);
    float sum = 0.f, zt = 0.f;
    const int t = bucket_of(target[m], bounds, K);
#pragma unroll
    for (int k = 0; k < SOFTMAX_MAX_K; ++k) {
      if (k < K) { sum += expf(z[k] - mx); if (k == t) zt = z[k]; }
    }
    acc += logf(sum) - (zt - mx);
  }
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = s_red[0] + s_red[1] + s_red[2] + s_red[3];
}
__global__ void softmax_ce_final_kernel(const double* __restrict__ tmp, int S, double count, float* __restrict__ loss) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  double s = 0.0;
  for (int i = 0; i < S; ++i) s += tmp[i];
  loss[0] = (float)(s / count);
}
// d(logits) = g/M * (softmax - onehot); padded class channels get 0
template <typename T>
__global__ __launch_bounds__(256) void softmax_ce_bwd_kernel(const T* __restrict__ logits, const float* __restrict__ target,
                                                              const float* __restrict__ bounds, int64_t M, int K, int stride,
                                                              const float* __restrict__ gscale, T* __restrict__ dlogits) {
  constexpr int N = Vec16<T>::N;
  const float g = gscale[0] / (float)M;
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    float z[SOFTMAX_MAX_K];
    load_logits(logits + m * stride, K, z);
    float mx = z[0];
#pragma unroll
    for (int k = 1; k < SOFTMAX_MAX_K; ++k) mx = fmaxf(mx, z[k]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SOFTMAX_MAX_K; ++k) { z[k] = (k < K) ? expf(z[k] - mx) : 0.f; sum += z[k]; }
    const int t = bucket_of(target[m], bounds, K);
    const float inv = 1.f / sum;
#pragma unroll
    for (int k = 0; k < SOFTMAX_MAX_K; k += N) {
      if (k < stride) {
        float v[N];
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = (k + j < K) ? g * (z[k + j] * inv - ((k + j == t) ? 1.f : 0.f)) : 0.f;
        Vec16<T>::store(dlogits + m * stride + k, v);
      }
    }
  }
}
// softmax_nested_sets_from_output, the lambda-independent part (:33-47): softmax over the classes, running sum,
//   lq = #(cumsum <= 0.05)/K, uq = #(cumsum <= 0.95)/K, pred = argmax/K, the collapse guards and the [0,1] clamp.
// out3 [N][3][P] fp32 planes (lq, pred, uq) of pixel m = n*P + i.
template <typename T>
__global__ __launch_bounds__(256) void softmax_summary_kernel(const T* __restrict__ logits, int64_t M, int64_t P, int K, int stride,
                                                               float* __restrict__ out3) {
  for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
    float z[SOFTMAX_MAX_K];
    load_logits(logits + m * stride, K, z);
    float mx = z[0];
#pragma unroll
    for (int k = 1; k < SOFTMAX_MAX_K; ++k) mx = fmaxf(mx, z[k]);
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < SOFTMAX_MAX_K; ++k) { if (k < K) { z[k] = expf(z[k] - mx); sum += z[k]; } }
    float cum = 0.f, best = -1.f;
    int n_lo = 0, n_hi = 0, arg = 0;
#pragma unroll
    for (int k = 0; k < SOFTMAX_MAX_K; ++k) {
      if (k < K) {
        const float pk = z[k] / sum;
        if (pk > best) { best = pk; arg = k; }               // first maximum, as torch.argmax
        cum += pk;
        n_lo += (cum <= 0.05f) ? 1 : 0;
        n_hi += (cum <= 0.95f) ? 1 : 0;
      }
    }
    const float Kf = (float)K, step = (float)(1.0 / (double)K);
    const float pred = (float)arg / Kf;
    float lq = (float)n_lo / Kf, uq = (float)n_hi / Kf;
    if (pred == lq) lq -= step;
    if (pred == uq) uq += step;
    lq = fminf(fmaxf(lq, 0.f), 1.f);
    uq = fminf(fmaxf(uq, 0.f), 1.f);
    const int64_t n = m / P, i = m - n * P;
    out3[(n * 3 + 0) * P + i] = lq;
    out3[(n * 3 + 1) * P + i] = pred;
    out3[(n * 3 + 2) * P + i] = uq;
  }
}

// ------------------------------------------------------------------------------------------------
// multi-tensor Adam (torch.optim.Adam defaults as used at core/scripts/train.py:120; no weight decay / amsgrad)
constexpr int ADAM_MAX_TENSORS = 24;
struct AdamArgs {
  float* p[ADAM_MAX_TENSORS]; const float* g[ADAM_MAX_TENSORS]; float* m[ADAM_MAX_TENSORS]; float* v[ADAM_MAX_TENSORS];
  int64_t start[ADAM_MAX_TENSORS + 1];     // prefix sums of sizes in units of 1024-element chunks
  int n;
  int64_t size[ADAM_MAX_TENSORS];
  float lr, beta1, beta2, eps, step_size, bc2_sqrt;
  const float* coef;       // null, or device {step_size, bc2_sqrt} written by adam_coef_kernel (step count kept on the device)
};
// the step count of a capturable optimizer lives on the device, so that a HIP graph holding the update replays with the right
// bias correction: ++*step, then step_size = lr / (1 - beta1^step) and sqrt(1 - beta2^step) in float64 as the host path
__global__ void adam_coef_kernel(long long* __restrict__ step, double lr, double beta1, double beta2, float* __restrict__ coef) {
  const long long t = *step + 1;
  *step = t;
  coef[0] = (float)(lr / (1.0 - pow(beta1, (double)t)));
  coef[1] = (float)sqrt(1.0 - pow(beta2, (double)t));
}
__global__ __launch_bounds__(256) void adam_kernel(AdamArgs a) {
  // find the tensor of this 1024-element chunk
  const int64_t chunk = blockIdx.x;
  int t = 0;
  while (t + 1 < a.n && chunk >= a.start[t + 1]) ++t;
  const int64_t off = (chunk - a.start[t]) * 1024;
  float* __restrict__ p = a.p[t]; const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t]; float* __restrict__ v = a.v[t];
  const int64_t n = a.size[t];
  const float step_size = a.coef ? a.coef[0] : a.step_size, bc2_sqrt = a.coef ? a.coef[1] : a.bc2_sqrt;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int64_t i = off + k * 256 + threadIdx.x;
    if (i < n) {
      const float gi = g[i];
      const float mi = m[i] + (1.f - a.beta1) * (gi - m[i]);               // exp_avg.lerp_(grad, 1-beta1)
      const float vi = a.beta2 * v[i] + (1.f - a.beta2) * gi * gi;         // mul_(beta2).addcmul_(g, g, 1-beta2)
      const float denom = sqrtf(vi) / bc2_sqrt + a.eps;
      m[i] = mi; v[i] = vi;
      p[i] = p[i] - step_size * (mi / denom);
    }
  }
}

template <typename F> int for_dtype(int dtype, F f) {
  if (dtype == IM2IM_BF16) return f((bf16_t*)nullptr);
  if (dtype == IM2IM_F32) return f((float*)nullptr);
  return fail_invalid("dtype");
}
inline dim3 row_grid(int rowvecs, int64_t rows) {
  int gx = (int)cdiv(rowvecs, 256);
  if (gx > 64) gx = 64;
  if (gx < 1) gx = 1;
  int64_t gy = rows;
  if (gy > 65535) gy = 65535;
  if (gy < 1) gy = 1;
  return dim3((unsigned)gx, (unsigned)gy);
}
inline int ew_blocks(int64_t n) { int64_t b = cdiv(n, 256); if (b > 256 * 32) b = 256 * 32; if (b < 1) b = 1; return (int)b; }

// BatchNorm-backward sums: partial[R][2][C] -> dgamma, dbeta, coef; one launch for few rows, two stages otherwise
int64_t g_bn_apply_keep_bytes = 0;                           // im2im_set_option("bn_apply_keep_mb", n): dz tensors up to n MB are written with ordinary (cacheable) stores
int g_pool_bwd_full = 1;                                     // im2im_set_option("pool_bwd_full", 0 / 1): branch-free bn_relu_pool_bwd for even extents
int g_bn_fused_small = 1;                                    // im2im_set_option("bn_fused_small", 0 / 1 / n): off / up to BN_FUSED_MAX_ROWS partial rows / up to n rows
inline int64_t bn_fused_rows() { return g_bn_fused_small <= 0 ? -1 : g_bn_fused_small == 1 ? BN_FUSED_MAX_ROWS : g_bn_fused_small; }
// im2im_set_option("bn_onelaunch", 0 / 1): [r6] statistics / backward sums of > bn_fused_small rows in ONE launch (needs `counters`).  DEFAULT OFF:
// measured slower than the two launches it replaces -- batch 78: 39.7 vs 38.3 ms per step, batch 10: 7.07 vs 6.42 (three alternating
// rounds on one box, profiles/r06_ab_experiments.txt section 1).  The ticket needs a device-scope release / acquire, which on this
// eight-XCD part writes back and invalidates the XCD's whole L2 (buffer_wbl2 / buffer_inv sc1) in every one of the ~64 blocks of each
// of the 34 launches: the conv kernels around them (and the weight gradients running beside them) lose their L2 contents.
int g_bn_onelaunch = 0;
inline int launch_bn_bwd_sums(const float* partial, int64_t R, int C, double count, double* tmp, float* dgamma, float* dbeta,
                              float* coef, int32_t* counters, hipStream_t stream) {
  if (R <= bn_fused_rows()) {
    hipLaunchKernelGGL(bn_bwd_sums_fused_kernel, dim3((unsigned)cdiv(C, 16)), dim3(1024), 0, stream, partial, R, C, count, dgamma, dbeta, coef);
    return check_launch("bn_bwd_sums_fused_kernel");
  }
  if (counters && g_bn_onelaunch && cdiv(2 * (int64_t)C, 64) <= IM2IM_BN_COUNTERS) {
    const int S = reduce_splits(R);
    hipLaunchKernelGGL(bn_bwd_sums_onelaunch_kernel, dim3((unsigned)cdiv(2 * (int64_t)C, 64), (unsigned)S), dim3(256), 0, stream, partial, R, C,
                       cdiv(R, S), tmp, counters, count, dgamma, dbeta, coef);
    return check_launch("bn_bwd_sums_onelaunch_kernel");
  }
  int rc;
  const int S = launch_reduce_stage1(partial, R, 2 * (int64_t)C, tmp, stream, &rc);
  if (rc) return rc;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, stream, (const double*)tmp, S, C, count, dgamma, dbeta, coef);
  return check_launch("bn_bwd_finalize_kernel");
}

}  // namespace
namespace im2im {
void set_bn_fused_small(int v) { g_bn_fused_small = v; }
void set_bn_onelaunch(int v) { g_bn_onelaunch = v; }
void set_pool_bwd_full(int v) { g_pool_bwd_full = v; }
void set_bn_apply_keep_mb(int v) { g_bn_apply_keep_bytes = (int64_t)v << 20; }
}

// ================================================================================================
extern "C" int64_t im2im_reduce_workspace_bytes(int64_t K) { return im2im::reduce_tmp_bytes(K); }

extern "C" int im2im_bn_finalize(const float* partial, int64_t R, int32_t C, int64_t count, const float* gamma,
                                 const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                 int32_t centered, float* mean_invstd, float* scale_shift, void* ws, int32_t* counters,
                                 int64_t* num_batches_tracked, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(partial && gamma && beta && mean_invstd && scale_shift && ws && R > 0 && C > 0 && count > 0);
  IM2IM_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  IM2IM_REQUIRE(!centered || running_mean);
  if (R <= bn_fused_rows()) {
    hipLaunchKernelGGL(bn_stats_fused_kernel, dim3((unsigned)cdiv(C, 16)), dim3(1024), 0, stream, partial, R, (int)C, gamma, beta,
                       running_mean, running_var, momentum, eps, (int)centered, mean_invstd, scale_shift, (long long*)num_batches_tracked);
    return check_launch("bn_stats_fused_kernel");
  }
  const int S = reduce_splits(R);
  if (counters && g_bn_onelaunch && cdiv(C, 64) <= IM2IM_BN_COUNTERS) {
    hipLaunchKernelGGL(bn_stats_onelaunch_kernel, dim3((unsigned)cdiv(C, 64), (unsigned)S), dim3(1024), 0, stream, partial, R, (int)C, cdiv(R, S),
                       (double*)ws, counters, gamma, beta, running_mean, running_var, momentum, eps, (int)centered, mean_invstd, scale_shift,
                       (long long*)num_batches_tracked);
    return check_launch("bn_stats_onelaunch_kernel");
  }
  hipLaunchKernelGGL(bn_stats_stage1_kernel, dim3((unsigned)cdiv(C, 64), (unsigned)S), dim3(1024), 0, stream, partial, R, (int)C,
                     cdiv(R, S), (double*)ws);
  if (int rc = check_launch("bn_stats_stage1_kernel")) return rc;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(C, 4)), dim3(256), 0, stream, (const double*)ws, S, (int)C,
                     (double)count, gamma, beta, running_mean, running_var, momentum, eps, (int)centered, mean_invstd, scale_shift,
                     (long long*)num_batches_tracked);
  return check_launch("bn_finalize_kernel");
}

extern "C" int im2im_bn_fold_eval(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                                  const float* conv_bias, float eps, int32_t C, float* scale_shift, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(gamma && beta && running_mean && running_var && scale_shift && C > 0);
  hipLaunchKernelGGL(bn_fold_eval_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, stream, gamma, beta, running_mean,
                     running_var, conv_bias, eps, (int)C, scale_shift);
  return check_launch("bn_fold_eval_kernel");
}

extern "C" int im2im_bn_relu_apply(const void* z, const float* scale_shift, void* a, int64_t M, int32_t C, int32_t dtype,
                                   im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(z && scale_shift && a && M > 0 && C > 0 && C % 8 == 0);
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const int64_t nvec = M * C / Vec16<T>::N;
    hipLaunchKernelGGL(bn_relu_apply_kernel<T>, dim3(ew_blocks(nvec)), dim3(256), 0, stream, (const T*)z, scale_shift, (T*)a, nvec, (int)C);
    return check_launch("bn_relu_apply_kernel");
  });
}

extern "C" int64_t im2im_bn_bwd_workspace_bytes(int64_t M, int32_t C) {
  const int64_t nblk = std::min<int64_t>(cdiv(M, 64), 2048);
  return nblk * 2 * C * (int64_t)sizeof(float) + reduce_tmp_bytes(2 * (int64_t)C) + 2 * (int64_t)C * sizeof(float);
}

extern "C" int im2im_bn_relu_bwd(const void* da, const void* z, const float* scale_shift, const float* mean_invstd, void* dz,
                                 float* dgamma, float* dbeta, int64_t M, int32_t C, int32_t dtype, void* ws, int64_t ws_bytes,
                                 int32_t* counters, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(da && z && scale_shift && mean_invstd && dz && dgamma && dbeta && ws && M > 0 && C > 0 && C % 8 == 0);
  IM2IM_REQUIRE(ws_bytes >= im2im_bn_bwd_workspace_bytes(M, C));
  IM2IM_REQUIRE(C <= 1024);                                   // one channel vector per thread
  const int64_t nblk = std::min<int64_t>(cdiv(M, 64), 2048);
  const int64_t rpb = cdiv(M, nblk);
  float* partial = (float*)ws;
  double* tmp = (double*)((char*)ws + nblk * 2 * C * sizeof(float));
  float* coef = (float*)((char*)tmp + reduce_tmp_bytes(2 * (int64_t)C));
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel<T>, dim3((unsigned)cdiv(M, rpb)), dim3(256), 0, stream,
                       (const T*)da, (const T*)z, scale_shift, mean_invstd, M, (int)C, rpb, partial);
    if (int rc = check_launch("bn_relu_bwd_reduce_kernel")) return rc;
    if (int rc = launch_bn_bwd_sums(partial, cdiv(M, rpb), (int)C, (double)M, tmp, dgamma, dbeta, coef, counters, stream)) return rc;
    const int64_t nvec = M * C / Vec16<T>::N;
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<T>, dim3(ew_blocks(nvec)), dim3(256), 0, stream, (const T*)da, (const T*)z,
                       scale_shift, mean_invstd, coef, (T*)dz, nvec, (int)C, (int)(nvec * 16 <= g_bn_apply_keep_bytes));
    return check_launch("bn_relu_bwd_apply_kernel");
  });
}

// The same BatchNorm+ReLU backward cut into phases over row ranges, so that a caller can run it on its own stream while
// the data-gradient kernels that produce `da` / consume `dz` work on another half of the batch (nn_ops.BnReluLazy):
//   phase 1: per-block partial sums of rows [row0,row1)   (row0 a multiple of im2im_bn_bwd_rows_per_block(M))
//   phase 2: reduce all partial rows -> dgamma, dbeta, coefficients (needs phase 1 over every row)
//   phase 4: dz rows [row0,row1)                          (needs phase 2)
// The block decomposition, and therefore every bit of the result, is that of im2im_bn_relu_bwd on the whole tensor.
extern "C" int64_t im2im_bn_bwd_rows_per_block(int64_t M) {
  if (M <= 0) return 0;
  const int64_t nblk = std::min<int64_t>(cdiv(M, 64), 2048);
  return cdiv(M, nblk);
}

extern "C" int im2im_bn_relu_bwd_phase(const void* da, const void* z, const float* scale_shift, const float* mean_invstd, void* dz,
                                       float* dgamma, float* dbeta, int64_t M, int32_t C, int32_t dtype, void* ws, int64_t ws_bytes,
                                       int32_t phase, int64_t row0, int64_t row1, int32_t* counters, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(da && z && scale_shift && mean_invstd && dz && dgamma && dbeta && ws && M > 0 && C > 0 && C % 8 == 0);
  IM2IM_REQUIRE(ws_bytes >= im2im_bn_bwd_workspace_bytes(M, C));
  IM2IM_REQUIRE(C <= 1024);
  IM2IM_REQUIRE(phase == 1 || phase == 2 || phase == 4);
  const int64_t nblk = std::min<int64_t>(cdiv(M, 64), 2048);
  const int64_t rpb = cdiv(M, nblk);
  IM2IM_REQUIRE(phase == 2 || (row0 >= 0 && row0 < row1 && row1 <= M));
  IM2IM_REQUIRE(phase != 1 || row0 % rpb == 0);
  float* partial = (float*)ws;
  double* tmp = (double*)((char*)ws + nblk * 2 * C * sizeof(float));
  float* coef = (float*)((char*)tmp + reduce_tmp_bytes(2 * (int64_t)C));
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    if (phase == 1) {
      const int64_t rows = row1 - row0;
      hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel<T>, dim3((unsigned)cdiv(rows, rpb)), dim3(256), 0, stream,
                         (const T*)da + row0 * C, (const T*)z + row0 * C, scale_shift, mean_invstd, rows, (int)C, rpb,
                         partial + (row0 / rpb) * 2 * C);
      return check_launch("bn_relu_bwd_reduce_kernel");
    }
    if (phase == 2) {
      return launch_bn_bwd_sums(partial, cdiv(M, rpb), (int)C, (double)M, tmp, dgamma, dbeta, coef, counters, stream);
    }
    const int64_t nvec = (row1 - row0) * C / Vec16<T>::N;
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<T>, dim3(ew_blocks(nvec)), dim3(256), 0, stream, (const T*)da + row0 * C,
                       (const T*)z + row0 * C, scale_shift, mean_invstd, coef, (T*)dz + row0 * C, nvec, (int)C, 0);
    return check_launch("bn_relu_bwd_apply_kernel");
  });
}

constexpr int POOL_BWD_MAX_BLOCKS = 6144, POOL_BWD_DEFAULT_BLOCKS = 2048;
int g_pool_bwd_blocks = POOL_BWD_DEFAULT_BLOCKS;             // im2im_set_option("pool_bwd_blocks", n in [1, 6144]; anything else = the default 2048): A/B -- 1,536 / 2,048 / 6,144 measured equal (r05_ab_experiments.txt section 6)
namespace im2im { void set_pool_bwd_blocks(int v) { g_pool_bwd_blocks = v > 0 && v <= POOL_BWD_MAX_BLOCKS ? v : POOL_BWD_DEFAULT_BLOCKS; } }
namespace {
inline dim3 pool_bwd_grid(int B, int H, int W, int vpr) {
  const int Hc = (H + 1) / 2, Wc = (W + 1) / 2;
  int gx = (int)cdiv((int64_t)Wc * vpr, 256);
  if (gx > 16) gx = 16;
  int64_t gy = std::min<int64_t>((int64_t)B * Hc, std::max<int64_t>(g_pool_bwd_blocks / gx, 1));
  return dim3((unsigned)gx, (unsigned)gy);
}
}  // namespace

extern "C" int64_t im2im_bn_relu_pool_bwd_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t C) {
  const int64_t nblk = POOL_BWD_MAX_BLOCKS + 16;                // upper bound of pool_bwd_grid's block count
  return nblk * 2 * C * (int64_t)sizeof(float) + reduce_tmp_bytes(2 * (int64_t)C) + 2 * (int64_t)C * sizeof(float);
}

extern "C" int im2im_bn_relu_pool_bwd(const void* da, const void* dpool, const void* z, const float* scale_shift,
                                      const float* mean_invstd, void* dz, float* dgamma, float* dbeta, int32_t B, int32_t H,
                                      int32_t W, int32_t C, int32_t dtype, void* ws, int64_t ws_bytes, int32_t* counters,
                                      im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(dpool && z && scale_shift && mean_invstd && dz && dgamma && dbeta && ws);
  IM2IM_REQUIRE(B > 0 && H >= 2 && W >= 2 && C > 0);
  IM2IM_REQUIRE(dtype == IM2IM_F32 || dtype == IM2IM_BF16);
  const int n = dtype == IM2IM_BF16 ? 8 : 4;
  IM2IM_REQUIRE(C % n == 0);
  const int vpr = C / n;
  IM2IM_REQUIRE((vpr & (vpr - 1)) == 0 && vpr <= 256);           // one fixed channel vector per thread
  IM2IM_REQUIRE(ws_bytes >= im2im_bn_relu_pool_bwd_workspace_bytes(B, H, W, C));
  const dim3 grid = pool_bwd_grid(B, H, W, vpr);
  const int64_t nblk = (int64_t)grid.x * grid.y;
  float* partial = (float*)ws;
  double* tmp = (double*)((char*)ws + (int64_t)(POOL_BWD_MAX_BLOCKS + 16) * 2 * C * sizeof(float));
  float* coef = (float*)((char*)tmp + reduce_tmp_bytes(2 * (int64_t)C));
  const double count = (double)B * H * W;
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const RowVec rv = make_rowvec(vpr);
    const bool full = da != nullptr && H % 2 == 0 && W % 2 == 0 && g_pool_bwd_full;
    if (full) hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<T, false, true>), grid, dim3(256), 0, stream, (const T*)da, (const T*)dpool, (const T*)z,
                                 scale_shift, mean_invstd, (const float*)nullptr, (T*)nullptr, partial, B, H, W, (int)C, rv);
    else hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<T, false, false>), grid, dim3(256), 0, stream, (const T*)da, (const T*)dpool, (const T*)z,
                            scale_shift, mean_invstd, (const float*)nullptr, (T*)nullptr, partial, B, H, W, (int)C, rv);
    if (int rc = check_launch("bn_relu_pool_bwd_kernel<reduce>")) return rc;
    if (int rc = launch_bn_bwd_sums(partial, nblk, (int)C, count, tmp, dgamma, dbeta, coef, counters, stream)) return rc;
    if (full) hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<T, true, true>), grid, dim3(256), 0, stream, (const T*)da, (const T*)dpool, (const T*)z,
                                 scale_shift, mean_invstd, (const float*)coef, (T*)dz, (float*)nullptr, B, H, W, (int)C, rv);
    else hipLaunchKernelGGL((bn_relu_pool_bwd_kernel<T, true, false>), grid, dim3(256), 0, stream, (const T*)da, (const T*)dpool, (const T*)z,
                            scale_shift, mean_invstd, (const float*)coef, (T*)dz, (float*)nullptr, B, H, W, (int)C, rv);
    return check_launch("bn_relu_pool_bwd_kernel<apply>");
  });
}

extern "C" int im2im_bn_relu_bwd_from_partial(const void* da, const void* z, const float* scale_shift, const float* mean_invstd,
                                              const float* partial, int64_t R, void* dz, float* dgamma, float* dbeta, int64_t M,
                                              int32_t C, int32_t dtype, void* ws, int64_t ws_bytes, int32_t* counters,
                                              im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(da && z && scale_shift && mean_invstd && partial && dz && dgamma && dbeta && ws && M > 0 && C > 0 && C % 8 == 0 && R > 0);
  IM2IM_REQUIRE(ws_bytes >= reduce_tmp_bytes(2 * (int64_t)C) + 2 * (int64_t)C * (int64_t)sizeof(float));
  IM2IM_REQUIRE(C <= 1024);
  double* tmp = (double*)ws;
  float* coef = (float*)((char*)tmp + reduce_tmp_bytes(2 * (int64_t)C));
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    if (int rc = launch_bn_bwd_sums(partial, R, (int)C, (double)M, tmp, dgamma, dbeta, coef, counters, stream)) return rc;
    const int64_t nvec = M * C / Vec16<T>::N;
    hipLaunchKernelGGL(bn_relu_bwd_apply_kernel<T>, dim3(ew_blocks(nvec)), dim3(256), 0, stream, (const T*)da, (const T*)z,
                       scale_shift, mean_invstd, coef, (T*)dz, nvec, (int)C, (int)(nvec * 16 <= g_bn_apply_keep_bytes));
    return check_launch("bn_relu_bwd_apply_kernel");
  });
}

// ------------------------------------------------------------------------------------------------ GroupNorm (extra)
extern "C" int im2im_groupnorm_finalize(const float* partial, int32_t B, int32_t tiles_per_image, int32_t C, int32_t G,
                                        const float* gamma, const float* beta, float eps, float* mean_rstd, float* scale_shift,
                                        im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(partial && gamma && beta && mean_rstd && scale_shift && B > 0 && tiles_per_image > 0 && C > 0 && G > 0 && C % G == 0);
  IM2IM_REQUIRE(B <= 65535);
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((unsigned)G, (unsigned)B), dim3(256), 0, stream, partial, (int)tiles_per_image, (int)C,
                     (int)G, gamma, beta, eps, mean_rstd, scale_shift);
  return check_launch("gn_finalize_kernel");
}

namespace {
inline int gn_blocks_per_image(int64_t B, int64_t HW) {
  int64_t n = std::min<int64_t>(cdiv(HW, 64), std::max<int64_t>(1, 2048 / B));
  return (int)std::max<int64_t>(n, 1);
}
}  // namespace

extern "C" int64_t im2im_groupnorm_stats_rows(int32_t B, int64_t HW) { return (int64_t)B * gn_blocks_per_image(B, HW); }

extern "C" int im2im_groupnorm_stats(const void* z, int32_t B, int64_t HW, int32_t C, int32_t dtype, float* partial,
                                     im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(z && partial && B > 0 && B <= 65535 && HW > 0 && C > 0 && C % 8 == 0 && C <= 2048);
  const int nblk = gn_blocks_per_image(B, HW);
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    IM2IM_REQUIRE(C / Vec16<T>::N <= 256);
    hipLaunchKernelGGL(gn_stats_kernel<T>, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, stream, (const T*)z, HW, (int)C,
                       cdiv(HW, nblk), partial);
    return check_launch("gn_stats_kernel");
  });
}

extern "C" int im2im_affine_relu_apply_per_image(const void* z, const float* scale_shift, void* a, int32_t B, int64_t HW,
                                                 int32_t C, int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(z && scale_shift && a && B > 0 && B <= 65535 && HW > 0 && C > 0 && C % 8 == 0);
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const int64_t nvec = HW * C / Vec16<T>::N;
    const int gx = (int)std::min<int64_t>(cdiv(nvec, 256), std::max<int64_t>(1, 8192 / B));
    hipLaunchKernelGGL(affine_relu_apply_img_kernel<T>, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, stream, (const T*)z, scale_shift,
                       (T*)a, nvec, (int)C);
    return check_launch("affine_relu_apply_img_kernel");
  });
}

extern "C" int64_t im2im_groupnorm_relu_bwd_workspace_bytes(int32_t B, int64_t HW, int32_t C) {
  return ((int64_t)B * gn_blocks_per_image(B, HW) * 2 * C + (int64_t)B * 4 * C) * (int64_t)sizeof(float);
}

extern "C" int im2im_groupnorm_relu_bwd(const void* da, const void* z, const float* scale_shift, const float* mean_rstd,
                                        const float* gamma, void* dz, float* dgamma, float* dbeta, int32_t B, int64_t HW,
                                        int32_t C, int32_t G, int32_t dtype, void* ws, int64_t ws_bytes, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(da && z && scale_shift && mean_rstd && gamma && dz && dgamma && dbeta && ws);
  IM2IM_REQUIRE(B > 0 && B <= 65535 && HW > 0 && C > 0 && C % 8 == 0 && C <= 1024 && G > 0 && C % G == 0);
  IM2IM_REQUIRE(ws_bytes >= im2im_groupnorm_relu_bwd_workspace_bytes(B, HW, C));
  const int nblk = gn_blocks_per_image(B, HW);
  float* partial = (float*)ws;
  float* coef = partial + (size_t)B * nblk * 2 * C;
  float* sums = coef + (size_t)B * 2 * C;
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    IM2IM_REQUIRE(256 % (C / Vec16<T>::N) == 0);               // the reduce kernel keeps one channel vector per thread
    hipLaunchKernelGGL(bn_relu_bwd_reduce_kernel<T>, dim3((unsigned)nblk, (unsigned)B), dim3(256), 0, stream, (const T*)da, (const T*)z,
                       scale_shift, mean_rstd, HW, (int)C, cdiv(HW, nblk), partial);
    if (int rc = check_launch("bn_relu_bwd_reduce_kernel")) return rc;
    hipLaunchKernelGGL(gn_bwd_finalize_kernel, dim3((unsigned)G, (unsigned)B), dim3(256), 0, stream, partial, nblk, (int)C, (int)G,
                       (double)(C / G) * (double)HW, gamma, coef, sums);
    if (int rc = check_launch("gn_bwd_finalize_kernel")) return rc;
    hipLaunchKernelGGL(gn_param_grad_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, stream, sums, (int)B, (int)C, dgamma, dbeta);
    if (int rc = check_launch("gn_param_grad_kernel")) return rc;
    const int64_t nvec = HW * C / Vec16<T>::N;
    const int gx = (int)std::min<int64_t>(cdiv(nvec, 256), std::max<int64_t>(1, 8192 / B));
    hipLaunchKernelGGL(gn_relu_bwd_apply_kernel<T>, dim3((unsigned)gx, (unsigned)B), dim3(256), 0, stream, (const T*)da, (const T*)z,
                       scale_shift, mean_rstd, coef, gamma, (T*)dz, nvec, (int)C);
    return check_launch("gn_relu_bwd_apply_kernel");
  });
}

extern "C" int im2im_head_activation_fwd(float* out, float* pre, int64_t B, int64_t P, int64_t img_stride, int64_t plane_offset,
                                         int32_t kind, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(out && pre && B > 0 && P > 0 && (kind == 0 || kind == 1));
  hipLaunchKernelGGL(head_act_fwd_kernel, dim3(ew_blocks(B * P)), dim3(256), 0, stream, out, pre, B, P, img_stride, plane_offset, (int)kind);
  return check_launch("head_act_fwd_kernel");
}

extern "C" int im2im_head_activation_bwd(float* dout, const float* pre, int64_t B, int64_t P, int64_t img_stride,
                                         int64_t plane_offset, int32_t kind, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(dout && pre && B > 0 && P > 0 && (kind == 0 || kind == 1));
  hipLaunchKernelGGL(head_act_bwd_kernel, dim3(ew_blocks(B * P)), dim3(256), 0, stream, dout, pre, B, P, img_stride, plane_offset, (int)kind);
  return check_launch("head_act_bwd_kernel");
}

extern "C" int im2im_depth_space2(const void* in, void* out, int64_t B, int32_t h, int32_t w, int32_t C, int32_t to_space,
                                  int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(in && out && B > 0 && h > 0 && w > 0 && C > 0);
  const int esz = dtype == IM2IM_BF16 ? 2 : 4;
  IM2IM_REQUIRE(dtype == IM2IM_BF16 || dtype == IM2IM_F32);
  IM2IM_REQUIRE((C * esz) % 16 == 0);
  const int vpc = C * esz / 16;
  hipLaunchKernelGGL(depth_space2_kernel, dim3(ew_blocks(B * h * w * 4 * vpc)), dim3(256), 0, stream, (const uint4*)in, (uint4*)out, B, (int)h,
                     (int)w, vpc, (int)to_space);
  return check_launch("depth_space2_kernel");
}

extern "C" int64_t im2im_colsum_workspace_bytes(int64_t M, int32_t C) {
  const int64_t nblk = std::min<int64_t>(cdiv(M, 64), 2048);
  return nblk * C * (int64_t)sizeof(float) + reduce_tmp_bytes(C);
}

extern "C" int im2im_colsum(const void* x, float* out, int64_t M, int32_t C, int32_t dtype, void* ws, int64_t ws_bytes,
                            im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(x && out && ws && M > 0 && C > 0 && C % 8 == 0);
  IM2IM_REQUIRE(ws_bytes >= im2im_colsum_workspace_bytes(M, C));
  IM2IM_REQUIRE(C <= 1024);
  const int64_t nblk = std::min<int64_t>(cdiv(M, 64), 2048);
  const int64_t rpb = cdiv(M, nblk);
  float* partial = (float*)ws;
  double* tmp = (double*)((char*)ws + nblk * C * sizeof(float));
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(colsum_partial_kernel<T>, dim3((unsigned)cdiv(M, rpb)), dim3(256), 0, stream, (const T*)x,
                       M, (int)C, rpb, partial);
    if (int rc = check_launch("colsum_partial_kernel")) return rc;
    int rc;
    const int S = launch_reduce_stage1(partial, cdiv(M, rpb), C, tmp, stream, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(sum_final_kernel, dim3((unsigned)cdiv(C, 256)), dim3(256), 0, stream, (const double*)tmp, S, (int64_t)C, out);
    return check_launch("sum_final_kernel");
  });
}

extern "C" int im2im_maxpool2_fwd(const void* x, const float* in_scale_shift, void* y, int32_t B, int32_t H, int32_t W,
                                  int32_t C, int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(x && y && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0);
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const RowVec rv = make_rowvec(C / Vec16<T>::N);
    hipLaunchKernelGGL(maxpool2_fwd_kernel<T>, row_grid((W / 2) * rv.vpr, B * (H / 2)), dim3(256), 0, stream, (const T*)x, in_scale_shift, (T*)y, B, H, W, C, rv);
    return check_launch("maxpool2_fwd_kernel");
  });
}

extern "C" int im2im_maxpool2_bwd(const void* x, const float* in_scale_shift, const void* dy, void* dx, int32_t B, int32_t H,
                                  int32_t W, int32_t C, int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(x && dy && dx && B > 0 && H >= 2 && W >= 2 && C > 0 && C % 8 == 0);
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const RowVec rv = make_rowvec(C / Vec16<T>::N);
    hipLaunchKernelGGL(maxpool2_bwd_kernel<T>, row_grid(((W + 1) / 2) * rv.vpr, B * ((H + 1) / 2)), dim3(256), 0, stream, (const T*)x, in_scale_shift, (const T*)dy, (T*)dx, B, H, W, C, rv);
    return check_launch("maxpool2_bwd_kernel");
  });
}

extern "C" int im2im_upsample2x_concat_fwd(const void* deep, const float* deep_scale_shift, const void* skip,
                                           const float* skip_scale_shift, void* out, int32_t B, int32_t h, int32_t w,
                                           int32_t Cd, int32_t H, int32_t W, int32_t Cs, int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(deep && out && B > 0 && h > 0 && w > 0 && H >= 2 * h && W >= 2 * w);
  IM2IM_REQUIRE(Cd > 0 && Cs >= 0 && Cd % 8 == 0 && Cs % 8 == 0);
  IM2IM_REQUIRE((Cs > 0) == (skip != nullptr));
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const bool tiled = Cd % UP_CC == 0 && Cs % Vec16<T>::N == 0;
    if (tiled) {
      const int tilesY = (int)cdiv(H, UPF_TR), tilesX = (int)cdiv(W, UPF_TC);
      const int64_t nblk = (int64_t)B * tilesY * tilesX * (Cd / UP_CC);
      IM2IM_REQUIRE(nblk < (1ll << 31));
      hipLaunchKernelGGL(up2x_fwd_tiled_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, stream, (const T*)deep, deep_scale_shift,
                         (T*)out, B, h, w, Cd, H, W, Cs + Cd, Cs, tilesY, tilesX);
      if (int rc = check_launch("up2x_fwd_tiled_kernel")) return rc;
      if (Cs == 0) return IM2IM_OK;
    }
    const RowVec rvd = make_rowvec(Cd / Vec16<T>::N), rvs = Cs > 0 ? make_rowvec(Cs / Vec16<T>::N) : rvd;
    dim3 grid = row_grid(W * (tiled ? rvs.vpr : std::max(rvs.vpr, rvd.vpr)), B * H);
    grid.z = (Cs > 0 && !tiled) ? 2 : 1;            // z == 0: the skip half; z == 1: the upsampled half unless the tiled kernel wrote it
    hipLaunchKernelGGL(upcat_fwd_kernel<T>, grid, dim3(256), 0, stream, (const T*)deep, deep_scale_shift, (const T*)skip, skip_scale_shift, (T*)out, B, h, w, Cd, H, W, Cs, rvs, rvd);
    return check_launch("upcat_fwd_kernel");
  });
}

extern "C" int im2im_upsample2x_concat_bwd(const void* dout, void* ddeep, void* dskip, int32_t B, int32_t h, int32_t w,
                                           int32_t Cd, int32_t H, int32_t W, int32_t Cs, int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(dout && ddeep && B > 0 && h > 0 && w > 0 && H >= 2 * h && W >= 2 * w);
  IM2IM_REQUIRE(Cd > 0 && Cs >= 0 && Cd % 8 == 0 && Cs % 8 == 0);
  IM2IM_REQUIRE((Cs > 0) == (dskip != nullptr));
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const bool tiled = Cd % UP_CC == 0 && Cs % Vec16<T>::N == 0;
    if (tiled) {
      const int tilesY = (int)cdiv(h, UPB_TR), tilesX = (int)cdiv(w, UPB_TC);
      const int64_t nblk = (int64_t)B * tilesY * tilesX * (Cd / UP_CC);
      IM2IM_REQUIRE(nblk < (1ll << 31));
      hipLaunchKernelGGL(up2x_bwd_tiled_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, stream, (const T*)dout, (T*)ddeep, B, h, w,
                         Cd, H, W, Cs + Cd, Cs, tilesY, tilesX);
      if (int rc = check_launch("up2x_bwd_tiled_kernel")) return rc;
      if (Cs == 0) return IM2IM_OK;
    }
    const RowVec rvd = make_rowvec(Cd / Vec16<T>::N), rvs = Cs > 0 ? make_rowvec(Cs / Vec16<T>::N) : rvd;
    hipLaunchKernelGGL(upcat_bwd_kernel<T>, row_grid(Cs > 0 ? W * rvs.vpr : w * rvd.vpr, (Cs > 0 ? B * H : 0) + (tiled ? 0 : B * h)), dim3(256), 0, stream, (const T*)dout, (T*)ddeep, (T*)dskip, B, h, w, Cd, H, W, Cs, rvs, rvd, tiled ? 0 : 1);
    return check_launch("upcat_bwd_kernel");
  });
}

extern "C" int64_t im2im_quantile_loss_workspace_bytes(void) { return 1024 * 3 * (int64_t)sizeof(float) + reduce_tmp_bytes(3); }

extern "C" int im2im_uq_loss_fwd(int32_t kind, const float* pa, const float* pb, const float* pc, const float* target, int64_t N,
                                 int64_t P, int64_t img_stride, float q_lo, float q_hi, float w0, float w1, float w2,
                                 float* loss, void* ws, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(kind >= IM2IM_LOSS_QUANTILE && kind <= IM2IM_LOSS_INN);
  IM2IM_REQUIRE(pa && pb && target && loss && ws && N > 0 && P > 0);
  IM2IM_REQUIRE((kind > IM2IM_LOSS_QUANTILE_L1 && kind != IM2IM_LOSS_INN) || pc != nullptr);
  LossArgs a{pa, pb, pc, target, N, P, img_stride, q_lo, q_hi, kind};
  float* partial = (float*)ws;
  double* tmp = (double*)((char*)ws + 1024 * 3 * sizeof(float));
  int64_t nblk = cdiv(N * P, 256 * 8);
  if (nblk > 1024) nblk = 1024;
  hipLaunchKernelGGL(qloss_partial_kernel, dim3((unsigned)nblk), dim3(256), 0, stream, a, partial);
  if (int rc = check_launch("qloss_partial_kernel")) return rc;
  int rc;
  const int S = launch_reduce_stage1(partial, nblk, 3, tmp, stream, &rc);
  if (rc) return rc;
  hipLaunchKernelGGL(qloss_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)tmp, S, (double)(N * P), w0, w1, w2, loss);
  return check_launch("qloss_final_kernel");
}

extern "C" int im2im_uq_loss_bwd(int32_t kind, const float* pa, const float* pb, const float* pc, const float* target, int64_t N,
                                 int64_t P, int64_t img_stride, float q_lo, float q_hi, float w0, float w1, float w2,
                                 const float* grad_out, float* d_a, float* d_b, float* d_c, int64_t d_stride,
                                 im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(kind >= IM2IM_LOSS_QUANTILE && kind <= IM2IM_LOSS_INN);
  IM2IM_REQUIRE(pa && pb && target && grad_out && N > 0 && P > 0);
  IM2IM_REQUIRE((kind > IM2IM_LOSS_QUANTILE_L1 && kind != IM2IM_LOSS_INN) || pc != nullptr);
  LossArgs a{pa, pb, pc, target, N, P, img_stride, q_lo, q_hi, kind};
  hipLaunchKernelGGL(qloss_bwd_kernel, dim3(ew_blocks(N * P)), dim3(256), 0, stream, a, grad_out, w0, w1, w2, d_a, d_b, d_c, d_stride);
  return check_launch("qloss_bwd_kernel");
}

extern "C" int im2im_quantile_loss_fwd(const float* lo, const float* mid, const float* hi, const float* target, int64_t N,
                                       int64_t P, int64_t img_stride, float q_lo, float q_hi, float w_lo, float w_hi,
                                       float w_mse, float* loss, void* ws, im2im_stream_t stream_) {
  return im2im_uq_loss_fwd(IM2IM_LOSS_QUANTILE, lo, mid, hi, target, N, P, img_stride, q_lo, q_hi, w_lo, w_hi, w_mse, loss, ws, stream_);
}

extern "C" int im2im_quantile_loss_bwd(const float* lo, const float* mid, const float* hi, const float* target, int64_t N,
                                       int64_t P, int64_t img_stride, float q_lo, float q_hi, float w_lo, float w_hi,
                                       float w_mse, const float* grad_out, float* d_lo, float* d_mid, float* d_hi,
                                       int64_t d_stride, im2im_stream_t stream_) {
  return im2im_uq_loss_bwd(IM2IM_LOSS_QUANTILE, lo, mid, hi, target, N, P, img_stride, q_lo, q_hi, w_lo, w_hi, w_mse, grad_out, d_lo,
                           d_mid, d_hi, d_stride, stream_);
}

extern "C" int im2im_softmax_ce_fwd(const void* logits, const float* target, const float* bounds, int64_t M, int32_t K,
                                   int32_t stride, int32_t dtype, float* loss, void* ws, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(logits && target && bounds && loss && ws && M > 0);
  IM2IM_REQUIRE(K >= 2 && K <= SOFTMAX_MAX_K && stride >= K && stride % 8 == 0 && stride <= SOFTMAX_MAX_K);
  float* partial = (float*)ws;
  double* tmp = (double*)((char*)ws + 1024 * 3 * sizeof(float));
  int64_t nblk = cdiv(M, 256 * 2);
  if (nblk > 1024) nblk = 1024;
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(softmax_ce_partial_kernel<T>, dim3((unsigned)nblk), dim3(256), 0, stream, (const T*)logits, target, bounds, M,
                       (int)K, (int)stride, partial);
    if (int rc = check_launch("softmax_ce_partial_kernel")) return rc;
    int rc;
    const int S = launch_reduce_stage1(partial, nblk, 1, tmp, stream, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(softmax_ce_final_kernel, dim3(1), dim3(64), 0, stream, (const double*)tmp, S, (double)M, loss);
    return check_launch("softmax_ce_final_kernel");
  });
}

extern "C" int im2im_softmax_ce_bwd(const void* logits, const float* target, const float* bounds, int64_t M, int32_t K,
                                   int32_t stride, int32_t dtype, const float* grad_out, void* dlogits, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(logits && target && bounds && grad_out && dlogits && M > 0);
  IM2IM_REQUIRE(K >= 2 && K <= SOFTMAX_MAX_K && stride >= K && stride % 8 == 0 && stride <= SOFTMAX_MAX_K);
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(softmax_ce_bwd_kernel<T>, dim3(ew_blocks(M)), dim3(256), 0, stream, (const T*)logits, target, bounds, M, (int)K,
                       (int)stride, grad_out, (T*)dlogits);
    return check_launch("softmax_ce_bwd_kernel");
  });
}

extern "C" int im2im_softmax_sets_summary(const void* logits, int64_t N, int64_t P, int32_t K, int32_t stride, int32_t dtype,
                                         float* out3, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(logits && out3 && N >= 0 && P > 0);
  IM2IM_REQUIRE(K >= 2 && K <= SOFTMAX_MAX_K && stride >= K && stride % 8 == 0 && stride <= SOFTMAX_MAX_K);
  if (N == 0) return IM2IM_OK;
  return for_dtype(dtype, [&](auto* tag) {
    using T = std::remove_pointer_t<decltype(tag)>;
    hipLaunchKernelGGL(softmax_summary_kernel<T>, dim3(ew_blocks(N * P)), dim3(256), 0, stream, (const T*)logits, N * P, P, (int)K,
                       (int)stride, out3);
    return check_launch("softmax_summary_kernel");
  });
}

namespace {
int adam_launch(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                const int64_t* sizes, float lr, float beta1, float beta2, float eps, float step_size, float bc2_sqrt, const float* coef,
                hipStream_t stream) {
  for (int base = 0; base < n_tensors; base += ADAM_MAX_TENSORS) {
    AdamArgs a;
    a.n = std::min(ADAM_MAX_TENSORS, n_tensors - base);
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.step_size = step_size;
    a.bc2_sqrt = bc2_sqrt;
    a.coef = coef;
    int64_t chunks = 0;
    for (int i = 0; i < a.n; ++i) {
      IM2IM_REQUIRE(params[base + i] && grads[base + i] && exp_avg[base + i] && exp_avg_sq[base + i] && sizes[base + i] >= 0);
      a.p[i] = params[base + i]; a.g[i] = grads[base + i]; a.m[i] = exp_avg[base + i]; a.v[i] = exp_avg_sq[base + i];
      a.size[i] = sizes[base + i];
      a.start[i] = chunks;
      chunks += cdiv(sizes[base + i], 1024);
    }
    a.start[a.n] = chunks;
    if (chunks == 0) continue;
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)chunks), dim3(256), 0, stream, a);
    if (int rc = check_launch("adam_kernel")) return rc;
  }
  return IM2IM_OK;
}
}  // namespace

extern "C" int im2im_adam_step(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                               float* const* exp_avg_sq, const int64_t* sizes, float lr, float beta1, float beta2, float eps,
                               int64_t step, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(n_tensors >= 0 && step >= 1);
  IM2IM_REQUIRE(n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && sizes));
  const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
  return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, sizes, lr, beta1, beta2, eps, (float)((double)lr / bc1),
                     (float)sqrt(bc2), nullptr, stream);
}

extern "C" int im2im_adam_step_dev(int32_t n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                                   float* const* exp_avg_sq, const int64_t* sizes, float lr, float beta1, float beta2, float eps,
                                   int64_t* step_dev, float* coef_dev, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(n_tensors >= 0 && step_dev && coef_dev);
  IM2IM_REQUIRE(n_tensors == 0 || (params && grads && exp_avg && exp_avg_sq && sizes));
  hipLaunchKernelGGL(adam_coef_kernel, dim3(1), dim3(1), 0, stream, (long long*)step_dev, (double)lr, (double)beta1, (double)beta2, coef_dev);
  if (int rc = check_launch("adam_coef_kernel")) return rc;
  return adam_launch(n_tensors, params, grads, exp_avg, exp_avg_sq, sizes, lr, beta1, beta2, eps, 0.f, 1.f, coef_dev, stream);
}
