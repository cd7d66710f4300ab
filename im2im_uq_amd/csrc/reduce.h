// Deterministic two-stage row reduction used by every "sum over all pixels" in the library
// (BatchNorm statistics, BN backward sums, small-conv weight gradients, bias gradients).
//   stage 1: partial[R][K] fp32  ->  tmp[S][K] fp64      (grid (ceil(K/64), S), no atomics)
//   stage 2: a per-use kernel sums the S rows of tmp in a fixed order and finishes the math.
#pragma once
#include "common.h"

namespace im2im {

constexpr int REDUCE_MAX_S = 64;   // must stay <= 64: the finalize kernels give one lane to each split row

inline int reduce_splits(int64_t R) {
  int64_t s = cdiv(R, 16);
  if (s > REDUCE_MAX_S) s = REDUCE_MAX_S;
  if (s < 1) s = 1;
  return (int)s;
}
inline int64_t reduce_tmp_bytes(int64_t K) { return (int64_t)REDUCE_MAX_S * K * (int64_t)sizeof(double); }

static __global__ __launch_bounds__(256) void reduce_rows_stage1_kernel(const float* __restrict__ partial, int64_t R, int64_t K,
                                                                  int64_t rows_per_split, double* __restrict__ tmp) {
  __shared__ double s_acc[4][64];
  const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
  const int64_t k = (int64_t)blockIdx.x * 64 + c;
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_split;
  const int64_t r1 = (r0 + rows_per_split < R) ? r0 + rows_per_split : R;
  double acc = 0.0;
  if (k < K)
    for (int64_t r = r0 + rl; r < r1; r += 4) acc += (double)partial[r * K + k];
  s_acc[rl][c] = acc;
  __syncthreads();
  if (rl == 0 && k < K) tmp[(int64_t)blockIdx.y * K + k] = s_acc[0][c] + s_acc[1][c] + s_acc[2][c] + s_acc[3][c];
}

// returns the number of tmp rows (S) written
inline int launch_reduce_stage1(const float* partial, int64_t R, int64_t K, double* tmp, hipStream_t stream, int* rc) {
  const int S = reduce_splits(R);
  const int64_t rps = cdiv(R, S);
  hipLaunchKernelGGL(reduce_rows_stage1_kernel, dim3((unsigned)cdiv(K, 64), (unsigned)S), dim3(256), 0, stream, partial, R, K,
                     rps, tmp);
  *rc = check_launch("reduce_rows_stage1_kernel");
  return S;
}

}  // namespace im2im
