// fastMRI input pipeline on the GPU: k-space * column mask -> centred orthonormal inverse 2-D DFT ->
// centre crop -> magnitude -> affine normalisation.  Replaces, per slice, apply_mask (core/datasets/fastmri/
// transforms.py:53-85), ifft2c_new (fftc.py:87-110), complex_center_crop (transforms.py:130-152), complex_abs
// (math_util.py:56-70) and the normalisation of FastMRIDataset.__getitem__ (FastMRIDataset.py:147-160), which the
// reference runs on the host inside the training thread (num_workers=0, core/scripts/train.py:104).
//
// MI355X design: the transform is evaluated as a PRUNED separable DFT, not an FFT.  Only the 320 x 320 centre of the
// 640 x 368 image is ever used, so the centred inverse DFT is two dense contractions with small matrices that hold the
// shifts, the 1/sqrt(n) and the crop:   T = X * W_C^T  (contract the columns),   I = W_R * T  (contract the rows),
// 1.1 GFLOP per slice, which the exact-fp32 MFMA GEMM of conv_mfma.hip (taps = 1) runs in ~10 us -- no bit-reversal
// passes, no radix restrictions (368 = 16*23 and 372 = 4*3*31 columns occur), any mask.  This file holds the
// HBM-bound glue around the two GEMMs: mask + pad, complex transposes, magnitude + normalise.
#include "common.h"

namespace {
using namespace im2im;

// out[(b, r)][Kp] = kspace[b][r][c][ri] * mask[b][c] + 0.0 (cols >= 2C and rows >= B*R zero)
__global__ __launch_bounds__(256) void mask_pack_kernel(const float2* __restrict__ ks, const float* __restrict__ mask,
                                                         int64_t mask_stride, float2* __restrict__ out, int64_t rows_in,
                                                         int64_t rows_out, int R, int C, int Kp2) {
  const int64_t total = rows_out * Kp2;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t row = i / Kp2;
    const int c = (int)(i % Kp2);
    float2 v = make_float2(0.f, 0.f);
    if (row < rows_in && c < C) {
      const float m = mask[(row / R) * mask_stride + c];
      const float2 k = ks[row * C + c];
      v = make_float2(k.x * m + 0.0f, k.y * m + 0.0f);
    }
    out[i] = v;
  }
}

// in [B][R][ld_in/2] complex (first X columns used) -> out [B][X][ld_out/2] complex (first R used, rest zero)
__global__ __launch_bounds__(256) void complex_transpose_kernel(const float2* __restrict__ in, float2* __restrict__ out, int R,
                                                                 int X, int ld_in, int ld_out) {
  __shared__ float2 tile[32][33];
  const int b = blockIdx.z;
  const int r0 = blockIdx.y * 32, x0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;                  // 32 x 8
  const float2* ib = in + (size_t)b * R * ld_in;
  float2* ob = out + (size_t)b * X * ld_out;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, x = x0 + tx;
    tile[ty + 8 * k][tx] = (r < R && x < X) ? ib[(size_t)r * ld_in + x] : make_float2(0.f, 0.f);
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + ty + 8 * k, r = r0 + tx;
    if (x < X && r < ld_out) ob[(size_t)x * ld_out + r] = (r < R) ? tile[tx][ty + 8 * k] : make_float2(0.f, 0.f);
  }
}

// in [B][X][ld_in/2] complex (first Y used; x-major) -> out [B][Y][X] = (sqrt(re^2 + im^2) - sub) / div
__global__ __launch_bounds__(256) void abs_norm_transpose_kernel(const float2* __restrict__ in, float* __restrict__ out, int X,
                                                                  int Y, int ld_in, float sub, float div) {
  __shared__ float tile[32][33];
  const int b = blockIdx.z;
  const int x0 = blockIdx.y * 32, y0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  const float2* ib = in + (size_t)b * X * ld_in;
  float* ob = out + (size_t)b * Y * X;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int x = x0 + ty + 8 * k, y = y0 + tx;
    float v = 0.f;
    if (x < X && y < Y) {
      const float2 c = ib[(size_t)x * ld_in + y];
      v = __fsqrt_rn(__fadd_rn(__fmul_rn(c.x, c.x), __fmul_rn(c.y, c.y)));     // (data ** 2).sum(-1).sqrt(), no fma
    }
    tile[ty + 8 * k][tx] = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int y = y0 + ty + 8 * k, x = x0 + tx;
    if (y < Y && x < X) ob[(size_t)y * X + x] = __fdiv_rn(__fsub_rn(tile[tx][ty + 8 * k], sub), div);
  }
}

// out [B][H][W] = (in [B][Hin][Win] centre crop - sub) / div
__global__ __launch_bounds__(256) void crop_affine_kernel(const float* __restrict__ in, float* __restrict__ out, int64_t B, int Hin,
                                                           int Win, int H, int W, float sub, float div) {
  const int y_from = (Hin - H) / 2, x_from = (Win - W) / 2;
  const int64_t total = B * H * W;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int x = (int)(i % W);
    const int64_t t = i / W;
    const int y = (int)(t % H);
    const int64_t b = t / H;
    out[i] = __fdiv_rn(__fsub_rn(in[(b * Hin + y_from + y) * Win + x_from + x], sub), div);
  }
}

inline int blocks_for(int64_t n) { int64_t b = cdiv(n, 256); return (int)std::min<int64_t>(std::max<int64_t>(b, 1), 256 * 32); }

}  // namespace

extern "C" int im2im_fastmri_mask_pack(const float* kspace, const float* mask, int64_t mask_stride, float* out, int64_t B,
                                       int32_t R, int32_t C, int64_t rows_out, int32_t Kp, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(kspace && mask && out && B > 0 && R > 0 && C > 0 && Kp >= 2 * C && Kp % 2 == 0 && rows_out >= B * R);
  hipLaunchKernelGGL(mask_pack_kernel, dim3(blocks_for(rows_out * (Kp / 2))), dim3(256), 0, stream, (const float2*)kspace, mask,
                     mask_stride, (float2*)out, B * R, rows_out, (int)R, (int)C, (int)(Kp / 2));
  return check_launch("mask_pack_kernel");
}

extern "C" int im2im_complex_transpose(const float* in, float* out, int32_t B, int32_t R, int32_t X, int32_t ld_in,
                                       int32_t ld_out, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(in && out && B > 0 && B <= 65535 && R > 0 && X > 0 && ld_in >= 2 * X && ld_out >= 2 * R && ld_in % 2 == 0 && ld_out % 2 == 0);
  hipLaunchKernelGGL(complex_transpose_kernel, dim3((unsigned)cdiv(X, 32), (unsigned)cdiv(ld_out / 2, 32), (unsigned)B), dim3(256), 0,
                     stream, (const float2*)in, (float2*)out, (int)R, (int)X, (int)(ld_in / 2), (int)(ld_out / 2));
  return check_launch("complex_transpose_kernel");
}

extern "C" int im2im_fastmri_abs_normalize(const float* in, float* out, int32_t B, int32_t X, int32_t Y, int32_t ld_in, float sub,
                                           float div, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(in && out && B > 0 && B <= 65535 && X > 0 && Y > 0 && ld_in >= 2 * Y && ld_in % 2 == 0);
  hipLaunchKernelGGL(abs_norm_transpose_kernel, dim3((unsigned)cdiv(Y, 32), (unsigned)cdiv(X, 32), (unsigned)B), dim3(256), 0, stream,
                     (const float2*)in, out, (int)X, (int)Y, (int)(ld_in / 2), sub, div);
  return check_launch("abs_norm_transpose_kernel");
}

extern "C" int im2im_center_crop_affine(const float* in, float* out, int64_t B, int32_t Hin, int32_t Win, int32_t H, int32_t W,
                                        float sub, float div, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(in && out && B > 0 && H > 0 && W > 0 && H <= Hin && W <= Win);
  hipLaunchKernelGGL(crop_affine_kernel, dim3(blocks_for(B * H * W)), dim3(256), 0, stream, in, out, B, (int)Hin, (int)Win, (int)H, (int)W,
                     sub, div);
  return check_launch("crop_affine_kernel");
}
