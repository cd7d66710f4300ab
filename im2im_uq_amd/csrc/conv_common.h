// Shared pieces of the MFMA implicit-GEMM convolution kernels (conv_mfma.hip, conv_fp8.hip): argument block, operand-fragment
// traits, moment merging, tile choice.
#pragma once
#include "common.h"
#include "dtypes.h"

namespace im2im {

struct ConvArgs {
  const void* x;        // [B][H][W][Ci]  T
  const void* w;        // [Co][TAPS][Ci] T
  const float* bias;    // [Co] or null
  const float* scale;   // [Co] or null  (mode affine)
  const float* shift;   // [Co] or null
  void* y;              // [B][H][W][Co]  T
  float* stats;         // [mtiles][3][Co] or null: per-tile (mean, M2, count) of the stored values
  int B, H, W, Ci, Co, tilesY, tilesX;
  int relu;             // apply ReLU after affine
  const float* center;  // [Co] or null: subtracted from the stored output (see im2im_conv_fwd)
  const float* in_ss;   // [2][Ci] or null: x holds the producer's PRE-BatchNorm output z; the operand staging applies
                        // a = max(z*scale + shift, 0) on the fly (the BatchNorm+ReLU pass is never materialised)
  // channel-split operands (the Up block's torch.cat([skip, up], 1) is never materialised, unet_parts.py:68):
  const void* x_hi;     // null, or: input channels [Ci_lo, Ci) live here (pixel stride Ci - Ci_lo == Ci_lo), [0, Ci_lo) in x
  const float* in_ss_hi;// [2][Ci - Ci_lo] or null: lazy BatchNorm+ReLU of x_hi (in_ss then describes x's Ci_lo channels)
  int Ci_lo;
  void* y_hi;           // null, or: output channels [Co_lo, Co) go here (pixel stride Co - Co_lo), [0, Co_lo) to y
  int Co_lo;
  // EPI 3 (data-gradient whose result is the gradient of a lazy BatchNorm+ReLU activation): the producer's pre-BN
  // output, its BatchNorm coefficients, and where this tile's partial sums of g and g*xhat go ([tiles][2][Co])
  const void* bn_z;
  const float* bn_ss;   // [2][Co] scale, shift
  const float* bn_mi;   // [2][Co] mean, invstd
  float* bn_partial;
  // GroupNorm producers have one (scale, shift) pair per IMAGE and channel: in_ss then points at [B][2][Ci] and this is the
  // stride between images (2*Ci); 0 = one pair per channel for the whole batch (BatchNorm).  Needs one image per tile.
  int in_ss_img;
  // split-K (conv_mfma.hip EPI 4): number of K splits (1 = off), input-channel chunks of 32 per split, fp32 partial sums
  // [ksplit][B*H*W][Co]
  int ksplit, kchunks;
  float* kpartial;
  int64_t kws_bytes;    // size of the caller's workspace behind kpartial
  // EPI 2 (eval): null, or [B][H/2][W/2][Co] T: MaxPool2d(2) of the stored result, taken from the epilogue's LDS tile (H, W even)
  void* pool_y;
  // EPI 5 (eval, Co == 64): the block's result is consumed ONLY by a 1x1 conv (OutConv, unet_parts.py:87-94): fuse_w [C1][64] T
  // (im2im_pack_conv_weight's wf for taps = 1), fuse_bias [C1] fp32 | null, fuse_y [B][H][W][C1] T; y itself is not written
  const void* fuse_w;
  const float* fuse_bias;
  void* fuse_y;
};

// Packed bf16 3x3 weights are FRAGMENT-MAJOR: the 32 rows x 16 reduction channels one lane-set of v_mfma_f32_32x32x16_bf16
// consumes form one contiguous 1 KiB block, lane (half*32 + row%32) owning 16 bytes, so a wave fetches a whole operand
// fragment from L2 with a single fully coalesced global_load_dwordx4 (conv_mfma.hip reads them straight into registers).
// Blocks are ordered [row block of 32][tap][32-channel chunk][k-step].  N x 9 x K logical tensor, K % 32 == 0, N % 32 == 0.
__host__ __device__ inline size_t wfrag_index(int n, int tap, int k, int K) {
  const int nch = K >> 5;
  return ((((size_t)(n >> 5) * 9 + tap) * nch + (k >> 5)) * 2 + ((k >> 4) & 1)) * 512 + ((((k >> 3) & 1) * 32 + (n & 31)) * 8) + (k & 7);
}
inline bool wfrag_layout(int elem_bytes, int taps, int Co, int Ci) { return elem_bytes == 2 && taps == 9 && Co % 32 == 0 && Ci % 32 == 0; }

template <typename T> struct Frag;
template <> struct Frag<bf16_t> {
  using AB = short8;
  static constexpr int KSTEPS = 2;            // 32 channels / 16 per MFMA
  static __device__ __forceinline__ AB load(const char* base, int ks, int half) {
    return *reinterpret_cast<const AB*>(base + ks * 32 + half * 16);
  }
  static __device__ __forceinline__ f32x16 mfma(AB a, AB b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
  }
};
template <> struct Frag<float> {
  using AB = float;
  static constexpr int KSTEPS = 16;           // 32 channels / 2 per MFMA
  static __device__ __forceinline__ AB load(const char* base, int ks, int half) {
    return *reinterpret_cast<const float*>(base + (ks * 2 + half) * 4);
  }
  static __device__ __forceinline__ f32x16 mfma(AB a, AB b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
};

// (count, mean, M2) of a union of two sample sets (Chan et al.); symmetric, so both partners of a shuffle agree
__device__ __forceinline__ void merge_moments_f32(float& n, float& m, float& q, float n2, float m2, float q2) {
  const float nn = n + n2;
  const float inv = nn > 0.f ? 1.f / nn : 0.f;
  const float d = m2 - m;
  q = q + q2 + d * d * (n * n2 * inv);
  m = (n * m + n2 * m2) * inv;
  n = nn;
}


// pixel-tile shape per problem: 16x16 for the large-extent levels, 8x8 patches of FOUR consecutive images for the deep,
// small-extent ones (40x40, 20x20) so that little of a tile hangs over the image edge while a weight tile is still
// amortised over 256 output pixels.
struct TileChoice { int tb, th, tw, bn; };
inline TileChoice pick_tile(int B, int H, int W, int Co, bool per_image = false) {
  const bool small = (H < 64 || W < 64) && !per_image;       // per_image: every tile (and its statistics row) lies in ONE image
  const int bn = (Co % 128 == 0) ? 128 : (Co % 64 == 0) ? 64 : 32;
  // 64 output channels (the full-resolution layers): 32 x 16 pixels, so that a wave owns 128 pixels x 64 channels and a weight
  // fragment read from L2 feeds four MFMAs as in the 128-wide tiles (conv_mfma.hip DIRECTW)
  if (!small && bn == 64 && !per_image && H % 32 == 0) return {1, 32, 16, bn};
  if (!small) return {1, 16, 16, bn};
  // [r3] small batches (the per-GPU share of a strong-scaled job): when four-image tiles would leave the chip's workgroup slots
  // more than half empty, two-image tiles (M = 128, 64 accumulators, three workgroups per CU) double the launch
  const long wgs4 = (long)cdiv(B, 4) * cdiv(H, 8) * cdiv(W, 8) * (Co / bn);
  return TileChoice{wgs4 < 384 ? 2 : 4, 8, 8, bn};
}

// conv_mfma.hip: A/B switch of the split-K path (im2im_set_option "conv_splitk", conv_wgrad.hip)
void set_conv_splitk(int v);
void set_bn_fused_small(int v);          // elementwise.hip: one-launch BatchNorm sums for few partial rows
void set_bn_onelaunch(int v);            // elementwise.hip [r6]: last-block-finalizes BatchNorm sums at every size
void set_pool_bwd_blocks(int v);
void set_bn_apply_keep_mb(int v);
void set_pool_bwd_full(int v);           // elementwise.hip: branch-free bn_relu_pool_bwd for even extents (A/B)
// conv_roll.hip [r5]: the persistent, software-pipelined kernel of the 64-output-channel full-resolution layers
void set_conv_roll(int v);               // A/B switch (im2im_set_option "conv_roll")
bool conv_roll64_eligible(const ConvArgs& a, const TileChoice& t, int taps, bool per_image);
int launch_conv_roll64(const ConvArgs& a, hipStream_t stream);

}  // namespace im2im
