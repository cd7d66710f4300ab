// fp8 (OCP e4m3) forward convolution for gfx950 on the block-scaled matrix instruction
// v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 per instruction, ~2x the bf16 MFMA rate: 3.7 vs 1.9-2.0 PFLOP/s measured on
// random operands with tools/hwprobe/fp8probe) -- BASELINE configs[4] "fp8 MFMA conv path".
//
// Same implicit-GEMM shape as conv_mfma.hip (3x3 pad-1, NHWC, M = 256 output pixels per workgroup, the halo patch staged
// once per channel chunk and reused by the 9 taps, weight tiles double-buffered), with these differences:
//   * activations stay bf16 in HBM (the producer's pre-BatchNorm z, or a plain tensor); the operand STAGING applies the
//     lazy BatchNorm+ReLU, multiplies by 2^4, clamps to the e4m3 range and converts to fp8 on the way into LDS
//     (v_cvt_pk_fp8_f32).  The 2^4 pre-scale puts post-BatchNorm activations (O(1)) in the middle of e4m3's exponent
//     window and is undone for free by the instruction's E8M0 block scale (scale_a = 127 - 4);
//   * weights are packed once per step to fp8 with one power-of-two scale per output channel (max |w| of the channel
//     -> (128, 256]), undone in the epilogue;
//   * a channel chunk is 64 channels = ONE MFMA k-step; an LDS row is still 80 bytes (64 fp8 + 16 pad), so the tile
//     geometry and the conflict-free ds_read_b128 pattern of the bf16 kernel carry over (halo rows padded by 96 B);
//   * the halo tile is double-buffered in LDS and the next chunk's halo is trickled in piece by piece across the 9 taps
//     (global load issued before a tap's MFMAs, converted and written after them): the fp8 MFMAs of a tap take ~512
//     cycles per wave, which covers the load latency and the ~30 VALU of a piece's BatchNorm + conversion, and only two
//     16-byte pieces are ever in flight per thread (the bf16 kernel holds a whole halo in registers).
// fp32 accumulation; bf16 output; the epilogue (bias / BatchNorm partial statistics / folded affine + ReLU) is the
// bf16 kernel's.
// [r3] The DATA-GRADIENT runs on the same kernel (GRAD = true): dx = conv(dz, flipped weights).  Gradients need range more
// than precision, so dz is staged as e5m2 ("bf8", the MFMA's A format 1) under a per-tensor power-of-two scale taken from
// the tensor's amax of the PREVIOUS step (delayed scaling: the kernel that consumes dz also records max|dz| for the next
// step while it converts, one atomic per wave, no extra pass over the tensor); the weights are e4m3 with one scale per
// output (= the layer's input) channel; both scales are undone in the epilogue.  The weight gradient stays bf16.
#include "common.h"
#include "dtypes.h"
#include <algorithm>
#include <type_traits>
#include <vector>

namespace {
using namespace im2im;

typedef int i32x8 __attribute__((ext_vector_type(8)));

constexpr float XSCALE = 16.f;       // activation pre-scale (2^4), undone by the MFMA's E8M0 scale_a
constexpr int XSCALE_E8M0 = 127 - 4;
constexpr float FP8_MAX = 448.f;

struct Fp8ConvArgs {
  const bf16_t* x;       // [B][H][W][Ci_lo or Ci]
  const bf16_t* x_hi;    // null, or input channels [Ci_lo, Ci)
  const float* in_ss;    // [2][Ci_lo or Ci] lazy BatchNorm+ReLU of x (or null)
  const float* in_ss_hi; // [2][Ci - Ci_lo]
  const unsigned char* w;  // [Co][9][Ci] e4m3
  const float* wscale;   // [Co] power-of-two scale of each output channel's weights
  const float* bias;     // [Co] or null
  const float* scale;    // [Co] or null (EPI 2)
  const float* shift;
  bf16_t* y;             // [B][H][W][Co]
  float* stats;          // [tiles][3][Co] or null
  int B, H, W, Ci, Co, Ci_lo, tilesY, tilesX, relu;
  // [r3] channel-split result (the Up block's data-gradient: d(skip) and d(up) are separate tensors) and the gradient form
  bf16_t* y_hi;          // null, or: output channels [Co_lo, Co) go here (pixel stride Co - Co_lo), [0, Co_lo) to y
  int Co_lo;
  const float* amax_in;  // GRAD: max |x| of this tensor at the previous step (device scalar) -> the staging scale
  float* amax_out;       // GRAD: receives max |x| of this launch (atomic max of the bit pattern; zeroed by the launch before)
  float* amax_zero;      // GRAD: slot the NEXT launch will accumulate into, zeroed here
};

__device__ __forceinline__ void merge_moments_f32(float& n, float& m, float& q, float n2, float m2, float q2) {
  const float nn = n + n2;
  const float inv = nn > 0.f ? 1.f / nn : 0.f;
  const float d = m2 - m;
  q = q + q2 + d * d * (n * n2 * inv);
  m = (n * m + n2 * m2) * inv;
  n = nn;
}

// 8 floats -> 8 e4m3 bytes (saturating: cvt_pk_fp8_f32 turns out-of-range values into NaN, so clamp first)
__device__ __forceinline__ uint2 to_fp8x8(const float (&v)[8]) {
  float c[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_fmed3f(v[k], -FP8_MAX, FP8_MAX);      // one v_med3_f32 per value
  int lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}

constexpr float BF8_MAX = 57344.f;
// 8 floats -> 8 e5m2 bytes (saturating)
__device__ __forceinline__ uint2 to_bf8x8(const float (&v)[8]) {
  float c[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_fmed3f(v[k], -BF8_MAX, BF8_MAX);
  int lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], 0, false);
  lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], lo, true);
  int hi = __builtin_amdgcn_cvt_pk_bf8_f32(c[4], c[5], 0, false);
  hi = __builtin_amdgcn_cvt_pk_bf8_f32(c[6], c[7], hi, true);
  return make_uint2((unsigned)lo, (unsigned)hi);
}
// staging scale of a gradient tensor from its (previous) amax: amax * scale lands in [2^13, 2^14), a factor 3.5 under
// e5m2's largest finite value, so a step-to-step growth of the gradients by that much still does not saturate
__device__ __forceinline__ float grad_scale(float amax) {
  if (!(amax > 0.f) || !(amax < 3.0e38f)) return 1.f;
  int e;
  frexpf(amax, &e);                                  // amax = f * 2^e, f in [0.5, 1)
  return ldexpf(1.f, 14 - e);
}

// Packed fp8 weights are fragment-major like the bf16 ones (conv_common.h wfrag_index): the 32 rows x 64 reduction channels one
// v_mfma_scale_f32_32x32x64_f8f6f4 consumes are one contiguous 2 KiB block = two 1 KiB parts; part p holds bytes 16p..16p+15 of
// every lane's 32 (lane = (k / 32 % 2) * 32 + row % 32), so a wave fetches a fragment with two fully coalesced 16-byte loads.
// Blocks are ordered [row block of 32][tap][64-channel chunk].  N x 9 x K logical tensor, N % 32 == 0, K % 64 == 0.
__host__ __device__ inline size_t wfrag8_index(int n, int tap, int k, int K) {
  return ((((size_t)(n >> 5) * 9 + tap) * (K >> 6) + (k >> 6)) * 2 + ((k >> 4) & 1)) * 1024 + ((((k >> 5) & 1) * 32 + (n & 31)) * 16) + (k & 15);
}

// EPI: 0 = (+bias) store; 1 = +bias, store, BatchNorm partial statistics; 2 = folded BatchNorm affine (+ReLU)
// GRAD: data-gradient form (e5m2 operand under a run-time scale, amax bookkeeping, optional split result)
template <int TB, int TH, int TW, int BN, int WM, int WN, int EPI, bool GRAD>
__global__ __launch_bounds__(256, 2) void conv_fp8_kernel(Fp8ConvArgs a) {
  using T = bf16_t;
  constexpr int HH = TH + 2, HWD = TW + 2, HPI = HH * HWD, HPX = TB * HPI;
  constexpr int MI = TH * TW;
  constexpr int KC = 64;                              // channels per chunk = one MFMA k-step
  constexpr int ROWB = KC + 16;                       // LDS row pitch (bytes): 64 fp8 + pad
  constexpr int M = TB * TH * TW;
  constexpr int MT = M / (32 * WM), NT = BN / (32 * WN);
  static_assert(WM * WN == 4 && MT >= 1 && NT >= 1, "tile split");
  constexpr int A_PIECES = HPX * 8;                   // 16-byte global pieces (8 bf16 channels) per halo chunk
  constexpr int A_ROUNDS = (A_PIECES + 255) / 256;
  constexpr int B_ROUNDS = (BN * 4 + 255) / 256;      // 16-byte pieces (16 fp8 channels) per weight tile
  constexpr int HROWB = HWD * ROWB + 96;              // halo rows padded: conflict-free ds_read_b128 (enumerated)
  constexpr int HIMGB = HH * HROWB;
  constexpr int A_BYTES = TB * HIMGB, B_BYTES = BN * ROWB;
  static_assert(A_ROUNDS <= 18, "two rounds per tap at most");
  // weight fragments straight from L2 into registers where a fragment feeds MT = 4 MFMAs (as conv_igemm_kernel's DIRECTW): no
  // weight tile in LDS, one barrier per chunk instead of one per tap
  constexpr bool DIRECTW = MT >= 4;
  constexpr int B_TOTAL = DIRECTW ? 0 : 2 * B_BYTES;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsA = smem;                                  // two halo buffers
  char* ldsB = smem + 2 * A_BYTES;                    // two weight buffers
  float* ldsS = reinterpret_cast<float*>(smem);       // stats scratch (after the main loop)
  float* ldsSS = reinterpret_cast<float*>(smem + 2 * A_BYTES + B_TOTAL);       // [2][Ci] lazy coefficients

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  // 1-D grid, tiles dealt to the 8 XCDs in contiguous bands with the channel blocks of a tile back to back (as conv_igemm_kernel)
  int tile_id, cob;
  {
    const int ncob = a.Co / BN;
    const int lin = blockIdx.x, total = (int)gridDim.x;
    const int band = (total >> 3) / ncob;
    if (lin < band * ncob * 8) {
      const int xcd = lin & 7, j = lin >> 3;
      tile_id = xcd * band + j / ncob;
      cob = j % ncob;
    } else {
      const int r = lin - band * ncob * 8;
      tile_id = band * 8 + r / ncob;
      cob = r % ncob;
    }
  }
  int mt_id = tile_id;
  const int tx_id = mt_id % a.tilesX; mt_id /= a.tilesX;
  const int ty_id = mt_id % a.tilesY;
  const int b0 = (mt_id / a.tilesY) * TB;
  const int y0 = ty_id * TH, x0 = tx_id * TW;
  const int n0 = cob * BN;
  const bool split_in = a.x_hi != nullptr;
  const int xstride = split_in ? a.Ci_lo : a.Ci;

  int aoff[MT], boff[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 32 + l31;
    aoff[mt] = (m / MI) * HIMGB + ((m % MI) / TW) * HROWB + (m % TW) * ROWB + half * 32;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) boff[nt] = ((wn * NT + nt) * 32 + l31) * ROWB + half * 32;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  const bool lazy_lo = a.in_ss != nullptr, lazy_hi = a.in_ss_hi != nullptr;
  if (lazy_lo || lazy_hi) {
    const int clo = split_in ? a.Ci_lo : a.Ci, chi = a.Ci - clo;
    if (lazy_lo) for (int i = tid; i < clo; i += 256) { ldsSS[i] = a.in_ss[i] * XSCALE; ldsSS[a.Ci + i] = a.in_ss[clo + i] * XSCALE; }
    if (lazy_hi) for (int i = tid; i < chi; i += 256) { ldsSS[clo + i] = a.in_ss_hi[i] * XSCALE; ldsSS[a.Ci + clo + i] = a.in_ss_hi[chi + i] * XSCALE; }
    __syncthreads();
  }

  float xs = XSCALE, amax_seen = 0.f;
  if constexpr (GRAD) {
    xs = grad_scale(*a.amax_in);
    if (a.amax_zero && blockIdx.x == 0 && tid == 0) *a.amax_zero = 0.f;
  }
  // ---- halo staging, one 16-byte piece (pixel, 8 channels) at a time.  part = tid % 8 is the same for all of a thread's pieces.
  const int part = tid & 7;
  const T* __restrict__ xg_tile = a.x + (size_t)b0 * a.H * a.W * xstride;
  const T* __restrict__ xh_tile = a.x_hi + (size_t)b0 * a.H * a.W * xstride;
  // `t` is the thread id, passed through an opaque asm once per chunk so that the compiler RE-COMPUTES the ~20 integer
  // operations of a piece's addresses at its tap (there are hundreds of idle VALU slots under a tap's MFMAs) instead of
  // keeping the 2 x A_ROUNDS offsets of the whole halo live across the loop, which the 128 accumulators leave no room for
  auto piece_src = [&](int chunk, int i, int t, int& loff) -> const T* {       // null: padding (zeros); loff < 0: no such piece
    const int p = i * 256 + t;
    const int px = p >> 3;
    loff = -1;
    if (A_PIECES % 256 != 0 && px >= HPX) return nullptr;
    const int tb = px / HPI, pi = px % HPI;
    const int hy = pi / HWD, hx = pi % HWD;
    loff = tb * HIMGB + hy * HROWB + hx * ROWB + (t & 7) * 8;
    const int yy = y0 + hy - 1, xx = x0 + hx - 1;
    if (b0 + tb >= a.B || yy < 0 || yy >= a.H || xx < 0 || xx >= a.W) return nullptr;
    const int c = chunk * KC;
    const T* base = (split_in && c >= a.Ci_lo) ? xh_tile + (c - a.Ci_lo) : xg_tile + c;
    return base + ((size_t)(tb * a.H + yy) * a.W + xx) * xstride + (t & 7) * 8;
  };
#ifndef IM2IM_FP8_ABL
#define IM2IM_FP8_ABL 0
#endif
  auto load_raw = [&](const T* src) __attribute__((always_inline)) -> uint4 {
#if IM2IM_FP8_ABL & 2     // measurement-only: ... and is half as many bytes
    const uint2 h = *reinterpret_cast<const uint2*>(src);
    return make_uint4(h.x, h.y, 0u, 0u);
#else
    return *reinterpret_cast<const uint4*>(src);
#endif
  };
  // lazy coefficients are read from LDS per piece (4 x ds_read_b128; registers are the scarce resource here), pre-scaled by 2^4
  auto convert_write = [&](const uint4& raw, bool real, int loff, char* dst, int chunk) {
    if (loff < 0) return;
    uint2 q = make_uint2(0u, 0u);
#if IM2IM_FP8_ABL & 1     // measurement-only: the operand arrives as fp8 bytes from its producer -- nothing to convert (values are garbage)
    if (real) q = make_uint2(raw.x, raw.y);
    *reinterpret_cast<uint2*>(dst + loff) = q;
    return;
#endif
    if (real) {
      float v[8];
      Vec16<T>::load(reinterpret_cast<const T*>(&raw), v);
      const bool lazy_cur = (split_in && chunk * KC >= a.Ci_lo) ? lazy_hi : lazy_lo;
      if (lazy_cur) {
        const int c0 = chunk * KC + part * 8;
        const float4 s0 = *reinterpret_cast<const float4*>(ldsSS + c0), s1 = *reinterpret_cast<const float4*>(ldsSS + c0 + 4);
        const float4 h0 = *reinterpret_cast<const float4*>(ldsSS + a.Ci + c0), h1 = *reinterpret_cast<const float4*>(ldsSS + a.Ci + c0 + 4);
        const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
        const float sh[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
        // 16 * max(z*scale + shift, 0) == max(z*(16 scale) + 16 shift, 0): ldsSS holds the coefficients times 2^4 (exact)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
      } else {
        if constexpr (GRAD) {
#pragma unroll
          for (int k = 0; k < 8; ++k) amax_seen = fmaxf(amax_seen, fabsf(v[k]));
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= xs;
      }
      if constexpr (GRAD) q = to_bf8x8(v); else q = to_fp8x8(v);
    }
    *reinterpret_cast<uint2*>(dst + loff) = q;
  };

  uint4 rb[2][B_ROUNDS];
  auto gload_B = [&](uint4 (&r)[B_ROUNDS], int chunk, int tap) {
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int n = p >> 2, pt = p & 3;
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((BN * 4) % 256 == 0 || n < BN) v = *reinterpret_cast<const uint4*>(a.w + wfrag8_index(n0 + n, tap, chunk * KC + pt * 16, a.Ci));
      r[i] = v;
    }
  };
  auto swrite_B = [&](const uint4 (&r)[B_ROUNDS], int buf) {
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int n = p >> 2, pt = p & 3;
      if ((BN * 4) % 256 == 0 || n < BN) *reinterpret_cast<uint4*>(ldsB + buf * B_BYTES + n * ROWB + pt * 16) = r[i];
    }
  };
  auto mfma_tap = [&](int toff, int abuf, const i32x8 (&fb)[NT]) __attribute__((always_inline)) {
    const char* pa = ldsA + abuf * A_BYTES + toff;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const uint4 lo = *reinterpret_cast<const uint4*>(pa + aoff[mt]);
      const uint4 hi = *reinterpret_cast<const uint4*>(pa + aoff[mt] + 16);
      const i32x8 fa = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if constexpr (GRAD)      // A = e5m2 (format 1), neutral block scales: the run-time tensor scale is undone in the epilogue
          acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb[nt], acc[mt][nt], 1, 0, 0, 127, 0, 127);
        else
          acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb[nt], acc[mt][nt], 0, 0, 0, XSCALE_E8M0, 0, 127);
    }
  };
  // DIRECTW: this wave's NT weight fragments of one (tap, chunk), from the fragment-major pack
  const size_t cob_stride8 = (size_t)9 * (a.Ci >> 6) * 2048;
  const unsigned char* __restrict__ wfr = a.w + (size_t)(n0 / 32 + wn * NT) * cob_stride8 + lane * 16;
  auto gload_F = [&](i32x8 (&f)[NT], int chunk, int tap) __attribute__((always_inline)) {
    const unsigned char* p = wfr + ((size_t)(tap * (a.Ci >> 6) + chunk) << 11);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const uint4 lo = *reinterpret_cast<const uint4*>(p + nt * cob_stride8);
      const uint4 hi = *reinterpret_cast<const uint4*>(p + nt * cob_stride8 + 1024);
      f[nt] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
  };
  auto compute = [&](int toff, int bbuf, int abuf) {
    const char* pa = ldsA + abuf * A_BYTES + toff;
    const char* pb = ldsB + bbuf * B_BYTES;
    i32x8 fb[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const uint4 lo = *reinterpret_cast<const uint4*>(pb + boff[nt]);
      const uint4 hi = *reinterpret_cast<const uint4*>(pb + boff[nt] + 16);
      fb[nt] = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
    }
    // one A fragment (8 registers) at a time, reused by the NT weight fragments: the 128 accumulators leave no room for four
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const uint4 lo = *reinterpret_cast<const uint4*>(pa + aoff[mt]);
      const uint4 hi = *reinterpret_cast<const uint4*>(pa + aoff[mt] + 16);
      const i32x8 fa = i32x8{(int)lo.x, (int)lo.y, (int)lo.z, (int)lo.w, (int)hi.x, (int)hi.y, (int)hi.z, (int)hi.w};
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        if constexpr (GRAD)      // A = e5m2 (format 1), neutral block scales: the run-time tensor scale is undone in the epilogue
          acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb[nt], acc[mt][nt], 1, 0, 0, 127, 0, 127);
        else
          acc[mt][nt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa, fb[nt], acc[mt][nt], 0, 0, 0, XSCALE_E8M0, 0, 127);
    }
  };

  const int nchunks = a.Ci / KC;
  // prologue: the whole first halo -- all its loads in flight at once (the accumulators are not live yet), then converted
  {
    uint4 raw[A_ROUNDS];
    int loffs[A_ROUNDS];
    bool real[A_ROUNDS];
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
      const T* src = piece_src(0, i, tid, loffs[i]);
      real[i] = src != nullptr;
      raw[i] = make_uint4(0, 0, 0, 0);
      if (real[i]) raw[i] = load_raw(src);
    }
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) convert_write(raw[i], real[i], loffs[i], ldsA, 0);
  }
  i32x8 fw[2][NT];                                   // DIRECTW: this tap's / the next tap's weight fragments
  if constexpr (DIRECTW) gload_F(fw[0], 0, 0);
  else { gload_B(rb[0], 0, 0); gload_B(rb[1], 0, 1); }
  auto chunk_body = [&](auto parity, int chunk) {
    constexpr int P0 = decltype(parity)::value;
    const bool more = chunk + 1 < nchunks;
    const int abuf = chunk & 1;
    char* nextA = ldsA + (abuf ^ 1) * A_BYTES;
    int tid_o = tid;
    asm volatile("" : "+v"(tid_o));
    if constexpr (DIRECTW) __syncthreads();            // the halo written during the previous chunk (or the prologue) is visible, its predecessor free
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int set = (P0 + tap) & 1;
      if constexpr (DIRECTW) {
        if (tap + 1 < 9) { if (set) gload_F(fw[0], chunk, tap + 1); else gload_F(fw[1], chunk, tap + 1); }
        else if (more) { if (set) gload_F(fw[0], chunk + 1, 0); else gload_F(fw[1], chunk + 1, 0); }
      } else {
      if (set) swrite_B(rb[1], 1); else swrite_B(rb[0], 0);
      __syncthreads();                                 // weight tile `set` (and, at tap 0, the halo written during the previous chunk) visible
      if (tap + 2 < 9) { if (set) gload_B(rb[1], chunk, tap + 2); else gload_B(rb[0], chunk, tap + 2); }
      else if (more) { if (set) gload_B(rb[1], chunk + 1, tap + 2 - 9); else gload_B(rb[0], chunk + 1, tap + 2 - 9); }
      }
      // trickle the next chunk's halo: rounds {tap, tap + 9} are loaded before this tap's MFMAs and written after them
      uint4 raw0 = make_uint4(0, 0, 0, 0), raw1 = make_uint4(0, 0, 0, 0);
      int loff0 = -1, loff1 = -1;
      bool real0 = false, real1 = false;
      if (more) {
        if (tap < A_ROUNDS) {
          const T* s0 = piece_src(chunk + 1, tap, tid_o, loff0);
          real0 = s0 != nullptr;
          if (real0) raw0 = load_raw(s0);
        }
        if (tap + 9 < A_ROUNDS) {
          const T* s1 = piece_src(chunk + 1, tap + 9, tid_o, loff1);
          real1 = s1 != nullptr;
          if (real1) raw1 = load_raw(s1);
        }
      }
      if constexpr (DIRECTW) { if (set) mfma_tap((tap / 3) * HROWB + (tap % 3) * ROWB, abuf, fw[1]); else mfma_tap((tap / 3) * HROWB + (tap % 3) * ROWB, abuf, fw[0]); }
      else compute((tap / 3) * HROWB + (tap % 3) * ROWB, set, abuf);
      if (more) {
        if (tap < A_ROUNDS) convert_write(raw0, real0, loff0, nextA, chunk + 1);
        if (tap + 9 < A_ROUNDS) convert_write(raw1, real1, loff1, nextA, chunk + 1);
      }
    }
  };
  for (int chunk = 0; chunk < nchunks; chunk += 2) {
    chunk_body(std::integral_constant<int, 0>{}, chunk);
    if (chunk + 1 < nchunks) chunk_body(std::integral_constant<int, 1>{}, chunk + 1);
  }

  // ---------------------------------------------------------------- epilogue (as conv_igemm_kernel, T = bf16)
  constexpr int EPP = 8;
  constexpr int WROWS = MT * 32, WCOLS = NT * 32;
  constexpr int WP = WCOLS * 2 + 16;
  constexpr int WBYTES = WROWS * WP;
  constexpr int EPR = WCOLS / EPP;
  constexpr int ROWS_PER_PASS = 64 / EPR;
  constexpr int PASSES = WROWS / ROWS_PER_PASS;
  const int ncol = n0 + wn * WCOLS;
  const bool to_hi = a.y_hi != nullptr && ncol >= a.Co_lo;
  T* __restrict__ yg = (to_hi ? a.y_hi : a.y) + (to_hi ? ncol - a.Co_lo : ncol);
  const int ystride = a.y_hi == nullptr ? a.Co : (to_hi ? a.Co - a.Co_lo : a.Co_lo);
  constexpr bool want_stats = (EPI == 1);
  if constexpr (GRAD) {
    if (a.amax_out) {                                  // max over the wave, one atomic per wave (positive floats order like their bits)
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) amax_seen = fmaxf(amax_seen, __shfl_xor(amax_seen, off, 64));
      // tens of thousands of waves hit ONE word: only those that would raise it pay for the atomic (a single address
      // serialises at ~90 atomics/us -- unconditionally this was 2/3 of the kernel's time); a stale read only costs an atomic
      if (lane == 0 && __float_as_uint(amax_seen) > __hip_atomic_load(reinterpret_cast<unsigned*>(a.amax_out), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))
        atomicMax(reinterpret_cast<unsigned*>(a.amax_out), __float_as_uint(amax_seen));
    }
  }
  const float inv_xs = GRAD ? 1.f / xs : 1.f;          // exact: xs is a power of two
  __syncthreads();
  char* wbuf = smem + wave * WBYTES;
  float st_n[NT], st_m[NT], st_q[NT];
  // as conv_igemm_kernel: a tile inside the batch and the image needs no per-value validity test (wave-uniform fast path);
  // an overhanging one keeps one validity bit per accumulator row
  const bool tile_full = (b0 + TB <= a.B) && (y0 + TH <= a.H) && (x0 + TW <= a.W);
  static_assert(MT * 16 <= 64, "one validity bit per accumulator row of the lane");
  unsigned long long okmask = ~0ull;
  float cnt = (float)(16 * MT);
  if constexpr (want_stats) {
    if (!tile_full) {
      okmask = 0ull;
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = wm * WROWS + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          const int bb = b0 + m / MI, yy = y0 + (m % MI) / TW, xx = x0 + m % TW;
          if (bb < a.B && yy < a.H && xx < a.W) okmask |= 1ull << (mt * 16 + r);
        }
      cnt = (float)__popcll(okmask);
    }
  }
  auto convert_tile = [&](auto full_tag) {
  constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int n = n0 + (wn * NT + nt) * 32 + l31;
    const float ws = a.wscale[n] * inv_xs;
    const float bias_v = a.bias ? a.bias[n] : 0.f;
    float sc2 = 1.f, sh2 = 0.f;
    if constexpr (EPI == 2) { sc2 = a.scale[n]; sh2 = a.shift[n]; }
    float s = 0.f, sq = 0.f;
    const float K = to_float(from_float<T>(acc[0][nt][0] * ws + bias_v));
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float v = acc[mt][nt][r] * ws + bias_v;
        if constexpr (EPI == 2) {
          v = v * sc2 + sh2;
          if (a.relu) v = fmaxf(v, 0.f);
        }
        const T tv = from_float<T>(v);
        *reinterpret_cast<T*>(wbuf + row * WP + (nt * 32 + l31) * 2) = tv;
        if constexpr (want_stats) {
          float d = to_float(tv) - K;
          if constexpr (!FULL) d = ((okmask >> (mt * 16 + r)) & 1ull) ? d : 0.f;
          s += d; sq += d * d;
        }
      }
    }
    if constexpr (want_stats) {
      const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
      st_n[nt] = cnt; st_m[nt] = K + s * inv; st_q[nt] = fmaxf(sq - s * s * inv, 0.f);
    }
  }
  };
  if (want_stats && tile_full) convert_tile(std::true_type{}); else convert_tile(std::false_type{});
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    const int row = pass * ROWS_PER_PASS + lane / EPR;
    const int piece = lane % EPR;
    const uint4 v = *reinterpret_cast<const uint4*>(wbuf + row * WP + piece * 16);
    const int m = wm * WROWS + row;
    const int bb = b0 + m / MI, yy = y0 + (m % MI) / TW, xx = x0 + m % TW;
    if (bb < a.B && yy < a.H && xx < a.W)
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(yg + (((size_t)bb * a.H + yy) * a.W + xx) * ystride + piece * EPP));
  }
  if (want_stats) {
    __syncthreads();
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nl = (wn * NT + nt) * 32 + l31;
      float n = st_n[nt], m = st_m[nt], q = st_q[nt];
      merge_moments_f32(n, m, q, __shfl_xor(n, 32, 64), __shfl_xor(m, 32, 64), __shfl_xor(q, 32, 64));
      if (half == 0) { ldsS[(wm * BN + nl) * 3 + 0] = n; ldsS[(wm * BN + nl) * 3 + 1] = m; ldsS[(wm * BN + nl) * 3 + 2] = q; }
    }
    __syncthreads();
    if (tid < BN) {
      float n = ldsS[tid * 3 + 0], m = ldsS[tid * 3 + 1], q = ldsS[tid * 3 + 2];
#pragma unroll
      for (int i = 1; i < WM; ++i)
        merge_moments_f32(n, m, q, ldsS[(i * BN + tid) * 3 + 0], ldsS[(i * BN + tid) * 3 + 1], ldsS[(i * BN + tid) * 3 + 2]);
      float* st = a.stats + (size_t)tile_id * 3 * a.Co;
      st[n0 + tid] = m;
      st[a.Co + n0 + tid] = q;
      st[2 * a.Co + n0 + tid] = n;
    }
  }
}

// w [Co][Ci][9] fp32 -> wq [Co][9][Ci] e4m3 with one power-of-two scale per output channel: max |w| -> (128, 256].  One block per
// output channel `co` of a [Co][Ci][taps] tensor.
__device__ __forceinline__ void pack_fp8_channel(const float* __restrict__ w, int Co, int Ci, int taps, int co,
                                                 unsigned char* __restrict__ wq, float* __restrict__ wscale, float* s_max) {
  const int per = Ci * taps;
  const float* wc = w + (size_t)co * per;
  float m = 0.f;
  for (int i = threadIdx.x; i < per; i += 256) m = fmaxf(m, fabsf(wc[i]));
  s_max[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + off]);
    __syncthreads();
  }
  const float amax = s_max[0];
  int e = 0;
  if (amax > 0.f) { frexpf(amax, &e); }                      // amax = f * 2^e, f in [0.5, 1)
  const float scale = amax > 0.f ? ldexpf(1.f, e - 8) : 1.f;  // amax / scale in [128, 256)
  const float inv = 1.f / scale;
  if (threadIdx.x == 0) wscale[co] = scale;
  for (int i = threadIdx.x; i < per; i += 256) {
    const int ci = i / taps, tp = i % taps;                  // source index (ci, tap)
    const float v = fminf(fmaxf(wc[i] * inv, -FP8_MAX), FP8_MAX);
    const int q = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
    if (taps == 9 && Ci % 64 == 0 && Co % 32 == 0) wq[wfrag8_index(co, tp, ci, Ci)] = (unsigned char)(q & 0xff);   // fragment-major
    else wq[((size_t)co * taps + tp) * Ci + ci] = (unsigned char)(q & 0xff);
  }
}
__global__ __launch_bounds__(256) void pack_weight_fp8_kernel(const float* __restrict__ w, int Ci, int taps,
                                                               unsigned char* __restrict__ wq, float* __restrict__ wscale) {
  __shared__ float s_max[256];
  pack_fp8_channel(w, (int)gridDim.x, Ci, taps, blockIdx.x, wq, wscale, s_max);
}

// data-gradient operand: w [Co][Ci][9] fp32 -> wq [Ci][9 taps, reversed][Co] e4m3 with one power-of-two scale per INPUT channel
// (the data-gradient's output channel): block = one ci
__device__ __forceinline__ void pack_fp8_dgrad_channel(const float* __restrict__ w, int Co, int Ci, int taps, int ci,
                                                       unsigned char* __restrict__ wq, float* __restrict__ wscale, float* s_max) {
  const int per = Co * taps;
  float m = 0.f;
  for (int i = threadIdx.x; i < per; i += 256) { const int co = i / taps, tp = i % taps; m = fmaxf(m, fabsf(w[((size_t)co * Ci + ci) * taps + tp])); }
  s_max[threadIdx.x] = m;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) s_max[threadIdx.x] = fmaxf(s_max[threadIdx.x], s_max[threadIdx.x + off]);
    __syncthreads();
  }
  const float amax = s_max[0];
  int e = 0;
  if (amax > 0.f) { frexpf(amax, &e); }
  const float scale = amax > 0.f ? ldexpf(1.f, e - 8) : 1.f;
  const float inv = 1.f / scale;
  if (threadIdx.x == 0) wscale[ci] = scale;
  for (int i = threadIdx.x; i < per; i += 256) {
    const int tp = i / Co, co = i % Co;                     // destination order (tap, co): coalesced bytes
    const float v = fminf(fmaxf(w[((size_t)co * Ci + ci) * taps + tp] * inv, -FP8_MAX), FP8_MAX);
    const int q = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
    if (taps == 9 && Co % 64 == 0 && Ci % 32 == 0) wq[wfrag8_index(ci, taps - 1 - tp, co, Co)] = (unsigned char)(q & 0xff);
    else wq[((size_t)ci * taps + (taps - 1 - tp)) * Co + co] = (unsigned char)(q & 0xff);
  }
}
__global__ __launch_bounds__(256) void pack_weight_fp8_dgrad_kernel(const float* __restrict__ w, int Co, int Ci, int taps,
                                                                     unsigned char* __restrict__ wq, float* __restrict__ wscale) {
  __shared__ float s_max[256];
  pack_fp8_dgrad_channel(w, Co, Ci, taps, blockIdx.x, wq, wscale, s_max);
}

// [r4] every fp8 operand of a model in ONE launch per kind (forward / data-gradient): the 17 + 14 per-layer pack launches of an
// fp8-mode training step were 0.46 ms of a 20 ms step.  Block = one channel of one tensor, found through the prefix table.
constexpr int FP8_PACK_MAX = 32;
struct Fp8PackMultiArgs {
  const float* w[FP8_PACK_MAX]; unsigned char* wq[FP8_PACK_MAX]; float* ws[FP8_PACK_MAX];
  int Co[FP8_PACK_MAX], Ci[FP8_PACK_MAX];
  int start[FP8_PACK_MAX + 1];       // prefix sums of the tensors' channel counts (Co forward, Ci data-gradient)
  int n;
};
template <bool DGRAD>
__global__ __launch_bounds__(256) void pack_weight_fp8_multi_kernel(Fp8PackMultiArgs a) {
  __shared__ float s_max[256];
  int t = 0;
  while (t + 1 < a.n && (int)blockIdx.x >= a.start[t + 1]) ++t;
  const int c = blockIdx.x - a.start[t];
  if constexpr (DGRAD) pack_fp8_dgrad_channel(a.w[t], a.Co[t], a.Ci[t], 9, c, a.wq[t], a.ws[t], s_max);
  else pack_fp8_channel(a.w[t], a.Co[t], a.Ci[t], 9, c, a.wq[t], a.ws[t], s_max);
}

// Tiled form for the fragment-major operands (N % 32 == 0, K % 64 == 0: every fp8 layer of the UNet), two launches per kind.
// PHASE 1, block = 8 (forward) / 16 (data-gradient) rows of one tensor: streams them once with coalesced float4 loads -> their scales.
// PHASE 2, block = (32-row block, 64-k chunk): re-reads (L2 / Infinity Cache) a [32 rows][64 k][9 taps] tile through LDS and writes
// every fragment part as 16-byte pieces of a contiguous 1 KiB run; `start` counts tiles (N / 32 * K / 64 per tensor).  The per-channel kernels above store single scattered bytes (and the data-gradient one gathers 36-byte runs): 0.18 +
// 0.28 ms per fp8-mode step on 31 M weights; same arithmetic per element, so the bytes and scales are identical.
//   element (n, tap, k) of the operand = w[n * sn + k * sk + tap]: forward n = co, k = ci (sn = 9 Ci, sk = 9), data-gradient
//   n = ci, k = co (sn = 9, sk = 9 Ci), taps reversed on output.
template <bool DGRAD, int PHASE>
__global__ __launch_bounds__(DGRAD ? 576 : 512) void pack_weight_fp8_tiled_kernel(Fp8PackMultiArgs a) {
  constexpr int NT = DGRAD ? 576 : 512;
  constexpr int PITCH = DGRAD ? 289 : 577;            // odd pitches: the gathers below are bank-conflict free
  extern __shared__ float tile[];                      // forward [32 n][577], data-gradient [64 k][289]
  __shared__ __attribute__((aligned(16))) float s_amax[PHASE == 1 && DGRAD ? 16 * 144 : 4];
  __shared__ float s_inv[32];
  int t = 0;
  while (t + 1 < a.n && (int)blockIdx.x >= a.start[t + 1]) ++t;
  const int Co = a.Co[t], Ci = a.Ci[t];
  const int K = DGRAD ? Co : Ci;
  const int rb = PHASE == 1 ? blockIdx.x - a.start[t] : (blockIdx.x - a.start[t]) / (K >> 6);
  const float* __restrict__ w = a.w[t];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n0 = rb * 32;
  // ---- phase 1: |w| maxima of the 32 rows -> scales
  if constexpr (PHASE == 1) {
    // block = P1_ROWS rows of one tensor (`start` counts those): forward 8 rows, one wave per (contiguous) row; data-gradient 16 rows
    // (input channels): per co their 144 values are contiguous -> thread = (float4 of the run, co mod 16)
    if constexpr (!DGRAD) {
      const int per4 = Ci * 9 / 4;
      const float4* row = reinterpret_cast<const float4*>(w + (size_t)(rb * 8 + wave) * Ci * 9);
      float m = 0.f;
#pragma unroll 4
      for (int i = lane; i < per4; i += 64) {
        const float4 v = row[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
      }
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
      if (lane == 0) {
        int e = 0;
        if (m > 0.f) { frexpf(m, &e); }
        a.ws[t][rb * 8 + wave] = m > 0.f ? ldexpf(1.f, e - 8) : 1.f;
      }
    } else {
      float4* s_part = reinterpret_cast<float4*>(s_amax);      // [16 co groups][36 float4]
      const int pos = tid % 36, grp = tid / 36;
      const float4* base = reinterpret_cast<const float4*>(w + (size_t)rb * 16 * 9) + pos;
      const size_t stride4 = (size_t)Ci * 9 / 4;
      float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
      for (int co = grp; co < Co; co += 16) {
        const float4 v = base[(size_t)co * stride4];
        m.x = fmaxf(m.x, fabsf(v.x)); m.y = fmaxf(m.y, fabsf(v.y)); m.z = fmaxf(m.z, fabsf(v.z)); m.w = fmaxf(m.w, fabsf(v.w));
      }
      s_part[grp * 36 + pos] = m;
      __syncthreads();
      if (tid < 16) {
        float r = 0.f;
        for (int g = 0; g < 16; ++g)
          for (int j = 0; j < 9; ++j) r = fmaxf(r, s_amax[g * 144 + tid * 9 + j]);
        int e = 0;
        if (r > 0.f) { frexpf(r, &e); }
        a.ws[t][rb * 16 + tid] = r > 0.f ? ldexpf(1.f, e - 8) : 1.f;
      }
    }
  } else {
    // ---- phase 2: one 64-k tile through LDS -> fragment parts
    if (tid < 32) s_inv[tid] = 1.f / a.ws[t][n0 + tid];
    unsigned char* out = a.wq[t] + (size_t)rb * 9 * (K >> 6) * 2048;
    const int chunk = (blockIdx.x - a.start[t]) % (K >> 6);
    const int k0 = chunk * 64;
    for (int i = tid; i < 4608; i += NT) {             // 4608 float4 = 32 x 64 x 9 floats
      if constexpr (!DGRAD) {
        const int r = i / 144, c = (i % 144) * 4;      // row r: 576 contiguous floats (64 ci x 9 taps)
        const float4 v = *reinterpret_cast<const float4*>(w + ((size_t)(n0 + r) * Ci + k0) * 9 + c);
        float* d = tile + r * PITCH + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      } else {
        const int kk = i / 72, c = (i % 72) * 4;       // co = k0 + kk: 288 contiguous floats (32 ci x 9 taps)
        const float4 v = *reinterpret_cast<const float4*>(w + ((size_t)(k0 + kk) * Ci + n0) * 9 + c);
        float* d = tile + kk * PITCH + c;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
      }
    }
    __syncthreads();
    for (int piece = tid; piece < 9 * 128; piece += NT) {
      const int tap = piece >> 7, part = (piece >> 6) & 1, l = piece & 63;
      const int r = l & 31, kb = (l >> 5) * 32 + part * 16;        // 16 consecutive k of row r
      const float inv = s_inv[r];
      unsigned q[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int kk = kb + j * 4 + u;
          const float x = DGRAD ? tile[kk * PITCH + r * 9 + tap] : tile[r * PITCH + kk * 9 + tap];
          v[u] = fminf(fmaxf(x * inv, -FP8_MAX), FP8_MAX);
        }
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], pk, true);
        q[j] = (unsigned)pk;
      }
      const int tap_out = DGRAD ? 8 - tap : tap;
      *reinterpret_cast<uint4*>(out + ((size_t)(tap_out * (K >> 6) + chunk) * 2 + part) * 1024 + l * 16) = make_uint4(q[0], q[1], q[2], q[3]);
    }
  }
}

template <int TB, int TH, int TW, int BN, int WM, int WN, int EPI, bool GRAD = false>
int launch_fp8(const Fp8ConvArgs& a_in, hipStream_t stream) {
  Fp8ConvArgs a = a_in;
  a.tilesY = (int)cdiv(a.H, TH);
  a.tilesX = (int)cdiv(a.W, TW);
  constexpr int ROWB = 80;
  constexpr bool DIRECTW = TB * TH * TW / (32 * WM) >= 4;                       // as in the kernel: no weight tile in LDS
  constexpr size_t smem_main = (size_t)2 * TB * (TH + 2) * ((TW + 2) * ROWB + 96) + (DIRECTW ? 0 : (size_t)2 * BN * ROWB);
  constexpr size_t smem_epi = (size_t)4 * (TB * TH * TW / WM) * ((BN / WN) * 2 + 16);
  const size_t smem_in = smem_main + ((a.in_ss || a.in_ss_hi) ? (size_t)2 * a.Ci * sizeof(float) : 0);
  const size_t smem = smem_in > smem_epi ? smem_in : smem_epi;
  auto kern = conv_fp8_kernel<TB, TH, TW, BN, WM, WN, EPI, GRAD>;
  static size_t attr_set = 0;
  if (smem > 64 * 1024 && smem > attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = smem;
  }
  dim3 grid((unsigned)((size_t)cdiv(a.B, TB) * a.tilesY * a.tilesX * (a.Co / BN)));
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a);
  return check_launch("conv_fp8_kernel");
}

template <int TB, int TH, int TW, int BN, int WM, int WN>
int launch_fp8_epi(const Fp8ConvArgs& a, hipStream_t stream) {
  if (a.amax_in) return launch_fp8<TB, TH, TW, BN, WM, WN, 0, true>(a, stream);
  if (a.stats) return launch_fp8<TB, TH, TW, BN, WM, WN, 1>(a, stream);
  if (a.scale) return launch_fp8<TB, TH, TW, BN, WM, WN, 2>(a, stream);
  return launch_fp8<TB, TH, TW, BN, WM, WN, 0>(a, stream);
}

}  // namespace

// small-extent layers (H or W < 64): 8x8 patches of IM2IM_FP8_SMALL_TB consecutive images per tile.  Two images (M = 128): 56 KB of LDS,
// two workgroups per CU -- measured 859 -> 1,186 TF at 40x40 512->512 against four images (92 KB, one workgroup per CU)
#ifndef IM2IM_FP8_SMALL_TB
#define IM2IM_FP8_SMALL_TB 2
#endif
namespace {
int dispatch_fp8(const Fp8ConvArgs& a, hipStream_t stream) {
  const bool small = (a.H < 64 || a.W < 64);
  const bool wide = a.Co % 128 == 0;
  if (!small) return wide ? launch_fp8_epi<1, 16, 16, 128, 2, 2>(a, stream) : launch_fp8_epi<1, 16, 16, 64, 4, 1>(a, stream);
  return wide ? launch_fp8_epi<IM2IM_FP8_SMALL_TB, 8, 8, 128, 2, 2>(a, stream) : launch_fp8_epi<IM2IM_FP8_SMALL_TB, 8, 8, 64, (IM2IM_FP8_SMALL_TB == 4 ? 4 : 2), (IM2IM_FP8_SMALL_TB == 4 ? 1 : 2)>(a, stream);
}
}  // namespace

extern "C" int64_t im2im_conv_fp8_stats_rows(int32_t B, int32_t H, int32_t W) {
  const bool small = (H < 64 || W < 64);
  return small ? im2im::cdiv(B, IM2IM_FP8_SMALL_TB) * im2im::cdiv(H, 8) * im2im::cdiv(W, 8) : (int64_t)B * im2im::cdiv(H, 16) * im2im::cdiv(W, 16);
}

extern "C" int im2im_pack_conv_weight_fp8(const float* w, int32_t Co, int32_t Ci, int32_t taps, void* wq, float* wscale,
                                          im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(w && wq && wscale && Co > 0 && Ci > 0 && taps > 0);
  hipLaunchKernelGGL(pack_weight_fp8_kernel, dim3((unsigned)Co), dim3(256), 0, stream, w, (int)Ci, (int)taps, (unsigned char*)wq, wscale);
  return im2im::check_launch("pack_weight_fp8_kernel");
}

extern "C" int im2im_pack_conv_weights_fp8_multi(int32_t n_tensors, const float* const* w, const int32_t* Co, const int32_t* Ci,
                                                 void* const* wq, float* const* wscale, int32_t dgrad, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (w && Co && Ci && wq && wscale)));
  // fragment-major tensors (every fp8 layer of the UNet) -> tiled kernel, heaviest row blocks first; the rest -> block per channel
  std::vector<int> tiled, plain;
  for (int i = 0; i < n_tensors; ++i) {
    IM2IM_REQUIRE(w[i] && wq[i] && wscale[i] && Co[i] > 0 && Ci[i] > 0);
    const int N = dgrad ? Ci[i] : Co[i], K = dgrad ? Co[i] : Ci[i];
    (N % 32 == 0 && K % 64 == 0 ? tiled : plain).push_back(i);
  }
  std::stable_sort(tiled.begin(), tiled.end(), [&](int x, int y) { return (dgrad ? Co[x] : Ci[x]) > (dgrad ? Co[y] : Ci[y]); });
  for (int pass = 0; pass < 2; ++pass) {
    const std::vector<int>& idx = pass == 0 ? tiled : plain;
    for (size_t base = 0; base < idx.size(); base += FP8_PACK_MAX) {
      Fp8PackMultiArgs a, a2;
      a.n = (int)std::min<size_t>(FP8_PACK_MAX, idx.size() - base);
      int blocks = 0, tiles = 0;
      for (int i = 0; i < a.n; ++i) {
        const int s = idx[base + i];
        a.w[i] = w[s]; a.wq[i] = (unsigned char*)wq[s]; a.ws[i] = wscale[s];
        a.Co[i] = Co[s]; a.Ci[i] = Ci[s];
        a.start[i] = blocks;
        const int N = dgrad ? Ci[s] : Co[s], K = dgrad ? Co[s] : Ci[s];
        blocks += pass == 0 ? N / (dgrad ? 16 : 8) : N;      // phase 1: 16 (data-gradient) / 8 rows per block; plain: one per block
        tiles += (N / 32) * (K / 64);
      }
      a.start[a.n] = blocks;
      if (pass == 0) {
        a2 = a;
        for (int i = 0, at = 0; i <= a.n; ++i) {
          a2.start[i] = at;
          if (i < a.n) at += ((dgrad ? a.Ci[i] : a.Co[i]) / 32) * ((dgrad ? a.Co[i] : a.Ci[i]) / 64);
        }
        static bool attr_set = false;
        if (!attr_set) {
          hipFuncSetAttribute(reinterpret_cast<const void*>(pack_weight_fp8_tiled_kernel<false, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 32 * 577 * 4);
          hipFuncSetAttribute(reinterpret_cast<const void*>(pack_weight_fp8_tiled_kernel<true, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 289 * 4);
          attr_set = true;
        }
        if (dgrad) {
          hipLaunchKernelGGL((pack_weight_fp8_tiled_kernel<true, 1>), dim3((unsigned)blocks), dim3(576), 0, stream, a);
          hipLaunchKernelGGL((pack_weight_fp8_tiled_kernel<true, 2>), dim3((unsigned)tiles), dim3(576), 64 * 289 * 4, stream, a2);
        } else {
          hipLaunchKernelGGL((pack_weight_fp8_tiled_kernel<false, 1>), dim3((unsigned)blocks), dim3(512), 0, stream, a);
          hipLaunchKernelGGL((pack_weight_fp8_tiled_kernel<false, 2>), dim3((unsigned)tiles), dim3(512), 32 * 577 * 4, stream, a2);
        }
        if (int rc = im2im::check_launch("pack_weight_fp8_tiled_kernel")) return rc;
      } else {
        if (dgrad) hipLaunchKernelGGL(pack_weight_fp8_multi_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        else hipLaunchKernelGGL(pack_weight_fp8_multi_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, stream, a);
        if (int rc = im2im::check_launch("pack_weight_fp8_multi_kernel")) return rc;
      }
    }
  }
  return IM2IM_OK;
}

extern "C" int im2im_conv_fwd_fp8(const void* x, const float* in_scale_shift, const void* x_hi, const float* in_scale_shift_hi,
                                  int32_t Ci_lo, const void* wq, const float* wscale, const float* bias, const float* scale,
                                  const float* shift, void* y, float* stats, int32_t B, int32_t H, int32_t W, int32_t Ci,
                                  int32_t Co, int32_t relu, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(x && wq && wscale && y);
  if (x_hi) {
    IM2IM_REQUIRE(Ci_lo > 0 && Ci_lo % 64 == 0 && Ci == 2 * Ci_lo);
  } else {
    IM2IM_REQUIRE(in_scale_shift_hi == nullptr);
    Ci_lo = Ci;
  }
  IM2IM_REQUIRE(B > 0 && H > 0 && W > 0);
  IM2IM_REQUIRE(Ci > 0 && Ci % 64 == 0 && Ci <= 2048);
  IM2IM_REQUIRE(Co > 0 && Co % 64 == 0);
  IM2IM_REQUIRE((scale == nullptr) == (shift == nullptr));
  IM2IM_REQUIRE(!(stats && scale));
  Fp8ConvArgs a{(const bf16_t*)x, (const bf16_t*)x_hi, in_scale_shift, in_scale_shift_hi, (const unsigned char*)wq, wscale, bias,
                scale, shift, (bf16_t*)y, stats, B, H, W, Ci, Co, Ci_lo, 0, 0, relu, nullptr, Co, nullptr, nullptr, nullptr};
  return dispatch_fp8(a, stream);
}

extern "C" int im2im_pack_conv_weight_fp8_dgrad(const float* w, int32_t Co, int32_t Ci, int32_t taps, void* wq, float* wscale,
                                                im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(w && wq && wscale && Co > 0 && Ci > 0 && taps > 0);
  hipLaunchKernelGGL(pack_weight_fp8_dgrad_kernel, dim3((unsigned)Ci), dim3(256), 0, stream, w, (int)Co, (int)Ci, (int)taps,
                     (unsigned char*)wq, wscale);
  return im2im::check_launch("pack_weight_fp8_dgrad_kernel");
}

extern "C" int im2im_conv_dgrad_fp8(const void* dz, const void* wq_d, const float* wscale_d, void* dx, void* dx_hi, int32_t Cx_lo,
                                    const float* amax_prev, float* amax_now, float* amax_next, int32_t B, int32_t H, int32_t W,
                                    int32_t Cz, int32_t Cx, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(dz && wq_d && wscale_d && dx && amax_prev);
  IM2IM_REQUIRE(B > 0 && H > 0 && W > 0);
  IM2IM_REQUIRE(Cz > 0 && Cz % 64 == 0 && Cz <= 2048);
  IM2IM_REQUIRE(Cx > 0 && Cx % 64 == 0);
  if (dx_hi) IM2IM_REQUIRE(Cx_lo > 0 && Cx_lo % 64 == 0 && Cx_lo < Cx && (Cx - Cx_lo) % 64 == 0);
  else Cx_lo = Cx;
  Fp8ConvArgs a{(const bf16_t*)dz, nullptr, nullptr, nullptr, (const unsigned char*)wq_d, wscale_d, nullptr, nullptr, nullptr,
                (bf16_t*)dx, nullptr, B, H, W, Cz, Cx, Cz, 0, 0, 0, (bf16_t*)dx_hi, Cx_lo, amax_prev, amax_now, amax_next};
  return dispatch_fp8(a, stream);
}
