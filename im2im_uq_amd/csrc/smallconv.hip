// Direct (VALU) 3x3 convolutions where one side has only a handful of channels -- memory-bound layers
// that would waste an MFMA tile (SURVEY K1 "first conv", K7 heads):
//   s2l : fp32 NCHW, CS<=8 channels  ->  NHWC T, CL in {32,64} channels
//         = first UNet conv (core/models/trunks/unet_parts.py:16 with Cin = n_in) incl. BatchNorm partial
//           statistics / folded eval BN + ReLU, and the data-gradient of the quantile heads;
//   l2s : NHWC T, CL channels -> fp32 NCHW planes, CS<=8 channels
//         = the three quantile heads written straight into the [B,3,C,H,W] output
//           (core/models/finallayers/quantile_layer.py:15-17,20);
//   wgrad : out[s][tap][l] = sum_px L[px][l] * S[s][px+tap]  -> weight gradients of both.
// 16x16-pixel tiles.  s2l: one thread per (pixel, 8-channel group) so a wave stores whole NHWC rows; l2s / wgrad: one
// thread per pixel / per wide-side channel with the small side's weights wave-uniform (scalar loads).
#include "common.h"
#include "dtypes.h"
#include "reduce.h"
#include <cstdlib>

namespace {
using namespace im2im;

constexpr int TS = 16;                 // tile side
constexpr int HS = TS + 2;             // halo side
constexpr int CS_MAX = 8;

struct S2LArgs {
  const float* in;      // [B][CS][H][W]
  const float* w;       // [CS][9][CL]
  const float* bias;    // [CL] | null
  const float* center;  // [CL] | null (subtracted from the stored output)
  const float* scale_shift;   // [2][CL] | null
  void* out;            // [B][H][W][CL] T
  float* stats;         // [blk][3][CL] | null: per-tile (mean, M2, count), see conv_mfma.hip
  int B, H, W, CS, tilesY, tilesX, relu, flip;
};

// Thread = (pixel, group of 8 output channels): the CL/8 lanes of one pixel write 16 B each, so a wave stores whole
// NHWC rows (8 x-adjacent pixels x 128 B for CL = 64) instead of 64 scattered 16-byte pieces.  A thread keeps its
// channel group for all its pixels of the 16x16 tile: with one input channel the 9x8 weights stay in registers
// (W_REGS), otherwise they are read from LDS; BatchNorm partial statistics are accumulated per thread and combined
// over the lanes that share a channel group (xor-shuffles), then over the four waves.
// LEAN [r5]: the training launch of the first conv as the bench runs it (one input plane, statistics, no folded affine / ReLU, the
// image a whole number of tiles): the epilogue's eval-only code and the bounds tests are compiled out and the tile's eight passes
// run as two rounds of four -- half the live accumulators, so three workgroups per CU instead of two hide each other's load and
// store phases.  Same arithmetic in the same order: bit-identical output and statistics.
// LEAN = 2: the same for the INFERENCE launch (folded BatchNorm affine + ReLU, no statistics).
template <typename T, int CL, bool W_REGS, int LEAN = 0>
__global__ __launch_bounds__(256, LEAN ? 3 : 1) void smallconv_s2l_kernel(S2LArgs a) {
  constexpr int G = CL / 8;                 // channel groups (lanes per pixel)
  constexpr int PPP = 256 / G;              // pixels per pass
  constexpr int PASSES = TS * TS / PPP;
  __shared__ float s_in[CS_MAX][HS][HS + 1];
  __shared__ __attribute__((aligned(16))) float s_w[W_REGS ? 1 : CS_MAX * 9 * CL];
  __shared__ float s_stat[4][3][CL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int t = blockIdx.x;
  const int tx_id = t % a.tilesX; t /= a.tilesX;
  const int ty_id = t % a.tilesY;
  const int b = t / a.tilesY;
  const int y0 = ty_id * TS, x0 = tx_id * TS;
  const float* inb = a.in + (size_t)b * a.CS * a.H * a.W;
  for (int i = tid; i < a.CS * HS * HS; i += 256) {
    const int s = i / (HS * HS), r = i % (HS * HS);
    const int hy = r / HS, hx = r % HS;
    const int yy = y0 + hy - 1, xx = x0 + hx - 1;
    s_in[s][hy][hx] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? inb[((size_t)s * a.H + yy) * a.W + xx] : 0.f;
  }
  const int g = tid % G, c0 = g * 8;
  float wreg[W_REGS ? 9 : 1][8];
  if constexpr (W_REGS) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int k = 0; k < 8; ++k) wreg[tap][k] = a.w[(size_t)(a.flip ? 8 - tap : tap) * CL + c0 + k];
  } else {
    for (int i = tid; i < a.CS * 9 * CL; i += 256) {
      const int s = i / (9 * CL), r = i % (9 * CL);
      const int tap = r / CL, l = r % CL;
      s_w[i] = a.w[((size_t)s * 9 + (a.flip ? 8 - tap : tap)) * CL + l];
    }
  }
  float b0[8], sc[8], sh[8], s1[8], s2[8], K[8];
  float cnt = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    b0[k] = (a.bias ? a.bias[c0 + k] : 0.f) - (a.center ? a.center[c0 + k] : 0.f);
    sc[k] = a.scale_shift ? a.scale_shift[c0 + k] : 1.f;
    sh[k] = a.scale_shift ? a.scale_shift[CL + c0 + k] : 0.f;
    s1[k] = 0.f; s2[k] = 0.f; K[k] = 0.f;
  }
  __syncthreads();
  T* outb = reinterpret_cast<T*>(a.out) + (size_t)b * a.H * a.W * CL + c0;
  // all of this thread's pixels advance together through (input channel, tap), so a weight vector fetched from LDS
  // (or held in registers) serves PASSES pixels; per output the summation order stays bias, then (s, tap) ascending
  const int q = tid / G;
  const int qy = q / TS, qx = q % TS;                    // pixel of pass p: (qy + p * PPP / TS, qx)
  if constexpr (LEAN) {
    static_assert(W_REGS && PASSES % 2 == 0, "one input plane, weights in registers");
    constexpr int HP = PASSES / 2;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      float av[HP][8];
#pragma unroll
      for (int p = 0; p < HP; ++p)
#pragma unroll
        for (int k = 0; k < 8; ++k) av[p][k] = b0[k];
#pragma unroll
      for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int p = 0; p < HP; ++p) {
          const float xv = s_in[0][qy + (r * HP + p) * (PPP / TS) + tap / 3][qx + tap % 3];
#pragma unroll
          for (int k = 0; k < 8; ++k) av[p][k] = __builtin_fmaf(xv, wreg[tap][k], av[p][k]);
        }
#pragma unroll
      for (int p = 0; p < HP; ++p) {
        const int ty = qy + (r * HP + p) * (PPP / TS);
        float acc[8];
        if constexpr (LEAN == 2) {
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] = fmaxf(av[p][k] * sc[k] + sh[k], 0.f);      // the generic epilogue's expression, a.relu set
        } else {
#pragma unroll
          for (int k = 0; k < 8; ++k) acc[k] = av[p][k];
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] = to_float(from_float<T>(acc[k]));
        T* o = outb + ((size_t)(y0 + ty) * a.W + x0 + qx) * CL;
        constexpr int N = Vec16<T>::N;
#pragma unroll
        for (int k = 0; k < 8; k += N) Vec16<T>::store_nt(o + k, acc + k);
        if constexpr (LEAN == 1) {
          if (r == 0 && p == 0) {
#pragma unroll
            for (int k = 0; k < 8; ++k) K[k] = acc[k];
          }
          cnt += 1.f;
#pragma unroll
          for (int k = 0; k < 8; ++k) { const float d = acc[k] - K[k]; s1[k] += d; s2[k] += d * d; }
        }
      }
    }
  } else {
  float accv[PASSES][8];
#pragma unroll
  for (int p = 0; p < PASSES; ++p)
#pragma unroll
    for (int k = 0; k < 8; ++k) accv[p][k] = b0[k];
  if constexpr (W_REGS) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int p = 0; p < PASSES; ++p) {
        const float xv = s_in[0][qy + p * (PPP / TS) + tap / 3][qx + tap % 3];
#pragma unroll
        for (int k = 0; k < 8; ++k) accv[p][k] = __builtin_fmaf(xv, wreg[tap][k], accv[p][k]);
      }
  } else {
    for (int s = 0; s < a.CS; ++s) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const float4 w0 = *reinterpret_cast<const float4*>(&s_w[(s * 9 + tap) * CL + c0]);
        const float4 w1 = *reinterpret_cast<const float4*>(&s_w[(s * 9 + tap) * CL + c0 + 4]);
#pragma unroll
        for (int p = 0; p < PASSES; ++p) {
          const float xv = s_in[s][qy + p * (PPP / TS) + tap / 3][qx + tap % 3];
          accv[p][0] = __builtin_fmaf(xv, w0.x, accv[p][0]); accv[p][1] = __builtin_fmaf(xv, w0.y, accv[p][1]);
          accv[p][2] = __builtin_fmaf(xv, w0.z, accv[p][2]); accv[p][3] = __builtin_fmaf(xv, w0.w, accv[p][3]);
          accv[p][4] = __builtin_fmaf(xv, w1.x, accv[p][4]); accv[p][5] = __builtin_fmaf(xv, w1.y, accv[p][5]);
          accv[p][6] = __builtin_fmaf(xv, w1.z, accv[p][6]); accv[p][7] = __builtin_fmaf(xv, w1.w, accv[p][7]);
        }
      }
    }
  }
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    const int ty = qy + pass * (PPP / TS), tx = qx;
    float acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = accv[pass][k];
    if (a.scale_shift) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = acc[k] * sc[k] + sh[k];
    }
    if (a.relu) {
#pragma unroll
      for (int k = 0; k < 8; ++k) acc[k] = fmaxf(acc[k], 0.f);
    }
    // round to the storage type first so the statistics describe what BatchNorm will normalise
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = to_float(from_float<T>(acc[k]));
    const int yy = y0 + ty, xx = x0 + tx;
    if (yy < a.H && xx < a.W) {
      T* o = outb + ((size_t)yy * a.W + xx) * CL;
      constexpr int N = Vec16<T>::N;
#pragma unroll
      for (int k = 0; k < 8; k += N) Vec16<T>::store_nt(o + k, acc + k);      // streamed: 1 GB per 78 images, not read again by this kernel
      // statistics relative to this thread's first valid value (no cancellation when forming M2 below)
      if (cnt == 0.f) {
#pragma unroll
        for (int k = 0; k < 8; ++k) K[k] = acc[k];
      }
      cnt += 1.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) { const float d = acc[k] - K[k]; s1[k] += d; s2[k] += d * d; }
    }
  }
  }   // !LEAN
  if (a.stats) {
    // thread -> (count, mean, M2); merged over the lanes that share the channel group, then over the four waves
    const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
    float mean[8], m2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { mean[k] = K[k] + s1[k] * inv; m2[k] = fmaxf(s2[k] - s1[k] * s1[k] * inv, 0.f); }
#pragma unroll
    for (int off = G; off < 64; off <<= 1) {
      const float n2 = __shfl_xor(cnt, off, 64);
      const float nn = cnt + n2;
      const float ninv = nn > 0.f ? 1.f / nn : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float mo = __shfl_xor(mean[k], off, 64), qo = __shfl_xor(m2[k], off, 64);
        const float d = mo - mean[k];
        m2[k] = m2[k] + qo + d * d * (cnt * n2 * ninv);
        mean[k] = (cnt * mean[k] + n2 * mo) * ninv;
      }
      cnt = nn;
    }
    if (lane < G) {
#pragma unroll
      for (int k = 0; k < 8; ++k) { s_stat[wave][0][c0 + k] = mean[k]; s_stat[wave][1][c0 + k] = m2[k]; s_stat[wave][2][c0 + k] = cnt; }
    }
    __syncthreads();
    if (tid < CL) {
      float n = s_stat[0][2][tid], m = s_stat[0][0][tid], q = s_stat[0][1][tid];
#pragma unroll
      for (int w = 1; w < 4; ++w) {
        const float n2 = s_stat[w][2][tid], mo = s_stat[w][0][tid], qo = s_stat[w][1][tid];
        const float nn = n + n2, ninv = nn > 0.f ? 1.f / nn : 0.f, d = mo - m;
        q = q + qo + d * d * (n * n2 * ninv);
        m = (n * m + n2 * mo) * ninv;
        n = nn;
      }
      float* st = a.stats + (size_t)blockIdx.x * 3 * CL;
      st[tid] = m; st[CL + tid] = q; st[2 * CL + tid] = n;
    }
  }
}

struct L2SArgs {
  const void* in;       // [B][H][W][CL] T
  const float* w;       // [CS][9][CL]
  const float* bias;    // [CS] | null
  float* out;           // [B][CS][H][W]
  int B, H, W, CS, tilesY, tilesX;
};

template <typename T, int CL>
__global__ __launch_bounds__(256) void smallconv_l2s_kernel(L2SArgs a) {
  constexpr int N = Vec16<T>::N;
  constexpr int PITCH = CL * (int)sizeof(T) + 16;
  __shared__ __attribute__((aligned(16))) char s_in[HS * HS * PITCH];
  const int tid = threadIdx.x;
  int t = blockIdx.x;
  const int tx_id = t % a.tilesX; t /= a.tilesX;
  const int ty_id = t % a.tilesY;
  const int b = t / a.tilesY;
  const int y0 = ty_id * TS, x0 = tx_id * TS;
  const T* inb = reinterpret_cast<const T*>(a.in) + (size_t)b * a.H * a.W * CL;
  constexpr int PPR = CL / N;
  for (int i = tid; i < HS * HS * PPR; i += 256) {
    const int px = i / PPR, part = i % PPR;
    const int yy = y0 + px / HS - 1, xx = x0 + px % HS - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
      v = *reinterpret_cast<const uint4*>(inb + ((size_t)yy * a.W + xx) * CL + part * N);
    *reinterpret_cast<uint4*>(s_in + px * PITCH + part * 16) = v;
  }
  __syncthreads();
  const int ty = tid / TS, tx = tid % TS;
  float acc[CS_MAX];
#pragma unroll
  for (int s = 0; s < CS_MAX; ++s) acc[s] = (a.bias && s < a.CS) ? a.bias[s] : 0.f;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const char* row = s_in + ((ty + tap / 3) * HS + tx + tap % 3) * PITCH;
#pragma unroll
    for (int part = 0; part < PPR; ++part) {
      float xv[N];
      Vec16<T>::load(reinterpret_cast<const T*>(row + part * 16), xv);
#pragma unroll
      for (int s = 0; s < CS_MAX; ++s) {
        if (s < a.CS) {
          const float* wr = a.w + ((size_t)s * 9 + tap) * CL + part * N;   // wave-uniform
#pragma unroll
          for (int k = 0; k < N; ++k) acc[s] += xv[k] * wr[k];
        }
      }
    }
  }
  const int yy = y0 + ty, xx = x0 + tx;
  if (yy < a.H && xx < a.W) {
#pragma unroll
    for (int s = 0; s < CS_MAX; ++s)
      if (s < a.CS) a.out[(((size_t)b * a.CS + s) * a.H + yy) * a.W + xx] = acc[s];
  }
}

// ------------------------------------------------------------------------------------------------------------------
// MFMA forms of the two "few channels on one side" convolutions that dominate the tail of the network at full
// resolution.  The VALU kernels above issue ~1 instruction per MAC and are issue-bound (the heads forward read its
// 0.5 GB at 1.2 TB/s); on the matrix cores the same work is a sliver of MFMA time even though only CS of the 32 output
// columns of a tile are real, and the kernels become HBM-bound.
//   l2s (heads forward):  out[px][s] = bias[s] + sum_{tap,c} F[px+tap][c] * w[s][tap][c]
//       A = the halo tile of the feature map in LDS (rows = pixels, as conv_mfma.hip), B = the CS weight rows.  bf16:
//       the fp32 weights enter as hi + lo bf16 pairs (two MFMAs), which keeps 16 mantissa bits of them -- the VALU
//       kernel multiplies bf16 features by fp32 weights, and the results agree to ~1e-5; fp32: v_mfma_f32_32x32x2_f32.
//   s2l with flip (heads data-gradient):  dF[px][c] = sum_{tap,s} g[s][px-tap] * w[s][tap][c]
//       K = (tap, s) with 8 s-slots per tap, so a lane's 8 k-values of one MFMA k-step are the 8 planes of ONE halo pixel:
//       one ds_read_b128 of the [halo px][8] gradient tile.
struct WgFrag {      // K-contiguous bf16 fragment out of a pixel-major LDS tile: hardware 4x16 transpose read (see conv_mfma.hip)
  static __device__ __forceinline__ short8 load(const char* p0, const char* p1) {
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(lds_char*)p0);
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(lds_char*)p1);
    short8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
};
template <typename T> struct SmallFrag;
template <> struct SmallFrag<bf16_t> {
  using AB = short8;
  static constexpr int KPS = 16;                      // k per MFMA step
  static __device__ __forceinline__ AB load(const char* p) { return *reinterpret_cast<const AB*>(p); }
  static __device__ __forceinline__ f32x16 mfma(AB a, AB b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
  }
};
template <> struct SmallFrag<float> {
  using AB = float;
  static constexpr int KPS = 2;
  static __device__ __forceinline__ AB load(const char* p) { return *reinterpret_cast<const float*>(p); }
  static __device__ __forceinline__ f32x16 mfma(AB a, AB b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
};

template <typename T, int CL>
__global__ __launch_bounds__(256) void smallconv_l2s_mfma_kernel(L2SArgs a) {
  constexpr int SZ = (int)sizeof(T);
  constexpr bool IS_BF16 = SZ == 2;
  constexpr int N = Vec16<T>::N;
  constexpr int PF = CL * SZ + 16;                    // halo pixel pitch
  constexpr int HROWB = HS * PF + (IS_BF16 ? 96 : 0); // halo row pitch (bf16: conflict-free ds_read_b128 across two tile rows)
  constexpr int F_BYTES = HS * HROWB;
  constexpr int PW = 9 * CL * SZ + 16;                // weight row pitch
  constexpr int NPARTS = IS_BF16 ? 2 : 1;             // bf16: hi + lo
  constexpr int W_BYTES = NPARTS * 8 * PW;
  constexpr int KSTEPS = CL / SmallFrag<T>::KPS;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsF = smem;
  char* ldsW = smem + F_BYTES;
  float* ldsO = reinterpret_cast<float*>(smem + F_BYTES + W_BYTES);     // [8][256]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  // persistent workgroups: the weights are converted ONCE per workgroup, the next tile's halo is in flight during the MFMAs
  // weights: w [CS][9][CL] fp32 -> rows s < 8 (zero beyond CS) of T, bf16 split into hi and lo = bf16(w - hi)
  for (int i = tid; i < 8 * 9 * CL; i += 256) {
    const int srow = i / (9 * CL), k = i % (9 * CL);
    const float v = srow < a.CS ? a.w[(size_t)srow * 9 * CL + k] : 0.f;
    if constexpr (IS_BF16) {
      const bf16_t hi = (bf16_t)v;
      const bf16_t lo = (bf16_t)(v - (float)hi);
      *reinterpret_cast<bf16_t*>(ldsW + srow * PW + k * 2) = hi;
      *reinterpret_cast<bf16_t*>(ldsW + 8 * PW + srow * PW + k * 2) = lo;
    } else {
      *reinterpret_cast<float*>(ldsW + srow * PW + k * 4) = v;
    }
  }
  constexpr int PPR = CL / N;
  constexpr int ROUNDS = (HS * HS * PPR + 255) / 256;
  uint4 r[ROUNDS];
  const int ntiles = a.B * a.tilesY * a.tilesX;
  auto gload = [&](int tile) {
    int t = tile;
    const int tx_id = t % a.tilesX; t /= a.tilesX;
    const int ty_id = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty_id * TS, x0 = tx_id * TS;
    const T* inb = reinterpret_cast<const T*>(a.in) + (size_t)b * a.H * a.W * CL;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int px = p / PPR, part = p % PPR;
      const int yy = y0 + px / HS - 1, xx = x0 + px % HS - 1;
      r[i] = make_uint4(0, 0, 0, 0);
      if (px < HS * HS && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W)
        r[i] = *reinterpret_cast<const uint4*>(inb + ((size_t)yy * a.W + xx) * CL + part * N);
    }
  };
  int aoff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) aoff[j] = (4 * wave + 2 * j + l31 / TS) * HROWB + (l31 % TS) * PF;
  // [r6] bf16: B-fragment columns 0-7 are the hi rows of the weights and columns 8-15 the lo rows (rows 8-15 of ldsW): hi and lo products
  // come out of ONE MFMA in separate accumulator columns and are added in the epilogue (was: two MFMAs per k-step and row tile)
  const int boff = (IS_BF16 ? (l31 & 15) : (l31 & 7)) * PW;
  if ((int)blockIdx.x < ntiles) gload(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx_id = t % a.tilesX; t /= a.tilesX;
    const int ty_id = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty_id * TS, x0 = tx_id * TS;
#pragma unroll
    for (int i = 0; i < ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int px = p / PPR, part = p % PPR;
      if (px < HS * HS) *reinterpret_cast<uint4*>(ldsF + (px / HS) * HROWB + (px % HS) * PF + part * 16) = r[i];
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) gload(tile + gridDim.x);
    // wave w: output tile rows 4w .. 4w+3 as two 32-pixel MFMA row tiles
    f32x16 acc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int toff = (tap / 3) * HROWB + (tap % 3) * PF;
#pragma unroll
      for (int ks = 0; ks < KSTEPS; ++ks) {
        const int koff = IS_BF16 ? ks * 32 + half * 16 : (ks * 2 + half) * 4;
        const auto fb = SmallFrag<T>::load(ldsW + boff + tap * CL * SZ + koff);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const auto fa = SmallFrag<T>::load(ldsF + aoff[j] + toff + koff);
          acc[j] = SmallFrag<T>::mfma(fa, fb, acc[j]);
        }
      }
    }
    {
      const float bv = (a.bias && l31 < a.CS) ? a.bias[l31] : 0.f;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = (q & 3) + 8 * (q >> 2) + 4 * half;              // pixel within the 32-pixel row tile
          float v = acc[j][q];
          if constexpr (IS_BF16) v += __shfl_down(acc[j][q], 8, 64);    // plane s: hi column s + lo column 8 + s
          if (l31 < a.CS) ldsO[l31 * 256 + (4 * wave + 2 * j) * TS + m] = v + bv;
        }
    }
    __syncthreads();                                    // results staged; every wave is done reading the halo tile
    for (int i = tid; i < a.CS * 256; i += 256) {
      const int sidx = i >> 8, px = i & 255;
      const int yy = y0 + px / TS, xx = x0 + px % TS;
      if (yy < a.H && xx < a.W) a.out[(((size_t)b * a.CS + sidx) * a.H + yy) * a.W + xx] = ldsO[i];
    }
  }
}

// data-gradient of the heads: g [B][CS][H][W] fp32, w [CS][9][CL] fp32 (forward layout, used flipped) -> dF [B][H][W][CL] T
template <typename T, int CL>
__global__ __launch_bounds__(256) void smallconv_s2l_dgrad_mfma_kernel(S2LArgs a) {
  constexpr int SZ = (int)sizeof(T);
  constexpr bool IS_BF16 = SZ == 2;
  constexpr int PD = 8 * SZ + (IS_BF16 ? 0 : 4);      // halo pixel pitch of the gradient tile: 8 plane slots (fp32: +4 B, odd dword pitch)
  constexpr int D_BYTES = (HS * HS + 2) * PD;
  constexpr int TAPS_P = IS_BF16 ? 10 : 9;            // bf16: k-steps of 16 = two taps; the tenth is zero
  constexpr int PW = TAPS_P * 8 * SZ + (IS_BF16 ? 16 : 4);   // weight row (one per output channel c): [tap][8 slots]
  constexpr int W_BYTES = CL * PW;
  constexpr int NT = CL / 32;
  constexpr int OP = CL * SZ + 16;                    // output staging pitch
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsD = smem;
  char* ldsW = smem + ((D_BYTES + 15) & ~15);
  char* ldsOut = ldsW + W_BYTES;                      // [256][OP]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int half = lane >> 5, l31 = lane & 31;
  // persistent workgroups: weights converted once.  row c, k = (tap', slot): w[slot][8 - tap'][c] (correlation over the flipped taps)
  for (int i = tid; i < CL * TAPS_P * 8; i += 256) {
    const int c = i / (TAPS_P * 8), k = i % (TAPS_P * 8);
    const int tp = k / 8, slot = k % 8;
    float v = 0.f;
    if (tp < 9 && slot < a.CS) v = a.w[((size_t)slot * 9 + (a.flip ? 8 - tp : tp)) * CL + c];
    *reinterpret_cast<T*>(ldsW + c * PW + k * SZ) = from_float<T>(v);
  }
  constexpr int G_ROUNDS = (HS * HS * 8 + 255) / 256;  // gradient halo elements per thread
  float gv[G_ROUNDS];
  const int ntiles = a.B * a.tilesY * a.tilesX;
  auto gload = [&](int tile) {
    int t = tile;
    const int tx_id = t % a.tilesX; t /= a.tilesX;
    const int ty_id = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty_id * TS, x0 = tx_id * TS;
    const float* gb = a.in + (size_t)b * a.CS * a.H * a.W;
#pragma unroll
    for (int i = 0; i < G_ROUNDS; ++i) {
      const int e = i * 256 + tid;
      const int slot = e / (HS * HS), px = e % (HS * HS);           // consecutive threads -> consecutive x of one plane
      const int yy = y0 + px / HS - 1, xx = x0 + px % HS - 1;
      gv[i] = 0.f;
      if (slot < a.CS && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) gv[i] = gb[((size_t)slot * a.H + yy) * a.W + xx];
    }
  };
  int prow[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) prow[j] = (4 * wave + 2 * j + l31 / TS) * HS + (l31 % TS);      // halo index of (ty, tx) minus (1, 1)
  if ((int)blockIdx.x < ntiles) gload(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx_id = t % a.tilesX; t /= a.tilesX;
    const int ty_id = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty_id * TS, x0 = tx_id * TS;
#pragma unroll
    for (int i = 0; i < G_ROUNDS; ++i) {
      const int e = i * 256 + tid;
      const int slot = e / (HS * HS), px = e % (HS * HS);
      if (slot < 8) *reinterpret_cast<T*>(ldsD + px * PD + slot * SZ) = from_float<T>(gv[i]);
    }
    __syncthreads();                                    // (also: the previous tile's output staging has been drained)
    if (tile + (int)gridDim.x < ntiles) gload(tile + gridDim.x);
    f32x16 acc[2][NT];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[j][n][q] = 0.f;
    if constexpr (IS_BF16) {
#pragma unroll
      for (int ks = 0; ks < 5; ++ks) {
        const int tp = 2 * ks + half;                    // this lane's tap of the k-step (tap 9: weights are zero)
        const int tpc = tp < 9 ? tp : 8;
        const int doff = ((tpc / 3) * HS + (tpc % 3)) * PD;
        short8 fb[NT];
#pragma unroll
        for (int n = 0; n < NT; ++n) fb[n] = *reinterpret_cast<const short8*>(ldsW + (n * 32 + l31) * PW + tp * 16);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const short8 fa = *reinterpret_cast<const short8*>(ldsD + prow[j] * PD + doff);
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[j][n] = SmallFrag<bf16_t>::mfma(fa, fb[n], acc[j][n]);
        }
      }
    } else {
#pragma unroll 4
      for (int ks = 0; ks < 36; ++ks) {                  // k = ks*2 + half -> (tap, slot)
        const int k = ks * 2 + half;
        const int tp = k / 8, slot = k % 8;
        const int doff = ((tp / 3) * HS + (tp % 3)) * PD + slot * 4;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float fa = *reinterpret_cast<const float*>(ldsD + prow[j] * PD + doff);
#pragma unroll
          for (int n = 0; n < NT; ++n) {
            const float fb = *reinterpret_cast<const float*>(ldsW + (n * 32 + l31) * PW + k * 4);
            acc[j][n] = SmallFrag<float>::mfma(fa, fb, acc[j][n]);
          }
        }
      }
    }
    // transpose through LDS and write whole NHWC rows
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int q = 0; q < 16; ++q) {
          const int m = (q & 3) + 8 * (q >> 2) + 4 * half;
          const int px = (4 * wave + 2 * j) * TS + m;
          *reinterpret_cast<T*>(ldsOut + px * OP + (n * 32 + l31) * SZ) = from_float<T>(acc[j][n][q]);
        }
    __syncthreads();                                    // staged; every wave is done reading the gradient tile
    constexpr int N = Vec16<T>::N;
    constexpr int PPR = CL / N;
    T* outb = reinterpret_cast<T*>(a.out) + (size_t)b * a.H * a.W * CL;
    for (int i = tid; i < 256 * PPR; i += 256) {
      const int px = i / PPR, part = i % PPR;
      const int yy = y0 + px / TS, xx = x0 + px % TS;
      if (yy < a.H && xx < a.W)
        *reinterpret_cast<uint4*>(outb + ((size_t)yy * a.W + xx) * CL + part * N) = *reinterpret_cast<const uint4*>(ldsOut + px * OP + part * 16);
    }
  }
}

template <typename T, int CL> constexpr size_t l2s_mfma_smem() {
  constexpr int SZ = (int)sizeof(T);
  return (size_t)HS * (HS * (CL * SZ + 16) + (SZ == 2 ? 96 : 0)) + (size_t)(SZ == 2 ? 2 : 1) * 8 * (9 * CL * SZ + 16) + 8 * 256 * sizeof(float);
}
template <typename T, int CL> constexpr size_t wgrad_mfma_smem() {
  constexpr int SZ = (int)sizeof(T);
  return (size_t)TS * TS * (CL * SZ + (SZ == 2 ? 16 : 4)) + (size_t)(SZ == 2 ? 2 : 1) * ((size_t)HS * HS * (SZ == 2 ? 16 : 36) + 128);
}
template <typename T, int CL> constexpr size_t s2l_dgrad_mfma_smem() {
  constexpr int SZ = (int)sizeof(T);
  constexpr size_t d = (((size_t)(HS * HS + 2) * (8 * SZ + (SZ == 2 ? 0 : 4))) + 15) & ~(size_t)15;
  return d + (size_t)CL * ((SZ == 2 ? 10 : 9) * 8 * SZ + (SZ == 2 ? 16 : 4)) + (size_t)256 * (CL * SZ + 16);
}

struct SWArgs {
  const float* S;       // [B][CS][H][W] fp32
  const void* L;        // [B][H][W][CL] T
  float* partial;       // [blk][CS*9*CL + CS]
  int B, H, W, CS, tilesY, tilesX;
};

// Persistent blocks: a block walks tiles t = blockIdx.x, += gridDim.x and keeps its sums in registers, so there is one
// partial row per block (not per tile) and one LDS reduction at the end.  Thread = (wide-side channel l, row group g):
// it walks its tile rows pixel by pixel with the 3x3 window of the small side sliding through registers (3 new LDS
// broadcast reads per pixel and input channel instead of 9).
// CSB = compile-time bound on the small side's channel count (1, 2, 4 or 8): the accumulators acc[CSB][9] must be
// register-resident, and a bound of 8 for a 1-channel input would cost the occupancy.
template <typename T, int CL, int CSB>
__global__ __launch_bounds__(256) void smallconv_wgrad_kernel(SWArgs a) {
  constexpr int N = Vec16<T>::N;
  constexpr int G = 256 / CL;
  constexpr int RPG = TS / G;                       // tile rows per group
  __shared__ __attribute__((aligned(16))) T s_L[TS * TS][CL];
  __shared__ float s_S[CS_MAX][HS][HS + 1];
  __shared__ float s_red[G][9][CL];
  __shared__ float s_b[G];
  const int tid = threadIdx.x;
  const int l = tid % CL, g = tid / CL;
  float acc[CSB][9];
  float bsum[CSB];
#pragma unroll
  for (int s = 0; s < CSB; ++s) {
    bsum[s] = 0.f;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) acc[s][tp] = 0.f;
  }
  const int ntiles = a.B * a.tilesY * a.tilesX;
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx_id = t % a.tilesX; t /= a.tilesX;
    const int ty_id = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty_id * TS, x0 = tx_id * TS;
    const float* Sb = a.S + (size_t)b * a.CS * a.H * a.W;
    const T* Lb = reinterpret_cast<const T*>(a.L) + (size_t)b * a.H * a.W * CL;
    __syncthreads();                                // the previous tile's readers are done
    for (int i = tid; i < a.CS * HS * HS; i += 256) {
      const int s = i / (HS * HS), r = i % (HS * HS);
      const int hy = r / HS, hx = r % HS;
      const int yy = y0 + hy - 1, xx = x0 + hx - 1;
      s_S[s][hy][hx] = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? Sb[((size_t)s * a.H + yy) * a.W + xx] : 0.f;
    }
    constexpr int PPR = CL / N;
    for (int i = tid; i < TS * TS * PPR; i += 256) {
      const int px = i / PPR, part = i % PPR;
      const int yy = y0 + px / TS, xx = x0 + px % TS;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (yy < a.H && xx < a.W) v = *reinterpret_cast<const uint4*>(Lb + ((size_t)yy * a.W + xx) * CL + part * N);
      *reinterpret_cast<uint4*>(&s_L[px][part * N]) = v;
    }
    __syncthreads();
#pragma unroll 1
    for (int rr = 0; rr < RPG; ++rr) {
      const int ty = g * RPG + rr;
#pragma unroll
      for (int s = 0; s < CSB; ++s) {
        if (s < a.CS) {
          float w[3][3];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) { w[dy][1] = s_S[s][ty + dy][0]; w[dy][2] = s_S[s][ty + dy][1]; }
#pragma unroll
          for (int tx = 0; tx < TS; ++tx) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) { w[dy][0] = w[dy][1]; w[dy][1] = w[dy][2]; w[dy][2] = s_S[s][ty + dy][tx + 2]; }
            const float lv = to_float(s_L[ty * TS + tx][l]);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) acc[s][tp] += lv * w[tp / 3][tp % 3];
            bsum[s] += w[1][1];
          }
        }
      }
    }
  }
  const int K = a.CS * 9 * CL + a.CS;
  float* out = a.partial + (size_t)blockIdx.x * K;
  for (int s = 0; s < a.CS; ++s) {
    __syncthreads();
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < CSB; ++q) if (q == s) v = acc[q][tp];
      s_red[g][tp][l] = v;
    }
    if (l == 0) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < CSB; ++q) if (q == s) v = bsum[q];
      s_b[g] = v;
    }
    __syncthreads();
    for (int i = tid; i < 9 * CL; i += 256) {
      const int tp = i / CL, ll = i % CL;
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < G; ++q) v += s_red[q][tp][ll];
      out[((size_t)s * 9 + tp) * CL + ll] = v;
    }
    if (tid == 0) {
      float v = 0.f;
#pragma unroll
      for (int q = 0; q < G; ++q) v += s_b[q];
      out[(size_t)a.CS * 9 * CL + s] = v;
    }
  }
}

// smallconv_wgrad_kernel with VL wide-side channels per thread (the first conv of the network, 1 or 2 input planes): a
// thread owns VL consecutive channels of one tile row, so the pixel's VL values arrive in ONE LDS read and the three new
// window values of the small side are shared by VL * 9 fused multiply-adds (the one-channel form issues 4 LDS reads and
// 19 VALU operations per 9 MACs: 0.48 ms for a 1 GB read).  Lanes sharing a channel group sit 16 apart in a wave: their
// sums meet by xor-shuffles, the four waves through LDS, in a fixed order.  Partial rows as smallconv_wgrad_kernel.
// FM [r6]: how the multiply-adds are issued.  The compiler pairs them into v_pk_fma_f32 and, for a small-side value that sits in the HIGH
// register of an aligned pair, selects it for the low lane with op_sel:[0,1,0].  That form returned wrong low-lane sums (the even channels,
// about one pixel's products off, a different result on every launch) whenever OTHER PROCESSES kept the GPU busy -- never in a process that
// had the GPU to itself, and not on every box: 35 % of the two-rank runs of tests/test_graph_ddp_gpu.py on one, 100 % of the launches beside
// two processes running training steps.  Bisected in place with explicit instruction forms (tools/debug_victim2.sh,
// profiles/r06_multiprocess_determinism.txt; 200 launches each beside two such processes):
//   0  compiler's choice (op_sel:[0,1,0] and op_sel_hi:[1,0,1] mixed)            200 / 200 launches differ
//   1  scalar v_fmac_f32                                                            0 / 200     <- what runs (DEFAULT)
//   2  even columns v_pk_fma_f32 op_sel_hi:[1,0,1], odd columns scalar              0 / 200
//   3  even columns scalar, odd columns v_pk_fma_f32 op_sel:[0,1,0]               200 / 200
//   4  both packed forms, explicit                                                200 / 200
//   5  small side stored twice in LDS, v_pk_fma_f32 without modifiers               0 / 200
// The register prefetch and the shuffles were excluded the same way.  Alone on the GPU the forms cost 0.296 (0) / 0.311 (1) / 0.316 (2) /
// 0.323 ms (5) at batch 78: the sums are not what limits the kernel, so the plain scalar form is the one kept.  Forms 2..5 need LF (the
// wide side as fp32 pairs in LDS; a bf16 tile is widened when it is written to LDS).  IM2IM_SWG_DBG=<form> selects one (CS = 1, CL = 64) in a
// library built with IM2IM_BUILD_EXPERIMENTAL=1; the default library holds form 1 only.
template <typename T, int CL, int CSB, int VL, bool LF = false, int FM = 1>
__global__ __launch_bounds__(256) void smallconv_wgrad_vec_kernel(SWArgs a) {
  constexpr int N = Vec16<T>::N;
  constexpr int LG = CL / VL;                       // lanes per tile row
  constexpr int G = 256 / LG;                       // tile rows in flight
  constexpr int RPG = TS / G;
  static_assert(LG == 16 && G == 16 && RPG == 1, "one tile row per 16-lane group");
  using LT = std::conditional_t<LF, float, T>;
  __shared__ __attribute__((aligned(16))) LT s_L[TS * TS][CL];
  constexpr int SW = FM == 5 ? 2 * (HS + 1) : (FM >= 2 ? HS + 2 : HS + 1);      // row pitch of the small side's halo tile (floats)
  constexpr int SD = FM == 5 ? 2 : 1;               // FM 5: every value twice, (v, v)
  static_assert(FM < 2 || LF, "explicit packed forms read the wide side as fp32 pairs");
  __shared__ __attribute__((aligned(16))) float s_S[CSB][HS][SW];
  __shared__ float s_red[4][CSB * 9 + 1][CL];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lq = tid % LG, ty = tid / LG;           // channel group, tile row
  using f2 = float __attribute__((ext_vector_type(2)));
  float acc[CSB][9][VL];
  float bsum[CSB];
#pragma unroll
  for (int s = 0; s < CSB; ++s) {
    bsum[s] = 0.f;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp)
#pragma unroll
      for (int k = 0; k < VL; ++k) acc[s][tp][k] = 0.f;
  }
  const int ntiles = a.B * a.tilesY * a.tilesX;
  constexpr int PPR = CL / N;
  constexpr int L_ROUNDS = TS * TS * PPR / 256;
  uint4 rl[L_ROUNDS];
  auto gload_L = [&](int tile) {
    int t = tile;
    const int tx_id = t % a.tilesX; t /= a.tilesX;
    const int ty_id = t % a.tilesY;
    const int b = t / a.tilesY;
    const T* Lb = reinterpret_cast<const T*>(a.L) + (size_t)b * a.H * a.W * CL;
#pragma unroll
    for (int i = 0; i < L_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int px = p / PPR, part = p % PPR;
      const int yy = ty_id * TS + px / TS, xx = tx_id * TS + px % TS;
      rl[i] = make_uint4(0, 0, 0, 0);
      if (yy < a.H && xx < a.W) rl[i] = *reinterpret_cast<const uint4*>(Lb + ((size_t)yy * a.W + xx) * CL + part * N);
    }
  };
  if ((int)blockIdx.x < ntiles) gload_L(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int tx_id = t % a.tilesX; t /= a.tilesX;
    const int ty_id = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty_id * TS, x0 = tx_id * TS;
    const float* Sb = a.S + (size_t)b * a.CS * a.H * a.W;
    __syncthreads();                                // the previous tile's readers are done
    for (int i = tid; i < a.CS * HS * HS; i += 256) {
      const int sidx = i / (HS * HS), r = i % (HS * HS);
      const int hy = r / HS, hx = r % HS;
      const int yy = y0 + hy - 1, xx = x0 + hx - 1;
      const float sv = (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) ? Sb[((size_t)sidx * a.H + yy) * a.W + xx] : 0.f;
      s_S[sidx][hy][hx * SD] = sv;
      if constexpr (SD == 2) s_S[sidx][hy][hx * 2 + 1] = sv;
    }
#pragma unroll
    for (int i = 0; i < L_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      if constexpr (std::is_same<LT, T>::value) *reinterpret_cast<uint4*>(&s_L[p / PPR][(p % PPR) * N]) = rl[i];
      else {
        float f[N];
        Vec16<T>::load(reinterpret_cast<const T*>(&rl[i]), f);
#pragma unroll
        for (int k = 0; k < N; k += 4) *reinterpret_cast<float4*>(&s_L[p / PPR][(p % PPR) * N + k]) = make_float4(f[k], f[k + 1], f[k + 2], f[k + 3]);
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) gload_L(tile + (int)gridDim.x);   // the next tile's wide side is in flight during the sums
#pragma unroll
    for (int s = 0; s < CSB; ++s) {
      if (s < a.CS) {
        if constexpr (FM < 2) {
          float w[3][3];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) { w[dy][1] = s_S[s][ty + dy][0]; w[dy][2] = s_S[s][ty + dy][1]; }
#pragma unroll
          for (int tx = 0; tx < TS; ++tx) {
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) { w[dy][0] = w[dy][1]; w[dy][1] = w[dy][2]; w[dy][2] = s_S[s][ty + dy][tx + 2]; }
            float lv[VL];
#pragma unroll
            for (int k = 0; k < VL; ++k) lv[k] = to_float(s_L[ty * TS + tx][lq * VL + k]);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp)
#pragma unroll
              for (int k = 0; k < VL; ++k) {
                if constexpr (FM == 1) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(acc[s][tp][k]) : "v"(lv[k]), "v"(w[tp / 3][tp % 3]));
                else acc[s][tp][k] = __builtin_fmaf(lv[k], w[tp / 3][tp % 3], acc[s][tp][k]);
              }
            bsum[s] += w[1][1];
          }
        } else {
          // the halo rows as aligned register pairs: FM 2..4 pr[dy][j] = (S[2j], S[2j+1]); FM 5 pr[dy][x] = (S[x], S[x])
          constexpr int NP = FM == 5 ? HS : HS / 2;
          f2 pr[3][NP];
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int j = 0; j < NP; ++j) pr[dy][j] = *reinterpret_cast<const f2*>(&s_S[s][ty + dy][2 * j]);
#pragma unroll
          for (int tx = 0; tx < TS; ++tx) {
            f2 lv2[VL / 2];
#pragma unroll
            for (int kp = 0; kp < VL / 2; ++kp) lv2[kp] = *reinterpret_cast<const f2*>(&s_L[ty * TS + tx][lq * VL + 2 * kp]);
#pragma unroll
            for (int tp = 0; tp < 9; ++tp) {
              const int x = tx + tp % 3;                                  // column of the halo row (compile-time after unrolling)
              const bool hi = FM != 5 && (x & 1);
              const f2 wp = FM == 5 ? pr[tp / 3][x] : pr[tp / 3][x / 2];
              const bool packed = FM == 4 || FM == 5 || (FM == 2 && !hi) || (FM == 3 && hi);
#pragma unroll
              for (int kp = 0; kp < VL / 2; ++kp) {
                f2 av = {acc[s][tp][2 * kp], acc[s][tp][2 * kp + 1]};
                if (packed) {
                  if (FM == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(av) : "v"(lv2[kp]), "v"(wp));
                  else if (hi) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(av) : "v"(lv2[kp]), "v"(wp));
                  else asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(av) : "v"(lv2[kp]), "v"(wp));
                } else {
                  const float wv = hi ? wp.y : wp.x;
                  float a0 = av.x, a1 = av.y;
                  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(lv2[kp].x), "v"(wv));
                  asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(lv2[kp].y), "v"(wv));
                  av.x = a0; av.y = a1;
                }
                acc[s][tp][2 * kp] = av.x; acc[s][tp][2 * kp + 1] = av.y;
              }
            }
            bsum[s] += FM == 5 ? pr[1][tx + 1].x : (((tx + 1) & 1) ? pr[1][(tx + 1) / 2].y : pr[1][(tx + 1) / 2].x);
          }
        }
      }
    }
  }
  // the four tile rows of a wave (lanes lq, lq+16, lq+32, lq+48), then the four waves
#pragma unroll
  for (int s = 0; s < CSB; ++s) {
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int k = 0; k < VL; ++k) acc[s][tp][k] += __shfl_xor(acc[s][tp][k], off, 64);
      bsum[s] += __shfl_xor(bsum[s], off, 64);
    }
  }
  __syncthreads();
  if (lane < LG) {
#pragma unroll
    for (int s = 0; s < CSB; ++s)
#pragma unroll
      for (int tp = 0; tp < 9; ++tp)
#pragma unroll
        for (int k = 0; k < VL; ++k) s_red[wave][s * 9 + tp][lq * VL + k] = acc[s][tp][k];
    if (lane == 0) {
#pragma unroll
      for (int s = 0; s < CSB; ++s) s_red[wave][CSB * 9][s] = bsum[s];
    }
  }
  __syncthreads();
  const int K = a.CS * 9 * CL + a.CS;
  float* out = a.partial + (size_t)blockIdx.x * K;
  for (int i = tid; i < a.CS * 9 * CL; i += 256) {
    const int row = i / CL, ll = i % CL;                       // row = s*9 + tap
    out[i] = ((s_red[0][row][ll] + s_red[1][row][ll]) + s_red[2][row][ll]) + s_red[3][row][ll];
  }
  if (tid < a.CS) out[(size_t)a.CS * 9 * CL + tid] = ((s_red[0][CSB * 9][tid] + s_red[1][CSB * 9][tid]) + s_red[2][CSB * 9][tid]) + s_red[3][CSB * 9][tid];
}

// MFMA form of the weight gradients above:  out[s][tap][l] = sum_px L[px][l] * S[s][px + off(tap)]  (+ sum_px S[s][px]).
// Per tap a GEMM with M = the wide side's channels l, N = the 8 plane slots of the small side, K = pixels.  Both operands
// live pixel-major in LDS ([px][channels]) but the matrix cores want k (= pixel) contiguous per lane: bf16 fragments are
// fetched with ds_read_b64_tr_b16 (as conv_wgrad_kernel in conv_mfma.hip), fp32 ones are single scalars.  The small side
// enters as hi + lo bf16 pairs (it is fp32 data: network input / loss gradient).  Persistent workgroups; wave w owns the
// taps {w, w+4, w+8}; partial rows in the layout of smallconv_wgrad_kernel, so the same reduction finishes them.
template <typename T, int CL>
__global__ __launch_bounds__(256) void smallconv_wgrad_mfma_kernel(SWArgs a) {
  constexpr int SZ = (int)sizeof(T);
  constexpr bool IS_BF16 = SZ == 2;
  constexpr int N = Vec16<T>::N;
  constexpr int PL = CL * SZ + (IS_BF16 ? 16 : 4);    // L tile pixel pitch
  constexpr int L_BYTES = TS * TS * PL;
  constexpr int PD = IS_BF16 ? 16 : 36;               // S tile: [halo px][8 slots]
  constexpr int D_BYTES = (HS * HS) * PD + 128;       // + slack: transposed reads of slots 8..31 run into the following pixels
  constexpr int NPARTS = IS_BF16 ? 2 : 1;
  constexpr int MT = CL / 32;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsL = smem;
  char* ldsD = smem + L_BYTES;                        // NPARTS tiles
  float* ldsR = reinterpret_cast<float*>(smem);       // final reduction scratch (after the loop)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int q = lane & 15;
  const int tr_col_b = (((lane >> 4) & 1) * 16 + (q & 3) * 4) * 2;
  const int tr_row = q >> 2;
  f32x16 acc[3][MT];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][m][r] = 0.f;
  constexpr int PPR = CL / N;
  constexpr int L_ROUNDS = (TS * TS * PPR + 255) / 256;
  constexpr int S_ROUNDS = (HS * HS * 8 + 255) / 256;
  uint4 rl[L_ROUNDS];
  float rs[S_ROUNDS], bs[S_ROUNDS];                    // bs: running sums of this thread's small-side elements (bias gradient)
#pragma unroll
  for (int i = 0; i < S_ROUNDS; ++i) bs[i] = 0.f;
  const int ntiles = a.B * a.tilesY * a.tilesX;
  auto gload = [&](int tile) {
    int t = tile;
    const int tx_id = t % a.tilesX; t /= a.tilesX;
    const int ty_id = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty_id * TS, x0 = tx_id * TS;
    const float* Sb = a.S + (size_t)b * a.CS * a.H * a.W;
    const T* Lb = reinterpret_cast<const T*>(a.L) + (size_t)b * a.H * a.W * CL;
#pragma unroll
    for (int i = 0; i < L_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int px = p / PPR, part = p % PPR;
      const int yy = y0 + px / TS, xx = x0 + px % TS;
      rl[i] = make_uint4(0, 0, 0, 0);
      if (px < TS * TS && yy < a.H && xx < a.W) rl[i] = *reinterpret_cast<const uint4*>(Lb + ((size_t)yy * a.W + xx) * CL + part * N);
    }
#pragma unroll
    for (int i = 0; i < S_ROUNDS; ++i) {
      const int e = i * 256 + tid;
      const int slot = e / (HS * HS), px = e % (HS * HS);
      const int yy = y0 + px / HS - 1, xx = x0 + px % HS - 1;
      rs[i] = 0.f;
      if (slot < a.CS && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) rs[i] = Sb[((size_t)slot * a.H + yy) * a.W + xx];
    }
  };
  if ((int)blockIdx.x < ntiles) gload(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                                   // the previous tile's readers are done
#pragma unroll
    for (int i = 0; i < L_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int px = p / PPR, part = p % PPR;
      if (px < TS * TS) *reinterpret_cast<uint4*>(ldsL + px * PL + part * 16) = rl[i];
    }
#pragma unroll
    for (int i = 0; i < S_ROUNDS; ++i) {
      const int e = i * 256 + tid;
      const int slot = e / (HS * HS), px = e % (HS * HS);
      if (slot < 8) {
        if constexpr (IS_BF16) {
          const bf16_t hi = (bf16_t)rs[i];
          *reinterpret_cast<bf16_t*>(ldsD + px * PD + slot * 2) = hi;
          *reinterpret_cast<bf16_t*>(ldsD + D_BYTES + px * PD + slot * 2) = (bf16_t)(rs[i] - (float)hi);
        } else {
          *reinterpret_cast<float*>(ldsD + px * PD + slot * 4) = rs[i];
        }
        const int hy = px / HS, hx = px % HS;
        if (hy >= 1 && hy <= TS && hx >= 1 && hx <= TS) bs[i] += rs[i];     // tile interior (zero beyond the image)
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) gload(tile + gridDim.x);
    // k-step = one tile row of 16 pixels
#ifndef IM2IM_SWG_UNROLL
#define IM2IM_SWG_UNROLL 1
#endif
#pragma unroll IM2IM_SWG_UNROLL
    for (int ty = 0; ty < TS; ++ty) {
      if constexpr (IS_BF16) {
        short8 fa[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          const char* p0 = ldsL + (ty * TS + half * 8 + tr_row) * PL + m * 64 + tr_col_b;
          fa[m] = WgFrag::load(p0, p0 + 4 * PL);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          const int tp = wave + 4 * i;
          if (tp < 9) {
            const char* d0 = ldsD + ((ty + tp / 3) * HS + (tp % 3) + half * 8 + tr_row) * PD + tr_col_b;
#pragma unroll
            for (int part = 0; part < NPARTS; ++part) {
              const short8 fb = WgFrag::load(d0 + part * D_BYTES, d0 + part * D_BYTES + 4 * PD);
#pragma unroll
              for (int m = 0; m < MT; ++m) acc[i][m] = SmallFrag<bf16_t>::mfma(fa[m], fb, acc[i][m]);
            }
          }
        }
      } else {
#pragma unroll 2
        for (int kk = 0; kk < 8; ++kk) {                 // two pixels per MFMA: x = 2 kk + half
          const int tx = 2 * kk + half;
          float fa[MT];
#pragma unroll
          for (int m = 0; m < MT; ++m) fa[m] = *reinterpret_cast<const float*>(ldsL + (ty * TS + tx) * PL + (m * 32 + l31) * 4);
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            const int tp = wave + 4 * i;
            if (tp < 9) {
              const float fb = *reinterpret_cast<const float*>(ldsD + ((ty + tp / 3) * HS + tx + (tp % 3)) * PD + (l31 & 7) * 4);
#pragma unroll
              for (int m = 0; m < MT; ++m) acc[i][m] = SmallFrag<float>::mfma(fa[m], fb, acc[i][m]);
            }
          }
        }
      }
    }
  }
  // partial row of this workgroup: out[(s*9 + tap)*CL + l], then the CS plane sums
  const int K = a.CS * 9 * CL + a.CS;
  float* out = a.partial + (size_t)blockIdx.x * K;
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int tp = wave + 4 * i;
    if (tp < 9 && l31 < a.CS) {
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int l = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          out[((size_t)l31 * 9 + tp) * CL + l] = acc[i][m][r];
        }
    }
  }
  __syncthreads();
  // plane sums: element e = i*256 + tid belongs to slot e / (HS*HS); summed per slot in a fixed order
#pragma unroll
  for (int i = 0; i < S_ROUNDS; ++i) ldsR[i * 256 + tid] = bs[i];
  __syncthreads();
  if (tid < a.CS) {
    float v = 0.f;
    for (int e = tid * HS * HS; e < (tid + 1) * HS * HS; ++e) v += ldsR[e];
    out[(size_t)a.CS * 9 * CL + tid] = v;
  }
}

// [r6] the bf16 / 32-channel form of the kernel above (the quantile heads' weight gradient: 3 planes x 32 channels, 0.397 ms at batch 78
// against 0.12 ms of traffic):
//   * the small side's hi and lo bf16 halves ride in ONE MFMA: the LDS row of a halo pixel holds 16 slots (hi in 0-7, lo in 8-15), so
//     columns 0-7 of the B fragment are the hi planes and 8-15 the lo planes; hi and lo products accumulate in separate accumulator
//     columns and are added once at the end -- half the matrix instructions of the kernel above;
//   * the host launches exactly one resident round of persistent workgroups (768 = 3 per CU by registers) instead of 2,048.
// MODE 1 (default): a wave owns the taps {w, w + 4, w + 8} over all 16 rows, as above (48 accumulators, 3 workgroups per CU): 0.293 ms at
// batch 78, 0.058 at batch 10.  MODE 0 (IM2IM_SMALLCONV_VALU bit 64): a wave owns half of the tile's rows and five (four) of the nine
// taps -- balanced MFMA counts (40 / 32 per wave instead of 48 / 32 / 32 / 32), but 80 accumulators leave two workgroups per CU and the two
// row halves have to be summed through LDS: 0.327 / 0.115 ms.  The matrix instructions were not the limit (profiles/r06_ab_experiments.txt 4).
constexpr size_t wgrad_rows_smem() { return (size_t)TS * TS * (32 * 2 + 16) + (size_t)HS * HS * 32 + 128; }
template <int MODE>      // 0: wave = (row half, tap group of 5 / 4), 80 accumulators; 1: wave = taps {w, w + 4, w + 8} over all rows, 48 accumulators
__global__ __launch_bounds__(256, MODE == 0 ? 2 : 3) void smallconv_wgrad_mfma_rows_kernel(SWArgs a) {
  using T = bf16_t;
  constexpr int CL = 32, N = 8;
  constexpr int PL = CL * 2 + 16;                      // L tile pixel pitch
  constexpr int L_BYTES = TS * TS * PL;
  constexpr int PD = 32;                               // S tile: [halo px][8 hi slots | 8 lo slots]
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsL = smem;
  char* ldsD = smem + L_BYTES;
  float* ldsR = reinterpret_cast<float*>(smem);        // final reduction scratch (after the loop)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int q = lane & 15;
  const int tr_col_b = (((lane >> 4) & 1) * 16 + (q & 3) * 4) * 2;
  const int tr_row = q >> 2;
  constexpr int NTP = MODE == 0 ? 5 : 3;               // taps of a wave.  MODE 0: tap group tg owns taps [5 tg, 5 tg + 5) of the nine
  const int rp = MODE == 0 ? (wave & 1) : 0, tg = MODE == 0 ? (wave >> 1) : wave;   // MODE 0: wave = (row half rp, tap group tg)
  constexpr int ROWS = MODE == 0 ? TS / 2 : TS;        // k-steps of a wave per tile
  f32x16 acc[NTP];
#pragma unroll
  for (int i = 0; i < NTP; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  constexpr int PPR = CL / N;
  constexpr int L_ROUNDS = (TS * TS * PPR + 255) / 256;
  constexpr int S_ROUNDS = (HS * HS * 8 + 255) / 256;
  uint4 rl[L_ROUNDS];
  float rs[S_ROUNDS], bs[S_ROUNDS];
#pragma unroll
  for (int i = 0; i < S_ROUNDS; ++i) bs[i] = 0.f;
  const int ntiles = a.B * a.tilesY * a.tilesX;
  auto gload = [&](int tile) {
    int t = tile;
    const int tx_id = t % a.tilesX; t /= a.tilesX;
    const int ty_id = t % a.tilesY;
    const int b = t / a.tilesY;
    const int y0 = ty_id * TS, x0 = tx_id * TS;
    const float* Sb = a.S + (size_t)b * a.CS * a.H * a.W;
    const T* Lb = reinterpret_cast<const T*>(a.L) + (size_t)b * a.H * a.W * CL;
#pragma unroll
    for (int i = 0; i < L_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int px = p / PPR, part = p % PPR;
      const int yy = y0 + px / TS, xx = x0 + px % TS;
      rl[i] = make_uint4(0, 0, 0, 0);
      if (px < TS * TS && yy < a.H && xx < a.W) rl[i] = *reinterpret_cast<const uint4*>(Lb + ((size_t)yy * a.W + xx) * CL + part * N);
    }
#pragma unroll
    for (int i = 0; i < S_ROUNDS; ++i) {
      const int e = i * 256 + tid;
      const int slot = e / (HS * HS), px = e % (HS * HS);
      const int yy = y0 + px / HS - 1, xx = x0 + px % HS - 1;
      rs[i] = 0.f;
      if (slot < a.CS && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) rs[i] = Sb[((size_t)slot * a.H + yy) * a.W + xx];
    }
  };
  // LDS byte offsets of this wave's taps relative to a tile row's first halo pixel (wave-uniform scalars)
  int tap_off[NTP];
#pragma unroll
  for (int i = 0; i < NTP; ++i) { const int tp = MODE == 0 ? tg * NTP + i : tg + 4 * i; tap_off[i] = ((tp / 3) * HS + (tp % 3)) * PD; }
  const int ntaps = MODE == 0 ? (tg == 0 ? NTP : 9 - NTP) : (tg == 0 ? 3 : 2);
  const int a_off = (half * 8 + tr_row) * PL + tr_col_b, d_off = (half * 8 + tr_row) * PD + tr_col_b;
  if ((int)blockIdx.x < ntiles) gload(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();                                   // the previous tile's readers are done
#pragma unroll
    for (int i = 0; i < L_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int px = p / PPR, part = p % PPR;
      if (px < TS * TS) *reinterpret_cast<uint4*>(ldsL + px * PL + part * 16) = rl[i];
    }
#pragma unroll
    for (int i = 0; i < S_ROUNDS; ++i) {
      const int e = i * 256 + tid;
      const int slot = e / (HS * HS), px = e % (HS * HS);
      if (slot < 8) {
        const bf16_t hi = (bf16_t)rs[i];
        *reinterpret_cast<bf16_t*>(ldsD + px * PD + slot * 2) = hi;
        *reinterpret_cast<bf16_t*>(ldsD + px * PD + 16 + slot * 2) = (bf16_t)(rs[i] - (float)hi);
        const int hy = px / HS, hx = px % HS;
        if (hy >= 1 && hy <= TS && hx >= 1 && hx <= TS) bs[i] += rs[i];     // tile interior (zero beyond the image)
      }
    }
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) gload(tile + gridDim.x);
#pragma unroll 2
    for (int r8 = 0; r8 < ROWS; ++r8) {                 // k-step = tile row ty: the A fragment feeds this wave's taps
      const int ty = rp * ROWS + r8;
      const char* p0 = ldsL + ty * TS * PL + a_off;
      const short8 fa = WgFrag::load(p0, p0 + 4 * PL);
      const char* drow = ldsD + ty * HS * PD + d_off;
#pragma unroll
      for (int i = 0; i < NTP; ++i)
        if (i < ntaps) {
          const short8 fb = WgFrag::load(drow + tap_off[i], drow + tap_off[i] + 4 * PD);
          acc[i] = SmallFrag<bf16_t>::mfma(fa, fb, acc[i]);
        }
    }
  }
  // hi + lo columns, then the two row halves' sums (rp = 0 first) through LDS:  red[(s*9 + tap)*CL + l]
  __syncthreads();
  const int K = a.CS * 9 * CL + a.CS;
  float* out = a.partial + (size_t)blockIdx.x * K;
#pragma unroll 1
  for (int w = 0; w < (MODE == 0 ? 2 : 1); ++w) {
    if (rp == w) {
#pragma unroll
      for (int i = 0; i < NTP; ++i) {
        const int tp = MODE == 0 ? tg * NTP + i : tg + 4 * i;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][r] + __shfl_down(acc[i][r], 8, 64);      // column s (hi) + column 8 + s (lo)
          if (i < ntaps && l31 < a.CS) {
            const int l = (r & 3) + 8 * (r >> 2) + 4 * half;
            float* dst = ldsR + ((size_t)l31 * 9 + tp) * CL + l;
            *dst = (w == 0) ? v : *dst + v;
          }
        }
      }
    }
    __syncthreads();
  }
  for (int i = tid; i < a.CS * 9 * CL; i += 256) out[i] = ldsR[i];
  __syncthreads();
  // plane sums: element e = i*256 + tid belongs to slot e / (HS*HS); summed per slot in a fixed order
#pragma unroll
  for (int i = 0; i < S_ROUNDS; ++i) ldsR[i * 256 + tid] = bs[i];
  __syncthreads();
  if (tid < a.CS) {
    float v = 0.f;
    for (int e = tid * HS * HS; e < (tid + 1) * HS * HS; ++e) v += ldsR[e];
    out[(size_t)a.CS * 9 * CL + tid] = v;
  }
}

// tmp[S][K] -> dw with index map, dbias tail
//   l_major != 0: dw[(l*CS + s)*9 + tap]        (first conv: weight [co=l][ci=s][tap])
//   l_major == 0: dw[(s*CL + l)*9 + (8 - tap)]  (heads: weight [co=s][ci=l][tap], correlation flipped)
__global__ __launch_bounds__(256) void smallconv_wgrad_final_kernel(const double* __restrict__ tmp, int S, int CS, int CL,
                                                                     int l_major, float* __restrict__ dw,
                                                                     float* __restrict__ dbias) {
  const int K = CS * 9 * CL + CS;
  const int k = blockIdx.x * 256 + threadIdx.x;
  if (k >= K) return;
  double v = 0.0;
  for (int i = 0; i < S; ++i) v += tmp[(size_t)i * K + k];
  if (k >= CS * 9 * CL) { if (dbias) dbias[k - CS * 9 * CL] = (float)v; return; }
  const int l = k % CL, tp = (k / CL) % 9, s = k / (9 * CL);
  if (l_major) dw[((size_t)l * CS + s) * 9 + tp] = (float)v;
  else dw[((size_t)s * CL + l) * 9 + (8 - tp)] = (float)v;
}

// IM2IM_SMALLCONV_VALU: bit mask that forces the VALU forms (A/B runs): 1 = heads forward, 2 = heads data-gradient, 4 = weight gradient;
// 16 = the generic form of smallconv_s2l_kernel also for the training launch of the first conv (LEAN off)
inline int valu_mask() {
  static const int m = [] { const char* e = getenv("IM2IM_SMALLCONV_VALU"); return e ? atoi(e) : 0; }();
  return m;
}

inline int swg_form() {      // IM2IM_SWG_DBG=<0..5>: multiply-add form of smallconv_wgrad_vec_kernel (CS = 1, CL = 64); unset: the default
  static const int m = [] { const char* e = getenv("IM2IM_SWG_DBG"); return e ? atoi(e) : -1; }();
  return m;
}

template <typename F> int for_dtype_cl(int dtype, int CL, F f) {
  if (dtype == IM2IM_BF16 && CL == 64) return f((bf16_t*)nullptr, std::integral_constant<int, 64>{});
  if (dtype == IM2IM_BF16 && CL == 32) return f((bf16_t*)nullptr, std::integral_constant<int, 32>{});
  if (dtype == IM2IM_F32 && CL == 64) return f((float*)nullptr, std::integral_constant<int, 64>{});
  if (dtype == IM2IM_F32 && CL == 32) return f((float*)nullptr, std::integral_constant<int, 32>{});
  return fail_invalid("small conv: dtype must be f32/bf16 and the wide side 32 or 64 channels");
}

}  // namespace

extern "C" int64_t im2im_smallconv_tiles(int32_t B, int32_t H, int32_t W) {
  return (int64_t)B * im2im::cdiv(H, TS) * im2im::cdiv(W, TS);
}

extern "C" int im2im_smallconv_s2l_fwd(const float* in, const float* w, const float* bias, const float* center,
                                       const float* scale_shift, void* out,
                                       float* stats, int32_t B, int32_t H, int32_t W, int32_t CS, int32_t CL, int32_t relu,
                                       int32_t flip, int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(in && w && out && B > 0 && H > 0 && W > 0 && CS >= 1 && CS <= CS_MAX);
  S2LArgs a{in, w, bias, center, scale_shift, out, stats, B, H, W, CS, (int)cdiv(H, TS), (int)cdiv(W, TS), relu, flip};
  return for_dtype_cl(dtype, CL, [&](auto* tag, auto cl) {
    using T = std::remove_pointer_t<decltype(tag)>;
    const dim3 grid((unsigned)(B * a.tilesY * a.tilesX));
    if (CS > 1 && !bias && !center && !scale_shift && !stats && !relu && !(valu_mask() & 2)) {
      // the heads' data-gradient (plain correlation into a wide NHWC tensor): matrix-core form
      constexpr int CLv = decltype(cl)::value;
      constexpr size_t smem = s2l_dgrad_mfma_smem<T, CLv>();
      auto kern = smallconv_s2l_dgrad_mfma_kernel<T, CLv>;
      if (smem > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_set = true; }
      }
      const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / smem));
      hipLaunchKernelGGL(kern, dim3(std::min<unsigned>(grid.x, 256u * per_cu)), dim3(256), smem, stream, a);
      return check_launch("smallconv_s2l_dgrad_mfma_kernel");
    }
    // (an exact-fp32 MFMA form of this layer was measured slower than the VALU kernel: K = 9*CS is too short to pay for the
    // LDS round trip of the accumulators -- 0.60 vs 0.50 ms at batch 78, 320x320, CS = 1)
    if (CS == 1 && a.stats && !a.scale_shift && !a.relu && H % TS == 0 && W % TS == 0 && !(valu_mask() & 16))
      hipLaunchKernelGGL((smallconv_s2l_kernel<T, decltype(cl)::value, true, 1>), grid, dim3(256), 0, stream, a);
    else if (CS == 1 && !a.stats && a.scale_shift && a.relu && H % TS == 0 && W % TS == 0 && !(valu_mask() & 16))
      hipLaunchKernelGGL((smallconv_s2l_kernel<T, decltype(cl)::value, true, 2>), grid, dim3(256), 0, stream, a);
    else if (CS == 1) hipLaunchKernelGGL((smallconv_s2l_kernel<T, decltype(cl)::value, true>), grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((smallconv_s2l_kernel<T, decltype(cl)::value, false>), grid, dim3(256), 0, stream, a);
    return check_launch("smallconv_s2l_kernel");
  });
}

extern "C" int im2im_smallconv_l2s_fwd(const void* in, const float* w, const float* bias, float* out, int32_t B, int32_t H,
                                       int32_t W, int32_t CL, int32_t CS, int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(in && w && out && B > 0 && H > 0 && W > 0 && CS >= 1 && CS <= CS_MAX);
  L2SArgs a{in, w, bias, out, B, H, W, CS, (int)cdiv(H, TS), (int)cdiv(W, TS)};
  return for_dtype_cl(dtype, CL, [&](auto* tag, auto cl) {
    using T = std::remove_pointer_t<decltype(tag)>;
    constexpr int CLv = decltype(cl)::value;
    if (valu_mask() & 1) {
      hipLaunchKernelGGL((smallconv_l2s_kernel<T, CLv>), dim3((unsigned)(B * a.tilesY * a.tilesX)), dim3(256), 0, stream, a);
      return check_launch("smallconv_l2s_kernel");
    }
    constexpr size_t smem = l2s_mfma_smem<T, CLv>();
    auto kern = smallconv_l2s_mfma_kernel<T, CLv>;
    if (smem > 64 * 1024) {
      static bool attr_set = false;
      if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_set = true; }
    }
    const int64_t ntiles = (int64_t)B * a.tilesY * a.tilesX;
    const int per_cu = (int)std::max<size_t>(1, std::min<size_t>(8, (160 * 1024) / smem));
    static const int l2s_blocks = [] { const char* e = getenv("IM2IM_L2S_BLOCKS"); return e ? atoi(e) : 0; }();     // A/B: persistent workgroups
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<int64_t>(ntiles, l2s_blocks > 0 ? l2s_blocks : 256 * per_cu)), dim3(256), smem, stream, a);
    return check_launch("smallconv_l2s_mfma_kernel");
  });
}

extern "C" int64_t im2im_smallconv_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t CS, int32_t CL) {
  const int64_t K = (int64_t)CS * 9 * CL + CS;
  return std::min<int64_t>(im2im_smallconv_tiles(B, H, W), 2048) * K * (int64_t)sizeof(float) + im2im::reduce_tmp_bytes(K);
}

extern "C" int im2im_smallconv_wgrad(const float* S, const void* L, float* dw, float* dbias, int32_t B, int32_t H, int32_t W,
                                     int32_t CS, int32_t CL, int32_t l_major, int32_t dtype, void* ws, int64_t ws_bytes,
                                     im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(S && L && dw && ws && B > 0 && H > 0 && W > 0 && CS >= 1 && CS <= CS_MAX);
  IM2IM_REQUIRE(ws_bytes >= im2im_smallconv_wgrad_workspace_bytes(B, H, W, CS, CL));
  // persistent blocks, one partial row each.  [r6] the heads' kernel (smallconv_wgrad_mfma_rows_kernel: 3 workgroups per CU by registers)
  // gets exactly one resident round of them (IM2IM_SWG_BLOCKS overrides: A/B)
  static const int swg_blocks = [] { const char* e = getenv("IM2IM_SWG_BLOCKS"); return e ? atoi(e) : 768; }();
  const bool rows_kernel = CS >= 3 && dtype == IM2IM_BF16 && CL == 32 && !(valu_mask() & (4 | 32));
  const int64_t nblk = std::min<int64_t>(im2im_smallconv_tiles(B, H, W), rows_kernel ? std::max(1, std::min(swg_blocks, 2048)) : 2048);
  const int64_t K = (int64_t)CS * 9 * CL + CS;
  float* partial = (float*)ws;
  double* tmp = (double*)((char*)ws + nblk * K * sizeof(float));
  SWArgs a{S, L, partial, B, H, W, CS, (int)cdiv(H, TS), (int)cdiv(W, TS)};
  return for_dtype_cl(dtype, CL, [&](auto* tag, auto cl) {
    using T = std::remove_pointer_t<decltype(tag)>;
    constexpr int CLv = decltype(cl)::value;
    if (rows_kernel && std::is_same<T, bf16_t>::value && CLv == 32) {
      // [r6] the heads' weight gradient: rows split over the waves, hi + lo halves in one MFMA (IM2IM_SMALLCONV_VALU bit 32 = the round-2 kernel)
      constexpr size_t smem = wgrad_rows_smem();
      static_assert(smem >= (size_t)((HS * HS * 8 + 255) / 256) * 256 * sizeof(float) && smem >= (size_t)CS_MAX * 9 * 32 * sizeof(float), "reduction scratch fits");
      if (valu_mask() & 64) hipLaunchKernelGGL(smallconv_wgrad_mfma_rows_kernel<0>, dim3((unsigned)nblk), dim3(256), smem, stream, a);
      else hipLaunchKernelGGL(smallconv_wgrad_mfma_rows_kernel<1>, dim3((unsigned)nblk), dim3(256), smem, stream, a);
    }
    else if (CS >= 3 && !(valu_mask() & 4)) {          // matrix cores pay from 3 planes up (CS = 1: 0.83 vs 0.48 ms, the VALU kernel wins)
      constexpr size_t smem = wgrad_mfma_smem<T, CLv>();
      static_assert(smem >= (size_t)((HS * HS * 8 + 255) / 256) * 256 * sizeof(float), "bias scratch fits");
      auto kern = smallconv_wgrad_mfma_kernel<T, CLv>;
      if (smem > 64 * 1024) {
        static bool attr_set = false;
        if (!attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem); attr_set = true; }
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), smem, stream, a);
    }
#ifdef IM2IM_BUILD_EXPERIMENTAL
    else if (CS == 1 && CLv == 64 && swg_form() >= 0 && !(valu_mask() & 8)) {      // the record of the bisect (tools/debug_victim2.sh)
#define IM2IM_SWG_CASE(F) if (swg_form() == F) hipLaunchKernelGGL((smallconv_wgrad_vec_kernel<T, 64, 1, 4, true, F>), dim3((unsigned)nblk), dim3(256), 0, stream, a); else
      IM2IM_SWG_CASE(0) IM2IM_SWG_CASE(1) IM2IM_SWG_CASE(2) IM2IM_SWG_CASE(3) IM2IM_SWG_CASE(4) IM2IM_SWG_CASE(5)
      return fail_invalid("IM2IM_SWG_DBG: forms 0..5");
#undef IM2IM_SWG_CASE
    }
#else
    else if (swg_form() >= 0) return fail_invalid("IM2IM_SWG_DBG needs a library built with IM2IM_BUILD_EXPERIMENTAL=1");
#endif
    else if (CS <= 2 && !(valu_mask() & 8) && (valu_mask() & 128)) {      // A/B: the wide side widened to fp32 when it is written to LDS
      if (CS == 1) hipLaunchKernelGGL((smallconv_wgrad_vec_kernel<T, CLv, 1, CLv / 16, true>), dim3((unsigned)nblk), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((smallconv_wgrad_vec_kernel<T, CLv, 2, CLv / 16, true>), dim3((unsigned)nblk), dim3(256), 0, stream, a);
    }
    else if (CS == 1 && !(valu_mask() & 8)) hipLaunchKernelGGL((smallconv_wgrad_vec_kernel<T, CLv, 1, CLv / 16>), dim3((unsigned)nblk), dim3(256), 0, stream, a);
    else if (CS == 2 && !(valu_mask() & 8)) hipLaunchKernelGGL((smallconv_wgrad_vec_kernel<T, CLv, 2, CLv / 16>), dim3((unsigned)nblk), dim3(256), 0, stream, a);
    else if (CS == 1) hipLaunchKernelGGL((smallconv_wgrad_kernel<T, CLv, 1>), dim3((unsigned)nblk), dim3(256), 0, stream, a);
    else if (CS == 2) hipLaunchKernelGGL((smallconv_wgrad_kernel<T, CLv, 2>), dim3((unsigned)nblk), dim3(256), 0, stream, a);
    else if (CS <= 4) hipLaunchKernelGGL((smallconv_wgrad_kernel<T, CLv, 4>), dim3((unsigned)nblk), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((smallconv_wgrad_kernel<T, CLv, 8>), dim3((unsigned)nblk), dim3(256), 0, stream, a);
    if (int rc = check_launch("smallconv_wgrad_kernel")) return rc;
    int rc;
    const int Sp = launch_reduce_stage1(partial, nblk, K, tmp, stream, &rc);
    if (rc) return rc;
    hipLaunchKernelGGL(smallconv_wgrad_final_kernel, dim3((unsigned)cdiv(K, 256)), dim3(256), 0, stream, (const double*)tmp, Sp,
                       (int)CS, (int)CL, (int)l_major, dw, dbias);
    return check_launch("smallconv_wgrad_final_kernel");
  });
}
