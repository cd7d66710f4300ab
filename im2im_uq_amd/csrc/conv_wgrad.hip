// Weight gradients of the MFMA convolutions and the weight packing for gfx950 (split from conv_mfma.hip so that the two
// halves compile in parallel): conv_wgrad_kernel (fp32, 1x1), conv_wgrad_pipe_kernel (bf16 3x3, software-pipelined 12-wave
// workgroups), the deterministic split-K reduction, and the packers of the forward / data-gradient weight operands.
// Replaces the weight half of autograd's backward of nn.Conv2d (core/models/trunks/unet_parts.py:16,19,90 under
// loss.backward(), core/scripts/train.py:159).
#include "conv_common.h"
#ifndef IM2IM_WGRAD_XCD
#define IM2IM_WGRAD_XCD 1
#endif
#ifndef IM2IM_WGRAD_ABL      // measurement-only: bit 0 = no global loads after the first tile, bit 1 = no LDS writes, bit 2 = no MFMA phase, bit 3 = loads of the same (cache-hot) tile, bit 4 = the dz half of the staging only for the first tile
#define IM2IM_WGRAD_ABL 0
#endif
#ifndef IM2IM_ROLL_DIST      // conv_wgrad_roll_kernel: operand prefetch distance in units (1 .. 3; measured 1,076 / 1,098 / 1,066 TF over the 13 layers)
#define IM2IM_ROLL_DIST 2
#endif
#ifndef IM2IM_ROLL_PIN       // ... sched_group_barrier pins "reads, then MFMAs" per unit
#define IM2IM_ROLL_PIN 1
#endif
#ifndef IM2IM_ROLL_SLOT      // ... where a tile's staging pieces go: 0 = spread over the k-steps, 1 = first half, 2 = second half
#define IM2IM_ROLL_SLOT 0
#endif
#include <string>
#include <type_traits>
#include <vector>

namespace {

using namespace im2im;

// ------------------------------------------------------------------------------------------------
// wgrad:  dW[co][tap][ci] = sum over pixels of dz[p][co] * x[p + tap][ci]
// Block = 64 co x 64 ci x all taps; K runs over pixel tiles (TH x TW), split across blockIdx.y.
// bf16: both operands need k (= pixel) contiguous per lane but live channel-contiguous in LDS, so
// they are fetched with ds_read_b64_tr_b16 (hardware 4x16 transpose, semantics verified by
// tools/hwprobe): lane q of a 16-lane group supplies row q>>2, 8-byte quad q&3 and receives column
// l&15.  fp32: v_mfma_f32_32x32x2_f32 takes one scalar per lane, read directly.
struct WgradArgs {
  const void* x;     // [B][H][W][Ci] T  (layer input)
  const void* dz;    // [B][H][W][Co] T
  float* partial;    // [nsplit][Co][TAPS][Ci] fp32
  int B, H, W, Ci, Co, tilesY, tilesX, ntiles, tiles_per_split;
  const float* x_ss; // [2][Ci] or null: x is the producer's pre-BatchNorm z; staging applies max(z*scale+shift, 0)
  const void* x_hi;  // null, or: input channels [Ci_lo, Ci) live here (see ConvArgs); x_ss_hi = its lazy coefficients
  const float* x_ss_hi;
  int Ci_lo;
};

// source tensor of a 64-channel input block: base pointer (at the block's first channel), pixel stride, lazy coefficients
template <typename T> struct WgradSrc {
  const T* x; int stride; const float* sc; const float* sh;
  __device__ __forceinline__ WgradSrc(const WgradArgs& a, int ci0) {
    const bool split = a.x_hi != nullptr, hi = split && ci0 >= a.Ci_lo;
    stride = split ? a.Ci_lo : a.Ci;
    const int c = hi ? ci0 - a.Ci_lo : ci0;
    x = reinterpret_cast<const T*>(hi ? a.x_hi : a.x) + c;
    const float* ss = hi ? a.x_ss_hi : a.x_ss;
    sc = ss ? ss + c : nullptr;
    sh = ss ? ss + stride + c : nullptr;
  }
};

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

template <typename T> struct WFrag;
template <> struct WFrag<bf16_t> {
  using AB = short8;
  static constexpr int KPX = 16;                    // pixels per MFMA k-step
  // rowbase: LDS byte address of pixel-row 0 of this k-step's 16-pixel run for this lane's half;
  // rows[i] = byte offset of pixel i (0..7) of the half relative to lds; col_b = byte offset of channel
  static __device__ __forceinline__ AB load(const char* p0, const char* p1) {
    short4v lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(lds_char*)p0);
    short4v hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4*)(lds_char*)p1);
    AB r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return r;
  }
  static __device__ __forceinline__ f32x16 mfma(AB a, AB b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(a), as_bf16x8(b), c, 0, 0, 0);
  }
};

template <typename T, int TH, int TW, int TAPS>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  constexpr int PAD = (TAPS == 9) ? 1 : 0;
  constexpr int HH = TH + 2 * PAD, HWD = TW + 2 * PAD, HPX = HH * HWD;
  constexpr int M = TH * TW;
  constexpr int CT = 64;                            // channels per tile (both co and ci)
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int EPP = 16 / (int)sizeof(T);
  constexpr int PPR = CT / EPP;                     // 8 (bf16) or 16 (fp32)
  constexpr int PB = IS_BF16 ? 192 : 272;           // LDS row pitch (bytes)
  constexpr int A_BYTES = M * PB;
  constexpr int A_ROUNDS = (M * PPR + 255) / 256, B_ROUNDS = (HPX * PPR + 255) / 256;
  constexpr int KPX = IS_BF16 ? 16 : 2;
  constexpr int KSTEPS = M / KPX;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsA = smem;                                // dz tile  [M][64 co]
  char* ldsB = smem + A_BYTES;                      // x halo   [HPX][64 ci]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wco = wave >> 1, wci = wave & 1;        // 2 x 2 waves, 32 co x 32 ci each
  const int half = lane >> 5, l31 = lane & 31;
  const int ci_tiles = (a.Ci + CT - 1) / CT;
  const int co0 = (blockIdx.x / ci_tiles) * CT, ci0 = (blockIdx.x % ci_tiles) * CT;   // Co / Ci may be 32 mod 64: masked
  const WgradSrc<T> xs(a, ci0);
  const T* __restrict__ xg = xs.x;
  const T* __restrict__ dzg = reinterpret_cast<const T*>(a.dz);

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // lane-constant pieces of the operand addresses
  //   bf16 tr-read: group g = lane>>4 -> channel sub-block (g&1)*16; lane q = lane&15 supplies
  //   pixel row (q>>2) of its 4-row block and the 8-byte quad (q&3).
  const int q = lane & 15;
  const int tr_col_b = (((lane >> 4) & 1) * 16 + (q & 3) * 4) * 2;   // byte offset of the quad's first channel
  const int tr_row = q >> 2;

  const int t_begin = blockIdx.y * a.tiles_per_split;
  const int t_end = min(t_begin + a.tiles_per_split, a.ntiles);
  // The next tile is fetched into registers while this one's MFMAs run, and only written to LDS after the barrier that ends
  // them -- the single-buffered loop spent its time waiting for loads (OutConv's 1x1 weight gradient: 0.51 -> 0.35 ms for
  // 1.5 GB; the fp32 3x3 weight gradients, one workgroup per CU with its 84 KB of LDS: 124 ms of the 305 ms fp32 step).
  // fp32 3x3 holds 20 pieces = 80 registers: fine, a lone workgroup per CU may use all 512.  bf16 3x3 is
  // conv_wgrad_pipe_kernel's job; this kernel is only its fallback there and stages straight into LDS.
  constexpr bool PREFETCH = (TAPS == 1) || !IS_BF16;
  uint4 ra[PREFETCH ? A_ROUNDS : 1], rb[PREFETCH ? B_ROUNDS : 1];
  auto fetch = [&](int t, auto&& put_a, auto&& put_b) {
    int tt = t;
    const int tx_id = tt % a.tilesX; tt /= a.tilesX;
    const int ty_id = tt % a.tilesY;
    const int b = tt / a.tilesY;
    const int y0 = ty_id * TH, x0 = tx_id * TW;
    const T* xb = xg + (size_t)b * a.H * a.W * xs.stride;
    const T* dzb = dzg + (size_t)b * a.H * a.W * a.Co;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int px = p / PPR, part = p % PPR;
      const int yy = y0 + px / TW, xx = x0 + px % TW;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (px < M && yy < a.H && xx < a.W && co0 + part * EPP < a.Co)
        v = *reinterpret_cast<const uint4*>(dzb + ((size_t)yy * a.W + xx) * a.Co + co0 + part * EPP);
      put_a(i, px, part, v);
    }
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i) {
      const int p = i * 256 + tid;
      const int px = p / PPR, part = p % PPR;
      const int yy = y0 + px / HWD - PAD, xx = x0 + px % HWD - PAD;
      uint4 v = make_uint4(0, 0, 0, 0);
      bool real = false;
      if (px < HPX && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W && ci0 + part * EPP < a.Ci) {
        v = *reinterpret_cast<const uint4*>(xb + ((size_t)yy * a.W + xx) * xs.stride + part * EPP);
        real = true;
      }
      put_b(i, px, part, v, real);
    }
  };
  auto lazy = [&](uint4& v, int part) {
    if (xs.sc) {
      float f[EPP];
      Vec16<T>::load(reinterpret_cast<const T*>(&v), f);
#pragma unroll
      for (int k = 0; k < EPP; ++k) f[k] = fmaxf(f[k] * xs.sc[part * EPP + k] + xs.sh[part * EPP + k], 0.f);
      Vec16<T>::store(reinterpret_cast<T*>(&v), f);
    }
  };
  unsigned b_real = 0;
  if constexpr (PREFETCH) {
    if (t_begin < t_end)
      fetch(t_begin, [&](int i, int, int, const uint4& v) { ra[i] = v; },
            [&](int i, int, int, const uint4& v, bool real) { rb[i] = v; b_real = real ? (b_real | (1u << i)) : (b_real & ~(1u << i)); });
  }
  for (int t = t_begin; t < t_end; ++t) {
    if (t != t_begin) __syncthreads();
    if constexpr (PREFETCH) {
#pragma unroll
      for (int i = 0; i < A_ROUNDS; ++i) {
        const int p = i * 256 + tid;
        if (p / PPR < M) *reinterpret_cast<uint4*>(ldsA + (p / PPR) * PB + (p % PPR) * 16) = ra[i];
      }
#pragma unroll
      for (int i = 0; i < B_ROUNDS; ++i) {
        const int p = i * 256 + tid;
        if (p / PPR < HPX) {
          uint4 v = rb[i];
          if ((b_real >> i) & 1) lazy(v, p % PPR);
          *reinterpret_cast<uint4*>(ldsB + (p / PPR) * PB + (p % PPR) * 16) = v;
        }
      }
    } else {
      fetch(t, [&](int, int px, int part, const uint4& v) { if (px < M) *reinterpret_cast<uint4*>(ldsA + px * PB + part * 16) = v; },
            [&](int, int px, int part, uint4 v, bool real) {
              if (real) lazy(v, part);
              if (px < HPX) *reinterpret_cast<uint4*>(ldsB + px * PB + part * 16) = v;
            });
    }
    __syncthreads();
    if constexpr (PREFETCH) {
      if (t + 1 < t_end)
        fetch(t + 1, [&](int i, int, int, const uint4& v) { ra[i] = v; },
              [&](int i, int, int, const uint4& v, bool real) { rb[i] = v; b_real = real ? (b_real | (1u << i)) : (b_real & ~(1u << i)); });
    }
#pragma unroll 2
    for (int ks = 0; ks < KSTEPS; ++ks) {
      if constexpr (IS_BF16) {
        // this lane's two 4-pixel row groups of the k-step: pixels m = ks*16 + half*8 + {0..3, 4..7} (+ tr_row)
        const int m0 = ks * 16 + half * 8 + tr_row, m1 = m0 + 4;
        const char* pa0 = ldsA + m0 * PB + wco * 64 + tr_col_b;
        const char* pa1 = ldsA + m1 * PB + wco * 64 + tr_col_b;
        const short8 fa = WFrag<bf16_t>::load(pa0, pa1);
        const int h0 = ((m0 / TW) * HWD + (m0 % TW)) * PB + wci * 64 + tr_col_b;
        const int h1 = ((m1 / TW) * HWD + (m1 % TW)) * PB + wci * 64 + tr_col_b;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          const int toff = (TAPS == 9) ? ((tp / 3) * HWD + (tp % 3)) * PB : 0;
          const short8 fb = WFrag<bf16_t>::load(ldsB + h0 + toff, ldsB + h1 + toff);
          acc[tp] = WFrag<bf16_t>::mfma(fa, fb, acc[tp]);
        }
      } else {
        const int m = ks * 2 + half;
        const float fa = *reinterpret_cast<const float*>(ldsA + m * PB + (wco * 32 + l31) * 4);
        const int hb = ((m / TW) * HWD + (m % TW)) * PB + (wci * 32 + l31) * 4;
#pragma unroll
        for (int tp = 0; tp < TAPS; ++tp) {
          const int toff = (TAPS == 9) ? ((tp / 3) * HWD + (tp % 3)) * PB : 0;
          const float fb = *reinterpret_cast<const float*>(ldsB + hb + toff);
          acc[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[tp], 0, 0, 0);
        }
      }
    }
  }
  // partial[split][co][tap][ci]
  float* out = a.partial + (size_t)blockIdx.y * a.Co * TAPS * a.Ci;
#pragma unroll
  for (int tp = 0; tp < TAPS; ++tp)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int co = co0 + wco * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
      const int ci = ci0 + wci * 32 + l31;
      if (co < a.Co && ci < a.Ci) out[((size_t)co * TAPS + tp) * a.Ci + ci] = acc[tp][r];
    }
}

// bf16 3x3 wgrad, software-pipelined: one workgroup of 12 waves per CU = 2x2 (co, ci) quadrants x 3 tap groups
// (kernel rows).  Each wave keeps only its 3 taps' accumulators (48 registers), so there is room to prefetch the
// NEXT pixel tile into registers while the MFMAs of the current one run from LDS; the tile is then written to the
// other LDS buffer and one barrier per tile separates the two.  (The single-buffered kernel above spends 63 % of
// its wave cycles waiting on memory; SQ_WAIT_ANY, profiles/.)
// COT = output channels per workgroup: 64 (a wave = 32 co x 32 ci x 3 taps, 48 accumulators) or 128 [r3] (a wave = 64 co x 32 ci x
// 3 taps, 96 accumulators: every x fragment feeds two MFMAs, 1.7 transposing LDS reads per MFMA instead of 2.7, and an x tile is
// fetched once per 128 output channels instead of once per 64).
template <int TH, int TW, int COT>
__global__ __launch_bounds__(768) void conv_wgrad_pipe_kernel(WgradArgs a) {
  using T = bf16_t;
  constexpr int NT = 768;
  constexpr int HH = TH + 2, HWD = TW + 2, HPX = HH * HWD;
  constexpr int M = TH * TW;
  // [r3] TH = 16 (256-pixel tiles, COT = 64 only): twice the MFMA work between two barriers, so a tile's loads have twice as long
  // to arrive (the 320x320 layers wait on HBM at every 128-pixel tile: cache-hot loads ran them 34 % faster), and 1.27x instead of
  // 1.41x halo.  Two such tiles only fit the LDS unpadded (128 B per pixel): instead of the 64-byte pad, the 64-byte half of a
  // pixel row is XOR-swizzled with bit 1 of the row index, which gives the transposing reads (4 consecutive rows x 64 B per 32
  // lanes) four distinct 16-bank windows again.
  constexpr bool SWZ = TH == 16;
  static_assert(!SWZ || COT == 64, "swizzled 256-pixel tiles: 64 output channels");
  constexpr int CT = 64, EPP = 8, PPR = 8, PB = SWZ ? 128 : 192;   // x: 64 input channels per workgroup, 128 B of data (+ 64 B pad) per pixel
  constexpr int CJ = COT / 64;                         // 32-channel co sub-blocks per wave
  constexpr int PA = SWZ ? COT * 2 : COT * 2 + 64;     // dz pixel pitch: 192 / 320 B (rows land on distinct 16-bank windows), 128 swizzled
  constexpr int PPRA = COT / 8;                        // 16-byte pieces per dz pixel
  constexpr int A_BYTES = M * PA, B_BYTES = HPX * PB, BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int A_ROUNDS = (M * PPRA + NT - 1) / NT, B_ROUNDS = (HPX * PPR + NT - 1) / NT;
  constexpr int KSTEPS = M / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave >> 2;                          // tap group = kernel row kh
  const int wco = (wave >> 1) & 1, wci = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int ci_tiles = a.Ci / CT;
  // Workgroups are dealt to the 8 XCDs round-robin in dispatch order (x fastest).  All channel blocks of one pixel
  // split read the same dz / x pixels, so they are renumbered to sit on ONE XCD and share them through its L2
  // instead of each XCD fetching them from HBM.
  int cb = blockIdx.x, split = blockIdx.y;
#if IM2IM_WGRAD_XCD
  if ((gridDim.y & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    split = (j / (int)gridDim.x) * 8 + xcd;
    cb = j % (int)gridDim.x;
  }
#endif
  const int co0 = (cb / ci_tiles) * COT, ci0 = (cb % ci_tiles) * CT;
  const WgradSrc<T> xs(a, ci0);
  const T* __restrict__ xg = xs.x;
  const T* __restrict__ dzg = reinterpret_cast<const T*>(a.dz);

  f32x16 acc[CJ][3];
#pragma unroll
  for (int j = 0; j < CJ; ++j)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

  const int q = lane & 15;
  const int tr_col_b = (((lane >> 4) & 1) * 16 + (q & 3) * 4) * 2;
  const int tr_row = q >> 2;

  // staging pieces of this thread (tile-independent parts)
  int a_px[A_ROUNDS], a_part[A_ROUNDS], b_px[B_ROUNDS], b_part[B_ROUNDS];
#pragma unroll
  for (int i = 0; i < A_ROUNDS; ++i) { const int p = i * NT + tid; a_px[i] = (p < M * PPRA) ? p / PPRA : -1; a_part[i] = p % PPRA; }
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) { const int p = i * NT + tid; b_px[i] = (p < HPX * PPR) ? p / PPR : -1; b_part[i] = p % PPR; }
  struct Stage { uint4 a[A_ROUNDS], b[B_ROUNDS]; unsigned valid; };   // one tile in flight; valid bit i: b[i] holds real pixels (not zero padding)
  // lazy BatchNorm coefficients of the 64 input channels: in registers (COT = 64) or, where the 96 accumulators leave no room
  // for 16 more live values, in LDS behind the tile buffers and read back per tile (COT = 128)
  constexpr bool SS_LDS = COT > 64;
  float xsc[SS_LDS ? 1 : EPP], xsh[SS_LDS ? 1 : EPP];   // this thread's pieces always cover channels ci0 + (tid % 8)*8 ...
  float* ldsSS = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);
  const bool lazy_x = xs.sc != nullptr;
  if (lazy_x) {
    if constexpr (SS_LDS) {
      if (tid < CT) { ldsSS[tid] = xs.sc[tid]; ldsSS[CT + tid] = xs.sh[tid]; }
      __syncthreads();
    } else {
#pragma unroll
      for (int k = 0; k < EPP; ++k) { xsc[k] = xs.sc[(tid % PPR) * EPP + k]; xsh[k] = xs.sh[(tid % PPR) * EPP + k]; }
    }
  }

  bool abl_first = true; (void)abl_first;
  // where tile t lies: first pixel, image base pointers
  struct TilePos { int y0, x0; const T* xb; const T* dzb; };
  auto tile_pos = [&](int t) __attribute__((always_inline)) -> TilePos {
    int tt = t;
    const int tx_id = tt % a.tilesX; tt /= a.tilesX;
    const int ty_id = tt % a.tilesY;
    const int b = tt / a.tilesY;
    return TilePos{ty_id * TH, tx_id * TW, xg + (size_t)b * a.H * a.W * xs.stride, dzg + (size_t)b * a.H * a.W * a.Co + co0};
  };
  // staging piece p of a tile: p < A_ROUNDS = the thread's dz piece p, else its x (halo) piece p - A_ROUNDS
  auto gload_piece = [&](const TilePos& tp, Stage& R, int p) __attribute__((always_inline)) {
    if (p < A_ROUNDS) {
      const int i = p;
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a_px[i] >= 0) {
        const int yy = tp.y0 + a_px[i] / TW, xx = tp.x0 + a_px[i] % TW;
        if (yy < a.H && xx < a.W && co0 + a_part[i] * EPP < a.Co)
          v = *reinterpret_cast<const uint4*>(tp.dzb + ((size_t)yy * a.W + xx) * a.Co + a_part[i] * EPP);
      }
      R.a[i] = v;
    } else {
      const int i = p - A_ROUNDS;
      uint4 v = make_uint4(0, 0, 0, 0);
      bool ok = false;
      if (b_px[i] >= 0) {
        const int yy = tp.y0 + b_px[i] / HWD - 1, xx = tp.x0 + b_px[i] % HWD - 1;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
          v = *reinterpret_cast<const uint4*>(tp.xb + ((size_t)yy * a.W + xx) * xs.stride + b_part[i] * EPP);
          ok = true;
        }
      }
      R.valid = (R.valid & ~(1u << i)) | ((ok ? 1u : 0u) << i);
      R.b[i] = v;
    }
  };
  auto swrite_piece = [&](int buf, const Stage& R, int p) __attribute__((always_inline)) {
    char* la = smem + buf * BUF_BYTES;
    char* lb = la + A_BYTES;
    if (p < A_ROUNDS) {
      const int i = p;
      if (a_px[i] >= 0) *reinterpret_cast<uint4*>(la + a_px[i] * PA + (SWZ ? (a_part[i] ^ (((a_px[i] >> 1) & 1) << 2)) : a_part[i]) * 16) = R.a[i];
    } else {
      const int i = p - A_ROUNDS;
      if (b_px[i] >= 0) {
        uint4 v = R.b[i];
        if (lazy_x && ((R.valid >> i) & 1)) {
          float f[EPP];
          Vec16<T>::load(reinterpret_cast<const T*>(&v), f);
          if constexpr (SS_LDS) {
            const int c0 = (tid % PPR) * EPP;
#pragma unroll
            for (int k = 0; k < EPP; ++k) f[k] = fmaxf(f[k] * ldsSS[c0 + k] + ldsSS[CT + c0 + k], 0.f);
          } else {
#pragma unroll
            for (int k = 0; k < EPP; ++k) f[k] = fmaxf(f[k] * xsc[k] + xsh[k], 0.f);
          }
          Vec16<T>::store(reinterpret_cast<T*>(&v), f);
        }
        *reinterpret_cast<uint4*>(lb + b_px[i] * PB + (SWZ ? (b_part[i] ^ (((b_px[i] >> 1) & 1) << 2)) : b_part[i]) * 16) = v;
      }
    }
  };
  constexpr int NPIECES = A_ROUNDS + B_ROUNDS;
  auto gload = [&](int t, Stage& R) __attribute__((always_inline)) {
    const TilePos tp = tile_pos(t);
    R.valid = 0;
#pragma unroll
    for (int p = 0; p < NPIECES; ++p) {
#if IM2IM_WGRAD_ABL & 16
      if (p < A_ROUNDS && !abl_first) continue;
#endif
      gload_piece(tp, R, p);
    }
  };
  auto swrite = [&](int buf, const Stage& R) __attribute__((always_inline)) {
#pragma unroll
    for (int p = 0; p < NPIECES; ++p) {
#if IM2IM_WGRAD_ABL & 16
      if (p < A_ROUNDS && !abl_first) continue;
#endif
      swrite_piece(buf, R, p);
    }
#if IM2IM_WGRAD_ABL & 16
    abl_first = false;
#endif
  };
  auto compute = [&](int buf, auto ks_lo_tag, auto ks_hi_tag) __attribute__((always_inline)) {
    constexpr int KS_LO = decltype(ks_lo_tag)::value, KS_HI = decltype(ks_hi_tag)::value;   // k-steps [KS_LO, KS_HI) of the tile
    const char* la = smem + buf * BUF_BYTES;
    const char* lb = la + A_BYTES + tg * HWD * PB;           // this wave's kernel row
    // k-step ks covers tile row ks (TW == 16): every address below is lane base + compile-time constant, so the fully
    // unrolled loop has no address arithmetic (it was ~5 VALU per MFMA when only partially unrolled)
    static_assert(TW == 16, "k-step == one 16-pixel tile row");
    // swizzled tiles: a lane's rows are (multiple of 4) + tr_row [+ 4], so bit 1 of the row index is bit 1 of tr_row for the dz
    // rows, and bit 1 of (c + tr_row) for halo pixel c + tr_row, c = (tg + ks) * 18 + kw (+ 8 * half): four per-lane variants
    const char* pa = la + (half * 8 + tr_row) * PA + (SWZ ? ((wco ^ ((tr_row >> 1) & 1)) << 6) : wco * (COT / 2) * 2) + tr_col_b;
    const char* pb = lb + (half * 8 + tr_row) * PB + (SWZ ? 0 : wci * 64 + tr_col_b);
    int xo[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) xo[j] = SWZ ? (((wci ^ (((((j + 2 * tg) & 3) + tr_row) >> 1) & 1)) << 6) + tr_col_b) : 0;
#pragma unroll
    for (int ks = KS_LO; ks < KS_HI; ++ks) {
      short8 fa[CJ];
#pragma unroll
      for (int j = 0; j < CJ; ++j) fa[j] = WFrag<bf16_t>::load(pa + j * 64 + ks * 16 * PA, pa + j * 64 + (ks * 16 + 4) * PA);
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int xj = xo[(2 * ks + kw) & 3];            // compile-time index after unrolling; 0 unswizzled
        const short8 fb = WFrag<bf16_t>::load(pb + xj + (ks * HWD + kw) * PB, pb + xj + (ks * HWD + kw + 4) * PB);
#pragma unroll
        for (int j = 0; j < CJ; ++j) acc[j][kw] = WFrag<bf16_t>::mfma(fa[j], fb, acc[j][kw]);
      }
    }
  };

  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(t_begin + a.tiles_per_split, a.ntiles);
  if (t_begin < t_end) {
    Stage R;
    gload(t_begin, R);
    swrite(0, R);
    __syncthreads();
    int cur = 0;
    for (int t = t_begin; t < t_end; ++t) {
      const bool more = t + 1 < t_end;
#if IM2IM_WGRAD_ABL & 8
      if (more) gload(t_begin, R);                     // same instruction stream, data always cache-hot
#elif !(IM2IM_WGRAD_ABL & 1)
      if (more) gload(t + 1, R);                       // in flight during the MFMAs below
#endif
#if !(IM2IM_WGRAD_ABL & 4)
      compute(cur, std::integral_constant<int, 0>{}, std::integral_constant<int, KSTEPS>{});
#endif
#if !(IM2IM_WGRAD_ABL & 2)
      if (more) swrite(cur ^ 1, R);
#endif
      __syncthreads();
      cur ^= 1;
    }
  }
  float* out = a.partial + (size_t)split * a.Co * 9 * a.Ci;
#pragma unroll
  for (int j = 0; j < CJ; ++j)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * (COT / 2) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ci = ci0 + wci * 32 + l31;
        if (co < a.Co) out[((size_t)co * 9 + tg * 3 + kw) * a.Ci + ci] = acc[j][kw][r];
      }
}

// ------------------------------------------------------------------------------------------------
// [r4b] conv_wgrad_pipe_kernel with the two phases of a tile dissolved into each other (needs Co % COT == 0 and 32-bit byte offsets
// within an image: launch_wgrad sends everything else to conv_wgrad_pipe_kernel; PARTIAL = tiles may hang over the image).  Same workgroup (12 waves = 2 x 2 (co, ci) quadrants
// x 3 kernel rows), same LDS images, same accumulation order -- bit-identical results -- but:
//  * ROLLING OPERAND PREFETCH.  The compiler's order for the MFMA phase is  ds_read_b64_tr_b16 x2 -> s_waitcnt lgkmcnt(0) -> MFMA x2:
//    a wave's LDS latency is hidden only by the two other waves of its SIMD.  Here the fragments of unit u+1 (unit = one
//    (k-step, kw): CJ MFMAs on one x fragment) are requested BEFORE the MFMAs of unit u issue (sched_group_barrier pins "reads,
//    then MFMAs"), so the latency also runs under the wave's own MFMAs.
//  * STAGING SPREAD OVER THE MFMA PHASE.  After k-step slot(p) of tile t this thread's piece p of tile t+1 -- requested at the same
//    point of tile t-1, one whole tile period in flight -- goes to the other LDS buffer and the request for piece p of tile t+2
//    follows at once.  No staging phase of all twelve waves in front of the barrier; after the barrier MFMA work issues at once.
//  * A VALU DIET.  With 48 MFMAs per wave and tile, ~7 VALU instructions per MFMA issue for free in the MFMA's shadow; the
//    pipe kernel spends 216 per tile on addresses, bounds and the lazy transform (the first version of this kernel 285: no gain;
//    without the lazy transform +7 %).  Here: buffer loads (uniform 64-bit tile base in the descriptor, one 32-bit per-thread
//    offset per piece computed ONCE; a halo piece outside the image gets offset 0xffffffff = out of range = the hardware returns
//    zeros: no clamps, no selects, no exec branches), LDS addresses computed once, a tile's edge test = one AND of a per-thread
//    nibble mask with a uniform nibble, the lazy BatchNorm+ReLU on packed fp32 pairs (v_pk_mul_f32, v_pk_add_f32,
//    v_cvt_pk_bf16_f32, ReLU as v_pk_max_i16 on the bf16 bit patterns -- same bits as max(x, 0) before the rounding).
//  * BRANCH-FREE tile iteration = ONE basic block: s_waitcnt vmcnt counts exactly (with exec branches around the loads the
//    compiler waits for vmcnt(0), i.e. also for the requests just issued).  Past the end of a split the last tile is requested
//    again (L2 hits) and written to the buffer nobody reads.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int TH, int TW, int COT, bool LAZY, bool PARTIAL>
__global__ __launch_bounds__(768) void conv_wgrad_roll_kernel(WgradArgs a) {
  using T = bf16_t;
  constexpr int NT = 768;
  constexpr int HH = TH + 2, HWD = TW + 2, HPX = HH * HWD;
  constexpr int M = TH * TW;
  constexpr bool SWZ = TH == 16;                       // unpadded XOR-swizzled 256-pixel tiles (see conv_wgrad_pipe_kernel)
  static_assert(!SWZ || COT == 64, "swizzled 256-pixel tiles: 64 output channels");
  static_assert(TW == 16, "k-step == one 16-pixel tile row");
  constexpr int CT = 64, EPP = 8, PPR = 8, PB = SWZ ? 128 : 192;
  constexpr int CJ = COT / 64;
  constexpr int PA = SWZ ? COT * 2 : COT * 2 + 64;
  constexpr int PPRA = COT / 8;
  constexpr int A_BYTES = M * PA, B_BYTES = HPX * PB, BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int A_PIECES = M * PPRA, B_PIECES = HPX * PPR;
  constexpr int A_ROUNDS = (A_PIECES + NT - 1) / NT, B_ROUNDS = (B_PIECES + NT - 1) / NT, NPIECES = A_ROUNDS + B_ROUNDS;
  static_assert(A_PIECES >= NT && B_PIECES >= NT, "a thread without a piece repeats its piece of the previous round");
  static_assert(B_ROUNDS <= 8, "one nibble of edge bits per halo piece");
  constexpr int A_PXR = NT / PPRA;                     // dz pixels per staging round: whole tile rows, an even number of pixel pairs
  static_assert(A_PXR % TW == 0 && (A_PXR / 2) % 2 == 0, "a dz round = whole tile rows; the swizzle parity repeats per round");
  constexpr int KSTEPS = M / 16;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave >> 2;                          // kernel row kh
  const int wco = (wave >> 1) & 1, wci = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int ci_tiles = a.Ci / CT;
  int cb = blockIdx.x, split = blockIdx.y;
#if IM2IM_WGRAD_XCD
  if ((gridDim.y & 7) == 0) {                        // all channel blocks of one pixel split on ONE XCD (see conv_wgrad_pipe_kernel)
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    split = (j / (int)gridDim.x) * 8 + xcd;
    cb = j % (int)gridDim.x;
  }
#endif
  const int co0 = (cb / ci_tiles) * COT, ci0 = (cb % ci_tiles) * CT;
  const WgradSrc<T> xs(a, ci0);
  const T* __restrict__ xg = xs.x;
  const T* __restrict__ dzg = reinterpret_cast<const T*>(a.dz) + co0;

  f32x16 acc[CJ][3];
#pragma unroll
  for (int j = 0; j < CJ; ++j)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

  const int q16 = lane & 15;
  const int tr_col_b = (((lane >> 4) & 1) * 16 + (q16 & 3) * 4) * 2;
  const int tr_row = q16 >> 2;

  // ---- per-thread staging constants (computed once) ----
  // dz: piece q = round * NT + tid -> pixel q / PPRA (a round = A_PXR / TW whole tile rows further down), 16-byte part q % PPRA.
  // Rounds 0 .. A_ROUNDS-2 share ONE byte offset (the round's rows go into the instruction's scalar offset) and ONE LDS address
  // (+ an immediate); a thread without a piece in the last round repeats its piece of the round before.
  const int a_pix = tid / PPRA, a_part = tid % PPRA;
  const int a_goff = ((a_pix / TW) * a.W + a_pix % TW) * a.Co * 2 + a_part * 16;               // bytes from the tile's first dz element
  const int a_round_b = (A_PXR / TW) * a.W * a.Co * 2;                                         // bytes per round (uniform)
  const int a_loff = a_pix * PA + (SWZ ? (a_part ^ (((a_pix >> 1) & 1) << 2)) : a_part) * 16;
  constexpr bool A_WRAP = A_ROUNDS * NT > A_PIECES;
  const int a_last = (A_WRAP && (A_ROUNDS - 1) * NT + tid >= A_PIECES) ? A_ROUNDS - 2 : A_ROUNDS - 1;   // round this thread stages last
  const int a_goff_last = a_goff + a_last * a_round_b, a_loff_last = a_loff + a_last * A_PXR * PA;
  // PARTIAL (H % TH or W % TW != 0: the 40x40 / 20x20 levels): tiles may hang over the image, so a piece's pixel (row << 16 | column,
  // tile / halo coordinates) is compared with the tile's uniform limits on packed 16-bit halves instead of the edge nibbles
  const int a_yx = ((a_pix / TW) << 16) | (a_pix % TW);
  const int a_yx_last = a_yx + ((a_last * (A_PXR / TW)) << 16);
  // x (halo): piece q -> halo pixel q / 8 (row q / 8 / 18, column q / 8 % 18), part q % 8; byte offset from the halo's first pixel
  // (y0 - 1, x0 - 1), LDS address, and four edge bits (top row, bottom row, left column, right column of the halo)
  int b_goff[B_ROUNDS], b_loff[B_ROUNDS], b_yx[PARTIAL ? B_ROUNDS : 1];
  unsigned b_edge = 0;
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) {
    int q = i * NT + tid;
    if ((i + 1) * NT > B_PIECES && q >= B_PIECES) q -= NT;
    const int pix = q / PPR, part = q % PPR, hy = pix / HWD, hx = pix % HWD;
    b_goff[i] = ((hy * a.W + hx) * xs.stride + part * EPP) * 2;
    b_loff[i] = A_BYTES + pix * PB + (SWZ ? (part ^ (((pix >> 1) & 1) << 2)) : part) * 16;
    b_edge |= (unsigned)((hy == 0) | ((hy == HH - 1) << 1) | ((hx == 0) << 2) | ((hx == HWD - 1) << 3)) << (4 * i);
    if constexpr (PARTIAL) b_yx[i] = (hy << 16) | hx;
  }

  // lazy BatchNorm coefficients of this thread's 8 channels as pairs: LDS behind the tile buffers (COT = 128: no registers
  // left) or registers.  LAZY = some source of the launch has coefficients; with a split input this workgroup's 64 channels may
  // come from the one that has none: scale 1, shift 0 and no ReLU then (wave-uniform mask, no branch)
  constexpr bool SS_LDS = COT > 64;
  f32x2 xsc[SS_LDS ? 1 : 4], xsh[SS_LDS ? 1 : 4];
  float* ldsSS = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);
  const bool lazy_x = LAZY && xs.sc != nullptr;
  const int relu_keep = lazy_x ? 0 : (int)0x80008000;       // v_pk_max_i16(v, relu_floor): floor 0 = ReLU, floor -32768 = identity
  if constexpr (LAZY) {
    if constexpr (SS_LDS) {
      if (tid < CT) { ldsSS[tid] = lazy_x ? xs.sc[tid] : 1.f; ldsSS[CT + tid] = lazy_x ? xs.sh[tid] : 0.f; }
      __syncthreads();
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int c = (tid % PPR) * EPP + 2 * k;
        xsc[k] = f32x2{lazy_x ? xs.sc[c] : 1.f, lazy_x ? xs.sc[c + 1] : 1.f};
        xsh[k] = f32x2{lazy_x ? xs.sh[c] : 0.f, lazy_x ? xs.sh[c + 1] : 0.f};
      }
    }
  }

  // ---- tile walk (wave-uniform) ----
  struct Pos { int tx, ty, b; };
  auto pos_of = [&](int t) __attribute__((always_inline)) -> Pos { Pos p; p.tx = t % a.tilesX; t /= a.tilesX; p.ty = t % a.tilesY; p.b = t / a.tilesY; return p; };
  auto advance = [&](Pos p, bool go) __attribute__((always_inline)) -> Pos {       // the next tile if go, else the same one; no branches
    const int wx = (p.tx + 1 == a.tilesX), wy = wx & (p.ty + 1 == a.tilesY);
    Pos n;
    n.tx = wx ? 0 : p.tx + 1;
    n.ty = wy ? 0 : p.ty + wx;
    n.b = p.b + wy;
    n.tx = go ? n.tx : p.tx; n.ty = go ? n.ty : p.ty; n.b = go ? n.b : p.b;
    return n;
  };
  // PARTIAL: inside <=> lo <= yx < hi on both halves: (yx - hi) negative and (yx - lo) non-negative per half
  auto outside = [&](int yx, int lo, int hi) __attribute__((always_inline)) -> bool {
    const s16x2 d1 = __builtin_bit_cast(s16x2, yx) - __builtin_bit_cast(s16x2, hi), d2 = __builtin_bit_cast(s16x2, yx) - __builtin_bit_cast(s16x2, lo);
    return ((__builtin_bit_cast(int, d1) & ~__builtin_bit_cast(int, d2)) & (int)0x80008000) != (int)0x80008000;
  };
  struct Src { __amdgpu_buffer_rsrc_t dz, x; unsigned bad; int a_hi, b_lo, b_hi; };   // descriptors based at the tile's first dz element / first halo pixel; bad: nibble i != 0 <=> halo piece i lies outside the image
  auto src_of = [&](Pos p) __attribute__((always_inline)) -> Src {
    const int y0 = p.ty * TH, x0 = p.tx * TW;
    const T* dzb = dzg + (((size_t)p.b * a.H + y0) * a.W + x0) * a.Co;
    const T* xb = xg + (((ptrdiff_t)p.b * a.H + y0 - 1) * a.W + x0 - 1) * (ptrdiff_t)xs.stride;     // may lie in front of the tensor: those pieces are "bad"
    const unsigned em = (unsigned)((y0 == 0) | ((y0 + TH == a.H) << 1) | ((x0 == 0) << 2) | ((x0 + TW == a.W) << 3));
    Src s;
    s.dz = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dzb), 0, 0x7fffffff, 0x00020000);
    s.x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb), 0, 0x7fffffff, 0x00020000);
    s.bad = b_edge & (em * 0x11111111u);
    s.a_hi = (min(a.H - y0, 0x7fff) << 16) | min(a.W - x0, 0x7fff);                  // dz pixel (row, column) limits of the tile
    s.b_lo = ((y0 == 0) << 16) | (x0 == 0);                                          // halo pixel limits (halo coordinates = tile + 1)
    s.b_hi = (min(a.H - y0 + 1, 0x7fff) << 16) | min(a.W - x0 + 1, 0x7fff);
    return s;
  };
  struct Stage { i32x4 v[NPIECES]; };                  // one tile in flight
  auto gload_piece = [&](const Src& s, Stage& R, auto p_tag) __attribute__((always_inline)) {
    constexpr int p = decltype(p_tag)::value;
    if constexpr (p < A_ROUNDS - 1) {
      int off = a_goff;
      if constexpr (PARTIAL) off = outside(a_yx + ((p * (A_PXR / TW)) << 16), 0, s.a_hi) ? -1 : off;
      R.v[p] = __builtin_amdgcn_raw_buffer_load_b128(s.dz, off, p * a_round_b, 0);
    } else if constexpr (p == A_ROUNDS - 1) {
      int off = a_goff_last;
      if constexpr (PARTIAL) off = outside(a_yx_last, 0, s.a_hi) ? -1 : off;
      R.v[p] = __builtin_amdgcn_raw_buffer_load_b128(s.dz, off, 0, 0);
    } else {
      constexpr int i = p - A_ROUNDS;
      bool bad;
      if constexpr (PARTIAL) bad = outside(b_yx[i], s.b_lo, s.b_hi); else bad = (s.bad >> (4 * i)) & 15u;
      const int off = bad ? -1 : b_goff[i];                              // 0xffffffff >= num_records: the load returns zeros
      R.v[p] = __builtin_amdgcn_raw_buffer_load_b128(s.x, off, 0, 0);
    }
  };
  struct Edge { unsigned bad; int b_lo, b_hi; };      // what swrite_piece needs of the tile it writes: which halo pieces are padding
  auto swrite_piece = [&](int bufoff, const Edge& e, const Stage& R, auto p_tag) __attribute__((always_inline)) {
    constexpr int p = decltype(p_tag)::value;
    i32x4 v = R.v[p];
    if constexpr (p < A_ROUNDS - 1) {
      *reinterpret_cast<i32x4*>(smem + bufoff + a_loff + p * A_PXR * PA) = v;
    } else if constexpr (p == A_ROUNDS - 1) {
      *reinterpret_cast<i32x4*>(smem + bufoff + a_loff_last) = v;
    } else {
      constexpr int i = p - A_ROUNDS;
      if constexpr (LAZY) {
        f32x2 sc[4], sh[4];
        if constexpr (SS_LDS) {
          const int c0 = (tid % PPR) * EPP;
          const float4 s0 = *reinterpret_cast<const float4*>(ldsSS + c0), s1 = *reinterpret_cast<const float4*>(ldsSS + c0 + 4);
          const float4 h0 = *reinterpret_cast<const float4*>(ldsSS + CT + c0), h1 = *reinterpret_cast<const float4*>(ldsSS + CT + c0 + 4);
          sc[0] = f32x2{s0.x, s0.y}; sc[1] = f32x2{s0.z, s0.w}; sc[2] = f32x2{s1.x, s1.y}; sc[3] = f32x2{s1.z, s1.w};
          sh[0] = f32x2{h0.x, h0.y}; sh[1] = f32x2{h0.z, h0.w}; sh[2] = f32x2{h1.x, h1.y}; sh[3] = f32x2{h1.z, h1.w};
        } else {
#pragma unroll
          for (int k = 0; k < 4; ++k) { sc[k] = xsc[k]; sh[k] = xsh[k]; }
        }
        bool bad;
        if constexpr (PARTIAL) bad = outside(b_yx[i], e.b_lo, e.b_hi); else bad = (e.bad >> (4 * i)) & 15u;
        const int keep = bad ? 0 : -1;                                  // zero padding stays exactly zero (not max(shift, 0))
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const unsigned u = (unsigned)v[k];
          f32x2 f = f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
          f = f * sc[k];
          f = f + sh[k];
          const bf16_t lo = (bf16_t)f[0], hi = (bf16_t)f[1];
          const unsigned r = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
          const s16x2 m = __builtin_elementwise_max(__builtin_bit_cast(s16x2, r), __builtin_bit_cast(s16x2, relu_keep));
          v[k] = __builtin_bit_cast(int, m) & keep;
        }
      }
      *reinterpret_cast<i32x4*>(smem + bufoff + b_loff[i]) = v;
    }
  };

  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(t_begin + a.tiles_per_split, a.ntiles);
  if (t_begin < t_end) {
    Stage R;
    Pos pos = pos_of(t_begin);
    Edge ew;                                            // edges of the tile whose pieces are written next
    {
      const Src s0 = src_of(pos);
      static_for<0, NPIECES>([&](auto p_tag) __attribute__((always_inline)) { gload_piece(s0, R, p_tag); });
      const Edge e0{s0.bad, s0.b_lo, s0.b_hi};
      static_for<0, NPIECES>([&](auto p_tag) __attribute__((always_inline)) { swrite_piece(0, e0, R, p_tag); });
      pos = advance(pos, t_begin + 1 < t_end);
      const Src s1 = src_of(pos);
      static_for<0, NPIECES>([&](auto p_tag) __attribute__((always_inline)) { gload_piece(s1, R, p_tag); });
      ew = Edge{s1.bad, s1.b_lo, s1.b_hi};
    }
    __syncthreads();
    int curoff = 0;
    for (int t = t_begin; t < t_end; ++t) {
      pos = advance(pos, t + 2 < t_end);
      const Src s2 = src_of(pos);                        // tile t+2 (the last tile again at the end of the split)
      const int nxtoff = BUF_BYTES - curoff;
      const char* la = smem + curoff;
      const char* lb = la + A_BYTES + tg * HWD * PB;           // this wave's kernel row
      const char* pa = la + (half * 8 + tr_row) * PA + (SWZ ? ((wco ^ ((tr_row >> 1) & 1)) << 6) : wco * (COT / 2) * 2) + tr_col_b;
      const char* pb = lb + (half * 8 + tr_row) * PB + (SWZ ? 0 : wci * 64 + tr_col_b);
      int xo[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) xo[j] = SWZ ? (((wci ^ (((((j + 2 * tg) & 3) + tr_row) >> 1) & 1)) << 6) + tr_col_b) : 0;
      // operand ring: the fragments of unit u + DIST are requested before the MFMAs of unit u issue
      constexpr int DIST = IM2IM_ROLL_DIST, NU = KSTEPS * 3;
      static_assert(DIST >= 1 && DIST <= 3, "the x ring holds DIST + 1 fragments, the dz ring two k-steps");
      short8 fa[2][CJ], fb[DIST + 1];
      auto req = [&](auto n_tag) __attribute__((always_inline)) {                  // request the fragments unit n needs
        constexpr int n = decltype(n_tag)::value, nks = n / 3, nkw = n % 3;
        if constexpr (n < NU) {
          if constexpr (nkw == 0) {
#pragma unroll
            for (int j = 0; j < CJ; ++j) fa[nks & 1][j] = WFrag<bf16_t>::load(pa + j * 64 + nks * 16 * PA, pa + j * 64 + (nks * 16 + 4) * PA);
          }
          const int xj = xo[(2 * nks + nkw) & 3];
          fb[n % (DIST + 1)] = WFrag<bf16_t>::load(pb + xj + (nks * HWD + nkw) * PB, pb + xj + (nks * HWD + nkw + 4) * PB);
        }
      };
      static_for<0, DIST>([&](auto n_tag) __attribute__((always_inline)) { req(n_tag); });
      static_for<0, NU>([&](auto u_tag) __attribute__((always_inline)) {
        constexpr int u = decltype(u_tag)::value, ks = u / 3, kw = u % 3, n = u + DIST;
        req(std::integral_constant<int, n>{});
#if IM2IM_ROLL_PIN
        if constexpr (n < NU) __builtin_amdgcn_sched_group_barrier(0x100, n % 3 == 0 ? 2 * CJ + 2 : 2, 0);
#endif
#pragma unroll
        for (int j = 0; j < CJ; ++j) acc[j][kw] = WFrag<bf16_t>::mfma(fa[ks & 1][j], fb[u % (DIST + 1)], acc[j][kw]);
#if IM2IM_ROLL_PIN
        __builtin_amdgcn_sched_group_barrier(0x008, CJ, 0);
#endif
        if constexpr (kw == 2) {
          static_for<0, NPIECES>([&](auto p_tag) __attribute__((always_inline)) {
            constexpr int p = decltype(p_tag)::value;
            if constexpr ((IM2IM_ROLL_SLOT == 0 ? (2 * p + 1) * KSTEPS / (2 * NPIECES) : IM2IM_ROLL_SLOT == 1 ? p * KSTEPS / (2 * NPIECES) : KSTEPS - 1 - (NPIECES - 1 - p) * KSTEPS / (2 * NPIECES)) == ks) {
              // SALU, MFMA and ds_read instructions may cross these two fences; ds_write, buffer_load and VALU (the transform of a
              // piece that has not arrived would pull its wait forward) may not: left alone the scheduler collects every
              // request at the end of the tile, a quarter period ahead of its use
              __builtin_amdgcn_sched_barrier(0x10c);
              swrite_piece(nxtoff, ew, R, p_tag);      // tile t+1
              gload_piece(s2, R, p_tag);               // tile t+2
              __builtin_amdgcn_sched_barrier(0x10c);
            }
          });
        }
      });
      ew = Edge{s2.bad, s2.b_lo, s2.b_hi};
      __syncthreads();
      curoff = nxtoff;
    }
  }
  float* out = a.partial + (size_t)split * a.Co * 9 * a.Ci;
#pragma unroll
  for (int j = 0; j < CJ; ++j)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * (COT / 2) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ci = ci0 + wci * 32 + l31;
        out[((size_t)co * 9 + tg * 3 + kw) * a.Ci + ci] = acc[j][kw][r];
      }
}

// ------------------------------------------------------------------------------------------------
// [r4] fp8 weight gradient (BASELINE configs[4] "fp8 MFMA conv path": the last third of the conv FLOPs that was still bf16):
//   dW[co][tap][ci] = sum over pixels of dz[p][co] * a[p + tap][ci]
// on v_mfma_scale_f32_32x32x64_f8f6f4 with K = 64 PIXELS per instruction: dz as OCP e5m2 under the tensor's delayed
// power-of-two scale (the same amax slot its fp8 data-gradient reads, conv_fp8.hip), the layer input a as e4m3 times 2^4 -- exactly
// the operand the fp8 FORWARD staged (lazy BatchNorm+ReLU with the coefficients pre-scaled, clamp, v_cvt_pk_fp8_f32) -- fp32
// accumulation; both scales are undone for free by the instruction's E8M0 block scales, so the partial sums and the split-K
// reduction are the bf16 kernel's.  Same workgroup shape as conv_wgrad_pipe_kernel (12 waves = 2 x 2 (co, ci) quadrants x 3
// kernel rows, 8 x 16-pixel tiles double-buffered in LDS, next tile prefetched into registers under the MFMAs), but the tiles
// are held as BYTES: [pixel][channel] fp8, and the K(= pixel)-contiguous fragments come from ds_read_b64_tr_b8 (8 x 16-byte
// block per 16 lanes: lane q supplies row q>>1, half q&1, receives column l&15 of the 8 rows -- tools/hwprobe/tr8probe.hip,
// profiles/r04_tr8probe.txt).  A lane's 32 fragment bytes = 32 consecutive pixels of ONE channel = two tile rows = 4 reads.
// Half the LDS bytes and half the MFMA cycles of the bf16 kernel per tile; the HBM side (bf16 operands) is unchanged, so the
// gain is on the layers whose operands are shared through L2 (>= 128 channels), not on the full-resolution ones.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) i32x2 lds_i32x2;
constexpr float FP8W_XSCALE = 16.f;          // activation pre-scale 2^4 (conv_fp8.hip XSCALE), undone by scale_b = 127 - 4
constexpr float FP8W_E4M3_MAX = 448.f, FP8W_E5M2_MAX = 57344.f;

struct Fp8WgradArgs {
  WgradArgs w;             // x = layer input (bf16), dz (bf16), partial, geometry, lazy coefficients, split input
  const float* amax_in;    // max |dz| of the previous step (device scalar): the staging scale 2^(14 - e), amax = f * 2^e
};

template <int COT>
__global__ __launch_bounds__(768) void conv_wgrad_fp8_kernel(Fp8WgradArgs fa_) {
  using T = bf16_t;
  const WgradArgs& a = fa_.w;
  constexpr int NT = 768;
  constexpr int TH = 8, TW = 16, HH = TH + 2, HWD = TW + 2, HPX = HH * HWD, M = TH * TW;
  constexpr int CT = 64, CJ = COT / 64;
  // byte pitches: the 8 rows of a transposing read (and the two channel sub-blocks of a half-wave) land on 16 distinct
  // 16-byte bank groups of the 256-byte LDS row with a pad of 32
  constexpr int PA = COT + 32, PB = CT + 32;
  constexpr int PPRA = COT / 8, PPR = CT / 8;          // 16-byte bf16 source pieces (8 channels) per pixel
  constexpr int A_BYTES = M * PA, B_BYTES = HPX * PB, BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int A_ROUNDS = (M * PPRA + NT - 1) / NT, B_ROUNDS = (HPX * PPR + NT - 1) / NT;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave >> 2;                          // kernel row kh
  const int wco = (wave >> 1) & 1, wci = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int ci_tiles = a.Ci / CT;
  int cb = blockIdx.x, split = blockIdx.y;
#if IM2IM_WGRAD_XCD
  if ((gridDim.y & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    split = (j / (int)gridDim.x) * 8 + xcd;
    cb = j % (int)gridDim.x;
  }
#endif
  const int co0 = (cb / ci_tiles) * COT, ci0 = (cb % ci_tiles) * CT;
  const WgradSrc<T> xs(a, ci0);
  const T* __restrict__ xg = xs.x;
  const T* __restrict__ dzg = reinterpret_cast<const T*>(a.dz);

  f32x16 acc[CJ][3];
#pragma unroll
  for (int j = 0; j < CJ; ++j)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

  // delayed scaling of dz: s = 2^(14 - e); the MFMA multiplies the A operand by 2^(scale_a - 127) = 1 / s
  float dzs = 1.f;
  int scale_a = 127;
  {
    const float amax = *fa_.amax_in;
    if (amax > 0.f && amax < 3.0e38f) {
      int e;
      frexpf(amax, &e);
      dzs = ldexpf(1.f, 14 - e);
      scale_a = 127 - (14 - e);
      if (scale_a < 1) { scale_a = 127; dzs = 1.f; }
    }
  }
  constexpr int scale_b = 127 - 4;

  int a_px[A_ROUNDS], a_part[A_ROUNDS], b_px[B_ROUNDS], b_part[B_ROUNDS];
#pragma unroll
  for (int i = 0; i < A_ROUNDS; ++i) { const int p = i * NT + tid; a_px[i] = (p < M * PPRA) ? p / PPRA : -1; a_part[i] = p % PPRA; }
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) { const int p = i * NT + tid; b_px[i] = (p < HPX * PPR) ? p / PPR : -1; b_part[i] = p % PPR; }
  struct Stage { uint4 a[A_ROUNDS], b[B_ROUNDS]; unsigned valid; };
  // lazy coefficients of the 64 input channels times 2^4, in LDS behind the tile buffers (16 * max(z*s + h, 0) == max(z*16s + 16h, 0))
  float* ldsSS = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);
  const bool lazy_x = xs.sc != nullptr;
  if (lazy_x) {
    if (tid < CT) { ldsSS[tid] = xs.sc[tid] * FP8W_XSCALE; ldsSS[CT + tid] = xs.sh[tid] * FP8W_XSCALE; }
    __syncthreads();
  }

  auto gload = [&](int t, Stage& R) __attribute__((always_inline)) {
    int tt = t;
    const int tx_id = tt % a.tilesX; tt /= a.tilesX;
    const int ty_id = tt % a.tilesY;
    const int b = tt / a.tilesY;
    const int y0 = ty_id * TH, x0 = tx_id * TW;
    const T* xb = xg + (size_t)b * a.H * a.W * xs.stride;
    const T* dzb = dzg + (size_t)b * a.H * a.W * a.Co + co0;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a_px[i] >= 0) {
        const int yy = y0 + a_px[i] / TW, xx = x0 + a_px[i] % TW;
        if (yy < a.H && xx < a.W) v = *reinterpret_cast<const uint4*>(dzb + ((size_t)yy * a.W + xx) * a.Co + a_part[i] * 8);
      }
      R.a[i] = v;
    }
    R.valid = 0;
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (b_px[i] >= 0) {
        const int yy = y0 + b_px[i] / HWD - 1, xx = x0 + b_px[i] % HWD - 1;
        if (yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
          v = *reinterpret_cast<const uint4*>(xb + ((size_t)yy * a.W + xx) * xs.stride + b_part[i] * 8);
          R.valid |= 1u << i;
        }
      }
      R.b[i] = v;
    }
  };
  auto to8 = [&](const float (&v)[8], bool e5m2) __attribute__((always_inline)) -> uint2 {
    float c[8];
    const float lim = e5m2 ? FP8W_E5M2_MAX : FP8W_E4M3_MAX;
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = __builtin_amdgcn_fmed3f(v[k], -lim, lim);
    int lo, hi;
    if (e5m2) {
      lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[0], c[1], 0, false); lo = __builtin_amdgcn_cvt_pk_bf8_f32(c[2], c[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_bf8_f32(c[4], c[5], 0, false); hi = __builtin_amdgcn_cvt_pk_bf8_f32(c[6], c[7], hi, true);
    } else {
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], 0, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], 0, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
    }
    return make_uint2((unsigned)lo, (unsigned)hi);
  };
  auto swrite = [&](int buf, const Stage& R) __attribute__((always_inline)) {
    char* la = smem + buf * BUF_BYTES;
    char* lb = la + A_BYTES;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i)
      if (a_px[i] >= 0) {
        float v[8];
        Vec16<T>::load(reinterpret_cast<const T*>(&R.a[i]), v);
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] *= dzs;
        *reinterpret_cast<uint2*>(la + a_px[i] * PA + a_part[i] * 8) = to8(v, true);
      }
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i) {
      if (b_px[i] >= 0) {
        uint2 q = make_uint2(0u, 0u);
        if ((R.valid >> i) & 1) {                         // zero padding stays exactly zero
          float v[8];
          Vec16<T>::load(reinterpret_cast<const T*>(&R.b[i]), v);
          if (lazy_x) {
            const int c0 = (tid % PPR) * 8;
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = fmaxf(v[k] * ldsSS[c0 + k] + ldsSS[CT + c0 + k], 0.f);
          } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] *= FP8W_XSCALE;
          }
          q = to8(v, false);
        }
        *reinterpret_cast<uint2*>(lb + b_px[i] * PB + b_part[i] * 8) = q;
      }
    }
  };
  // fragment addressing: 16-lane group g = lane >> 4 reads channel sub-block (g & 1) * 16 of the wave's 32, pixels (g >> 1) * 32 ...
  // of the k-step (= MFMA lane half); lane q of the group supplies pixel row q >> 1 and the 8-byte half q & 1 of the block
  const int g = lane >> 4, q = lane & 15;
  auto tr8 = [&](const char* p) __attribute__((always_inline)) -> i32x2 {
    return __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(lds_char*)p);
  };
  auto compute = [&](int buf) __attribute__((always_inline)) {
    const char* la = smem + buf * BUF_BYTES;
    const char* lb = la + A_BYTES + tg * HWD * PB;
    const char* pa = la + ((g >> 1) * 32 + (q >> 1)) * PA + wco * (COT / 2) + (g & 1) * 16 + (q & 1) * 8;
    const char* pb = lb + ((g >> 1) * 2 * HWD + (q >> 1)) * PB + wci * 32 + (g & 1) * 16 + (q & 1) * 8;
#pragma unroll 1                                          // (unrolled, the scheduler hoists both k-steps' 40 fragment reads: 94 spilled registers)
    for (int ks = 0; ks < M / 64; ++ks) {
      i32x8 fa[CJ];
#pragma unroll
      for (int j = 0; j < CJ; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const i32x2 v = tr8(pa + j * 32 + (ks * 64 + 8 * i) * PA);
          fa[j][2 * i] = v[0]; fa[j][2 * i + 1] = v[1];
        }
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        i32x8 fb;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const i32x2 v = tr8(pb + ((ks * 4 + (i >> 1)) * HWD + (i & 1) * 8 + kw) * PB);
          fb[2 * i] = v[0]; fb[2 * i + 1] = v[1];
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j)
          acc[j][kw] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[j], fb, acc[j][kw], 1, 0, 0, scale_a, 0, scale_b);
      }
    }
  };

  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(t_begin + a.tiles_per_split, a.ntiles);
  if (t_begin < t_end) {
    Stage R;
    gload(t_begin, R);
    swrite(0, R);
    __syncthreads();
    int cur = 0;
    for (int t = t_begin; t < t_end; ++t) {
      const bool more = t + 1 < t_end;
      if (more) gload(t + 1, R);
      compute(cur);
      if (more) swrite(cur ^ 1, R);
      __syncthreads();
      cur ^= 1;
    }
  }
  float* out = a.partial + (size_t)split * a.Co * 9 * a.Ci;
#pragma unroll
  for (int j = 0; j < CJ; ++j)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * (COT / 2) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ci = ci0 + wci * 32 + l31;
        if (co < a.Co) out[((size_t)co * 9 + tg * 3 + kw) * a.Ci + ci] = acc[j][kw][r];
      }
}

// [r4b] conv_wgrad_fp8_kernel rebuilt the way conv_wgrad_roll_kernel rebuilt the bf16 kernel (same LDS byte images, fragments and
// accumulation order as conv_wgrad_fp8_kernel: bit-identical partial sums): a tile iteration is ONE basic block; the next unit's x
// fragment (4 transposing reads) is requested before a unit's MFMAs issue; this thread's piece p of tile t+1 is converted and written
// after unit p of tile t and the request for its piece p of tile t+2 follows (a whole tile period in flight); buffer loads with
// offsets computed once, out-of-image halo pieces through offset 0xffffffff = zero fill; the conversions on packed fp32 pairs
// (v_pk_mul_f32 / v_pk_add_f32, clamp + ReLU in one v_med3_f32, v_cvt_pk_{bf8,fp8}_f32).  The conversion of bf16 operands to fp8 is
// what this kernel's time goes to: ~24 VALU instructions per dz piece and ~30 per x piece against 12 MFMAs of 64 cycles per tile.
template <int COT, bool LAZY, bool PARTIAL>
__global__ __launch_bounds__(768) void conv_wgrad_fp8_roll_kernel(Fp8WgradArgs fa_) {
  using T = bf16_t;
  const WgradArgs& a = fa_.w;
  constexpr int NT = 768;
  constexpr int TH = 8, TW = 16, HH = TH + 2, HWD = TW + 2, HPX = HH * HWD, M = TH * TW;
  constexpr int CT = 64, CJ = COT / 64;
  constexpr int PA = COT + 32, PB = CT + 32;           // byte pitches of the fp8 tiles (see conv_wgrad_fp8_kernel)
  constexpr int PPRA = COT / 8, PPR = CT / 8;          // 16-byte bf16 source pieces (8 channels) per pixel
  constexpr int A_BYTES = M * PA, B_BYTES = HPX * PB, BUF_BYTES = A_BYTES + B_BYTES;
  constexpr int A_PIECES = M * PPRA, B_PIECES = HPX * PPR;
  constexpr int A_ROUNDS = (A_PIECES + NT - 1) / NT, B_ROUNDS = (B_PIECES + NT - 1) / NT, NPIECES = A_ROUNDS + B_ROUNDS;
  static_assert(A_PIECES >= NT && B_PIECES >= NT, "a thread without a piece repeats its piece of the previous round");
  constexpr int A_PXR = NT / PPRA;                     // dz pixels per staging round: whole tile rows
  static_assert(A_PXR % TW == 0, "a dz round = whole tile rows");
  constexpr int KSTEPS = M / 64, NU = KSTEPS * 3;      // k-step = 64 pixels = four tile rows; unit = one (k-step, kw)
  static_assert(NPIECES <= NU, "one staging piece per unit");

  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tg = wave >> 2;                          // kernel row kh
  const int wco = (wave >> 1) & 1, wci = wave & 1;
  const int half = lane >> 5, l31 = lane & 31;
  const int ci_tiles = a.Ci / CT;
  int cb = blockIdx.x, split = blockIdx.y;
#if IM2IM_WGRAD_XCD
  if ((gridDim.y & 7) == 0) {
    const int lin = blockIdx.x + gridDim.x * blockIdx.y;
    const int xcd = lin & 7, j = lin >> 3;
    split = (j / (int)gridDim.x) * 8 + xcd;
    cb = j % (int)gridDim.x;
  }
#endif
  const int co0 = (cb / ci_tiles) * COT, ci0 = (cb % ci_tiles) * CT;
  const WgradSrc<T> xs(a, ci0);
  const T* __restrict__ xg = xs.x;
  const T* __restrict__ dzg = reinterpret_cast<const T*>(a.dz) + co0;

  f32x16 acc[CJ][3];
#pragma unroll
  for (int j = 0; j < CJ; ++j)
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][t][r] = 0.f;

  // delayed scaling of dz: s = 2^(14 - e); the MFMA multiplies the A operand by 2^(scale_a - 127) = 1 / s
  float dzs = 1.f;
  int scale_a = 127;
  {
    const float amax = *fa_.amax_in;
    if (amax > 0.f && amax < 3.0e38f) {
      int e;
      frexpf(amax, &e);
      dzs = ldexpf(1.f, 14 - e);
      scale_a = 127 - (14 - e);
      if (scale_a < 1) { scale_a = 127; dzs = 1.f; }
    }
  }
  constexpr int scale_b = 127 - 4;

  // ---- per-thread staging constants (see conv_wgrad_roll_kernel) ----
  const int a_pix = tid / PPRA, a_part = tid % PPRA;
  const int a_goff = ((a_pix / TW) * a.W + a_pix % TW) * a.Co * 2 + a_part * 16;
  const int a_round_b = (A_PXR / TW) * a.W * a.Co * 2;
  const int a_loff = a_pix * PA + a_part * 8;
  constexpr bool A_WRAP = A_ROUNDS * NT > A_PIECES;
  // (the last round's offsets are not kept in registers: a_goff + (wrapped ? A_ROUNDS - 2 : A_ROUNDS - 1) rounds, from an opaque
  // thread id so that the compiler does not hoist them out of the tile loop and spill them)
  auto last_round = [&]() __attribute__((always_inline)) -> int { int t = tid; asm("" : "+v"(t)); return (A_WRAP && (A_ROUNDS - 1) * NT + t >= A_PIECES) ? A_ROUNDS - 2 : A_ROUNDS - 1; };
  const int a_yx = ((a_pix / TW) << 16) | (a_pix % TW);
  int b_goff[B_ROUNDS], b_loff[B_ROUNDS], b_yx[PARTIAL ? B_ROUNDS : 1];
  unsigned b_edge = 0;
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) {
    int q = i * NT + tid;
    if ((i + 1) * NT > B_PIECES && q >= B_PIECES) q -= NT;
    const int pix = q / PPR, part = q % PPR, hy = pix / HWD, hx = pix % HWD;
    b_goff[i] = ((hy * a.W + hx) * xs.stride + part * 8) * 2;
    b_loff[i] = A_BYTES + pix * PB + part * 8;
    b_edge |= (unsigned)((hy == 0) | ((hy == HH - 1) << 1) | ((hx == 0) << 2) | ((hx == HWD - 1) << 3)) << (4 * i);
    if constexpr (PARTIAL) b_yx[i] = (hy << 16) | hx;
  }

  // coefficients of the x operand's conversion, times the activation pre-scale 2^4, in LDS behind the tile buffers: lazy BatchNorm
  // (scale, shift) or (16, 0); lower clamp 0 (= the ReLU) or -448
  float* ldsSS = reinterpret_cast<float*>(smem + 2 * BUF_BYTES);
  const bool lazy_x = LAZY && xs.sc != nullptr;
  const float x_lo = lazy_x ? 0.f : -FP8W_E4M3_MAX;
  if constexpr (LAZY) {
    if (tid < CT) { ldsSS[tid] = (lazy_x ? xs.sc[tid] : 1.f) * FP8W_XSCALE; ldsSS[CT + tid] = lazy_x ? xs.sh[tid] * FP8W_XSCALE : 0.f; }
    __syncthreads();
  }

  struct Pos { int tx, ty, b; };
  auto pos_of = [&](int t) __attribute__((always_inline)) -> Pos { Pos p; p.tx = t % a.tilesX; t /= a.tilesX; p.ty = t % a.tilesY; p.b = t / a.tilesY; return p; };
  auto advance = [&](Pos p, bool go) __attribute__((always_inline)) -> Pos {
    const int wx = (p.tx + 1 == a.tilesX), wy = wx & (p.ty + 1 == a.tilesY);
    Pos n;
    n.tx = wx ? 0 : p.tx + 1;
    n.ty = wy ? 0 : p.ty + wx;
    n.b = p.b + wy;
    n.tx = go ? n.tx : p.tx; n.ty = go ? n.ty : p.ty; n.b = go ? n.b : p.b;
    return n;
  };
  auto outside = [&](int yx, int lo, int hi) __attribute__((always_inline)) -> bool {
    const s16x2 d1 = __builtin_bit_cast(s16x2, yx) - __builtin_bit_cast(s16x2, hi), d2 = __builtin_bit_cast(s16x2, yx) - __builtin_bit_cast(s16x2, lo);
    return ((__builtin_bit_cast(int, d1) & ~__builtin_bit_cast(int, d2)) & (int)0x80008000) != (int)0x80008000;
  };
  struct Src { __amdgpu_buffer_rsrc_t dz, x; unsigned bad; int a_hi, b_lo, b_hi; };
  auto src_of = [&](Pos p) __attribute__((always_inline)) -> Src {
    const int y0 = p.ty * TH, x0 = p.tx * TW;
    const T* dzb = dzg + (((size_t)p.b * a.H + y0) * a.W + x0) * a.Co;
    const T* xb = xg + (((ptrdiff_t)p.b * a.H + y0 - 1) * a.W + x0 - 1) * (ptrdiff_t)xs.stride;
    const unsigned em = (unsigned)((y0 == 0) | ((y0 + TH == a.H) << 1) | ((x0 == 0) << 2) | ((x0 + TW == a.W) << 3));
    Src s;
    s.dz = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dzb), 0, 0x7fffffff, 0x00020000);
    s.x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb), 0, 0x7fffffff, 0x00020000);
    s.bad = b_edge & (em * 0x11111111u);
    s.a_hi = (min(a.H - y0, 0x7fff) << 16) | min(a.W - x0, 0x7fff);
    s.b_lo = ((y0 == 0) << 16) | (x0 == 0);
    s.b_hi = (min(a.H - y0 + 1, 0x7fff) << 16) | min(a.W - x0 + 1, 0x7fff);
    return s;
  };
  struct Stage { i32x4 v[NPIECES]; };
  auto gload_piece = [&](const Src& s, Stage& R, auto p_tag) __attribute__((always_inline)) {
    constexpr int p = decltype(p_tag)::value;
    if constexpr (p < A_ROUNDS - 1) {
      int off = a_goff;
      if constexpr (PARTIAL) off = outside(a_yx + ((p * (A_PXR / TW)) << 16), 0, s.a_hi) ? -1 : off;
      R.v[p] = __builtin_amdgcn_raw_buffer_load_b128(s.dz, off, p * a_round_b, 0);
    } else if constexpr (p == A_ROUNDS - 1) {
      const int lr = last_round();
      int off = a_goff + lr * a_round_b;
      if constexpr (PARTIAL) off = outside(a_yx + ((lr * (A_PXR / TW)) << 16), 0, s.a_hi) ? -1 : off;
      R.v[p] = __builtin_amdgcn_raw_buffer_load_b128(s.dz, off, 0, 0);
    } else {
      constexpr int i = p - A_ROUNDS;
      bool bad;
      if constexpr (PARTIAL) bad = outside(b_yx[i], s.b_lo, s.b_hi); else bad = (s.bad >> (4 * i)) & 15u;
      R.v[p] = __builtin_amdgcn_raw_buffer_load_b128(s.x, bad ? -1 : b_goff[i], 0, 0);
    }
  };
  struct Edge { unsigned bad; int b_lo, b_hi; };
  auto swrite_piece = [&](int bufoff, const Edge& e, const Stage& R, auto p_tag) __attribute__((always_inline)) {
    constexpr int p = decltype(p_tag)::value;
    const i32x4 v = R.v[p];
    f32x2 f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { const unsigned u = (unsigned)v[k]; f[k] = f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)}; }
    int lo, hi;
    if constexpr (p < A_ROUNDS) {                      // dz: e5m2 under the delayed scale (a zero-filled piece stays zero)
      const f32x2 sc = f32x2{dzs, dzs};
#pragma unroll
      for (int k = 0; k < 4; ++k) { f[k] = f[k] * sc; f[k][0] = __builtin_amdgcn_fmed3f(f[k][0], -FP8W_E5M2_MAX, FP8W_E5M2_MAX); f[k][1] = __builtin_amdgcn_fmed3f(f[k][1], -FP8W_E5M2_MAX, FP8W_E5M2_MAX); }
      lo = __builtin_amdgcn_cvt_pk_bf8_f32(f[0][0], f[0][1], 0, false); lo = __builtin_amdgcn_cvt_pk_bf8_f32(f[1][0], f[1][1], lo, true);
      hi = __builtin_amdgcn_cvt_pk_bf8_f32(f[2][0], f[2][1], 0, false); hi = __builtin_amdgcn_cvt_pk_bf8_f32(f[3][0], f[3][1], hi, true);
      const int dst = a_loff + (p < A_ROUNDS - 1 ? p : last_round()) * A_PXR * PA;
      *reinterpret_cast<uint2*>(smem + bufoff + dst) = make_uint2((unsigned)lo, (unsigned)hi);
    } else {                                           // x: e4m3 of 16 * (lazy BatchNorm+ReLU of z | x)
      constexpr int i = p - A_ROUNDS;
      if constexpr (LAZY) {
        const int c0 = (tid % PPR) * 8;
        {                                              // (in two halves: all sixteen coefficients at once are sixteen registers the wide form lacks)
          const float4 s0 = *reinterpret_cast<const float4*>(ldsSS + c0), h0 = *reinterpret_cast<const float4*>(ldsSS + CT + c0);
          f[0] = f[0] * f32x2{s0.x, s0.y}; f[1] = f[1] * f32x2{s0.z, s0.w};
          f[0] = f[0] + f32x2{h0.x, h0.y}; f[1] = f[1] + f32x2{h0.z, h0.w};
        }
        if constexpr (COT > 64) __builtin_amdgcn_sched_barrier(0);
        {
          const float4 s1 = *reinterpret_cast<const float4*>(ldsSS + c0 + 4), h1 = *reinterpret_cast<const float4*>(ldsSS + CT + c0 + 4);
          f[2] = f[2] * f32x2{s1.x, s1.y}; f[3] = f[3] * f32x2{s1.z, s1.w};
          f[2] = f[2] + f32x2{h1.x, h1.y}; f[3] = f[3] + f32x2{h1.z, h1.w};
        }
      } else {
        const f32x2 sc = f32x2{FP8W_XSCALE, FP8W_XSCALE};
#pragma unroll
        for (int k = 0; k < 4; ++k) f[k] = f[k] * sc;
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { f[k][0] = __builtin_amdgcn_fmed3f(f[k][0], x_lo, FP8W_E4M3_MAX); f[k][1] = __builtin_amdgcn_fmed3f(f[k][1], x_lo, FP8W_E4M3_MAX); }
      lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[0][0], f[0][1], 0, false); lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[1][0], f[1][1], lo, true);
      hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[2][0], f[2][1], 0, false); hi = __builtin_amdgcn_cvt_pk_fp8_f32(f[3][0], f[3][1], hi, true);
      if constexpr (LAZY) {                            // zero padding stays exactly zero (not max(shift, 0))
        bool bad;
        if constexpr (PARTIAL) bad = outside(b_yx[i], e.b_lo, e.b_hi); else bad = (e.bad >> (4 * i)) & 15u;
        const int keep = bad ? 0 : -1;
        lo &= keep; hi &= keep;
      }
      *reinterpret_cast<uint2*>(smem + bufoff + b_loff[i]) = make_uint2((unsigned)lo, (unsigned)hi);
    }
  };

  // fragment addressing (conv_wgrad_fp8_kernel): 16-lane group g reads channel sub-block (g & 1) * 16 of the wave's 32, pixels
  // (g >> 1) * 32 ... of the k-step; lane q of the group supplies pixel row q >> 1 and the 8-byte half q & 1 of the block
  const int g = lane >> 4, q = lane & 15;
  auto tr8 = [&](const char* ptr) __attribute__((always_inline)) -> i32x2 { return __builtin_amdgcn_ds_read_tr8_b64_v2i32((lds_i32x2*)(lds_char*)ptr); };

  const int t_begin = split * a.tiles_per_split;
  const int t_end = min(t_begin + a.tiles_per_split, a.ntiles);
  if (t_begin < t_end) {
    Stage R;
    Pos pos = pos_of(t_begin);
    Edge ew;
    {
      const Src s0 = src_of(pos);
      static_for<0, NPIECES>([&](auto p_tag) __attribute__((always_inline)) { gload_piece(s0, R, p_tag); });
      const Edge e0{s0.bad, s0.b_lo, s0.b_hi};
      static_for<0, NPIECES>([&](auto p_tag) __attribute__((always_inline)) { swrite_piece(0, e0, R, p_tag); });
      pos = advance(pos, t_begin + 1 < t_end);
      const Src s1 = src_of(pos);
      static_for<0, NPIECES>([&](auto p_tag) __attribute__((always_inline)) { gload_piece(s1, R, p_tag); });
      ew = Edge{s1.bad, s1.b_lo, s1.b_hi};
    }
    __syncthreads();
    int curoff = 0;
    for (int t = t_begin; t < t_end; ++t) {
      pos = advance(pos, t + 2 < t_end);
      const Src s2 = src_of(pos);
      const int nxtoff = BUF_BYTES - curoff;
      const char* la = smem + curoff;
      const char* lb = la + A_BYTES + tg * HWD * PB;
      const char* pa = la + ((g >> 1) * 32 + (q >> 1)) * PA + wco * (COT / 2) + (g & 1) * 16 + (q & 1) * 8;
      const char* pb = lb + ((g >> 1) * 2 * HWD + (q >> 1)) * PB + wci * 32 + (g & 1) * 16 + (q & 1) * 8;
      // x fragments: two sets, the next unit's requested before this unit's MFMAs issue -- or ONE (COT = 128: 96 accumulators leave no
      // registers for the second), refilled right after the unit's MFMAs were issued, like the dz fragments
      constexpr int FB_SETS = COT > 64 ? 1 : 2;
      i32x8 fa[CJ], fb[FB_SETS];
      auto load_fa = [&](auto ks_tag) __attribute__((always_inline)) {
        constexpr int ks = decltype(ks_tag)::value;
#pragma unroll
        for (int j = 0; j < CJ; ++j)
#pragma unroll
          for (int i = 0; i < 4; ++i) { const i32x2 w = tr8(pa + j * 32 + (ks * 64 + 8 * i) * PA); fa[j][2 * i] = w[0]; fa[j][2 * i + 1] = w[1]; }
      };
      auto load_fb = [&](auto n_tag) __attribute__((always_inline)) {
        constexpr int n = decltype(n_tag)::value, ks = n / 3, kw = n % 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) { const i32x2 w = tr8(pb + ((ks * 4 + (i >> 1)) * HWD + (i & 1) * 8 + kw) * PB); fb[n % FB_SETS][2 * i] = w[0]; fb[n % FB_SETS][2 * i + 1] = w[1]; }
      };
      load_fa(std::integral_constant<int, 0>{});
      load_fb(std::integral_constant<int, 0>{});
      static_for<0, NU>([&](auto u_tag) __attribute__((always_inline)) {
        constexpr int u = decltype(u_tag)::value, ks = u / 3, kw = u % 3;
        if constexpr (FB_SETS == 2 && u + 1 < NU) {
          load_fb(std::integral_constant<int, u + 1>{});
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        }
#pragma unroll
        for (int j = 0; j < CJ; ++j)
          acc[j][kw] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(fa[j], fb[u % FB_SETS], acc[j][kw], 1, 0, 0, scale_a, 0, scale_b);
        __builtin_amdgcn_sched_group_barrier(0x008, CJ, 0);
        if constexpr (FB_SETS == 1 && u + 1 < NU) load_fb(std::integral_constant<int, u + 1>{});
        if constexpr (kw == 2 && ks + 1 < KSTEPS) load_fa(std::integral_constant<int, ks + 1>{});     // one dz fragment set: refilled after its last MFMAs
        if constexpr (u < NPIECES) {
          // (see conv_wgrad_roll_kernel; with 96 accumulators the coefficient ds_reads must not be hoisted out of their piece either)
          __builtin_amdgcn_sched_barrier(COT > 64 ? 0x00c : 0x10c);
          swrite_piece(nxtoff, ew, R, u_tag);          // piece u of tile t+1
          gload_piece(s2, R, u_tag);                   // piece u of tile t+2
          __builtin_amdgcn_sched_barrier(COT > 64 ? 0x00c : 0x10c);
        }
      });
      ew = Edge{s2.bad, s2.b_lo, s2.b_hi};
      __syncthreads();
      curoff = nxtoff;
    }
  }
  float* out = a.partial + (size_t)split * a.Co * 9 * a.Ci;
#pragma unroll
  for (int j = 0; j < CJ; ++j)
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = co0 + wco * (COT / 2) + j * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int ci = ci0 + wci * 32 + l31;
        out[((size_t)co * 9 + tg * 3 + kw) * a.Ci + ci] = acc[j][kw][r];
      }
}

// sum partial[nsplit][Co][TAPS][Ci] over splits and write torch layout dw[Co][Ci][TAPS].  Block = 64 outputs x 4 split
// lanes (lane s adds splits s, s+4, ... in order, the four partial sums are combined in a fixed order): deterministic,
// and the many-split / few-output case (the 1x1 OutConv) does not serialise on one thread per output.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int nsplit, int Co, int Ci,
                                                            int taps, float* __restrict__ dw) {
  // thread (ol, sl): FOUR consecutive outputs (one 16-byte load per split slab), the slabs sl, sl+4, ... in ascending order;
  // the four slab groups are then added as (0+1)+(2+3) -- the order of the one-float-per-thread version it replaces, so the
  // bits are the same; up to four slabs' loads are in flight per thread.
  __shared__ float4 s_acc[4][64];
  const size_t total = (size_t)Co * taps * Ci;            // a multiple of 4 (Ci % 32 == 0)
  const size_t total4 = total / 4;
  const float4* __restrict__ p4 = reinterpret_cast<const float4*>(partial);
  const int ol = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (size_t base = (size_t)blockIdx.x * 64; base < total4; base += (size_t)gridDim.x * 64) {
    const size_t i = base + ol;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total4) {
      int k = sl;
      for (; k + 12 < nsplit; k += 16) {
        const float4 a = p4[(size_t)k * total4 + i], b = p4[(size_t)(k + 4) * total4 + i];
        const float4 c = p4[(size_t)(k + 8) * total4 + i], d = p4[(size_t)(k + 12) * total4 + i];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
        s.x += b.x; s.y += b.y; s.z += b.z; s.w += b.w;
        s.x += c.x; s.y += c.y; s.z += c.z; s.w += c.w;
        s.x += d.x; s.y += d.y; s.z += d.z; s.w += d.w;
      }
      for (; k < nsplit; k += 4) {
        const float4 a = p4[(size_t)k * total4 + i];
        s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
      }
    }
    s_acc[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && i < total4) {
      const float4 a = s_acc[0][ol], b = s_acc[1][ol], c = s_acc[2][ol], d = s_acc[3][ol];
      const float v[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)};
      const size_t e = i * 4;                               // four consecutive ci of one (co, tap): Ci % 4 == 0
      const int ci = (int)(e % Ci);
      const size_t r = e / Ci;
      const int tp = (int)(r % taps);
      const size_t co = r / taps;
#pragma unroll
      for (int j = 0; j < 4; ++j) dw[(co * Ci + ci + j) * taps + tp] = v[j];
    }
    __syncthreads();
  }
}

// [r6] the same reduction for many layers in one launch: block b belongs to the tensor whose block range holds it and runs
// wgrad_reduce_kernel's loop with that tensor's own block count -- per output the same slabs in the same order, the same bits.
// 18 launches of 8-50 us per training step become one (the slabs stay in per-layer workspaces until the backward pass ends).
constexpr int REDUCE_MULTI_MAX = 32;
struct ReduceMultiArgs {
  const float* partial[REDUCE_MULTI_MAX];
  float* dw[REDUCE_MULTI_MAX];
  int nsplit[REDUCE_MULTI_MAX], Co[REDUCE_MULTI_MAX], Ci[REDUCE_MULTI_MAX], taps[REDUCE_MULTI_MAX];
  int first_block[REDUCE_MULTI_MAX + 1];                   // tensor t owns blocks [first_block[t], first_block[t+1])
  int n;
};
__global__ __launch_bounds__(256) void wgrad_reduce_multi_kernel(ReduceMultiArgs a) {
  __shared__ float4 s_acc[4][64];
  int t = 0;
  while (t + 1 < a.n && (int)blockIdx.x >= a.first_block[t + 1]) ++t;
  const int blk = (int)blockIdx.x - a.first_block[t], nblk = a.first_block[t + 1] - a.first_block[t];
  const int nsplit = a.nsplit[t], Ci = a.Ci[t], taps = a.taps[t];
  float* __restrict__ dw = a.dw[t];
  const size_t total4 = (size_t)a.Co[t] * taps * Ci / 4;
  const float4* __restrict__ p4 = reinterpret_cast<const float4*>(a.partial[t]);
  const int ol = threadIdx.x & 63, sl = threadIdx.x >> 6;
  for (size_t base = (size_t)blk * 64; base < total4; base += (size_t)nblk * 64) {
    const size_t i = base + ol;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (i < total4) {
      int k = sl;
      for (; k + 12 < nsplit; k += 16) {
        const float4 q0 = p4[(size_t)k * total4 + i], q1 = p4[(size_t)(k + 4) * total4 + i];
        const float4 q2 = p4[(size_t)(k + 8) * total4 + i], q3 = p4[(size_t)(k + 12) * total4 + i];
        s.x += q0.x; s.y += q0.y; s.z += q0.z; s.w += q0.w;
        s.x += q1.x; s.y += q1.y; s.z += q1.z; s.w += q1.w;
        s.x += q2.x; s.y += q2.y; s.z += q2.z; s.w += q2.w;
        s.x += q3.x; s.y += q3.y; s.z += q3.z; s.w += q3.w;
      }
      for (; k < nsplit; k += 4) {
        const float4 q = p4[(size_t)k * total4 + i];
        s.x += q.x; s.y += q.y; s.z += q.z; s.w += q.w;
      }
    }
    s_acc[sl][ol] = s;
    __syncthreads();
    if (sl == 0 && i < total4) {
      const float4 q0 = s_acc[0][ol], q1 = s_acc[1][ol], q2 = s_acc[2][ol], q3 = s_acc[3][ol];
      const float v[4] = {(q0.x + q1.x) + (q2.x + q3.x), (q0.y + q1.y) + (q2.y + q3.y), (q0.z + q1.z) + (q2.z + q3.z), (q0.w + q1.w) + (q2.w + q3.w)};
      const size_t e = i * 4;
      const int ci = (int)(e % Ci);
      const size_t r = e / Ci;
      const int tp = (int)(r % taps);
      const size_t co = r / taps;
#pragma unroll
      for (int j = 0; j < 4; ++j) dw[(co * Ci + ci + j) * taps + tp] = v[j];
    }
    __syncthreads();
  }
}

template <typename T>
__global__ __launch_bounds__(256) void pack_weight_kernel(const float* __restrict__ w, int Co, int Ci, int taps,
                                                           T* __restrict__ wf, T* __restrict__ wd) {
  const size_t total = (size_t)Co * Ci * taps;
  const bool frag = sizeof(T) == 2 && taps == 9 && Co % 32 == 0 && Ci % 32 == 0;      // wfrag_layout (conv_common.h)
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    // i indexes the logical wf: (co, tp, ci)
    const int ci = (int)(i % Ci);
    const size_t r = i / Ci;
    const int tp = (int)(r % taps);
    const size_t co = r / taps;
    const T v = from_float<T>(w[(co * Ci + ci) * taps + tp]);
    if (frag) {
      wf[wfrag_index((int)co, tp, ci, Ci)] = v;
      if (wd) wd[wfrag_index(ci, taps - 1 - tp, (int)co, Co)] = v;
    } else {
      wf[i] = v;
      if (wd) wd[((size_t)ci * taps + (taps - 1 - tp)) * Co + co] = v;
    }
  }
}

// every conv weight of a model in ONE launch (18 per-layer pack launches per training step were 6 % of the launches of a
// batch-10 step): block = a 1024-element chunk of one tensor, found through the prefix table
constexpr int PACK_MAX_TENSORS = 32;
struct PackMultiArgs {
  const float* w[PACK_MAX_TENSORS]; void* wf[PACK_MAX_TENSORS]; void* wd[PACK_MAX_TENSORS];
  int Co[PACK_MAX_TENSORS], Ci[PACK_MAX_TENSORS], taps[PACK_MAX_TENSORS];
  int start[PACK_MAX_TENSORS + 1];          // prefix sums of sizes in 1024-element chunks
  int n;
};
template <typename T>
__global__ __launch_bounds__(256) void pack_weight_multi_kernel(PackMultiArgs a) {
  const int chunk = blockIdx.x;
  int t = 0;
  while (t + 1 < a.n && chunk >= a.start[t + 1]) ++t;
  const int Co = a.Co[t], Ci = a.Ci[t], taps = a.taps[t];
  const bool frag = sizeof(T) == 2 && taps == 9 && Co % 32 == 0 && Ci % 32 == 0;
  const size_t total = (size_t)Co * Ci * taps;
  const float* __restrict__ w = a.w[t];
  T* __restrict__ wf = reinterpret_cast<T*>(a.wf[t]);
  T* __restrict__ wd = reinterpret_cast<T*>(a.wd[t]);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const size_t i = (size_t)(chunk - a.start[t]) * 1024 + k * 256 + threadIdx.x;      // indexes wf: (co, tp, ci)
    if (i < total) {
      const int ci = (int)(i % Ci);
      const size_t r = i / Ci;
      const int tp = (int)(r % taps);
      const size_t co = r / taps;
      const T v = from_float<T>(w[(co * Ci + ci) * taps + tp]);
      if (frag) {
        wf[wfrag_index((int)co, tp, ci, Ci)] = v;
        if (wd) wd[wfrag_index(ci, taps - 1 - tp, (int)co, Co)] = v;
      } else {
        wf[i] = v;
        if (wd) wd[((size_t)ci * taps + (taps - 1 - tp)) * Co + co] = v;
      }
    }
  }
}

// Fragment-major (bf16 3x3, Co % 32 == 0, Ci % 32 == 0) tensors: one block per 32 co x 32 ci x 9 taps.  The 32 x 288 source
// floats are contiguous per output channel (w[co][ci][tap]) and are read as whole rows into LDS; every 16-byte piece of the
// two packed operands (wf: 8 consecutive ci of one co; wd: 8 consecutive co of one ci, taps reversed -- conv_common.h
// wfrag_index) is then written by one thread, a tap's 2 KiB fragment pair as one contiguous run.  The element-per-thread
// kernel above read with stride 9 and scattered 2-byte stores: 152 us for the 17 M weights of the UNet, this one ~35.
__global__ __launch_bounds__(256) void pack_weight_frag_multi_kernel(PackMultiArgs a) {
  constexpr int PITCH = 289;                                   // floats per co row in LDS (odd: the strided reads spread over the banks)
  __shared__ float sw[32 * PITCH];
  const int blk = blockIdx.x;
  int t = 0;
  while (t + 1 < a.n && blk >= a.start[t + 1]) ++t;
  const int Co = a.Co[t], Ci = a.Ci[t];
  const int local = blk - a.start[t];
  const int nci = Ci >> 5;
  const int cob = local / nci, cib = local % nci;
  const float* __restrict__ w = a.w[t] + ((size_t)cob * 32 * Ci + (size_t)cib * 32) * 9;
  for (int e = threadIdx.x; e < 32 * 288; e += 256) {
    const int r = e / 288, c = e % 288;
    sw[r * PITCH + c] = w[(size_t)r * Ci * 9 + c];
  }
  __syncthreads();
  bf16_t* __restrict__ wf = reinterpret_cast<bf16_t*>(a.wf[t]);
  bf16_t* __restrict__ wd = reinterpret_cast<bf16_t*>(a.wd[t]);
  const int nco = Co >> 5;
  for (int e = threadIdx.x; e < 9 * 128; e += 256) {
    const int tp = e >> 7, p = e & 127;
    const int ks = p >> 6, lane = p & 63, n = lane & 31, k0 = ks * 16 + (lane >> 5) * 8;
    {   // wf: row n = co, k = ci
      alignas(16) bf16_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = from_float<bf16_t>(sw[n * PITCH + (k0 + j) * 9 + tp]);
      *reinterpret_cast<uint4*>(wf + ((((size_t)cob * 9 + tp) * nci + cib) << 10) + (p << 3)) = *reinterpret_cast<const uint4*>(v);
    }
    if (wd) {   // wd: row n = ci, k = co, tap reversed
      alignas(16) bf16_t v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = from_float<bf16_t>(sw[(k0 + j) * PITCH + n * 9 + tp]);
      *reinterpret_cast<uint4*>(wd + ((((size_t)cib * 9 + (8 - tp)) * nco + cob) << 10) + (p << 3)) = *reinterpret_cast<const uint4*>(v);
    }
  }
}

}  // namespace

namespace {
int g_wgrad_co128 = 1;      // A/B switch (im2im_set_option "wgrad_co128")
int g_wgrad_tile16 = 1;     // A/B switch "wgrad_tile16": 256-pixel tiles for the 64-output-channel form
int g_wgrad_fp8_co128 = 1;  // A/B switch "wgrad_fp8_co128": the fp8 weight gradient's 128-output-channel form where Co % 128 == 0
int g_wgrad_roll = 1;       // A/B switch "wgrad_roll": conv_wgrad_roll_kernel (rolling operand prefetch, staging spread over the MFMA phase)
template <typename T, int TAPS>
int launch_wgrad(const void* x, const float* x_ss, const void* x_hi, const float* x_ss_hi, int Ci_lo, const void* dz, float* partial,
                 int64_t partial_bytes, float* dw, int B, int H, int W, int Ci, int Co, int target_wgs, int32_t* nsplit_out, hipStream_t stream) {
  const int g_wgrad_wgs = target_wgs > 0 ? target_wgs : 256;     // [r6] an argument since ABI 3 (was a process-global option)
  constexpr int TH = 8, TW = 16;
  constexpr int PAD = (TAPS == 9) ? 1 : 0;
  constexpr bool IS_BF16 = sizeof(T) == 2;
  constexpr int PB = IS_BF16 ? 192 : 272;
  constexpr size_t smem = (size_t)(TH * TW + (TH + 2 * PAD) * (TW + 2 * PAD)) * PB;
  WgradArgs a{x, dz, partial, B, H, W, Ci, Co, (int)cdiv(H, TH), (int)cdiv(W, TW), 0, 0, x_ss, x_hi, x_ss_hi, Ci_lo};
  a.ntiles = B * a.tilesY * a.tilesX;
  const int cblocks = (int)cdiv(Co, 64) * (int)cdiv(Ci, 64);
  const size_t wsz = (size_t)Co * TAPS * Ci * sizeof(float);
  int64_t max_split = partial_bytes / (int64_t)wsz;
  if (max_split < 1) return fail_invalid("wgrad: workspace smaller than one weight-sized slab");
  int64_t nsplit = cdiv((IS_BF16 && TAPS == 9) ? g_wgrad_wgs : (TAPS == 1 ? 1536 : 512), cblocks);   // pipelined kernel: one workgroup per CU; 1x1: latency-bound, many small blocks
  if (nsplit > a.ntiles) nsplit = a.ntiles;
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  a.tiles_per_split = (int)cdiv(a.ntiles, nsplit);
  nsplit = cdiv(a.ntiles, a.tiles_per_split);
  const bool pipe = IS_BF16 && TAPS == 9 && Ci % 64 == 0;   // the pipelined kernel has no channel masking
  if (pipe) {
   if constexpr (IS_BF16 && TAPS == 9) {
    const bool wide = g_wgrad_co128 && Co % 128 == 0;         // 128 output channels per workgroup (see the kernel)
    if (wide) {
      const int cb128 = (Co / 128) * (Ci / 64);
      nsplit = cdiv(g_wgrad_wgs, cb128);
      if (nsplit > a.ntiles) nsplit = a.ntiles;
      if (nsplit > max_split) nsplit = max_split;
      if (nsplit < 1) nsplit = 1;
      a.tiles_per_split = (int)cdiv(a.ntiles, nsplit);
      nsplit = cdiv(a.ntiles, a.tiles_per_split);
      constexpr size_t smem128 = 2 * ((size_t)TH * TW * (128 * 2 + 64) + (size_t)(TH + 2) * (TW + 2) * 192) + 512;
      const bool lazy = x_ss != nullptr || x_ss_hi != nullptr;
      const bool roll = g_wgrad_roll && (int64_t)(H + 16) * W * std::max(Ci, Co) * 2 < (1ll << 31);   // 32-bit byte offsets within an image
      const bool partial = H % TH != 0 || W % TW != 0;                                             // tiles hang over the image
      auto kern = !roll ? conv_wgrad_pipe_kernel<TH, TW, 128>
                  : partial ? (lazy ? conv_wgrad_roll_kernel<TH, TW, 128, true, true> : conv_wgrad_roll_kernel<TH, TW, 128, false, true>)
                            : (lazy ? conv_wgrad_roll_kernel<TH, TW, 128, true, false> : conv_wgrad_roll_kernel<TH, TW, 128, false, false>);
      static bool attr_set = false;
      if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_pipe_kernel<TH, TW, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_roll_kernel<TH, TW, 128, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_roll_kernel<TH, TW, 128, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_roll_kernel<TH, TW, 128, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_roll_kernel<TH, TW, 128, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
        attr_set = true;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)cb128, (unsigned)nsplit), dim3(768), smem128, stream, a);
      if (int rc = check_launch("conv_wgrad_pipe_kernel<128>")) return rc;
    } else if (g_wgrad_tile16 && H >= 64 && W >= 64) {
      // 64 output channels at the large-extent levels: 256-pixel tiles in the swizzled LDS layout (see the kernel)
      WgradArgs a16 = a;
      a16.tilesY = (int)cdiv(H, 16); a16.tilesX = (int)cdiv(W, 16);
      a16.ntiles = B * a16.tilesY * a16.tilesX;
      if (nsplit > a16.ntiles) nsplit = a16.ntiles;
      a16.tiles_per_split = (int)cdiv(a16.ntiles, nsplit);
      nsplit = cdiv(a16.ntiles, a16.tiles_per_split);
      constexpr size_t smem16 = 2 * ((size_t)16 * 16 * 128 + (size_t)18 * 18 * 128);
      const bool lazy = x_ss != nullptr || x_ss_hi != nullptr;
      const bool roll = g_wgrad_roll && Co % 64 == 0 && (int64_t)(H + 16) * W * std::max(Ci, Co) * 2 < (1ll << 31);
      const bool partial = H % 16 != 0 || W % 16 != 0;
      auto kern = !roll ? conv_wgrad_pipe_kernel<16, 16, 64>
                  : partial ? (lazy ? conv_wgrad_roll_kernel<16, 16, 64, true, true> : conv_wgrad_roll_kernel<16, 16, 64, false, true>)
                            : (lazy ? conv_wgrad_roll_kernel<16, 16, 64, true, false> : conv_wgrad_roll_kernel<16, 16, 64, false, false>);   // (the roll kernel has no output-channel masking)
      static bool attr_set = false;
      if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_pipe_kernel<16, 16, 64>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_roll_kernel<16, 16, 64, true, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_roll_kernel<16, 16, 64, false, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_roll_kernel<16, 16, 64, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);
        hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_roll_kernel<16, 16, 64, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);
        attr_set = true;
      }
      hipLaunchKernelGGL(kern, dim3((unsigned)cblocks, (unsigned)nsplit), dim3(768), smem16, stream, a16);
      if (int rc = check_launch("conv_wgrad_pipe_kernel<16,16,64>")) return rc;
    } else {
    constexpr size_t smem2 = 2 * smem;                        // double-buffered tiles, one 12-wave workgroup per CU
    auto kern = conv_wgrad_pipe_kernel<TH, TW, 64>;
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem2);
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)cblocks, (unsigned)nsplit), dim3(768), smem2, stream, a);
    if (int rc = check_launch("conv_wgrad_pipe_kernel")) return rc;
    }
   }
  } else {
    auto kern = conv_wgrad_kernel<T, TH, TW, TAPS>;
    static bool attr_set = false;
    if (!attr_set && smem > 64 * 1024) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)cblocks, (unsigned)nsplit), dim3(256), smem, stream, a);
    if (int rc = check_launch("conv_wgrad_kernel")) return rc;
  }
  if (nsplit_out) *nsplit_out = (int32_t)nsplit;
  if (!dw) return IM2IM_OK;                                   // the slabs stay in the workspace for im2im_wgrad_reduce_multi
  const size_t total = (size_t)Co * TAPS * Ci;
  int blocks = (int)std::min<size_t>(cdiv(total / 4, 64), 8192);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, partial, (int)nsplit, Co, Ci, TAPS, dw);
  return check_launch("wgrad_reduce_kernel");
}
}  // namespace

extern "C" int64_t im2im_conv_wgrad_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t taps) {
  if (Ci <= 0 || Co <= 0 || Ci % 32 || Co % 32) return -1;
  const int64_t ntiles = (int64_t)B * im2im::cdiv(H, 8) * im2im::cdiv(W, 16);
  const int64_t cblocks = im2im::cdiv(Co, 64) * im2im::cdiv(Ci, 64);
  int64_t nsplit = im2im::cdiv(taps == 1 ? 1536 : 512, cblocks);
  if (nsplit > ntiles) nsplit = ntiles;
  if (nsplit < 1) nsplit = 1;
  return nsplit * (int64_t)Co * taps * Ci * (int64_t)sizeof(float);
}

extern "C" int im2im_conv_wgrad(const void* x, const float* x_scale_shift, const void* dz, float* dw, void* workspace,
                                int64_t workspace_bytes,
                                int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t dtype,
                                im2im_stream_t stream_) {
  return im2im_conv_wgrad_split(x, x_scale_shift, nullptr, nullptr, Ci, dz, dw, workspace, workspace_bytes, B, H, W, Ci, Co, taps,
                                dtype, 0, nullptr, stream_);
}

extern "C" int im2im_conv_wgrad_split(const void* x, const float* x_scale_shift, const void* x_hi, const float* x_scale_shift_hi,
                                      int32_t Ci_lo, const void* dz, float* dw, void* workspace, int64_t workspace_bytes,
                                      int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t dtype,
                                      int32_t target_wgs, int32_t* nsplit, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(x && dz && workspace && (dw || nsplit));
  IM2IM_REQUIRE(target_wgs >= 0 && target_wgs <= 4096);
  if (x_hi) {
    IM2IM_REQUIRE(Ci_lo > 0 && Ci_lo % 64 == 0 && Ci == 2 * Ci_lo);   // a 64-channel block never straddles the two sources
  } else {
    IM2IM_REQUIRE(x_scale_shift_hi == nullptr);
    Ci_lo = Ci;
  }
  IM2IM_REQUIRE(B > 0 && H > 0 && W > 0);
  IM2IM_REQUIRE(Ci > 0 && Ci % 32 == 0);
  IM2IM_REQUIRE(Co > 0 && Co % 32 == 0);
  IM2IM_REQUIRE(taps == 9 || taps == 1);
  IM2IM_REQUIRE(dtype == IM2IM_F32 || dtype == IM2IM_BF16);
  float* partial = reinterpret_cast<float*>(workspace);
  if (dtype == IM2IM_BF16)
    return taps == 9 ? launch_wgrad<bf16_t, 9>(x, x_scale_shift, x_hi, x_scale_shift_hi, Ci_lo, dz, partial, workspace_bytes, dw, B, H, W, Ci, Co, target_wgs, nsplit, stream)
                     : launch_wgrad<bf16_t, 1>(x, x_scale_shift, x_hi, x_scale_shift_hi, Ci_lo, dz, partial, workspace_bytes, dw, B, H, W, Ci, Co, target_wgs, nsplit, stream);
  return taps == 9 ? launch_wgrad<float, 9>(x, x_scale_shift, x_hi, x_scale_shift_hi, Ci_lo, dz, partial, workspace_bytes, dw, B, H, W, Ci, Co, target_wgs, nsplit, stream)
                   : launch_wgrad<float, 1>(x, x_scale_shift, x_hi, x_scale_shift_hi, Ci_lo, dz, partial, workspace_bytes, dw, B, H, W, Ci, Co, target_wgs, nsplit, stream);
}

extern "C" int im2im_wgrad_reduce_multi(int32_t n_tensors, const float* const* slabs, const int32_t* nsplit, const int32_t* Co,
                                        const int32_t* Ci, const int32_t* taps, float* const* dw, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (slabs && nsplit && Co && Ci && taps && dw)));
  for (int t0 = 0; t0 < n_tensors; t0 += REDUCE_MULTI_MAX) {
    ReduceMultiArgs a;
    a.n = std::min(n_tensors - t0, REDUCE_MULTI_MAX);
    int nblocks = 0;
    for (int j = 0; j < a.n; ++j) {
      const int i = t0 + j;
      IM2IM_REQUIRE(slabs[i] && dw[i] && nsplit[i] > 0 && Co[i] > 0 && Ci[i] > 0 && Ci[i] % 4 == 0 && (taps[i] == 9 || taps[i] == 1));
      a.partial[j] = slabs[i]; a.dw[j] = dw[i]; a.nsplit[j] = nsplit[i]; a.Co[j] = Co[i]; a.Ci[j] = Ci[i]; a.taps[j] = taps[i];
      a.first_block[j] = nblocks;
      const size_t total4 = (size_t)Co[i] * taps[i] * Ci[i] / 4;
      nblocks += (int)std::min<size_t>(cdiv(total4, 64), 2048);
    }
    a.first_block[a.n] = nblocks;
    for (int j = a.n + 1; j <= REDUCE_MULTI_MAX; ++j) a.first_block[j] = nblocks;
    hipLaunchKernelGGL(wgrad_reduce_multi_kernel, dim3((unsigned)nblocks), dim3(256), 0, stream, a);
    if (int rc = check_launch("wgrad_reduce_multi_kernel")) return rc;
  }
  return IM2IM_OK;
}

extern "C" int im2im_pack_conv_weight(const float* w, int32_t Co, int32_t Ci, int32_t taps, int32_t dtype, void* wf,
                                      void* wd, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(w && wf && Co > 0 && Ci > 0 && taps > 0);
  IM2IM_REQUIRE(dtype == IM2IM_F32 || dtype == IM2IM_BF16);
  const size_t total = (size_t)Co * Ci * taps;
  int blocks = (int)std::min<size_t>(im2im::cdiv(total, 256), 4096);
  if (dtype == IM2IM_BF16)
    hipLaunchKernelGGL(pack_weight_kernel<bf16_t>, dim3(blocks), dim3(256), 0, stream, w, Co, Ci, taps, (bf16_t*)wf, (bf16_t*)wd);
  else
    hipLaunchKernelGGL(pack_weight_kernel<float>, dim3(blocks), dim3(256), 0, stream, w, Co, Ci, taps, (float*)wf, (float*)wd);
  return im2im::check_launch("pack_weight_kernel");
}

// run-time switches for within-process A/B measurements (tools/, bench).  Not a reference interface.
extern "C" int im2im_set_option(const char* key, int32_t value) {
  IM2IM_REQUIRE(key != nullptr);
  if (std::string(key) == "conv_splitk") { im2im::set_conv_splitk(value); return IM2IM_OK; }
  if (std::string(key) == "wgrad_co128") { g_wgrad_co128 = value; return IM2IM_OK; }
  if (std::string(key) == "wgrad_tile16") { g_wgrad_tile16 = value; return IM2IM_OK; }
  if (std::string(key) == "wgrad_roll") { g_wgrad_roll = value; return IM2IM_OK; }
  if (std::string(key) == "wgrad_fp8_co128") { g_wgrad_fp8_co128 = value; return IM2IM_OK; }
  if (std::string(key) == "bn_fused_small") { im2im::set_bn_fused_small(value); return IM2IM_OK; }
  if (std::string(key) == "bn_onelaunch") { im2im::set_bn_onelaunch(value); return IM2IM_OK; }
  if (std::string(key) == "pool_bwd_full") { im2im::set_pool_bwd_full(value); return IM2IM_OK; }
  if (std::string(key) == "pool_bwd_blocks") { im2im::set_pool_bwd_blocks(value); return IM2IM_OK; }
  if (std::string(key) == "bn_apply_keep_mb") { im2im::set_bn_apply_keep_mb(value); return IM2IM_OK; }
#ifdef IM2IM_BUILD_EXPERIMENTAL
  if (std::string(key) == "conv_roll") { im2im::set_conv_roll(value); return IM2IM_OK; }
#else
  if (std::string(key) == "conv_roll")       // conv_roll.hip is only in libraries built with IM2IM_BUILD_EXPERIMENTAL=1: 0 = already the case
    return value == 0 ? IM2IM_OK : im2im::fail_invalid("conv_roll: this library was built without the experimental kernels (IM2IM_BUILD_EXPERIMENTAL=1)");
#endif
  return im2im::fail_invalid("unknown option");
}

extern "C" int im2im_pack_conv_weights_multi(int32_t n_tensors, const float* const* w, const int32_t* Co, const int32_t* Ci,
                                             const int32_t* taps, int32_t dtype, void* const* wf, void* const* wd,
                                             im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (w && Co && Ci && taps && wf && wd)));
  IM2IM_REQUIRE(dtype == IM2IM_F32 || dtype == IM2IM_BF16);
  // fragment-major tensors (bf16 3x3, multiples of 32 channels) go to the block-per-32x32 kernel, the rest to the chunked one
  std::vector<int> frag, plain;
  for (int i = 0; i < n_tensors; ++i) {
    IM2IM_REQUIRE(w[i] && wf[i] && Co[i] > 0 && Ci[i] > 0 && taps[i] > 0);
    (dtype == IM2IM_BF16 && wfrag_layout(2, taps[i], Co[i], Ci[i]) ? frag : plain).push_back(i);
  }
  for (int pass = 0; pass < 2; ++pass) {
    const std::vector<int>& idx = pass == 0 ? frag : plain;
    for (size_t base = 0; base < idx.size(); base += PACK_MAX_TENSORS) {
      PackMultiArgs a;
      a.n = (int)std::min<size_t>(PACK_MAX_TENSORS, idx.size() - base);
      int units = 0;
      for (int i = 0; i < a.n; ++i) {
        const int j = idx[base + i];
        a.w[i] = w[j]; a.wf[i] = wf[j]; a.wd[i] = wd[j];
        a.Co[i] = Co[j]; a.Ci[i] = Ci[j]; a.taps[i] = taps[j];
        a.start[i] = units;
        units += pass == 0 ? (Co[j] / 32) * (Ci[j] / 32) : (int)im2im::cdiv((int64_t)Co[j] * Ci[j] * taps[j], 1024);
      }
      a.start[a.n] = units;
      if (units == 0) continue;
      if (pass == 0) hipLaunchKernelGGL(pack_weight_frag_multi_kernel, dim3((unsigned)units), dim3(256), 0, stream, a);
      else if (dtype == IM2IM_BF16) hipLaunchKernelGGL(pack_weight_multi_kernel<bf16_t>, dim3((unsigned)units), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL(pack_weight_multi_kernel<float>, dim3((unsigned)units), dim3(256), 0, stream, a);
      if (int rc = im2im::check_launch("pack_weight_multi_kernel")) return rc;
    }
  }
  return IM2IM_OK;
}

// ------------------------------------------------------------------------------------------------ fp8 weight gradient (ABI)
extern "C" int im2im_conv_wgrad_fp8(const void* x, const float* x_scale_shift, const void* x_hi, const float* x_scale_shift_hi,
                                    int32_t Ci_lo, const void* dz, const float* amax_prev, float* dw, void* workspace,
                                    int64_t workspace_bytes, int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                                    int32_t target_wgs, int32_t* nsplit_out, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(x && dz && (dw || nsplit_out) && workspace && amax_prev);
  IM2IM_REQUIRE(target_wgs >= 0 && target_wgs <= 4096);
  const int g_wgrad_wgs = target_wgs > 0 ? target_wgs : 256;
  if (x_hi) {
    IM2IM_REQUIRE(Ci_lo > 0 && Ci_lo % 64 == 0 && Ci == 2 * Ci_lo);
  } else {
    IM2IM_REQUIRE(x_scale_shift_hi == nullptr);
    Ci_lo = Ci;
  }
  IM2IM_REQUIRE(B > 0 && H > 0 && W > 0);
  IM2IM_REQUIRE(Ci > 0 && Ci % 64 == 0 && Co > 0 && Co % 64 == 0);
  constexpr int TH = 8, TW = 16;
  WgradArgs a{x, dz, reinterpret_cast<float*>(workspace), B, H, W, Ci, Co, (int)cdiv(H, TH), (int)cdiv(W, TW), 0, 0, x_scale_shift,
              x_hi, x_scale_shift_hi, Ci_lo};
  a.ntiles = B * a.tilesY * a.tilesX;
  const bool wide = g_wgrad_fp8_co128 && Co % 128 == 0;
  const int cot = wide ? 128 : 64;
  const int cblocks = (Co / cot) * (Ci / 64);
  const size_t wsz = (size_t)Co * 9 * Ci * sizeof(float);
  const int64_t max_split = workspace_bytes / (int64_t)wsz;
  if (max_split < 1) return fail_invalid("wgrad_fp8: workspace smaller than one weight-sized slab");
  int64_t nsplit = cdiv(g_wgrad_wgs, cblocks);
  if (nsplit > a.ntiles) nsplit = a.ntiles;
  if (nsplit > max_split) nsplit = max_split;
  if (nsplit < 1) nsplit = 1;
  a.tiles_per_split = (int)cdiv(a.ntiles, nsplit);
  nsplit = cdiv(a.ntiles, a.tiles_per_split);
  Fp8WgradArgs fa{a, amax_prev};
  const bool roll = g_wgrad_roll && (int64_t)(H + 16) * W * std::max(Ci, Co) * 2 < (1ll << 31);   // 32-bit byte offsets within an image
  const bool lazy = x_scale_shift != nullptr || x_scale_shift_hi != nullptr, partial = H % TH != 0 || W % TW != 0;
  if (wide) {
    constexpr size_t smem = 2 * ((size_t)TH * TW * (128 + 32) + (size_t)(TH + 2) * (TW + 2) * 96) + 512;
    auto kern = !roll ? conv_wgrad_fp8_kernel<128>
                : partial ? (lazy ? conv_wgrad_fp8_roll_kernel<128, true, true> : conv_wgrad_fp8_roll_kernel<128, false, true>)
                          : (lazy ? conv_wgrad_fp8_roll_kernel<128, true, false> : conv_wgrad_fp8_roll_kernel<128, false, false>);
    static bool attr_set = false;
    if (!attr_set) {
      const void* ks[] = {reinterpret_cast<const void*>(conv_wgrad_fp8_kernel<128>), reinterpret_cast<const void*>(conv_wgrad_fp8_roll_kernel<128, true, true>),
                          reinterpret_cast<const void*>(conv_wgrad_fp8_roll_kernel<128, false, true>), reinterpret_cast<const void*>(conv_wgrad_fp8_roll_kernel<128, true, false>),
                          reinterpret_cast<const void*>(conv_wgrad_fp8_roll_kernel<128, false, false>)};
      for (const void* k : ks) hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)cblocks, (unsigned)nsplit), dim3(768), smem, stream, fa);
  } else {
    constexpr size_t smem = 2 * ((size_t)TH * TW * (64 + 32) + (size_t)(TH + 2) * (TW + 2) * 96) + 512;
    auto kern = !roll ? conv_wgrad_fp8_kernel<64>
                : partial ? (lazy ? conv_wgrad_fp8_roll_kernel<64, true, true> : conv_wgrad_fp8_roll_kernel<64, false, true>)
                          : (lazy ? conv_wgrad_fp8_roll_kernel<64, true, false> : conv_wgrad_fp8_roll_kernel<64, false, false>);
    static bool attr_set = false;
    if (!attr_set) {
      const void* ks[] = {reinterpret_cast<const void*>(conv_wgrad_fp8_kernel<64>), reinterpret_cast<const void*>(conv_wgrad_fp8_roll_kernel<64, true, true>),
                          reinterpret_cast<const void*>(conv_wgrad_fp8_roll_kernel<64, false, true>), reinterpret_cast<const void*>(conv_wgrad_fp8_roll_kernel<64, true, false>),
                          reinterpret_cast<const void*>(conv_wgrad_fp8_roll_kernel<64, false, false>)};
      for (const void* k : ks) hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
      attr_set = true;
    }
    hipLaunchKernelGGL(kern, dim3((unsigned)cblocks, (unsigned)nsplit), dim3(768), smem, stream, fa);
  }
  if (int rc = check_launch("conv_wgrad_fp8_kernel")) return rc;
  if (nsplit_out) *nsplit_out = (int32_t)nsplit;
  if (!dw) return IM2IM_OK;
  const size_t total = (size_t)Co * 9 * Ci;
  int blocks = (int)std::min<size_t>(cdiv(total / 4, 64), 8192);
  hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<float*>(workspace), (int)nsplit, Co, Ci, 9, dw);
  return check_launch("wgrad_reduce_kernel");
}
