// MFMA implicit-GEMM convolution for gfx950: NHWC activations, 3x3 pad-1 or 1x1.
//
// Replaces the nn.Conv2d calls of the reference UNet (core/models/trunks/unet_parts.py:16,19,90)
// and, through the same kernels, their autograd backward (dgrad = forward kernel on tap-flipped,
// transposed weights; wgrad = its own kernel).
//
// GEMM view:  M = output pixels (a TH x TW patch of one image per workgroup), N = output channels,
// K = taps * Cin.  Per workgroup (256 threads = 4 waves):
//   * the input halo patch (TH+2)x(TW+2) x 32 channels is staged once per Cin-chunk into LDS and
//     re-used by all 9 taps (a tap is just an LDS address offset);
//   * weights: bf16 3x3 tiles whose waves own 128 pixels x 64 channels (the 128-wide tiles, the 32x16x64 tile) read each
//     32 x 16 operand fragment straight from L2 into registers, one tap ahead, from a fragment-major pack (conv_common.h
//     wfrag_index) -- no weight LDS traffic, two barriers per chunk [r3]; every other variant (fp32, 1x1, 4 x 1-wave tiles)
//     double-buffers the weight tile [BN][32] of one (tap, Cin-chunk) in LDS, the next one prefetched into registers while
//     the MFMAs of the current one issue;
//   * v_mfma_f32_32x32x16_bf16 (bf16 in, fp32 accumulate) or v_mfma_f32_32x32x2_f32 (exact fp32,
//     parity mode); operand/result lane maps verified on hardware (tools/hwprobe).
//   * epilogue: + bias, optional folded-BN affine + ReLU (eval), store, and per-channel partial
//     sums / sums of squares for train-mode BatchNorm (deterministic, no atomics).
// LDS rows are padded by 16 B so ds_read_b128 operand fetches are (nearly) conflict free.
#include "conv_common.h"
#ifndef IM2IM_SETPRIO
#define IM2IM_SETPRIO 0
#endif
#ifndef IM2IM_IGEMM_COB_INNER
#define IM2IM_IGEMM_COB_INNER 1
#endif
#ifndef IM2IM_IGEMM_XCD_BANDS
#define IM2IM_IGEMM_XCD_BANDS 1
#endif
#ifndef IM2IM_WGRAD_XCD
#define IM2IM_WGRAD_XCD 1
#endif
#include <string>
#include <type_traits>
// measurement-only builds (tools/ab_build.sh, never the shipped library): bit 0 = no BatchNorm statistics in the epilogue,
// bit 1 = the epilogue is one sum + one store per lane, bit 2 = no lazy BatchNorm+ReLU on the staged input
#ifndef IM2IM_ABLATE
#define IM2IM_ABLATE 0
#endif
#ifndef IM2IM_WGRAD_ABL      // measurement-only: bit 0 = no global loads after the first tile, bit 1 = no LDS writes, bit 2 = no MFMA phase, bit 3 = loads of the same (cache-hot) tile, bit 4 = the dz half of the staging only for the first tile
#define IM2IM_WGRAD_ABL 0
#endif

namespace {

using namespace im2im;

// workgroups per CU a variant is compiled for, by its accumulator registers per lane: bf16 128 -> 2 (256 registers each),
// 64 or fewer -> 3 (168); fp32 (operand buffers twice as large) 128 -> 1 (the whole 512-entry file), fewer -> 2
template <typename T, int ACC> constexpr int igemm_wgs_per_cu() { return sizeof(T) == 2 ? (ACC >= 128 ? 2 : 3) : (ACC >= 128 ? 1 : 2); }

// EPI: 0 = (+bias) store only [data-gradient, 1x1 conv]; 1 = +bias, store, BatchNorm partial statistics [train forward];
//      2 = folded BatchNorm affine + ReLU [eval forward];  3 = data-gradient that also starts the BatchNorm+ReLU backward
//      of the layer it differentiates into: while the rows go out, the same lanes read the producer's z at the same
//      addresses and accumulate sum(g) and sum(g*xhat), g = da*[z*scale+shift > 0] -- bn_relu_bwd_reduce without its
//      own pass over da and z.  Compile-time so the 128 values per lane pay only for what
//      the launch needs (the generic epilogue was ~10 VALU per value; the data-gradient needs ~1).
// bf16: two workgroups per CU (256 registers per lane each), three for the 64- and 32-channel-wide tiles (168 registers).  fp32: the operand buffers are twice as large (84-94 KB for
// the 128-wide tiles), so only one workgroup fits a CU anyway -- it may then use the whole 512-entry register file
// instead of spilling 150-540 registers to scratch as it did under the two-workgroup bound.
template <typename T, int TB, int TH, int TW, int BN, int WM, int WN, int TAPS, int EPI>
__global__ __launch_bounds__(256, (igemm_wgs_per_cu<T, TB * TH * TW / (32 * WM) * (BN / (32 * WN)) * 16>())) void conv_igemm_kernel(ConvArgs a) {
  constexpr int PAD = (TAPS == 9) ? 1 : 0;
  constexpr int HH = TH + 2 * PAD, HWD = TW + 2 * PAD, HPI = HH * HWD, HPX = TB * HPI;   // halo pixels per image / per tile
  constexpr int MI = TH * TW;                         // output pixels per image in the tile
  constexpr int KC = 32;
  constexpr int EPP = 16 / (int)sizeof(T);          // elements per 16-B piece
  constexpr int PPR = KC / EPP;                     // pieces per row
  constexpr int ROWB = KC * (int)sizeof(T) + 16;    // padded LDS row pitch (bytes)
  constexpr int M = TB * TH * TW;
  constexpr int MT = M / (32 * WM), NT = BN / (32 * WN);
  static_assert(WM * WN == 4, "4 waves");
  static_assert(MT >= 1 && NT >= 1 && M % (32 * WM) == 0 && BN % (32 * WN) == 0, "tile split");
  constexpr int A_ROUNDS = (HPX * PPR + 255) / 256;
  constexpr int B_ROUNDS = (BN * PPR + 255) / 256;
  // halo rows are padded by 96 B (bf16) so that the two image rows a 32-lane operand fetch spans land on disjoint
  // bank groups: ds_read_b128 of the A fragments becomes conflict-free (it was 2-way for 16-wide, 3-way for 8-wide
  // tiles; found by enumerating the hardware's 16-lane groups).
  constexpr int HROWB = HWD * ROWB + (sizeof(T) == 2 ? 96 : 0);
  constexpr int HIMGB = HH * HROWB;
  constexpr int A_BYTES = TB * HIMGB, B_BYTES = BN * ROWB;
  constexpr bool FRAGW = sizeof(T) == 2 && TAPS == 9;   // packed weights are fragment-major (conv_common.h wfrag_index)
  // ... and read straight from L2 into registers where a fragment feeds MT = 4 MFMAs.  With MT = 2 (4 x 1 waves on a 64-wide
  // tile, three workgroups per CU) the same loads saturate the CU's load path (16 KiB per workgroup and tap): 750 vs 847 TF on
  // the 320x320 64->64 data-gradient; such tiles stage the weights through LDS once per workgroup.  (A 2 x 2 wave split of the
  // 64-wide tile, MT = 4 / NT = 1 with direct weights, measured 6-14 % slower than that on the three 64-output-channel layers.)
  constexpr bool DIRECTW = FRAGW && MT >= 4;
  constexpr int NBUF = 2;                           // weight buffers (one tap each, double-buffered)

  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* ldsA = smem;
  char* ldsB = smem + A_BYTES;                      // the weight buffers
  float* ldsS = reinterpret_cast<float*>(smem);     // stats scratch, re-uses A after the main loop
  float* ldsSS = reinterpret_cast<float*>(smem + A_BYTES + NBUF * B_BYTES);   // [2][Ci] input scale/shift (lazy BN+ReLU)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;

#if IM2IM_IGEMM_XCD_BANDS
  // workgroups go to the 8 XCDs round-robin in dispatch order: give every XCD a contiguous band of tiles, so that the halo
  // pixels neighbouring tiles share are found in ITS L2 instead of being fetched by eight different ones
#if IM2IM_IGEMM_COB_INNER
  // 1-D grid over (tile, output-channel block): within an XCD's band the channel blocks of one tile run back to back, so
  // the tile's input halo comes from HBM once and from that XCD's L2 for the other blocks
  int tile_id, cob;
  {
    const int ncob = a.Co / BN;
    const int lin = blockIdx.x, total = (int)gridDim.x;
    const int band = (total >> 3) / ncob;                 // tiles per XCD band
    if (lin < band * ncob * 8) {
      const int xcd = lin & 7, j = lin >> 3;
      tile_id = xcd * band + j / ncob;
      cob = j % ncob;
    } else {                                              // leftover tiles: plain order
      const int r = lin - band * ncob * 8;
      tile_id = band * 8 + r / ncob;
      cob = r % ncob;
    }
  }
#else
  int tile_id = blockIdx.x;
  {
    const int band = (int)gridDim.x >> 3;
    if (tile_id < band * 8) tile_id = (tile_id & 7) * band + (tile_id >> 3);
  }
  const int cob = blockIdx.y;
#endif
#else
  const int tile_id = blockIdx.x, cob = blockIdx.y;
#endif
  int mt_id = tile_id;
  const int tx_id = mt_id % a.tilesX; mt_id /= a.tilesX;
  const int ty_id = mt_id % a.tilesY;
  const int b0 = (mt_id / a.tilesY) * TB;             // first image of this tile (TB images share the weight tiles)
  const int y0 = ty_id * TH, x0 = tx_id * TW;
  const int n0 = cob * BN;

  const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
  const bool split_in = a.x_hi != nullptr;
  const int xstride = split_in ? a.Ci_lo : a.Ci;       // pixel stride of the source tensor(s)

  // per-lane LDS byte offsets of this lane's A rows / B rows
  int aoff[MT], boff[NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int m = (wm * MT + mt) * 32 + l31;
    aoff[mt] = (m / MI) * HIMGB + ((m % MI) / TW) * HROWB + (m % TW) * ROWB;
  }
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) boff[nt] = ((wn * NT + nt) * 32 + l31) * ROWB;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  uint4 ra[A_ROUNDS], rb[2][B_ROUNDS];

  // Per-lane element offsets of this thread's staging pieces, computed ONCE (uniform base + 32-bit lane offset lets
  // the compiler use scalar-base addressing and keeps the unrolled loop free of index arithmetic).
  //   halo piece i:  xg_tile + a_goff[i] + chunk*KC      (a_goff < 0: outside the image/batch -> zeros)
  //   weight piece i: wg_tile + b_goff[i] + tap*Ci + chunk*KC
  const T* __restrict__ xg_tile = xg + (size_t)b0 * a.H * a.W * xstride;
  const T* __restrict__ xh_tile = reinterpret_cast<const T*>(a.x_hi) + (size_t)b0 * a.H * a.W * xstride;
  const T* __restrict__ wg_tile = wg + (size_t)n0 * TAPS * a.Ci;
  int a_goff[A_ROUNDS], a_loff[A_ROUNDS], b_goff[B_ROUNDS], b_loff[B_ROUNDS];
#pragma unroll
  for (int i = 0; i < A_ROUNDS; ++i) {
    const int p = i * 256 + tid;
    const int px = p / PPR, part = p % PPR;
    const int tb = px / HPI, pi = px % HPI;
    const int yy = y0 + pi / HWD - PAD, xx = x0 + pi % HWD - PAD;
    const bool ok = px < HPX && b0 + tb < a.B && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
    a_goff[i] = ok ? (((tb * a.H + yy) * a.W + xx) * xstride + part * EPP) : -1;
    a_loff[i] = (px < HPX) ? tb * HIMGB + (pi / HWD) * HROWB + (pi % HWD) * ROWB + part * 16 : -1;
  }
#pragma unroll
  for (int i = 0; i < B_ROUNDS; ++i) {
    const int p = i * 256 + tid;
    const int n = p / PPR, part = p % PPR;
    if constexpr (FRAGW) b_goff[i] = (n < BN) ? (int)(wfrag_index(n0 + n, 0, part * EPP, a.Ci) - (size_t)n0 * TAPS * a.Ci) : -1;   // fragment-major pack
    else b_goff[i] = (n < BN) ? (n * TAPS * a.Ci + part * EPP) : -1;
    b_loff[i] = (n < BN) ? n * ROWB + part * 16 : -1;
  }

  auto gload_A = [&](int chunk) __attribute__((always_inline)) {
    const int c = chunk * KC;
    const T* src = (split_in && c >= a.Ci_lo) ? xh_tile + (c - a.Ci_lo) : xg_tile + c;
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a_goff[i] >= 0) v = *reinterpret_cast<const uint4*>(src + a_goff[i]);
      ra[i] = v;
    }
  };
  const bool lazy_lo = a.in_ss != nullptr, lazy_hi = a.in_ss_hi != nullptr;
  if (lazy_lo || lazy_hi) {
    // ldsSS = [scale over the Ci logical channels][shift ...]; a source without coefficients is never looked up
    const int clo = split_in ? a.Ci_lo : a.Ci, chi = a.Ci - clo;
    const float* __restrict__ ss_lo = a.in_ss + (size_t)b0 * a.in_ss_img;      // per-image coefficients (GroupNorm): TB == 1
    if (lazy_lo) for (int i = tid; i < clo; i += 256) { ldsSS[i] = ss_lo[i]; ldsSS[a.Ci + i] = ss_lo[clo + i]; }
    if (lazy_hi) for (int i = tid; i < chi; i += 256) { ldsSS[clo + i] = a.in_ss_hi[i]; ldsSS[a.Ci + clo + i] = a.in_ss_hi[chi + i]; }
    __syncthreads();
  }
  auto swrite_A = [&](int chunk) __attribute__((always_inline)) {
    const bool lazy_in = (split_in && chunk * KC >= a.Ci_lo) ? lazy_hi : lazy_lo;
    if (lazy_in && !(IM2IM_ABLATE & 4)) {
      // this thread's pieces all cover the same EPP channels of the chunk: chunk*KC + (tid % PPR)*EPP ...
      const int c0 = chunk * KC + (tid % PPR) * EPP;
      float sc[EPP], sh[EPP];
#pragma unroll
      for (int k = 0; k < EPP; ++k) { sc[k] = ldsSS[c0 + k]; sh[k] = ldsSS[a.Ci + c0 + k]; }
#pragma unroll
      for (int i = 0; i < A_ROUNDS; ++i) {
        if (a_goff[i] >= 0) {                        // padding / out-of-image pieces stay exactly zero
          float v[EPP];
          Vec16<T>::load(reinterpret_cast<const T*>(&ra[i]), v);
#pragma unroll
          for (int k = 0; k < EPP; ++k) v[k] = fmaxf(v[k] * sc[k] + sh[k], 0.f);
          Vec16<T>::store(reinterpret_cast<T*>(&ra[i]), v);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i)
      if ((HPX * PPR) % 256 == 0 || a_loff[i] >= 0) *reinterpret_cast<uint4*>(ldsA + a_loff[i]) = ra[i];
  };
  auto gload_B = [&](uint4 (&r)[B_ROUNDS], int chunk, int tap) {
    const T* src = wg_tile + (FRAGW ? ((tap * (a.Ci >> 5) + chunk) << 10) : (tap * a.Ci + chunk * KC));
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if ((BN * PPR) % 256 == 0 || b_goff[i] >= 0) v = *reinterpret_cast<const uint4*>(src + b_goff[i]);
      r[i] = v;
    }
  };
  auto swrite_B = [&](const uint4 (&r)[B_ROUNDS], int buf) {
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i)
      if ((BN * PPR) % 256 == 0 || b_loff[i] >= 0) *reinterpret_cast<uint4*>(ldsB + buf * B_BYTES + b_loff[i]) = r[i];
  };
  auto compute = [&](int toff, int buf) __attribute__((always_inline)) {
    const char* pa = ldsA + toff;
    const char* pb = ldsB + buf * B_BYTES;
#pragma unroll
    for (int ks = 0; ks < Frag<T>::KSTEPS; ++ks) {
      typename Frag<T>::AB fa[MT], fb[NT];
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) fa[mt] = Frag<T>::load(pa + aoff[mt], ks, half);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fb[nt] = Frag<T>::load(pb + boff[nt], ks, half);
#if IM2IM_SETPRIO
      __builtin_amdgcn_s_setprio(1);               // the co-resident wave of the other workgroup is in its staging phase
#endif
#pragma unroll
      for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Frag<T>::mfma(fa[mt], fb[nt], acc[mt][nt]);
#if IM2IM_SETPRIO
      __builtin_amdgcn_s_setprio(0);
#endif
    }
  };

  const int nchunks = a.Ci / KC;
  if constexpr (DIRECTW) {
    // bf16 3x3, 128-channel-wide tiles: the weight operand never touches LDS.  The packed weights are fragment-major (conv_common.h): a wave reads
    // each 32 x 16 operand fragment of its NT channel blocks straight from L2 into registers with one coalesced 1 KiB load, one
    // tap ahead of its use.  Only the halo goes through LDS, so a chunk's nine taps (18 k-steps) run between two barriers
    // instead of ten, with no weight ds_write / ds_read at all (tools/hwprobe/directb_probe.hip measured the loop shape).
    using AB = typename Frag<T>::AB;
    AB fw[2][2][NT];                                   // [register set][k-step][channel block]
    const size_t cob_stride = (size_t)9 * nchunks * 1024;                 // elements per 32-row block
    const T* __restrict__ wfr = wg + (size_t)(n0 / 32 + wn * NT) * cob_stride + lane * 8;
    auto gload_F = [&](AB (&f)[2][NT], int chunk, int tap) __attribute__((always_inline)) {
      const T* p = wfr + ((size_t)(tap * nchunks + chunk) << 10);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) f[ks][nt] = *reinterpret_cast<const AB*>(p + nt * cob_stride + ks * 512);
    };
    auto compute_F = [&](int toff, const AB (&f)[2][NT]) __attribute__((always_inline)) {
      const char* pa = ldsA + toff;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        AB fa[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) fa[mt] = Frag<T>::load(pa + aoff[mt], ks, half);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) acc[mt][nt] = Frag<T>::mfma(fa[mt], f[ks][nt], acc[mt][nt]);
      }
    };
    gload_A(0);
    gload_F(fw[0], 0, 0);
    auto chunk_body = [&](auto parity, int chunk) __attribute__((always_inline)) {
      constexpr int P0 = decltype(parity)::value;        // 9 taps per chunk: the register-set parity of a chunk's first tap alternates
      const bool more = chunk + 1 < nchunks;
      if (chunk) __syncthreads();                      // everyone done reading the previous halo
      swrite_A(chunk);
      __syncthreads();
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        constexpr int dummy = 0; (void)dummy;
        const int set = (P0 + tap) & 1;                  // compile-time after unrolling
        if (tap + 1 < 9) { if (set) gload_F(fw[0], chunk, tap + 1); else gload_F(fw[1], chunk, tap + 1); }
        else if (more) { if (set) gload_F(fw[0], chunk + 1, 0); else gload_F(fw[1], chunk + 1, 0); }
        if (tap == 6 && more) gload_A(chunk + 1);        // the next chunk's halo, three taps early
        if (set) compute_F((tap / 3) * HROWB + (tap % 3) * ROWB, fw[1]); else compute_F((tap / 3) * HROWB + (tap % 3) * ROWB, fw[0]);
      }
    };
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
      chunk_body(std::integral_constant<int, 0>{}, chunk);
      if (chunk + 1 < nchunks) chunk_body(std::integral_constant<int, 1>{}, chunk + 1);
    }
  } else if constexpr (TAPS == 9) {
    // 9 taps fully unrolled (every register-set / LDS-buffer index and tap offset is a compile-time constant).
    // Weight tiles are prefetched TWO iterations ahead: iteration `it` writes register set it&1 (loaded at it-2)
    // to LDS buffer it&1 and immediately re-issues that set for it+2.  9 is odd, so the parity of a chunk's first
    // tap alternates: the body is instantiated for both parities.  The next chunk's halo is fetched 3 taps early.
    gload_A(0);
    gload_B(rb[0], 0, 0);
    gload_B(rb[1], 0, 1);
    auto chunk_body = [&](auto parity, int chunk) {
      constexpr int P0 = decltype(parity)::value;
      const bool more = chunk + 1 < nchunks;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        constexpr int dummy = 0; (void)dummy;
        const int set = (P0 + tap) & 1;                // compile-time after unrolling
        if (tap == 0) {
          if (chunk) __syncthreads();                // everyone done reading the previous halo
          swrite_A(chunk);
        }
        if (set) swrite_B(rb[1], 1); else swrite_B(rb[0], 0);
        __syncthreads();
        if (tap + 2 < 9) { if (set) gload_B(rb[1], chunk, tap + 2); else gload_B(rb[0], chunk, tap + 2); }
        else if (more) { if (set) gload_B(rb[1], chunk + 1, tap + 2 - 9); else gload_B(rb[0], chunk + 1, tap + 2 - 9); }
        if (tap == 6 && more) gload_A(chunk + 1);
        compute((tap / 3) * HROWB + (tap % 3) * ROWB, set);
      }
    };
    for (int chunk = 0; chunk < nchunks; chunk += 2) {
      chunk_body(std::integral_constant<int, 0>{}, chunk);
      if (chunk + 1 < nchunks) chunk_body(std::integral_constant<int, 1>{}, chunk + 1);
    }
  } else {
    gload_A(0);
    gload_B(rb[0], 0, 0);
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      if (chunk) __syncthreads();
      swrite_A(chunk);
      if (chunk & 1) swrite_B(rb[1], 1); else swrite_B(rb[0], 0);
      __syncthreads();
      if (chunk + 1 < nchunks) {
        gload_A(chunk + 1);
        if (chunk & 1) gload_B(rb[0], chunk + 1, 0); else gload_B(rb[1], chunk + 1, 0);
      }
      compute(0, chunk & 1);
    }
  }

  // ---------------------------------------------------------------- epilogue
  // The MFMA result layout gives each lane ONE channel of 16 pixel rows; storing that directly is 2 bytes per lane
  // per store instruction (issue-bound).  Instead every wave transposes its sub-tile through a private LDS region
  // ([pixel][channel], rows padded by 16 B) and writes whole 16-byte pieces of NHWC rows.
  constexpr int WROWS = MT * 32, WCOLS = NT * 32;
  constexpr int WP = WCOLS * (int)sizeof(T) + 16;             // padded row pitch of the wave's LDS tile
  constexpr int WBYTES = WROWS * WP;
  constexpr int EPR = WCOLS / EPP;                            // 16-byte pieces per row
  constexpr int ROWS_PER_PASS = 64 / EPR;
  // this wave's WCOLS output channels start at ncol; with a split destination they all belong to one of the tensors
  const int ncol = n0 + wn * WCOLS;
  const bool to_hi = a.y_hi != nullptr && ncol >= a.Co_lo;
  T* __restrict__ yg = reinterpret_cast<T*>(to_hi ? a.y_hi : a.y) + (to_hi ? ncol - a.Co_lo : ncol);
  const int ystride = a.y_hi == nullptr ? a.Co : (to_hi ? a.Co - a.Co_lo : a.Co_lo);
  constexpr bool want_stats = (EPI == 1) && !(IM2IM_ABLATE & 1);
#if IM2IM_ABLATE & 2
  {
    float t = 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) t += acc[mt][nt][r];
    yg[(size_t)tile_id * 256 + tid] = from_float<T>(t);
    return;
  }
#endif
  // EPI 3: the producer's z pieces this lane will need in the row-copy loop below are requested NOW, so their HBM
  // latency runs under the accumulator conversion and the LDS transpose instead of stalling each pass
  constexpr int PASSES = WROWS / ROWS_PER_PASS;
  uint4 zreg[EPI == 3 ? PASSES : 1];
  if constexpr (EPI == 3) {
    const T* __restrict__ bzp = reinterpret_cast<const T*>(a.bn_z) + ncol + (lane % EPR) * EPP;
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      const int m = wm * WROWS + pass * ROWS_PER_PASS + lane / EPR;
      const int bb = b0 + m / MI, yy = y0 + (m % MI) / TW, xx = x0 + m % TW;
      zreg[pass] = make_uint4(0, 0, 0, 0);
      if (bb < a.B && yy < a.H && xx < a.W)
        zreg[pass] = *reinterpret_cast<const uint4*>(bzp + (((size_t)bb * a.H + yy) * a.W + xx) * a.Co);
    }
  }
  __syncthreads();                                           // every wave is done reading the operand buffers
  char* wbuf = smem + wave * WBYTES;
  // BatchNorm partial statistics of this lane's values of channel nt: count, mean and M2 (sum of squared deviations).
  // Sums are taken relative to K = the lane's first value of the channel (a sample of the data, or of its zero-padded
  // continuation in an overhanging tile), so M2 = q - s^2/n has no catastrophic cancellation however far the channel
  // mean is from zero.  The count only depends on the lane's rows: computed once.
  float st_n[NT], st_m[NT], st_q[NT];
  // A tile that lies completely inside the batch and the image (every tile of the 320 / 160 / 80 / 40-pixel levels) needs no
  // per-value validity test: the test compiled to an exec-mask branch around each of the 128 values (the statistics epilogue
  // was 4,663 instructions against 1,900 for the whole main loop of a 64-channel layer).  Wave-uniform choice, same arithmetic.
  const bool tile_full = (b0 + TB <= a.B) && (y0 + TH <= a.H) && (x0 + TW <= a.W);
  static_assert(MT * 16 <= 64, "one validity bit per accumulator row of the lane");
  unsigned long long okmask = ~0ull;                         // bit mt*16+r: this lane's row (mt, r) is a pixel of the image
  float cnt = (float)(16 * MT);
  if constexpr (want_stats) {
    if (!tile_full) {                                        // overhanging tile: ONE pass over the rows, kept as a bit mask
      okmask = 0ull;                                         // (64 separately hoisted predicates spilled registers)
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
  auto convert_tile = [&](auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nl = (wn * NT + nt) * 32 + l31;                  // channel within the block tile
      const int n = n0 + nl;
      const float bias_v = (a.bias ? a.bias[n] : 0.f) - (a.center ? a.center[n] : 0.f);
      float sc = 1.f, sh = 0.f;
      if constexpr (EPI == 2) { sc = a.scale[n]; sh = a.shift[n]; }
      float s = 0.f, sq = 0.f;
      const float K = to_float(from_float<T>(acc[0][nt][0] + bias_v));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = acc[mt][nt][r] + bias_v;
          if constexpr (EPI == 2) {
            v = v * sc + sh;
            if (a.relu) v = fmaxf(v, 0.f);
          }
          const T tv = from_float<T>(v);
          *reinterpret_cast<T*>(wbuf + row * WP + (nt * 32 + l31) * (int)sizeof(T)) = tv;
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
  // wave-private region: LDS operations of one wave complete in issue order, no barrier needed
  float bsc[EPP], bsh[EPP], bmu[EPP], bis[EPP], bs1[EPP], bs2[EPP];      // EPI 3: this lane's EPP channels (fixed over the passes)
  if constexpr (EPI == 3) {
    const int c0 = ncol + (lane % EPR) * EPP;
#pragma unroll
    for (int k = 0; k < EPP; ++k) {
      bsc[k] = a.bn_ss[c0 + k]; bsh[k] = a.bn_ss[a.Co + c0 + k];
      bmu[k] = a.bn_mi[c0 + k]; bis[k] = a.bn_mi[a.Co + c0 + k];
      bs1[k] = 0.f; bs2[k] = 0.f;
    }
  }
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    const int row = pass * ROWS_PER_PASS + lane / EPR;
    const int piece = lane % EPR;
    const uint4 v = *reinterpret_cast<const uint4*>(wbuf + row * WP + piece * 16);
    const int m = wm * WROWS + row;
    const int bb = b0 + m / MI, yy = y0 + (m % MI) / TW, xx = x0 + m % TW;
    if (bb < a.B && yy < a.H && xx < a.W) {
      const size_t off = (((size_t)bb * a.H + yy) * a.W + xx) * ystride + piece * EPP;
      // streamed: the tile is not read again by this kernel and is far larger than L2 (+2 % over the 13 BASELINE layers)
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(yg + off));
      if constexpr (EPI == 3) {
        float g[EPP], zz[EPP];
        Vec16<T>::load(reinterpret_cast<const T*>(&v), g);
        Vec16<T>::load(reinterpret_cast<const T*>(&zreg[pass]), zz);
#pragma unroll
        for (int k = 0; k < EPP; ++k) {
          const float gg = (zz[k] * bsc[k] + bsh[k] > 0.f) ? g[k] : 0.f;      // same test as bn_relu_bwd_reduce_kernel
          bs1[k] += gg;
          bs2[k] += gg * ((zz[k] - bmu[k]) * bis[k]);
        }
      }
    }
  }
  if constexpr (EPI == 3) {
    // lanes piece, piece + EPR, ... hold the same channels: butterfly over them, then the WM waves through LDS
#pragma unroll
    for (int off = EPR; off < 64; off <<= 1) {
#pragma unroll
      for (int k = 0; k < EPP; ++k) { bs1[k] += __shfl_xor(bs1[k], off, 64); bs2[k] += __shfl_xor(bs2[k], off, 64); }
    }
    __syncthreads();                                         // the scratch aliases wave 0's tile
    if (lane < EPR) {
#pragma unroll
      for (int k = 0; k < EPP; ++k) {
        const int nl = wn * WCOLS + lane * EPP + k;
        ldsS[(wm * BN + nl) * 2 + 0] = bs1[k];
        ldsS[(wm * BN + nl) * 2 + 1] = bs2[k];
      }
    }
    __syncthreads();
    if (tid < BN) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < WM; ++i) { s1 += ldsS[(i * BN + tid) * 2 + 0]; s2 += ldsS[(i * BN + tid) * 2 + 1]; }
      float* pr = a.bn_partial + (size_t)tile_id * 2 * a.Co;
      pr[n0 + tid] = s1;
      pr[a.Co + n0 + tid] = s2;
    }
  }
  if (want_stats) {
    __syncthreads();                                         // the stats scratch aliases wave 0's tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nl = (wn * NT + nt) * 32 + l31;
      // merge the two half-waves (rows r and r+4 of every 8), then the WM waves that share this channel
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
  auto gload = [&](int t, Stage& R) __attribute__((always_inline)) {
    int tt = t;
    const int tx_id = tt % a.tilesX; tt /= a.tilesX;
    const int ty_id = tt % a.tilesY;
    const int b = tt / a.tilesY;
    const int y0 = ty_id * TH, x0 = tx_id * TW;
    const T* xb = xg + (size_t)b * a.H * a.W * xs.stride;
    const T* dzb = dzg + (size_t)b * a.H * a.W * a.Co + co0;
#if IM2IM_WGRAD_ABL & 16
    if (abl_first)
#endif
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (a_px[i] >= 0) {
        const int yy = y0 + a_px[i] / TW, xx = x0 + a_px[i] % TW;
        if (yy < a.H && xx < a.W && co0 + a_part[i] * EPP < a.Co)
          v = *reinterpret_cast<const uint4*>(dzb + ((size_t)yy * a.W + xx) * a.Co + a_part[i] * EPP);
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
          v = *reinterpret_cast<const uint4*>(xb + ((size_t)yy * a.W + xx) * xs.stride + b_part[i] * EPP);
          R.valid |= 1u << i;
        }
      }
      R.b[i] = v;
    }
  };
  auto swrite = [&](int buf, const Stage& R) __attribute__((always_inline)) {
    char* la = smem + buf * BUF_BYTES;
    char* lb = la + A_BYTES;
#if IM2IM_WGRAD_ABL & 16
    if (abl_first)
#endif
#pragma unroll
    for (int i = 0; i < A_ROUNDS; ++i)
      if (a_px[i] >= 0) *reinterpret_cast<uint4*>(la + a_px[i] * PA + (SWZ ? (a_part[i] ^ (((a_px[i] >> 1) & 1) << 2)) : a_part[i]) * 16) = R.a[i];
#if IM2IM_WGRAD_ABL & 16
    abl_first = false;
#endif
#pragma unroll
    for (int i = 0; i < B_ROUNDS; ++i) {
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
  auto compute = [&](int buf) __attribute__((always_inline)) {
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
    for (int ks = 0; ks < KSTEPS; ++ks) {
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
      compute(cur);
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

template <typename T, int TB, int TH, int TW, int BN, int WM, int WN, int TAPS, int EPI>
int launch_conv_epi(const ConvArgs& a_in, hipStream_t stream) {
  ConvArgs a = a_in;
  a.tilesY = (int)cdiv(a.H, TH);
  a.tilesX = (int)cdiv(a.W, TW);
  constexpr int PAD = (TAPS == 9) ? 1 : 0;
  constexpr int ROWB = 32 * (int)sizeof(T) + 16;
  constexpr size_t smem_main = (size_t)TB * (TH + 2 * PAD) * ((TW + 2 * PAD) * ROWB + (sizeof(T) == 2 ? 96 : 0)) + (size_t)2 * BN * ROWB;
  constexpr size_t smem_epi = (size_t)4 * (TB * TH * TW / WM) * ((BN / WN) * sizeof(T) + 16);   // 4 wave-private output tiles
  const size_t smem_in = smem_main + ((a.in_ss || a.in_ss_hi) ? (size_t)2 * a.Ci * sizeof(float) : 0);
  const size_t smem = smem_in > smem_epi ? smem_in : smem_epi;
  static_assert(smem_epi >= (size_t)WM * BN * 3 * 4, "stats scratch fits");
  auto kern = conv_igemm_kernel<T, TB, TH, TW, BN, WM, WN, TAPS, EPI>;
  static size_t attr_set = 0;
  if (smem > 64 * 1024 && smem > attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = smem;
  }
#if IM2IM_IGEMM_XCD_BANDS && IM2IM_IGEMM_COB_INNER
  dim3 grid((unsigned)((size_t)cdiv(a.B, TB) * a.tilesY * a.tilesX * (a.Co / BN)));
#else
  dim3 grid((unsigned)((size_t)cdiv(a.B, TB) * a.tilesY * a.tilesX), (unsigned)(a.Co / BN));
#endif
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a);
  return check_launch("conv_igemm_kernel");
}

template <typename T, int TB, int TH, int TW, int BN, int WM, int WN, int TAPS>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
  if (a.bn_partial) return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 3>(a, stream);
  if (a.stats) return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 1>(a, stream);
  if (a.scale) return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 2>(a, stream);
  return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 0>(a, stream);
}

template <typename T, int TAPS>
int dispatch_conv(const ConvArgs& a, hipStream_t stream, bool per_image = false) {
  if constexpr (std::is_same<T, bf16_t>::value && TAPS == 9) {
    if (!per_image) {                                        // the 8-wave ping-pong kernel takes the shapes it covers
      const int rc = launch_conv_pp(a, stream);
      if (rc != 1) return rc;
    }
  }
  const TileChoice t = pick_tile(a.B, a.H, a.W, a.Co, per_image);
  if (t.tb == 1) {
    if (t.bn == 128) return launch_conv<T, 1, 16, 16, 128, 2, 2, TAPS>(a, stream);
    if (t.bn == 64 && t.th == 32) return launch_conv<T, 1, 32, 16, 64, 4, 1, TAPS>(a, stream);
    if (t.bn == 64) return launch_conv<T, 1, 16, 16, 64, 4, 1, TAPS>(a, stream);
    return launch_conv<T, 1, 16, 16, 32, 4, 1, TAPS>(a, stream);
  }
  if (t.tb == 2) {
    if (t.bn == 128) return launch_conv<T, 2, 8, 8, 128, 2, 2, TAPS>(a, stream);
    if (t.bn == 64) return launch_conv<T, 2, 8, 8, 64, 4, 1, TAPS>(a, stream);
    return launch_conv<T, 2, 8, 8, 32, 4, 1, TAPS>(a, stream);
  }
  if (t.bn == 128) return launch_conv<T, 4, 8, 8, 128, 2, 2, TAPS>(a, stream);
  if (t.bn == 64) return launch_conv<T, 4, 8, 8, 64, 4, 1, TAPS>(a, stream);
  return launch_conv<T, 4, 8, 8, 32, 4, 1, TAPS>(a, stream);
}

}  // namespace

extern "C" int64_t im2im_conv_stats_rows(int32_t B, int32_t H, int32_t W, int32_t Co) {
  const TileChoice t = pick_tile(B, H, W, Co);
  return im2im::cdiv(B, t.tb) * im2im::cdiv(H, t.th) * im2im::cdiv(W, t.tw);
}

extern "C" int64_t im2im_conv_tiles_per_image(int32_t H, int32_t W) { return im2im::cdiv(H, 16) * im2im::cdiv(W, 16); }

extern "C" int im2im_conv_fwd(const void* x, const float* in_scale_shift, const void* w, const float* bias, const float* center,
                              const float* scale, const float* shift, void* y, float* stats, int32_t B, int32_t H, int32_t W,
                              int32_t Ci, int32_t Co, int32_t taps, int32_t relu, int32_t dtype, im2im_stream_t stream_) {
  return im2im_conv_fwd_split(x, in_scale_shift, nullptr, nullptr, Ci, w, bias, center, scale, shift, y, nullptr, Co, stats, B, H, W,
                              Ci, Co, taps, relu, dtype, stream_);
}

namespace {
int conv_fwd_impl(const void* x, const float* in_scale_shift, int in_ss_img, bool per_image, const void* x_hi,
                  const float* in_scale_shift_hi, int32_t Ci_lo, const void* w, const float* bias, const float* center,
                  const float* scale, const float* shift, void* y, void* y_hi, int32_t Co_lo, float* stats, int32_t B, int32_t H,
                  int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t relu, int32_t dtype, im2im_stream_t stream_);
}

extern "C" int im2im_conv_fwd_per_image(const void* x, const float* in_scale_shift_per_image, const void* w, const float* bias,
                                        void* y, float* stats, int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co,
                                        int32_t taps, int32_t dtype, im2im_stream_t stream_) {
  return conv_fwd_impl(x, in_scale_shift_per_image, in_scale_shift_per_image ? 2 * Ci : 0, true, nullptr, nullptr, Ci, w, bias,
                       nullptr, nullptr, nullptr, y, nullptr, Co, stats, B, H, W, Ci, Co, taps, 0, dtype, stream_);
}

extern "C" int im2im_conv_fwd_split(const void* x, const float* in_scale_shift, const void* x_hi, const float* in_scale_shift_hi,
                                    int32_t Ci_lo, const void* w, const float* bias, const float* center, const float* scale,
                                    const float* shift, void* y, void* y_hi, int32_t Co_lo, float* stats, int32_t B, int32_t H,
                                    int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t relu, int32_t dtype,
                                    im2im_stream_t stream_) {
  return conv_fwd_impl(x, in_scale_shift, 0, false, x_hi, in_scale_shift_hi, Ci_lo, w, bias, center, scale, shift, y, y_hi, Co_lo,
                       stats, B, H, W, Ci, Co, taps, relu, dtype, stream_);
}

namespace {
int conv_fwd_impl(const void* x, const float* in_scale_shift, int in_ss_img, bool per_image, const void* x_hi,
                  const float* in_scale_shift_hi, int32_t Ci_lo, const void* w, const float* bias, const float* center,
                  const float* scale, const float* shift, void* y, void* y_hi, int32_t Co_lo, float* stats, int32_t B, int32_t H,
                  int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t relu, int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(x && w && y);
  if (x_hi) {
    IM2IM_REQUIRE(Ci_lo > 0 && Ci_lo % 32 == 0 && Ci == 2 * Ci_lo);   // both sources share one pixel stride
  } else {
    IM2IM_REQUIRE(in_scale_shift_hi == nullptr);
    Ci_lo = Ci;
  }
  if (y_hi) {
    IM2IM_REQUIRE(Co_lo > 0 && Co_lo % 64 == 0 && Co_lo < Co && (Co - Co_lo) % 64 == 0);   // a wave's 32/64 columns never straddle
    IM2IM_REQUIRE(stats == nullptr && scale == nullptr);
  } else {
    Co_lo = Co;
  }
  IM2IM_REQUIRE(B > 0 && H > 0 && W > 0);
  IM2IM_REQUIRE(Ci > 0 && Ci % 32 == 0);
  IM2IM_REQUIRE(Co > 0 && Co % 32 == 0);
  IM2IM_REQUIRE(taps == 9 || taps == 1);
  IM2IM_REQUIRE(dtype == IM2IM_F32 || dtype == IM2IM_BF16);
  IM2IM_REQUIRE((scale == nullptr) == (shift == nullptr));
  IM2IM_REQUIRE(!(stats && scale));                              // statistics describe the raw conv output
  IM2IM_REQUIRE(Ci <= 2048);
  IM2IM_REQUIRE(in_ss_img == 0 || (per_image && x_hi == nullptr));   // per-image coefficients need one image per tile
  ConvArgs a{x, w, bias, scale, shift, y, stats, B, H, W, Ci, Co, 0, 0, relu, center, in_scale_shift,
             x_hi, in_scale_shift_hi, Ci_lo, y_hi, Co_lo, nullptr, nullptr, nullptr, nullptr, in_ss_img};
  if (dtype == IM2IM_BF16) return taps == 9 ? dispatch_conv<bf16_t, 9>(a, stream, per_image) : dispatch_conv<bf16_t, 1>(a, stream, per_image);
  return taps == 9 ? dispatch_conv<float, 9>(a, stream, per_image) : dispatch_conv<float, 1>(a, stream, per_image);
}
}  // namespace

extern "C" int im2im_conv_dgrad_bn(const void* dz, const void* wd, void* dx, const void* bn_z, const float* bn_scale_shift,
                                   const float* bn_mean_invstd, float* bn_partial, int32_t B, int32_t H, int32_t W, int32_t Ci,
                                   int32_t Co, int32_t taps, int32_t dtype, im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(dz && wd && dx && bn_z && bn_scale_shift && bn_mean_invstd && bn_partial);
  IM2IM_REQUIRE(B > 0 && H > 0 && W > 0);
  IM2IM_REQUIRE(Ci > 0 && Ci % 32 == 0 && Ci <= 2048);
  IM2IM_REQUIRE(Co > 0 && Co % 32 == 0);
  IM2IM_REQUIRE(taps == 9 || taps == 1);
  IM2IM_REQUIRE(dtype == IM2IM_F32 || dtype == IM2IM_BF16);
  ConvArgs a{dz, wd, nullptr, nullptr, nullptr, dx, nullptr, B, H, W, Ci, Co, 0, 0, 0, nullptr, nullptr,
             nullptr, nullptr, Ci, nullptr, Co, bn_z, bn_scale_shift, bn_mean_invstd, bn_partial, 0};
  if (dtype == IM2IM_BF16) return taps == 9 ? dispatch_conv<bf16_t, 9>(a, stream) : dispatch_conv<bf16_t, 1>(a, stream);
  return taps == 9 ? dispatch_conv<float, 9>(a, stream) : dispatch_conv<float, 1>(a, stream);
}

namespace {
int g_wgrad_co128 = 1;      // A/B switch (im2im_set_option "wgrad_co128")
int g_wgrad_tile16 = 1;     // A/B switch "wgrad_tile16": 256-pixel tiles for the 64-output-channel form
template <typename T, int TAPS>
int launch_wgrad(const void* x, const float* x_ss, const void* x_hi, const float* x_ss_hi, int Ci_lo, const void* dz, float* partial,
                 int64_t partial_bytes, float* dw, int B, int H, int W, int Ci, int Co, hipStream_t stream) {
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
  int64_t nsplit = cdiv((IS_BF16 && TAPS == 9) ? 256 : (TAPS == 1 ? 1536 : 512), cblocks);   // pipelined kernel: one workgroup per CU; 1x1: latency-bound, many small blocks
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
      nsplit = cdiv(256, cb128);
      if (nsplit > a.ntiles) nsplit = a.ntiles;
      if (nsplit > max_split) nsplit = max_split;
      if (nsplit < 1) nsplit = 1;
      a.tiles_per_split = (int)cdiv(a.ntiles, nsplit);
      nsplit = cdiv(a.ntiles, a.tiles_per_split);
      constexpr size_t smem128 = 2 * ((size_t)TH * TW * (128 * 2 + 64) + (size_t)(TH + 2) * (TW + 2) * 192) + 512;
      auto kern = conv_wgrad_pipe_kernel<TH, TW, 128>;
      static bool attr_set = false;
      if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem128);
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
      auto kern = conv_wgrad_pipe_kernel<16, 16, 64>;
      static bool attr_set = false;
      if (!attr_set) {
        hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16);
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
                                dtype, stream_);
}

extern "C" int im2im_conv_wgrad_split(const void* x, const float* x_scale_shift, const void* x_hi, const float* x_scale_shift_hi,
                                      int32_t Ci_lo, const void* dz, float* dw, void* workspace, int64_t workspace_bytes,
                                      int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t dtype,
                                      im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(x && dz && dw && workspace);
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
    return taps == 9 ? launch_wgrad<bf16_t, 9>(x, x_scale_shift, x_hi, x_scale_shift_hi, Ci_lo, dz, partial, workspace_bytes, dw, B, H, W, Ci, Co, stream)
                     : launch_wgrad<bf16_t, 1>(x, x_scale_shift, x_hi, x_scale_shift_hi, Ci_lo, dz, partial, workspace_bytes, dw, B, H, W, Ci, Co, stream);
  return taps == 9 ? launch_wgrad<float, 9>(x, x_scale_shift, x_hi, x_scale_shift_hi, Ci_lo, dz, partial, workspace_bytes, dw, B, H, W, Ci, Co, stream)
                   : launch_wgrad<float, 1>(x, x_scale_shift, x_hi, x_scale_shift_hi, Ci_lo, dz, partial, workspace_bytes, dw, B, H, W, Ci, Co, stream);
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

// run-time switches for within-process A/B measurements (tools/, bench): "conv_pp" = bit 0: ping-pong kernel for the
// 128-wide tiles, bit 1: for the 64-wide tiles.  Not a reference interface.
extern "C" int im2im_set_option(const char* key, int32_t value) {
  IM2IM_REQUIRE(key != nullptr);
  if (std::string(key) == "conv_pp") { im2im::set_conv_pp_mode(value); return IM2IM_OK; }
  if (std::string(key) == "wgrad_co128") { g_wgrad_co128 = value; return IM2IM_OK; }
  if (std::string(key) == "wgrad_tile16") { g_wgrad_tile16 = value; return IM2IM_OK; }
  return im2im::fail_invalid("unknown option");
}

extern "C" int im2im_pack_conv_weights_multi(int32_t n_tensors, const float* const* w, const int32_t* Co, const int32_t* Ci,
                                             const int32_t* taps, int32_t dtype, void* const* wf, void* const* wd,
                                             im2im_stream_t stream_) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(n_tensors >= 0 && (n_tensors == 0 || (w && Co && Ci && taps && wf && wd)));
  IM2IM_REQUIRE(dtype == IM2IM_F32 || dtype == IM2IM_BF16);
  for (int base = 0; base < n_tensors; base += PACK_MAX_TENSORS) {
    PackMultiArgs a;
    a.n = std::min(PACK_MAX_TENSORS, n_tensors - base);
    int chunks = 0;
    for (int i = 0; i < a.n; ++i) {
      IM2IM_REQUIRE(w[base + i] && wf[base + i] && Co[base + i] > 0 && Ci[base + i] > 0 && taps[base + i] > 0);
      a.w[i] = w[base + i]; a.wf[i] = wf[base + i]; a.wd[i] = wd[base + i];
      a.Co[i] = Co[base + i]; a.Ci[i] = Ci[base + i]; a.taps[i] = taps[base + i];
      a.start[i] = chunks;
      chunks += (int)im2im::cdiv((int64_t)Co[base + i] * Ci[base + i] * taps[base + i], 1024);
    }
    a.start[a.n] = chunks;
    if (chunks == 0) continue;
    if (dtype == IM2IM_BF16) hipLaunchKernelGGL(pack_weight_multi_kernel<bf16_t>, dim3((unsigned)chunks), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(pack_weight_multi_kernel<float>, dim3((unsigned)chunks), dim3(256), 0, stream, a);
    if (int rc = im2im::check_launch("pack_weight_multi_kernel")) return rc;
  }
  return IM2IM_OK;
}
