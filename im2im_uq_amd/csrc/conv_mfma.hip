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
#include <vector>
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
//      4 = split-K partial: this workgroup reduces only the input-channel chunks [ksp*kchunks, (ksp+1)*kchunks) and stores its raw
//      fp32 accumulators to kpartial[ksp][pixel][Co]; conv_splitk_reduce_kernel adds the splits in order, adds the bias, rounds,
//      stores and takes the BatchNorm statistics.  For launches that would leave most of the chip empty (a strong-scaled job's
//      per-GPU batch of ~10 at the 40x40 / 20x20 levels: 180-250 workgroups of one wave per SIMD with K = 4,608-9,216).
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
  int ksp = 0;                                              // EPI 4: which K split this workgroup reduces
  {
    const int ncob = a.Co / BN;
    int lin = blockIdx.x, total = (int)gridDim.x;
    if constexpr (EPI == 4) { total /= a.ksplit; ksp = lin / total; lin -= ksp * total; }
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
  const int c_begin = (EPI == 4) ? ksp * a.kchunks : 0;
  const int c_end = (EPI == 4) ? min(c_begin + a.kchunks, nchunks) : nchunks;
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
    gload_A(c_begin);
    gload_F(fw[0], c_begin, 0);
    auto chunk_body = [&](auto parity, int chunk) __attribute__((always_inline)) {
      constexpr int P0 = decltype(parity)::value;        // 9 taps per chunk: the register-set parity of a chunk's first tap alternates
      const bool more = chunk + 1 < c_end;
      if (chunk != c_begin) __syncthreads();           // everyone done reading the previous halo
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
    for (int chunk = c_begin; chunk < c_end; chunk += 2) {
      chunk_body(std::integral_constant<int, 0>{}, chunk);
      if (chunk + 1 < c_end) chunk_body(std::integral_constant<int, 1>{}, chunk + 1);
    }
  } else if constexpr (TAPS == 9) {
    // 9 taps fully unrolled (every register-set / LDS-buffer index and tap offset is a compile-time constant).
    // Weight tiles are prefetched TWO iterations ahead: iteration `it` writes register set it&1 (loaded at it-2)
    // to LDS buffer it&1 and immediately re-issues that set for it+2.  9 is odd, so the parity of a chunk's first
    // tap alternates: the body is instantiated for both parities.  The next chunk's halo is fetched 3 taps early.
    gload_A(c_begin);
    gload_B(rb[0], c_begin, 0);
    gload_B(rb[1], c_begin, 1);
    auto chunk_body = [&](auto parity, int chunk) {
      constexpr int P0 = decltype(parity)::value;
      const bool more = chunk + 1 < c_end;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        constexpr int dummy = 0; (void)dummy;
        const int set = (P0 + tap) & 1;                // compile-time after unrolling
        if (tap == 0) {
          if (chunk != c_begin) __syncthreads();     // everyone done reading the previous halo
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
    for (int chunk = c_begin; chunk < c_end; chunk += 2) {
      chunk_body(std::integral_constant<int, 0>{}, chunk);
      if (chunk + 1 < c_end) chunk_body(std::integral_constant<int, 1>{}, chunk + 1);
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
  if constexpr (EPI == 4) {
    // split-K partial: raw fp32 accumulators, one channel per lane -> 32 lanes write 128 contiguous bytes of a pixel row
    float* __restrict__ kp = a.kpartial + (size_t)ksp * a.B * a.H * a.W * a.Co + n0;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (wm * MT + mt) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        const int bb = b0 + m / MI, yy = y0 + (m % MI) / TW, xx = x0 + m % TW;
        if (bb < a.B && yy < a.H && xx < a.W) {
          float* row = kp + (((size_t)bb * a.H + yy) * a.W + xx) * a.Co;
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) row[(wn * NT + nt) * 32 + l31] = acc[mt][nt][r];
        }
      }
    return;
  }
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
      if constexpr (EPI == 2 || EPI == 5) { sc = a.scale[n]; sh = a.shift[n]; }
      float s = 0.f, sq = 0.f;
      const float K = to_float(from_float<T>(acc[0][nt][0] + bias_v));
#pragma unroll
      for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          float v = acc[mt][nt][r] + bias_v;
          if constexpr (EPI == 2 || EPI == 5) {
            v = v * sc + sh;
            if (a.relu) v = fmaxf(v, 0.f);
          }
          const T tv = from_float<T>(v);
          *reinterpret_cast<T*>(wbuf + row * WP + (nt * 32 + l31) * (int)sizeof(T)) = tv;
          if constexpr (want_stats) {
            float d;
            // fp32: the compiler paired these subtractions into v_pk_add_f32 op_sel:[0,1] (K broadcast from the high register of a pair),
            // a form whose low lane is not reliable beside other processes (profiles/r06_multiprocess_determinism.txt): one scalar v_sub each
            if constexpr (sizeof(T) == 4) asm("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(to_float(tv)), "v"(K));
            else d = to_float(tv) - K;
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
  if constexpr (EPI == 5) {
    // [r4] eval mode, last block of the trunk: its 64-channel result is consumed by OutConv's 1x1 convolution only
    // (unet.py:45-46).  The wave holds all 64 channels of its pixels in its LDS tile (rounded to the storage type, as the
    // separate kernel would read them back): the 1x1 conv is 4 k-steps of MFMAs per 32 pixels from that tile, in the order
    // conv_igemm<taps = 1> takes them (two 32-channel chunks), + bias, one rounding -- same bits -- and only the C1-channel
    // feature map reaches HBM: the 64-channel tensor is neither written nor read back, and the 1x1 launch disappears.
    static_assert(EPI != 5 || (NT == 2 && WN == 1), "the wave owns all 64 output channels");
    constexpr int C1 = 32;
    const char* w1 = reinterpret_cast<const char*>(a.fuse_w) + (size_t)l31 * 64 * sizeof(T);
    f32x16 acc2[MT];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc2[mt][r] = 0.f;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int ks = 0; ks < Frag<T>::KSTEPS; ++ks) {
        const typename Frag<T>::AB fb = Frag<T>::load(w1 + c * 32 * (int)sizeof(T), ks, half);
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
          const typename Frag<T>::AB fa = Frag<T>::load(wbuf + (mt * 32 + l31) * WP + c * 32 * (int)sizeof(T), ks, half);
          acc2[mt] = Frag<T>::mfma(fa, fb, acc2[mt]);
        }
      }
    // every fragment read above has returned before the MFMA that used it; LDS operations of one wave complete in issue order, so
    // the same region can take the C1-channel tile now: [pixel][C1], pitch + 16 B
    constexpr int WP2 = C1 * (int)sizeof(T) + 16;
    const float b1 = a.fuse_bias ? a.fuse_bias[l31] : 0.f;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        *reinterpret_cast<T*>(wbuf + row * WP2 + l31 * (int)sizeof(T)) = from_float<T>(acc2[mt][r] + b1);
      }
    constexpr int EPR2 = C1 / EPP, RPP2 = 64 / EPR2;
    T* __restrict__ fy = reinterpret_cast<T*>(a.fuse_y);
#pragma unroll
    for (int pass = 0; pass < WROWS / RPP2; ++pass) {
      const int row = pass * RPP2 + lane / EPR2, piece = lane % EPR2;
      const uint4 v = *reinterpret_cast<const uint4*>(wbuf + row * WP2 + piece * 16);
      const int m = wm * WROWS + row;
      const int bb = b0 + m / MI, yy = y0 + (m % MI) / TW, xx = x0 + m % TW;
      if (bb < a.B && yy < a.H && xx < a.W)
        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(fy + (((size_t)bb * a.H + yy) * a.W + xx) * C1 + piece * EPP));
    }
    return;
  }
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
      if constexpr (EPI == 2) {
        // [r4] eval mode, a block whose output also feeds MaxPool2d(2) (unet_parts.py:33-36): the 2x2 maximum comes from this wave's
        // LDS tile (the window's other three pixels are rows +1, +TW, +TW+1 of the same tile: tiles start on even pixels and a
        // wave owns an even number of whole tile rows), so the separate pooling pass never re-reads the activation from HBM
        if (a.pool_y) {
          const int yi = (m % MI) / TW, xi = m % TW;
          if (!((yi | xi) & 1) && yy + 1 < a.H && xx + 1 < a.W) {
            float p00[EPP], p01[EPP], p10[EPP], p11[EPP];
            Vec16<T>::load(reinterpret_cast<const T*>(&v), p00);
            Vec16<T>::load(reinterpret_cast<const T*>(wbuf + (row + 1) * WP + piece * 16), p01);
            Vec16<T>::load(reinterpret_cast<const T*>(wbuf + (row + TW) * WP + piece * 16), p10);
            Vec16<T>::load(reinterpret_cast<const T*>(wbuf + (row + TW + 1) * WP + piece * 16), p11);
#pragma unroll
            for (int k = 0; k < EPP; ++k) p00[k] = fmaxf(fmaxf(p00[k], p01[k]), fmaxf(p10[k], p11[k]));      // maxpool2_fwd_kernel's order
            T* py = reinterpret_cast<T*>(a.pool_y) + ((((size_t)bb * (a.H >> 1) + (yy >> 1)) * (a.W >> 1) + (xx >> 1)) * a.Co) + ncol + piece * EPP;
            Vec16<T>::store_nt(py, p00);
          }
        }
      }
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

// Second half of a split-K convolution: y = round(sum over splits (in order) + bias), plus the per-tile BatchNorm partial
// statistics the one-kernel epilogue would have produced (one row per conv tile: mean, M2, count of the STORED values).
// Block = one conv tile's pixels x 64 output channels; thread = 4 channels (one 16-byte load per split) x one of 16 pixel
// lanes; the lanes' moments are merged through LDS in a fixed order.  Deterministic, no atomics.
template <typename T, bool STATS>
__global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(ConvArgs a, int TB, int TH, int TW) {
  __shared__ float s_st[16][64][3];
  const int cv = threadIdx.x & 15, pl = threadIdx.x >> 4;
  const int c0 = blockIdx.y * 64 + cv * 4;
  int t = blockIdx.x;
  const int tx = t % a.tilesX; t /= a.tilesX;
  const int ty = t % a.tilesY;
  const int b0 = (t / a.tilesY) * TB, y0 = ty * TH, x0 = tx * TW;
  const int MI = TH * TW, M = TB * MI;
  const size_t slab = (size_t)a.B * a.H * a.W * a.Co;
  float bias[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) bias[k] = (a.bias ? a.bias[c0 + k] : 0.f) - (a.center ? a.center[c0 + k] : 0.f);
  const bool to_hi = a.y_hi != nullptr && c0 >= a.Co_lo;
  T* __restrict__ yg = reinterpret_cast<T*>(to_hi ? a.y_hi : a.y) + (to_hi ? c0 - a.Co_lo : c0);
  const int ystride = a.y_hi == nullptr ? a.Co : (to_hi ? a.Co - a.Co_lo : a.Co_lo);
  float n = 0.f, K[4] = {0.f, 0.f, 0.f, 0.f}, sm[4] = {0.f, 0.f, 0.f, 0.f}, sq[4] = {0.f, 0.f, 0.f, 0.f};
  for (int m = pl; m < M; m += 16) {
    const int bb = b0 + m / MI, yy = y0 + (m % MI) / TW, xx = x0 + m % TW;
    if (bb >= a.B || yy >= a.H || xx >= a.W) continue;
    const size_t pix = ((size_t)bb * a.H + yy) * a.W + xx;
    const float* __restrict__ src = a.kpartial + pix * a.Co + c0;
    float4 v = *reinterpret_cast<const float4*>(src);
    for (int sp = 1; sp < a.ksplit; ++sp) {
      const float4 u = *reinterpret_cast<const float4*>(src + sp * slab);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    float o[4] = {v.x + bias[0], v.y + bias[1], v.z + bias[2], v.w + bias[3]};
    T tv[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) tv[k] = from_float<T>(o[k]);
    if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(yg + pix * ystride) = *reinterpret_cast<const uint2*>(tv);
    else *reinterpret_cast<uint4*>(yg + pix * ystride) = *reinterpret_cast<const uint4*>(tv);
    if constexpr (STATS) {
      if (n == 0.f) {
#pragma unroll
        for (int k = 0; k < 4; ++k) K[k] = to_float(tv[k]);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { const float d = to_float(tv[k]) - K[k]; sm[k] += d; sq[k] += d * d; }
      n += 1.f;
    }
  }
  if constexpr (STATS) {
    const float inv = n > 0.f ? 1.f / n : 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      s_st[pl][cv * 4 + k][0] = n;
      s_st[pl][cv * 4 + k][1] = K[k] + sm[k] * inv;
      s_st[pl][cv * 4 + k][2] = fmaxf(sq[k] - sm[k] * sm[k] * inv, 0.f);
    }
    __syncthreads();
    if (threadIdx.x < 64) {
      float nn = s_st[0][threadIdx.x][0], mm = s_st[0][threadIdx.x][1], qq = s_st[0][threadIdx.x][2];
#pragma unroll
      for (int i = 1; i < 16; ++i) merge_moments_f32(nn, mm, qq, s_st[i][threadIdx.x][0], s_st[i][threadIdx.x][1], s_st[i][threadIdx.x][2]);
      float* st = a.stats + (size_t)blockIdx.x * 3 * a.Co + blockIdx.y * 64 + threadIdx.x;
      st[0] = mm; st[a.Co] = qq; st[2 * a.Co] = nn;
    }
  }
}

int g_conv_splitk = 3;      // A/B switch (im2im_set_option "conv_splitk"): 0 = never split, n = aim at n * 256 workgroups
// K splits for a launch of `wgs` workgroups reducing `nchunks` 32-channel chunks: only launches that leave most of the chip's
// 512 workgroup slots empty AND carry a long reduction (K = 9*Ci >= 4,608) -- the fp32 partial sums cost 8*ksplit bytes per
// output element, which a short reduction does not pay back
inline int splitk_choice(long wgs, int nchunks) {
  if (g_conv_splitk <= 0 || wgs >= 384 || nchunks < 16) return 1;
  long ks = (256L * g_conv_splitk + wgs / 2) / wgs;
  if (ks > nchunks / 4) ks = nchunks / 4;
  if (ks > 8) ks = 8;
  return ks < 2 ? 1 : (int)ks;
}

// bytes of fp32 partial sums a (B,H,W,Ci,Co) launch would use when split, 0 when it is never split
inline int64_t conv_splitk_bytes(int B, int H, int W, int Ci, int Co, int taps, bool per_image) {
  if (taps != 9 || per_image || Ci % 32 || Co % 64) return 0;
  const TileChoice t = pick_tile(B, H, W, Co, per_image);
  if (t.tb < 2 || t.bn < 64) return 0;
  const long wgs = (long)cdiv(B, t.tb) * cdiv(H, t.th) * cdiv(W, t.tw) * (Co / t.bn);
  const int ks = splitk_choice(wgs, Ci / 32);
  return ks > 1 ? (int64_t)ks * B * H * W * Co * (int64_t)sizeof(float) : 0;
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
  const size_t smem = (EPI == 4 || smem_in > smem_epi) ? smem_in : smem_epi;      // the split-K partial store needs no LDS
  static_assert(smem_epi >= (size_t)WM * BN * 3 * 4, "stats scratch fits");
  auto kern = conv_igemm_kernel<T, TB, TH, TW, BN, WM, WN, TAPS, EPI>;
  static size_t attr_set = 0;
  if (smem > 64 * 1024 && smem > attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = smem;
  }
#if IM2IM_IGEMM_XCD_BANDS && IM2IM_IGEMM_COB_INNER
  dim3 grid((unsigned)((size_t)cdiv(a.B, TB) * a.tilesY * a.tilesX * (a.Co / BN) * (EPI == 4 ? a.ksplit : 1)));
#else
  static_assert(EPI != 4, "split-K needs the 1-D grid");
  dim3 grid((unsigned)((size_t)cdiv(a.B, TB) * a.tilesY * a.tilesX), (unsigned)(a.Co / BN));
#endif
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, stream, a);
  if (int rc = check_launch("conv_igemm_kernel")) return rc;
  if constexpr (EPI == 4) {
    const dim3 rgrid((unsigned)((size_t)cdiv(a.B, TB) * a.tilesY * a.tilesX), (unsigned)(a.Co / 64));
    if (a.stats) hipLaunchKernelGGL((conv_splitk_reduce_kernel<T, true>), rgrid, dim3(256), 0, stream, a, TB, TH, TW);
    else hipLaunchKernelGGL((conv_splitk_reduce_kernel<T, false>), rgrid, dim3(256), 0, stream, a, TB, TH, TW);
    return check_launch("conv_splitk_reduce_kernel");
  }
  return IM2IM_OK;
}

template <typename T, int TB, int TH, int TW, int BN, int WM, int WN, int TAPS>
int launch_conv(const ConvArgs& a, hipStream_t stream) {
  if constexpr (TAPS == 9 && TB >= 2 && BN >= 64) {
    if (a.ksplit > 1) return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 4>(a, stream);
  }
  if constexpr (TAPS == 9 && BN == 64 && WN == 1) {
    if (a.fuse_y) return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 5>(a, stream);
  }
  if (a.fuse_y) return fail_invalid("fused 1x1 tail: 64 output channels only");
  if (a.bn_partial) return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 3>(a, stream);
  if (a.stats) return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 1>(a, stream);
  if (a.scale) return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 2>(a, stream);
  return launch_conv_epi<T, TB, TH, TW, BN, WM, WN, TAPS, 0>(a, stream);
}

template <typename T, int TAPS>
int dispatch_conv(const ConvArgs& a_in, hipStream_t stream, bool per_image = false) {
  ConvArgs a = a_in;
  const TileChoice t = pick_tile(a.B, a.H, a.W, a.Co, per_image);
  const int64_t need = conv_splitk_bytes(a.B, a.H, a.W, a.Ci, a.Co, TAPS, per_image);
  if (need > 0 && a.kpartial && a.kws_bytes >= need && !a.scale && !a.bn_partial) {
    const long wgs = (long)cdiv(a.B, t.tb) * cdiv(a.H, t.th) * cdiv(a.W, t.tw) * (a.Co / t.bn);
    a.ksplit = splitk_choice(wgs, a.Ci / 32);
    a.kchunks = (int)cdiv(a.Ci / 32, a.ksplit);
    a.ksplit = (int)cdiv(a.Ci / 32, a.kchunks);
  } else {
    a.ksplit = 1;
  }
#ifdef IM2IM_BUILD_EXPERIMENTAL
  if constexpr (sizeof(T) == 2) {
    if (conv_roll64_eligible(a, t, TAPS, per_image)) return launch_conv_roll64(a, stream);      // [r5] conv_roll.hip (build.py EXPERIMENTAL)
  }
#endif
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
                  int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t relu, int32_t dtype, im2im_stream_t stream_,
                  void* ws = nullptr, int64_t ws_bytes = 0, void* pool_y = nullptr, const void* fuse_w = nullptr,
                  const float* fuse_bias = nullptr, void* fuse_y = nullptr);
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

extern "C" int64_t im2im_conv_splitk_workspace_bytes(int32_t B, int32_t H, int32_t W, int32_t Ci, int32_t Co, int32_t taps) {
  if (B <= 0 || H <= 0 || W <= 0 || Ci <= 0 || Co <= 0) return 0;
  return conv_splitk_bytes(B, H, W, Ci, Co, taps, false);
}

extern "C" int im2im_conv_fwd_split_ws(const void* x, const float* in_scale_shift, const void* x_hi, const float* in_scale_shift_hi,
                                       int32_t Ci_lo, const void* w, const float* bias, const float* center, const float* scale,
                                       const float* shift, void* y, void* y_hi, int32_t Co_lo, float* stats, int32_t B, int32_t H,
                                       int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t relu, int32_t dtype,
                                       void* workspace, int64_t workspace_bytes, im2im_stream_t stream_) {
  return conv_fwd_impl(x, in_scale_shift, 0, false, x_hi, in_scale_shift_hi, Ci_lo, w, bias, center, scale, shift, y, y_hi, Co_lo,
                       stats, B, H, W, Ci, Co, taps, relu, dtype, stream_, workspace, workspace_bytes);
}

extern "C" int im2im_conv_fwd_eval_pool(const void* x, const void* x_hi, int32_t Ci_lo, const void* w, const float* scale,
                                        const float* shift, void* y, void* pool_y, int32_t B, int32_t H, int32_t W, int32_t Ci,
                                        int32_t Co, int32_t dtype, im2im_stream_t stream_) {
  IM2IM_REQUIRE(pool_y != nullptr);
  return conv_fwd_impl(x, nullptr, 0, false, x_hi, nullptr, Ci_lo, w, nullptr, nullptr, scale, shift, y, nullptr, Co, nullptr, B, H, W,
                       Ci, Co, 9, 1, dtype, stream_, nullptr, 0, pool_y);
}

extern "C" int im2im_conv_fwd_eval_tail(const void* x, const void* x_hi, int32_t Ci_lo, const void* w, const float* scale,
                                        const float* shift, const void* w1, const float* b1, void* f_out, int32_t B, int32_t H,
                                        int32_t W, int32_t Ci, int32_t C1, int32_t dtype, im2im_stream_t stream_) {
  IM2IM_REQUIRE(w1 && f_out && C1 == 32);
  return conv_fwd_impl(x, nullptr, 0, false, x_hi, nullptr, Ci_lo, w, nullptr, nullptr, scale, shift, nullptr, nullptr, 64, nullptr, B, H, W,
                       Ci, 64, 9, 1, dtype, stream_, nullptr, 0, nullptr, w1, b1, f_out);
}

namespace im2im { void set_conv_splitk(int v) { g_conv_splitk = v; } }

namespace {
int conv_fwd_impl(const void* x, const float* in_scale_shift, int in_ss_img, bool per_image, const void* x_hi,
                  const float* in_scale_shift_hi, int32_t Ci_lo, const void* w, const float* bias, const float* center,
                  const float* scale, const float* shift, void* y, void* y_hi, int32_t Co_lo, float* stats, int32_t B, int32_t H,
                  int32_t W, int32_t Ci, int32_t Co, int32_t taps, int32_t relu, int32_t dtype, im2im_stream_t stream_,
                  void* ws, int64_t ws_bytes, void* pool_y, const void* fuse_w, const float* fuse_bias, void* fuse_y) {
  hipStream_t stream = (hipStream_t)stream_;
  IM2IM_REQUIRE(x && w && (y || fuse_y));
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
             x_hi, in_scale_shift_hi, Ci_lo, y_hi, Co_lo, nullptr, nullptr, nullptr, nullptr, in_ss_img,
             1, 0, reinterpret_cast<float*>(ws), ws ? ws_bytes : 0, pool_y, fuse_w, fuse_bias, fuse_y};
  if (fuse_y) IM2IM_REQUIRE(fuse_w && scale && relu && Co == 64 && taps == 9 && y_hi == nullptr && !per_image && pool_y == nullptr);
  if (pool_y) IM2IM_REQUIRE(scale && relu && y_hi == nullptr && H % 2 == 0 && W % 2 == 0 && !per_image);
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
             nullptr, nullptr, Ci, nullptr, Co, bn_z, bn_scale_shift, bn_mean_invstd, bn_partial, 0, 1, 0, nullptr, 0, nullptr, nullptr, nullptr, nullptr};
  if (dtype == IM2IM_BF16) return taps == 9 ? dispatch_conv<bf16_t, 9>(a, stream) : dispatch_conv<bf16_t, 1>(a, stream);
  return taps == 9 ? dispatch_conv<float, 9>(a, stream) : dispatch_conv<float, 1>(a, stream);
}
