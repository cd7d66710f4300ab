// [r5] Persistent, software-pipelined 3x3 convolution for the 64-output-channel full-resolution layers (bf16).
//
// Replaces conv_igemm_kernel<bf16, 1, 32, 16, 64, 4, 1, 9, EPI> (conv_mfma.hip) for the nn.Conv2d calls of the reference's first and
// last UNet levels (core/models/trunks/unet_parts.py:16,19 as instantiated at core/models/trunks/unet.py:20,29) and their
// data-gradients: 64 -> 64 and 128 -> 64 channels at 320 x 320 / 160 x 160, 40 % of the network's convolution FLOPs with
// K = 576 or 1,152 only.  That kernel ran them at 0.33 of the MFMA peak with the matrix pipe busy 37 % of the time
// (profiles/r04_pmc_mfma_busy.json): a workgroup lives for ONE tile -- halo from HBM, two 144-MFMA chunks, 64 KB of output --
// so a third of its life is the start-up latency and the store tail, hidden only by the one co-resident workgroup; and its
// weight fragments share the in-order vmcnt queue with the halo loads, so a weight wait behind a halo request is an HBM wait.
//
// Here the recipe of conv_wgrad_roll_kernel (conv_wgrad.hip), which runs the SAME shape at 0.64 pipe-busy:
//   * PERSISTENT: 2 workgroups per CU (4 waves, 128 accumulators each) walk a contiguous run of tiles; consecutive runs sit
//     on one XCD (neighbouring tiles share halo rows through its L2).
//   * UNIT = (tile, 16 input channels): 9 taps x 8 MFMAs per wave between two barriers.  BOTH operands of unit n come from LDS
//     buffer n & 1: the (32+2) x (16+2)-pixel halo (32 B per pixel, the two 16-byte halves XOR-swizzled by bit 3 of the halo
//     column) and the unit's 9 x 64 x 16 weights as eighteen 1 KiB MFMA fragments copied verbatim from the fragment-major pack
//     (conv_common.h wfrag_index: a fragment read is base + lane * 16, conflict-free by construction).
//   * STAGING SPREAD OVER THE MFMA PHASE: a thread owns 5 halo + 5 weight 16-byte pieces per unit.  After tap slot(p) of unit n
//     its piece p of unit n+1 -- requested at the same point of unit n-1, one whole unit period in flight -- is transformed
//     (lazy BatchNorm+ReLU) and written to the other buffer, and the request for piece p of unit n+2 follows at once.  Nothing
//     a wave consumes comes straight from a global load, so no wait is ever coupled to a younger HBM request.
//   * buffer loads with per-thread offsets computed once, out-of-image halo pieces through offset 0xffffffff (hardware zero
//     fill), edge tests = one AND of a per-thread nibble mask with a uniform nibble, one basic block per unit.
//   * lane -> pixel map of an MFMA row block chosen so that each of ds_read_b128's 16-lane groups ({0-3,12-15,20-27}, ...)
//     reads 16 CONSECUTIVE pixels of one image row: with the swizzle every operand fetch is bank-conflict free for every tap.
//   * epilogue through the unit buffer that has just been freed (two rounds of 64 rows per wave); the next tile's first unit
//     is already in the other buffer and its second one in flight, so the first MFMA after the epilogue issues at once.
// Accumulation order over K is (16-channel slice, tap) instead of (32-channel chunk, tap, 16-channel step): results differ
// from conv_igemm_kernel's in the last bits (fp32 sums in another order), not in what they are.
#include "conv_common.h"
#include <type_traits>
#include <mutex>
#ifndef IM2IM_CROLL_PIN
#define IM2IM_CROLL_PIN 1
#endif
#ifndef IM2IM_CROLL_ABL      // measurement-only builds (tools/ab_build_one.sh): bit 0 = no output stores, bit 1 = no epilogue at all, bit 2 = no
#define IM2IM_CROLL_ABL 0    // global loads inside the unit loop, bit 3 = no staging writes inside the unit loop, bit 4 = no barrier per unit
#endif

namespace {

using namespace im2im;
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int I, int N, typename F> __device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) { f(std::integral_constant<int, I>{}); static_for<I + 1, N>(f); }
}

constexpr int R_TH = 32, R_TW = 16, R_BN = 64;
constexpr int R_HH = R_TH + 2, R_HWD = R_TW + 2, R_HPX = R_HH * R_HWD;     // 34 x 18 = 612 halo pixels
constexpr int R_SLOT = 32;                                                 // bytes per halo pixel and unit (16 channels)
constexpr int R_HROW = R_HWD * R_SLOT;                                     // 576
constexpr int R_HALO_B = R_HPX * R_SLOT;                                   // 19,584
constexpr int R_W_B = 18 * 1024;                                           // 9 taps x 2 row blocks of 32 output channels
constexpr int R_DUMMY = R_HALO_B + R_W_B;                                  // 16 bytes nobody reads (threads without a piece write here)
constexpr int R_UNIT_B = R_HALO_B + R_W_B + 16;                            // 38,032
constexpr int R_H_PIECES = R_HPX * 2, R_H_ROUNDS = 5, R_W_ROUNDS = 5, R_NP = R_H_ROUNDS + R_W_ROUNDS;
constexpr int R_WP = R_BN * 2 + 16;                                        // epilogue tile row pitch (bytes)
constexpr int R_EROWS = 64;                                                // epilogue rows per wave and round
static_assert(4 * R_EROWS * R_WP <= R_UNIT_B, "the epilogue tiles of the four waves fit one unit buffer");

// EPI: 0 = (+bias) store [data-gradient]; 1 = +bias, store, BatchNorm partial statistics [train forward]; 2 = folded BatchNorm
// affine + ReLU [eval forward]  (conv_mfma.hip's numbering).  LAZY: some source of the launch has lazy coefficients.
template <int EPI, bool LAZY>
__global__ __launch_bounds__(256, 2) void conv_roll64_kernel(ConvArgs a, int nitems) {
  using T = bf16_t;
  constexpr int MT = 4, NT = 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* ldsSS = reinterpret_cast<float*>(smem + 2 * R_UNIT_B);            // [2][Ci] scale, shift of the logical input channels

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int NU = a.Ci >> 4;                                                // units per item
  const bool split_in = a.x_hi != nullptr;
  const int NU_lo = (split_in ? a.Ci_lo : a.Ci) >> 4;
  const int xstride = split_in ? a.Ci_lo : a.Ci;
  const int nch2 = (a.Ci >> 5) * 2;                                        // 1 KiB fragments per (row block, tap)
  const int ncob = a.Co / R_BN;

  // ---- this workgroup's run of items (item = (pixel tile, 64-channel output block), output block innermost) ----
  int run = blockIdx.x;
  const int G = gridDim.x;
  if ((G & 7) == 0) run = (run & 7) * (G >> 3) + (run >> 3);               // workgroups go to XCDs round-robin: a band of runs per XCD
  const int it_begin = (int)(((long long)run * nitems) / G), it_end = (int)(((long long)(run + 1) * nitems) / G);
  if (it_begin >= it_end) return;
  const int N = (it_end - it_begin) * NU;                                  // units of this run

  // ---- operand fetch addresses (LDS, relative to the unit buffer) ----
  // MFMA row l31 of a 32-row block = pixel (row pair rs, column xx) with the hardware's 16-lane groups mapped to whole rows
  const int xx = l31 & 15;
  const int rs = ((l31 >> 4) ^ ((xx >> 2) ^ (xx >> 3))) & 1;               // rows {0-3,12-15 | 20-27} -> rs 0, {4-11 | 16-19,28-31} -> rs 1
  int A_dx[3];
#pragma unroll
  for (int dx = 0; dx < 3; ++dx) {
    const int hx = xx + dx;
    A_dx[dx] = (wave * 8 + rs) * R_HROW + hx * R_SLOT + ((half ^ ((hx >> 3) & 1)) << 4);
  }
  const int W_l = R_HALO_B + lane * 16;

  // ---- per-thread staging constants ----
  // halo: a round = 7 halo rows of 36 16-byte pieces (thread t -> row t / 36, column (t % 36) >> 1, half t & 1; threads 252-255 repeat
  // thread 251), so the rounds differ by a UNIFORM byte stride in memory (the load's scalar offset) and in LDS (an immediate):
  // one byte offset and one LDS address per thread.  Round 4 holds rows 28-33: its seventh row group has no piece -- it requests
  // offset 0xffffffff (no memory access, zeros) and writes them to 16 spare bytes.  (A negative byte offset compensated by the
  // scalar offset does NOT work: the range check looks at the vector offset alone and returned zeros for row 27.)
  const int tq = tid < 252 ? tid : 251;
  const int h_rowg = tq / 36, h_hx = (tq % 36) >> 1, h_half = tq & 1;
  const int h_g0 = ((h_rowg * a.W + h_hx) * xstride + h_half * 8) * 2;
  const int h_rstep = 7 * a.W * xstride * 2;                              // bytes per round (uniform)
  const int h_glast = h_rowg == 6 ? -1 : h_g0;
  const int h_l0 = (h_rowg * R_HWD + h_hx) * R_SLOT + ((h_half ^ ((h_hx >> 3) & 1)) << 4);
  const int h_llast = h_rowg == 6 ? R_DUMMY : h_l0 + 4 * 7 * R_HROW;
  unsigned h_edge = 0;                                                     // nibble i: top row, bottom row, left column, right column of the halo
#pragma unroll
  for (int i = 0; i < R_H_ROUNDS; ++i) {
    const int hy = i * 7 + h_rowg;                                         // (row 34 of round 4 does not exist: no edge bits, see h_glast)
    h_edge |= (unsigned)((hy == 0) | ((hy == R_HH - 1) << 1) | ((h_hx == 0) << 2) | ((h_hx == R_HWD - 1) << 3)) << (4 * i);
  }
  // weight piece: 1 KiB fragment (round * 4 + wave) = tap * 2 + row block, 16 bytes at lane * 16; rounds differ by a uniform
  // byte stride, so the per-thread part is lane * 16 only.  Fragments 18, 19 do not exist: waves 2, 3 repeat round 3.
  const int w_voff = lane * 16;
  const int w_frag0 = (wave & 1) * 9 + (wave >> 1);                        // (row block * 9 + tap) of round 0
  const int w_last = wave < 2 ? 4 : 3;
  const int w_loff = R_HALO_B + wave * 1024 + lane * 16;

  if constexpr (LAZY) {
    const int clo = split_in ? a.Ci_lo : a.Ci, chi = a.Ci - clo;
    for (int i = tid; i < clo; i += 256) { ldsSS[i] = a.in_ss ? a.in_ss[i] : 1.f; ldsSS[a.Ci + i] = a.in_ss ? a.in_ss[clo + i] : 0.f; }
    for (int i = tid; i < chi; i += 256) { ldsSS[clo + i] = a.in_ss_hi ? a.in_ss_hi[i] : 1.f; ldsSS[a.Ci + clo + i] = a.in_ss_hi ? a.in_ss_hi[chi + i] : 0.f; }
    __syncthreads();
  }

  // ---- item / unit walk (wave-uniform) ----
  struct Cur { int tx, ty, b, cob, u; };
  auto cur_of = [&](int item) __attribute__((always_inline)) -> Cur {
    Cur c; c.u = 0; c.cob = item % ncob; int t = item / ncob;
    c.tx = t % a.tilesX; t /= a.tilesX; c.ty = t % a.tilesY; c.b = t / a.tilesY; return c;
  };
  auto next_unit = [&](Cur c, bool go) __attribute__((always_inline)) -> Cur {        // the next unit if go, else the same one; no branches
    const int wu = (c.u + 1 == NU), wc = wu & (c.cob + 1 == ncob), wx = wc & (c.tx + 1 == a.tilesX), wy = wx & (c.ty + 1 == a.tilesY);
    Cur n;
    n.u = wu ? 0 : c.u + 1;
    n.cob = wc ? 0 : c.cob + wu;
    n.tx = wx ? 0 : c.tx + wc;
    n.ty = wy ? 0 : c.ty + wx;
    n.b = c.b + wy;
    n.u = go ? n.u : c.u; n.cob = go ? n.cob : c.cob; n.tx = go ? n.tx : c.tx; n.ty = go ? n.ty : c.ty; n.b = go ? n.b : c.b;
    return n;
  };
  const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ xh = reinterpret_cast<const T*>(a.x_hi);
  const char* __restrict__ wgb = reinterpret_cast<const char*>(a.w);
  struct Src { __amdgpu_buffer_rsrc_t x, w; int xs, ws; unsigned bad; int u; int relu_floor; };
  auto src_of = [&](Cur c) __attribute__((always_inline)) -> Src {
    const int y0 = c.ty * R_TH, x0 = c.tx * R_TW;
    const bool hi = c.u >= NU_lo;
    const T* xb = (hi ? xh : xg) + (((ptrdiff_t)c.b * a.H + y0 - 1) * a.W + x0 - 1) * (ptrdiff_t)xstride;   // may lie in front of the tensor: those pieces are "bad"
    const unsigned em = (unsigned)((y0 == 0) | ((y0 + R_TH == a.H) << 1) | ((x0 == 0) << 2) | ((x0 + R_TW == a.W) << 3));
    Src s;
    // (pinned to scalar registers: left to itself the compiler keeps this base in vector registers in the LAZY instantiations and
    // wraps every halo load in a waterfall loop, which also cuts the unit into a dozen basic blocks)
    const unsigned long long xbv = reinterpret_cast<unsigned long long>(xb);
    xb = reinterpret_cast<const T*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(xbv >> 32)) << 32) |
                                    (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)xbv));
    s.x = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(xb), 0, 0x7fffffff, 0x00020000);
    s.w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(wgb + (size_t)c.cob * 2 * 9 * nch2 * 1024), 0, 0x7fffffff, 0x00020000);
    s.xs = __builtin_amdgcn_readfirstlane((hi ? c.u - NU_lo : c.u) * 32);
    s.ws = __builtin_amdgcn_readfirstlane(c.u * 1024);
    s.bad = h_edge & (em * 0x11111u);
    s.u = c.u;
    const bool lazy_src = hi ? (a.in_ss_hi != nullptr) : (a.in_ss != nullptr);
    s.relu_floor = lazy_src ? 0 : (int)0x80008000;                         // v_pk_max_i16 floor: 0 = ReLU, -32768 = identity
    return s;
  };
  struct Stage { i32x4 v[R_NP]; };                                         // one unit in flight; piece k: even = halo round k/2, odd = weight round k/2
  auto gload_piece = [&](const Src& s, Stage& R, auto k_tag) __attribute__((always_inline)) {
    constexpr int k = decltype(k_tag)::value, i = k >> 1;
    if constexpr ((k & 1) == 0) {
      const int off = ((s.bad >> (4 * i)) & 15u) ? -1 : (i == R_H_ROUNDS - 1 ? h_glast : h_g0);   // 0xffffffff >= num_records: the load returns zeros
      R.v[k] = __builtin_amdgcn_raw_buffer_load_b128(s.x, off, s.xs + i * h_rstep, 0);
    } else {
      const int ie = (i == R_W_ROUNDS - 1) ? w_last : i;
      R.v[k] = __builtin_amdgcn_raw_buffer_load_b128(s.w, w_voff, s.ws + (w_frag0 + 2 * ie) * nch2 * 1024, 0);
    }
  };
  struct Wr { unsigned bad; int u; int relu_floor; };                      // what swrite_piece needs of the unit it writes
  f32x2 sc[4], sh[4];
  auto load_coefs = [&](int u) __attribute__((always_inline)) {            // this thread's 8 channels of unit u: (tid & 1) * 8 ...
    if constexpr (LAZY) {
      const float* p = ldsSS + u * 16 + h_half * 8;
      const float4 s0 = *reinterpret_cast<const float4*>(p), s1 = *reinterpret_cast<const float4*>(p + 4);
      const float4 h0 = *reinterpret_cast<const float4*>(p + a.Ci), h1 = *reinterpret_cast<const float4*>(p + a.Ci + 4);
      sc[0] = f32x2{s0.x, s0.y}; sc[1] = f32x2{s0.z, s0.w}; sc[2] = f32x2{s1.x, s1.y}; sc[3] = f32x2{s1.z, s1.w};
      sh[0] = f32x2{h0.x, h0.y}; sh[1] = f32x2{h0.z, h0.w}; sh[2] = f32x2{h1.x, h1.y}; sh[3] = f32x2{h1.z, h1.w};
    }
  };
  auto swrite_piece = [&](int bufoff, const Wr& e, const Stage& R, auto k_tag) __attribute__((always_inline)) {
    constexpr int k = decltype(k_tag)::value, i = k >> 1;
    i32x4 v = R.v[k];
    if constexpr ((k & 1) == 0) {
      if constexpr (LAZY) {
        const int keep = ((e.bad >> (4 * i)) & 15u) ? 0 : -1;              // zero padding stays exactly zero (not max(shift, 0))
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const unsigned u = (unsigned)v[j];
          f32x2 f = f32x2{__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)};
          f = f * sc[j];
          f = f + sh[j];
          const bf16_t lo = (bf16_t)f[0], hi = (bf16_t)f[1];
          const unsigned r = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
          const s16x2 m = __builtin_elementwise_max(__builtin_bit_cast(s16x2, r), __builtin_bit_cast(s16x2, e.relu_floor));
          v[j] = __builtin_bit_cast(int, m) & keep;
        }
      }
      if constexpr (i < R_H_ROUNDS - 1) *reinterpret_cast<i32x4*>(smem + bufoff + h_l0 + i * 7 * R_HROW) = v;
      else *reinterpret_cast<i32x4*>(smem + bufoff + h_llast) = v;
    } else {
      if constexpr (i < R_W_ROUNDS - 1) *reinterpret_cast<i32x4*>(smem + bufoff + w_loff + i * 4096) = v;
      else *reinterpret_cast<i32x4*>(smem + bufoff + w_loff + w_last * 4096) = v;
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  // ---- prologue: unit 0 into buffer 0, unit 1 requested ----
  Stage R;
  Cur ld = cur_of(it_begin);                 // load cursor
  Cur ep = ld;                               // the item whose accumulators are being built (epilogue coordinates)
  Wr wr;
  {
    const Src s0 = src_of(ld);
    static_for<0, R_NP>([&](auto k_tag) __attribute__((always_inline)) { gload_piece(s0, R, k_tag); });
    const Wr w0{s0.bad, s0.u, s0.relu_floor};
    load_coefs(w0.u);
    static_for<0, R_NP>([&](auto k_tag) __attribute__((always_inline)) { swrite_piece(0, w0, R, k_tag); });
    ld = next_unit(ld, 1 < N);
    const Src s1 = src_of(ld);
    static_for<0, R_NP>([&](auto k_tag) __attribute__((always_inline)) { gload_piece(s1, R, k_tag); });
    wr = Wr{s1.bad, s1.u, s1.relu_floor};
  }
  __syncthreads();

  int curoff = 0, u_in_item = 0;
  for (int n = 0; n < N; ++n) {
    ld = next_unit(ld, n + 2 < N);
    const Src s2 = src_of(ld);                                             // unit n + 2 (the last unit again at the end of the run)
    const int nxtoff = R_UNIT_B - curoff;
    load_coefs(wr.u);
    const char* pa0 = smem + curoff + A_dx[0];
    const char* pa1 = smem + curoff + A_dx[1];
    const char* pa2 = smem + curoff + A_dx[2];
    const char* pw = smem + curoff + W_l;
    short8 fa[2][MT], fb[2][NT];
    // operand ring: the weight fragments of tap t + 1 and its first pixel fragment are requested before the MFMAs of tap t issue,
    // pixel fragment m + 1 of tap t + 1 after the MFMAs of row block m: 5 pixel + 4 weight fragments live instead of 8 + 4
    auto req_a = [&](auto t_tag, auto m_tag) __attribute__((always_inline)) {
      constexpr int t = decltype(t_tag)::value, mt = decltype(m_tag)::value;
      if constexpr (t < 9) {
        constexpr int dy = t / 3, dx = t % 3;
        const char* pa = dx == 0 ? pa0 : dx == 1 ? pa1 : pa2;
        fa[t & 1][mt] = *reinterpret_cast<const short8*>(pa + (mt * 2 + dy) * R_HROW);
      }
    };
    auto req_b = [&](auto t_tag) __attribute__((always_inline)) {
      constexpr int t = decltype(t_tag)::value;
      if constexpr (t < 9) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) fb[t & 1][nt] = *reinterpret_cast<const short8*>(pw + (t * 2 + nt) * 1024);
      }
    };
    req_b(std::integral_constant<int, 0>{});
    static_for<0, MT>([&](auto m_tag) __attribute__((always_inline)) { req_a(std::integral_constant<int, 0>{}, m_tag); });
    static_for<0, 9>([&](auto t_tag) __attribute__((always_inline)) {
      constexpr int t = decltype(t_tag)::value;
      req_b(std::integral_constant<int, t + 1>{});
      req_a(std::integral_constant<int, t + 1>{}, std::integral_constant<int, 0>{});
#if IM2IM_CROLL_PIN
      if constexpr (t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, NT + 1, 0);
#endif
      static_for<0, MT>([&](auto m_tag) __attribute__((always_inline)) {
        constexpr int mt = decltype(m_tag)::value;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa[t & 1][mt]), as_bf16x8(fb[t & 1][nt]), acc[mt][nt], 0, 0, 0);
#if IM2IM_CROLL_PIN
        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
#endif
        if constexpr (mt + 1 < MT) {
          req_a(std::integral_constant<int, t + 1>{}, std::integral_constant<int, mt + 1>{});
#if IM2IM_CROLL_PIN
          if constexpr (t + 1 < 9) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#endif
        }
      });
      static_for<0, R_NP>([&](auto k_tag) __attribute__((always_inline)) {
        constexpr int k = decltype(k_tag)::value;
        if constexpr ((2 * k + 1) * 9 / (2 * R_NP) == t) {
          // SALU, MFMA and ds_read may cross these fences; ds_write, buffer_load and VALU may not (conv_wgrad_roll_kernel)
          __builtin_amdgcn_sched_barrier(0x10c);
          if constexpr (!(IM2IM_CROLL_ABL & 8)) swrite_piece(nxtoff, wr, R, k_tag);          // unit n + 1
          if constexpr (!(IM2IM_CROLL_ABL & 4)) gload_piece(s2, R, k_tag);                   // unit n + 2
          __builtin_amdgcn_sched_barrier(0x10c);
        }
      });
    });
    wr = Wr{s2.bad, s2.u, s2.relu_floor};
    if constexpr (!(IM2IM_CROLL_ABL & 16)) __syncthreads();
    ++u_in_item;
    if (u_in_item == NU) {
      u_in_item = 0;
      // ------------------------------------------------------------ epilogue of item `ep`, through the buffer just freed
      char* wbuf = smem + curoff + wave * (R_EROWS * R_WP);
      float* ldsS = reinterpret_cast<float*>(smem + curoff);
      const int n0 = ep.cob * R_BN;
      const int y0 = ep.ty * R_TH, x0 = ep.tx * R_TW;
      const bool to_hi = a.y_hi != nullptr && n0 >= a.Co_lo;
      T* __restrict__ yg = reinterpret_cast<T*>(to_hi ? a.y_hi : a.y) + (to_hi ? n0 - a.Co_lo : n0);
      const int ystride = a.y_hi == nullptr ? a.Co : (to_hi ? a.Co - a.Co_lo : a.Co_lo);
#if IM2IM_CROLL_ABL & 2
      {
        float t = 0.f;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
          for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { t += acc[mt][nt][r]; acc[mt][nt][r] = 0.f; }
        if (t == 123.456f) yg[tid] = from_float<T>(t);
        const int wc = (ep.cob + 1 == ncob), wx = wc & (ep.tx + 1 == a.tilesX), wy = wx & (ep.ty + 1 == a.tilesY);
        ep.cob = wc ? 0 : ep.cob + 1;
        const int tx = wx ? 0 : ep.tx + wc, ty = wy ? 0 : ep.ty + wx;
        ep.tx = tx; ep.ty = ty; ep.b += wy;
        curoff = nxtoff;
        continue;
      }
#endif
      float bias_v[NT], esc[NT], esh[NT], Kc[NT], s_[NT], sq_[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const int c = n0 + nt * 32 + l31;
        bias_v[nt] = (a.bias ? a.bias[c] : 0.f) - (a.center ? a.center[c] : 0.f);
        esc[nt] = 1.f; esh[nt] = 0.f;
        if constexpr (EPI == 2) { esc[nt] = a.scale[c]; esh[nt] = a.shift[c]; }
        Kc[nt] = to_float(from_float<T>(acc[0][nt][0] + bias_v[nt]));     // statistics are taken relative to a sample (conv_mfma.hip)
        s_[nt] = 0.f; sq_[nt] = 0.f;
      }
#pragma unroll
      for (int rd = 0; rd < 2; ++rd) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int ml = 0; ml < 2; ++ml)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              // MFMA row (r & 3) + 8 * (r >> 2) + 4 * half of the block = pixel (row pair, column) by the lane map above
              const int col = (r & 3) + 8 * ((r >> 2) & 1) + 4 * half;
              const int rp = ((r >> 3) ^ half ^ ((r >> 2) & 1)) & 1;
              const int row = ml * 32 + rp * 16 + col;
              float v = acc[rd * 2 + ml][nt][r] + bias_v[nt];
              if constexpr (EPI == 2) {
                v = v * esc[nt] + esh[nt];
                if (a.relu) v = fmaxf(v, 0.f);
              }
              const T tv = from_float<T>(v);
              *reinterpret_cast<T*>(wbuf + row * R_WP + (nt * 32 + l31) * 2) = tv;
              if constexpr (EPI == 1) { const float d = to_float(tv) - Kc[nt]; s_[nt] += d; sq_[nt] += d * d; }
              acc[rd * 2 + ml][nt][r] = 0.f;
            }
        // wave-private region: LDS operations of one wave complete in issue order, no barrier needed
#pragma unroll
        for (int pass = 0; pass < R_EROWS / 8; ++pass) {
          const int row = pass * 8 + (lane >> 3), piece = lane & 7;
          const uint4 v = *reinterpret_cast<const uint4*>(wbuf + row * R_WP + piece * 16);
          const int m = wave * 128 + rd * 64 + row;
          const size_t off = (((size_t)ep.b * a.H + y0 + (m >> 4)) * a.W + x0 + (m & 15)) * ystride + piece * 8;
          if (!(IM2IM_CROLL_ABL & 1) || v.x == 0x12345678u) __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(yg + off));
        }
      }
      if constexpr (EPI == 1) {
        __syncthreads();                                       // the statistics scratch aliases wave 0's tile
        const float cnt = 64.f;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          float nn = cnt, mm = Kc[nt] + s_[nt] * (1.f / cnt), qq = fmaxf(sq_[nt] - s_[nt] * s_[nt] * (1.f / cnt), 0.f);
          merge_moments_f32(nn, mm, qq, __shfl_xor(nn, 32, 64), __shfl_xor(mm, 32, 64), __shfl_xor(qq, 32, 64));
          const int nl = nt * 32 + l31;
          if (half == 0) { ldsS[(wave * R_BN + nl) * 3 + 0] = nn; ldsS[(wave * R_BN + nl) * 3 + 1] = mm; ldsS[(wave * R_BN + nl) * 3 + 2] = qq; }
        }
        __syncthreads();
        if (tid < R_BN) {
          float nn = ldsS[tid * 3 + 0], mm = ldsS[tid * 3 + 1], qq = ldsS[tid * 3 + 2];
#pragma unroll
          for (int i = 1; i < 4; ++i)
            merge_moments_f32(nn, mm, qq, ldsS[(i * R_BN + tid) * 3 + 0], ldsS[(i * R_BN + tid) * 3 + 1], ldsS[(i * R_BN + tid) * 3 + 2]);
          const int tile_id = (ep.b * a.tilesY + ep.ty) * a.tilesX + ep.tx;
          float* st = a.stats + (size_t)tile_id * 3 * a.Co;
          st[n0 + tid] = mm;
          st[a.Co + n0 + tid] = qq;
          st[2 * a.Co + n0 + tid] = nn;
        }
      }
      __syncthreads();                                         // the next unit's staging writes go into this buffer
      // advance the epilogue cursor by one item
      {
        const int wc = (ep.cob + 1 == ncob), wx = wc & (ep.tx + 1 == a.tilesX), wy = wx & (ep.ty + 1 == a.tilesY);
        ep.cob = wc ? 0 : ep.cob + 1;
        const int tx = wx ? 0 : ep.tx + wc, ty = wy ? 0 : ep.ty + wx;
        ep.tx = tx; ep.ty = ty; ep.b += wy;
      }
    }
    curoff = nxtoff;
  }
}

// A/B switch (im2im_set_option "conv_roll", env IM2IM_CONV_ROLL): 0 = conv_igemm_kernel for every launch (DEFAULT), 1 = every eligible launch
// here, 2 = as 1 with one workgroup per CU (a measurement), 3 = the data-gradients only.  Default off because the chip gives the
// kernel's cycle savings back as clock: 64 -> 64 @ 320 x 320 data-gradient, batch 78, sustained: 1,104 k cycles at 1.37 GHz against
// 1,194 k at 1.52 GHz for conv_igemm (MFMA pipe busy 0.51 vs 0.47), i.e. the same energy per tile under the power limit; the
// training step is 39.2 vs 39.0 ms with it (profiles/r05_ab_experiments.txt sections 1-2)
int g_conv_roll = 0;

template <int EPI>
int launch_roll_epi(const ConvArgs& a, int nitems, int grid, size_t smem, hipStream_t stream) {
  const bool lazy = a.in_ss != nullptr || a.in_ss_hi != nullptr;
  auto k1 = conv_roll64_kernel<EPI, true>;
  auto k0 = conv_roll64_kernel<EPI, false>;
  // the dynamic-LDS limit is a per-device function attribute: remembered per device, set under a lock, and a refusal (a part with
  // less LDS than gfx950's 160 KB) is an error code for THIS launch, not a later launch failure
  static std::mutex mu;
  static size_t attr_set[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return fail_invalid("conv_roll64: device index");
  {
    std::lock_guard<std::mutex> lock(mu);
    if (smem > attr_set[dev]) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k1), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess ||
          hipFuncSetAttribute(reinterpret_cast<const void*>(k0), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess) {
        (void)hipGetLastError();
        return fail_invalid("conv_roll64: the device refuses the kernel's dynamic LDS size");
      }
      attr_set[dev] = smem;
    }
  }
  if (lazy) hipLaunchKernelGGL(k1, dim3((unsigned)grid), dim3(256), smem, stream, a, nitems);
  else hipLaunchKernelGGL(k0, dim3((unsigned)grid), dim3(256), smem, stream, a, nitems);
  return check_launch("conv_roll64_kernel");
}

}  // namespace

namespace im2im {

void set_conv_roll(int v) { g_conv_roll = v; }

// does this launch go to conv_roll64_kernel?  (bf16, 3x3, the 32 x 16 x 64 tile with nothing hanging over the image, no split-K,
// no fused BatchNorm-backward sums / max-pool / 1x1 tail in the epilogue, one (scale, shift) pair per channel)
bool conv_roll64_eligible(const ConvArgs& a, const TileChoice& t, int taps, bool per_image) {
  if (g_conv_roll == 0) return false;
  {   // two workgroups per CU at 2 x 38 KB + coefficients: only parts with gfx950's LDS take it; the others keep conv_igemm_kernel
    static int lds_ok = -1;
    if (lds_ok < 0) {
      int dev = 0, lds = 0;
      lds_ok = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess &&
                (size_t)lds >= (size_t)2 * R_UNIT_B + (size_t)2 * 512 * sizeof(float)) ? 1 : 0;
    }
    if (!lds_ok) return false;
  }
  if (g_conv_roll == 3 && (a.in_ss || a.in_ss_hi || a.stats || a.scale)) return false;      // 3 = the data-gradients only
  return g_conv_roll != 0 && taps == 9 && !per_image && t.tb == 1 && t.th == R_TH && t.tw == R_TW && t.bn == R_BN && a.H % R_TH == 0 &&
         a.W % R_TW == 0 && a.Ci % 32 == 0 && a.Ci <= 512 && a.Co % R_BN == 0 && a.ksplit <= 1 && a.bn_partial == nullptr &&
         a.fuse_y == nullptr && a.pool_y == nullptr && a.in_ss_img == 0 && (a.x_hi == nullptr || a.Ci_lo % 16 == 0);
}

int launch_conv_roll64(const ConvArgs& a_in, hipStream_t stream) {
  ConvArgs a = a_in;
  a.tilesY = a.H / R_TH;
  a.tilesX = a.W / R_TW;
  const long long items = (long long)a.B * a.tilesY * a.tilesX * (a.Co / R_BN);
  if (items <= 0 || items > 0x7fffffffLL / 64) return fail_invalid("conv_roll64: item count");
  int grid = (int)std::min<long long>(items, g_conv_roll == 2 ? 256 : 512);     // two resident workgroups per CU (option value 2: one, a measurement)
  if (grid >= 8) grid &= ~7;
  const size_t smem = (size_t)2 * R_UNIT_B + (size_t)2 * a.Ci * sizeof(float);
  if (a.stats) return launch_roll_epi<1>(a, (int)items, grid, smem, stream);
  if (a.scale) return launch_roll_epi<2>(a, (int)items, grid, smem, stream);
  return launch_roll_epi<0>(a, (int)items, grid, smem, stream);
}

}  // namespace im2im
