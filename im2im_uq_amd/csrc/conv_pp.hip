// 8-wave "ping-pong" form of the bf16 3x3 implicit-GEMM convolution (conv_mfma.hip has the 4-wave form and the full
// description of the GEMM view, the halo staging, the lazy BatchNorm and the epilogues; this file follows it piece by
// piece).  Replaces the same nn.Conv2d calls of the reference UNet (core/models/trunks/unet_parts.py:16,19) and their
// autograd data-gradient.
//
// Why: in the 4-wave kernel two independent workgroups share a CU (two waves per SIMD).  Each wave alternates between
// fetching MFMA operands from LDS (latency-bound) and issuing MFMAs, and nothing keeps the two waves of a SIMD in
// complementary phases: they convoy, both fetch, then both queue on the one matrix pipe -- SQ counters show the pipe
// busy ~54 % of the time (profiles/r02_pmc_conv_sq_counters.txt).  Here ONE workgroup of 8 waves holds two GROUPS of four;
// each group owns its own output tile, halo buffer, weight buffers and accumulators exactly like a 4-wave workgroup, but
// the groups are phase-locked one phase (half a k-step) apart by the workgroup barrier:
//
//     phase 2k   : group 0  L(k): ds_read the fragments of k-step k, stage weights / halo    group 1  M(k-1): 8 MFMAs, registers only
//     phase 2k+1 : group 0  M(k): 8 MFMAs from registers                                  group 1  L(k)
//
// so on every SIMD one wave is always in a register-only MFMA segment while its partner does the LDS / global work
// (MI355X_MICROARCH.md "Two waves per SIMD": a rendezvous structure pays when the merged interval is complementary --
// matrix beside memory).  Both groups run the same instruction stream; group 1 starts with one extra barrier and group 0
// ends with one, so every wave executes the same number of barriers.
#include "conv_common.h"
#include <cstdlib>
#include <type_traits>
#include <utility>

namespace {

using namespace im2im;

#ifdef IM2IM_PP_TRACE
// debug build (tools/ab_build.sh trace -DIM2IM_PP_TRACE): wave 0 of each group of workgroup 0 keeps the s_memtime stamp of every
// phase boundary in LDS and dumps them to ConvArgs::bn_partial (unused by this kernel) -> tools/pp_trace.py prints phase lengths
#define PP_STAMP() do { if (tr_on) { const unsigned long long ts_ = __builtin_amdgcn_s_memtime(); \
    if (lane == 0 && tr_n < 2048 && 8 * tr_n + 8 <= 16384) *reinterpret_cast<unsigned long long*>(smem + tr_base + 8 * tr_n) = ts_; ++tr_n; } } while (0)
#else
#define PP_STAMP() do {} while (0)
#endif

// compile-time loop: f(std::integral_constant<int, 0>{}), ..., f(std::integral_constant<int, N-1>{})
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(f, std::make_integer_sequence<int, N>{}); }

// lazy BatchNorm+ReLU of the two bf16 of one dword, on the register itself (going through a pointer to the staging array,
// or a nested lambda, kept the array / the closures on the stack); same arithmetic and rounding as Vec16<bf16_t> +
// conv_mfma.hip's staging.  ok == false (padding / outside the image): exact zeros.
__device__ __forceinline__ unsigned lazy_pair(unsigned w, float sc0, float sh0, float sc1, float sh1, bool lazy_in, bool ok) {
  const float lo = __uint_as_float(w << 16), hi = __uint_as_float(w & 0xffff0000u);
  const float rl = fmaxf(lo * sc0 + sh0, 0.f), rh = fmaxf(hi * sc1 + sh1, 0.f);
  const bf16_t bl = (bf16_t)(lazy_in ? rl : lo), bh = (bf16_t)(lazy_in ? rh : hi);
  const unsigned packed = (unsigned)__builtin_bit_cast(unsigned short, bl) | ((unsigned)__builtin_bit_cast(unsigned short, bh) << 16);
  return ok ? packed : 0u;
}

__device__ __forceinline__ void wg_barrier() {
  // LDS traffic of this wave complete (stores visible, fragment reads landed), then the workgroup barrier; the scheduling
  // fences keep the compiler from sliding MFMAs / LDS reads across the phase boundary (MFMAs are register-only
  // instructions, a plain barrier does not order them)
  __builtin_amdgcn_sched_barrier(0);
#ifdef IM2IM_PP_PRIO
  __builtin_amdgcn_s_setprio(0);
#endif
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
}

// LAZY: some input carries lazy BatchNorm+ReLU coefficients (compile-time so that launches without any -- the data-gradients --
// do not pay for the transform, and so that no branch splits the scheduling region of an M phase)
template <int TB, int TH, int TW, int BN, int WM, int WN, int EPI, bool LAZY>
__global__ __launch_bounds__(512, 2) void conv_pp_kernel(ConvArgs a) {
  using T = bf16_t;
  constexpr int TAPS = 9, PAD = 1;
  constexpr int HH = TH + 2 * PAD, HWD = TW + 2 * PAD, HPI = HH * HWD, HPX = TB * HPI;
  constexpr int MI = TH * TW;
  constexpr int KC = 32, EPP = 8, PPR = 4, ROWB = KC * 2 + 16;
  constexpr int M = TB * TH * TW;
  constexpr int MT = M / (32 * WM), NT = BN / (32 * WN);
  static_assert(WM * WN == 4 && M == 256, "4 waves per group, 256 pixels per group tile");
  constexpr int A_ROUNDS = (HPX * PPR + 255) / 256;
  constexpr int B_ROUNDS = (BN * PPR + 255) / 256;
  static_assert((BN * PPR) % 256 == 0, "weight tile = whole rounds");
  constexpr int HROWB = HWD * ROWB + 96;
  constexpr int HIMGB = HH * HROWB;
  constexpr int A_BYTES = TB * HIMGB, B_BYTES = BN * ROWB;
  constexpr int WROWS = MT * 32, WCOLS = NT * 32;
  constexpr int WP = WCOLS * 2 + 16;
  constexpr int WBYTES = WROWS * WP;

  // scalar copies of the fields the (nested) lambdas below use: a lambda that captures the by-value argument block itself
  // makes the compiler keep the whole block on the stack and re-load fields from scratch in the main loop
  const int aCi = a.Ci, aCi_lo = a.Ci_lo, aB = a.B, aH = a.H, aW = a.W;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x;
  // the two groups must pair up on the SIMDs (one wave of each group per SIMD).  pp_flags bit 0 picks which waves form a
  // group: 0 = waves 0-3 / 4-7 (pairs w, w+4 share a SIMD when waves are dealt round-robin), 1 = even / odd waves
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = (a.pp_flags & 1) ? (wv & 1) : (wv >> 2);            // group 0 / 1
  const int wave = (a.pp_flags & 1) ? (wv >> 1) : (wv & 3);         // wave within the group
  const int lane = tid & 63;
  const int t = wave * 64 + lane;                                    // thread index within the group
  const int wm = wave / WN, wn = wave % WN;
  const int half = lane >> 5, l31 = lane & 31;
  // this group's LDS region.  Everything below addresses LDS as `smem + 32-bit byte offset` formed at the access: a char*
  // kept in a variable can lose its address space on the way through the register allocator (seen: the A-fragment reads
  // compiled to flat_load through a spilled 64-bit pointer)
  const int gbase = g * a.pp_group_bytes;
  const int oA = gbase, oB = gbase + A_BYTES, oSS = gbase + A_BYTES + 2 * B_BYTES;
  auto lds_f32 = [&](int byte_off) __attribute__((always_inline)) -> float& { return *reinterpret_cast<float*>(smem + byte_off); };
#ifdef IM2IM_PP_TRACE
  const bool tr_on = a.bn_partial != nullptr && blockIdx.x == 0 && wave == 0;
  const int tr_base = gbase + A_BYTES + 2 * B_BYTES + 4096;       // free during the main loop (the epilogue staging reuses it: dumped before)
  int tr_n = 0;
#endif

  // (tile, output-channel block) of this group: the two groups of a workgroup take neighbouring items of an XCD's band
  // (conv_mfma.hip: workgroup i runs on XCD i mod 8; a band's tiles share halo pixels through that XCD's L2 and the
  // channel blocks of a tile run back to back)
  int tile_id, cob;
  bool ghost = false;
  {
    const int ncob = a.Co / BN;
    const int ntiles = a.pp_ntiles;
    const int band = ntiles >> 3;
    const int per_xcd = band * ncob, ppx = (per_xcd + 1) >> 1;
    const int lin = blockIdx.x;
    if (lin < ppx * 8) {
      const int xcd = lin & 7, j2 = 2 * (lin >> 3) + g;
      ghost = j2 >= per_xcd;
      const int j = ghost ? per_xcd - 1 : j2;
      tile_id = xcd * band + j / ncob;
      cob = j % ncob;
    } else {
      const int left = (ntiles - band * 8) * ncob;
      const int j2 = 2 * (lin - ppx * 8) + g;
      ghost = j2 >= left;
      const int j = ghost ? left - 1 : j2;
      tile_id = band * 8 + j / ncob;
      cob = j % ncob;
    }
  }
  int mt_id = tile_id;
  const int tx_id = mt_id % a.tilesX; mt_id /= a.tilesX;
  const int ty_id = mt_id % a.tilesY;
  const int b0 = (mt_id / a.tilesY) * TB;
  const int y0 = ty_id * TH, x0 = tx_id * TW;
  const int n0 = cob * BN;

  const T* __restrict__ xg = reinterpret_cast<const T*>(a.x);
  const T* __restrict__ wg = reinterpret_cast<const T*>(a.w);
  const bool split_in = a.x_hi != nullptr;
  const int xstride = split_in ? aCi_lo : aCi;

  // LDS byte offsets of this lane's operand rows: one base per operand, the other fragments sit at compile-time distances
  // (32 consecutive pixels of a tile = 2 rows of 16 / 4 rows of 8; 32 output channels = 32 weight rows)
  static_assert((TW == 16 && TH == 16) || (TW == 8 && TH == 8), "tile shapes with constant fragment distances");
  int aoff0;
  {
    const int m = wm * MT * 32 + l31;
    aoff0 = (m / MI) * HIMGB + ((m % MI) / TW) * HROWB + (m % TW) * ROWB + half * 16;
  }
  auto aoff_rel = [](int mt) -> int {                 // fragment mt relative to fragment 0 (MT*32 divides MI or is a multiple of it)
    const int m = mt * 32;
    return (m / MI) * HIMGB + ((m % MI) / TW) * HROWB;
  };
  const int boff0 = (wn * NT * 32 + l31) * ROWB + half * 16;

  f32x16 acc[MT][NT];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

  uint4 ra[A_ROUNDS];
  uint4 rb0 = make_uint4(0, 0, 0, 0), rb1 = rb0;     // weight prefetch registers (named, not an array: the array form stayed a stack object)
  static_assert(B_ROUNDS <= 2, "weight tile = one or two rounds");
  static_assert(A_ROUNDS <= 7, "halo pieces are loaded at taps 0.., transformed two taps later, all before tap 8 ends");

  const T* __restrict__ xg_tile = xg + (size_t)b0 * aH * aW * xstride;
  const T* __restrict__ xh_tile = reinterpret_cast<const T*>(a.x_hi) + (size_t)b0 * aH * aW * xstride;
  const T* __restrict__ wg_tile = wg + (size_t)n0 * TAPS * aCi;
  const bool lazy_lo = a.in_ss != nullptr, lazy_hi = a.in_ss_hi != nullptr;
  const int oOFF = oSS + (LAZY ? 2 * aCi * 4 : 0);
  // halo piece i of this thread = pixel i*64 + t/4 of the halo patch, 16-byte part t%4.  Its (global element offset, LDS byte
  // offset) pair -- -1 for pieces outside the image / the patch -- is worked out ONCE per tile and parked in LDS (8 bytes per
  // piece and thread, read back by the same thread): in the main loop a piece then costs one ds_read_b64 instead of ~20 VALU,
  // and no registers between uses (the register file is the scarce thing: 128 accumulators + fragments + prefetches).
  const int part = t % PPR;
  auto piece = [&](int i, int& goff, int& loff) __attribute__((always_inline)) {
    int hx = (t / PPR) % HWD + (i * 64) % HWD;
    int hy = (t / PPR) / HWD + (i * 64) / HWD;
    if (hx >= HWD) { hx -= HWD; hy += 1; }
    const int tb = (hy * (65536 / HH + 1)) >> 16;           // hy / HH for the small values that occur
    hy -= tb * HH;
    const int yy = y0 + hy - PAD, xx = x0 + hx - PAD;
    const bool inpatch = (HPX * PPR) % 256 == 0 || (i * 64 + t / PPR) < HPX;
    const bool ok = inpatch && b0 + tb < aB && (unsigned)yy < (unsigned)aH && (unsigned)xx < (unsigned)aW;
    goff = ok ? (((tb * aH + yy) * aW + xx) * xstride + part * EPP) : -1;
    // pieces beyond the patch (the last, partly filled round) are written to a per-thread dump slot instead of being masked
    // off: an exec-masked ds_write cannot be scheduled between MFMAs
    loff = inpatch ? tb * HIMGB + hy * HROWB + hx * ROWB + part * 16 : (oOFF - oA) + A_ROUNDS * 256 * 8 + t * 16;
  };
  auto piece_offsets = [&](int i) __attribute__((always_inline)) -> int2 {
    return *reinterpret_cast<const int2*>(smem + oOFF + (i * 256 + t) * 8);
  };
  // weight piece i of this thread: row n = i*64 + t/4, 16-byte part t%4 -> global offset b_goff0 + i*64*9*Ci, LDS b_loff0 + i*64*ROWB
  // (fragment-major packed weights, conv_common.h wfrag_index: row n0 + i*64 + t/4, channels (t%4)*8.. of the chunk)
  const int b_goff0 = (int)wfrag_index(n0 + t / PPR, 0, (t % PPR) * EPP, aCi);
  const int b_loff0 = (t / PPR) * ROWB + (t % PPR) * 16;
  const int b_gstep = 64 * TAPS * aCi;                 // 64 rows = two row blocks of 9 * (Ci/32) * 1024 elements

  if (LAZY) {
    const int clo = split_in ? aCi_lo : aCi, chi = aCi - clo;
    if (lazy_lo) for (int i = t; i < clo; i += 256) { lds_f32(oSS + 4 * i) = a.in_ss[i]; lds_f32(oSS + 4 * (aCi + i)) = a.in_ss[clo + i]; }
    if (lazy_hi) for (int i = t; i < chi; i += 256) { lds_f32(oSS + 4 * (clo + i)) = a.in_ss_hi[i]; lds_f32(oSS + 4 * (aCi + clo + i)) = a.in_ss_hi[chi + i]; }
    __syncthreads();
  }
  // ---- halo staging in three steps per piece, each small enough to ride in the issue gaps of this wave's own MFMAs:
  //   load  : global -> register (clamped address; validity is re-read at the next step)
  //   xform : out-of-image pieces become exact zeros; lazy BatchNorm+ReLU max(z*scale+shift, 0) of the rest (conv_mfma.hip)
  //   write : register -> the halo buffer (only when every wave of the group has finished reading the previous chunk)
  auto halo_src = [&](int chunk) __attribute__((always_inline)) -> const T* {
    const int c = chunk * KC;
    return (split_in && c >= aCi_lo) ? xh_tile + (c - aCi_lo) : xg_tile + c;
  };
  auto halo_load = [&](const int i, int chunk) __attribute__((always_inline)) {
    const int2 o = piece_offsets(i);
    ra[i] = *reinterpret_cast<const uint4*>(halo_src(chunk) + max(o.x, 0));
  };
  // BatchNorm coefficients of this thread's 8 channels of the chunk being staged: four 16-byte LDS reads once per chunk
  // (named registers, live while its pieces are transformed), not sixteen scalar reads per piece
  f32x4 sc_a = {0.f, 0.f, 0.f, 0.f}, sc_b = sc_a, sh_a = sc_a, sh_b = sc_a;
  bool lazy_cur = false;
  auto load_coeffs = [&](int chunk) __attribute__((always_inline)) {
    if constexpr (LAZY) {
      // (bit arithmetic on values: `cond ? lazy_hi : lazy_lo` on the captured flags became a select between their ADDRESSES,
      // which pinned every captured variable to the stack)
      const int in_hi = (int)split_in & (int)(chunk * KC >= aCi_lo);
      lazy_cur = ((in_hi & (int)lazy_hi) | ((in_hi ^ 1) & (int)lazy_lo)) != 0;
      const int c0 = chunk * KC + part * EPP;
      sc_a = *reinterpret_cast<const f32x4*>(smem + oSS + 4 * c0);
      sc_b = *reinterpret_cast<const f32x4*>(smem + oSS + 4 * c0 + 16);
      sh_a = *reinterpret_cast<const f32x4*>(smem + oSS + 4 * (aCi + c0));
      sh_b = *reinterpret_cast<const f32x4*>(smem + oSS + 4 * (aCi + c0) + 16);
    }
  };
  auto halo_xform = [&](int i) __attribute__((always_inline)) {
    const bool ok = piece_offsets(i).x >= 0;
    if constexpr (LAZY) {
      ra[i].x = lazy_pair(ra[i].x, sc_a[0], sh_a[0], sc_a[1], sh_a[1], lazy_cur, ok);
      ra[i].y = lazy_pair(ra[i].y, sc_a[2], sh_a[2], sc_a[3], sh_a[3], lazy_cur, ok);
      ra[i].z = lazy_pair(ra[i].z, sc_b[0], sh_b[0], sc_b[1], sh_b[1], lazy_cur, ok);
      ra[i].w = lazy_pair(ra[i].w, sc_b[2], sh_b[2], sc_b[3], sh_b[3], lazy_cur, ok);
    } else {
      ra[i].x = ok ? ra[i].x : 0u; ra[i].y = ok ? ra[i].y : 0u; ra[i].z = ok ? ra[i].z : 0u; ra[i].w = ok ? ra[i].w : 0u;
    }
    // pin the transformed piece HERE: left alone, the compiler sinks the whole computation down to its only use -- the LDS
    // write in the last M phase of the chunk -- and that phase then carries six transforms
    asm volatile("" : "+v"(ra[i].x), "+v"(ra[i].y), "+v"(ra[i].z), "+v"(ra[i].w));
  };
  auto halo_write = [&](int i) __attribute__((always_inline)) {
    *reinterpret_cast<uint4*>(smem + oA + piece_offsets(i).y) = ra[i];
  };
  auto gload_B = [&](int gt) __attribute__((always_inline)) {                       // gt = global tap index = chunk*9 + tap
    const int chunk = gt / 9, tap = gt - chunk * 9;
    const T* src = wg + ((tap * (aCi >> 5) + chunk) << 10) + b_goff0;
    rb0 = *reinterpret_cast<const uint4*>(src);
    if constexpr (B_ROUNDS > 1) rb1 = *reinterpret_cast<const uint4*>(src + b_gstep);
  };
  auto swrite_B = [&](int buf) __attribute__((always_inline)) {
    *reinterpret_cast<uint4*>(smem + oB + buf * B_BYTES + b_loff0) = rb0;
    if constexpr (B_ROUNDS > 1) *reinterpret_cast<uint4*>(smem + oB + buf * B_BYTES + b_loff0 + 64 * ROWB) = rb1;
  };
  short8 fa[MT], fb[NT];
  auto load_frags = [&](int toff, int buf, int ks) __attribute__((always_inline)) {
    const int pa = oA + toff + aoff0 + ks * 32;
    const int pb = oB + buf * B_BYTES + boff0 + ks * 32;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) fa[mt] = *reinterpret_cast<const short8*>(smem + pa + aoff_rel(mt));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) fb[nt] = *reinterpret_cast<const short8*>(smem + pb + nt * 32 * ROWB);
  };
  auto mfmas = [&]() __attribute__((always_inline)) {
#ifdef IM2IM_PP_PRIO
    __builtin_amdgcn_s_setprio(2);                     // A/B build: this wave's priority up for the length of its M phase (guide T5)
#endif
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
        acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16x8(fa[mt]), as_bf16x8(fb[nt]), acc[mt][nt], 0, 0, 0);
  };
  // the service instructions of an M phase go BETWEEN this wave's MFMAs (a 32x32x16 MFMA occupies the matrix pipe for 32
  // cycles; up to ~5 other instructions issue in that gap for free, MI355X_MICROARCH.md), not before or after them: measured
  // with s_memtime stamps, the same instructions in the L phase -- beside the PARTNER's MFMAs -- doubled that phase
  auto interleave = [&](auto reads_first) __attribute__((always_inline)) {
    // LDS reads the service code needs (offsets, BatchNorm coefficients) go out first, their consumers come after two MFMAs
    constexpr int R = decltype(reads_first)::value;
    if constexpr (R > 0) __builtin_amdgcn_sched_group_barrier(0x100, R, 0);
    static_for<MT * NT>([&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                              // one MFMA
      if constexpr (k >= 1 || R == 0)
        __builtin_amdgcn_sched_group_barrier(0x002 | 0x004 | 0x010 | 0x080, 6, 0);    // up to six VALU / SALU / VMEM / DS
    });
  };

  const int nchunks = aCi / KC;
  // ---------------------------------------------------------------- prologue: offsets table, first halo, first two weight tiles
#pragma unroll
  for (int i = 0; i < A_ROUNDS; ++i) {
    int goff, loff;
    piece(i, goff, loff);
    *reinterpret_cast<int2*>(smem + oOFF + (i * 256 + t) * 8) = make_int2(goff, loff);
  }
  // (compile-time piece indices everywhere: an `ra[i]` with a loop variable kept the array on the stack)
  static_for<A_ROUNDS>([&](auto ic) __attribute__((always_inline)) { halo_load(decltype(ic)::value, 0); });
  gload_B(0);
  load_coeffs(0);
  static_for<A_ROUNDS>([&](auto ic) __attribute__((always_inline)) {
    halo_xform(decltype(ic)::value); halo_write(decltype(ic)::value); __builtin_amdgcn_sched_barrier(0);
  });
  swrite_B(0);
  gload_B(1);
  PP_STAMP();
  wg_barrier(); PP_STAMP();
  if (g == 1) wg_barrier();                          // group 1 runs one phase behind group 0
  // ---------------------------------------------------------------- main loop
  // ONE loop over pairs of chunks (9 taps is odd, so the weight-buffer parity of a chunk's first tap alternates: two
  // instantiations of the body).  Ci % 64 == 0 (launcher), so chunks come in pairs.  Every tap does the same things --
  // the weight prefetch of the last two taps of the last chunk re-reads the last tile (clamped index, harmless) -- and
  // only the halo service is under a (scalar, uniform) branch: few code shapes, one accumulator assignment for the
  // register allocator to keep (an earlier version with five specialised bodies spilled accumulators).
  const int last_gt = nchunks * 9 - 1;
  auto chunk_body = [&](auto parity, int chunk) __attribute__((always_inline)) {
    constexpr int P0 = decltype(parity)::value;
    const int nxt = min(chunk + 1, nchunks - 1);
    static_for<9>([&](auto tapc) __attribute__((always_inline)) {
      constexpr int tap = decltype(tapc)::value;
      constexpr int buf = (P0 + tap) & 1;
      const int gt = chunk * 9 + tap;
      // A tap is two k-steps of 16 channels; each k-step is one L phase (its MT + NT fragments out of LDS, nothing else) and
      // one M phase (MT*NT MFMAs with the staging work of this group in their issue gaps).
      const int toff = (tap / 3) * HROWB + (tap % 3) * ROWB;
      load_frags(toff, buf, 0);
      wg_barrier(); PP_STAMP();
      // next chunk's halo: piece `tap` leaves for the registers, piece tap-2 is transformed.  No branch around it (a branch
      // would end the scheduling region and keep these instructions out of the MFMA gaps): on the last chunk the same
      // chunk is staged once more (clamped index; its lines are in L2) and never read.
      if constexpr (tap < A_ROUNDS) halo_load(tap, nxt);
      if constexpr (tap == 1) load_coeffs(nxt);
      if constexpr (tap >= 2 && tap - 2 < A_ROUNDS) halo_xform(tap - 2);
      mfmas();
      interleave(std::integral_constant<int, (tap < A_ROUNDS ? 1 : 0) + ((tap >= 2 && tap - 2 < A_ROUNDS) ? 1 : 0) + ((tap == 1 && LAZY) ? 4 : 0)>{});
      wg_barrier(); PP_STAMP();
      load_frags(toff, buf, 1);
      wg_barrier(); PP_STAMP();
      swrite_B(buf ^ 1);                               // weights of the next tap (loaded one tap ago) into the other buffer
      gload_B(min(gt + 2, last_gt));                   // and the ones after that on their way
      if constexpr (tap == 8) {                        // every wave of the group has read the old halo (barrier above)
        static_for<A_ROUNDS>([&](auto ic) __attribute__((always_inline)) { halo_write(decltype(ic)::value); });
      }
      mfmas();
      interleave(std::integral_constant<int, (tap == 8 ? A_ROUNDS : 0)>{});
      wg_barrier(); PP_STAMP();
    });
  };
  for (int chunk = 0; chunk < nchunks; chunk += 2) {
    chunk_body(std::integral_constant<int, 0>{}, chunk);
    chunk_body(std::integral_constant<int, 1>{}, chunk + 1);
  }
  if (g == 0) wg_barrier();
#ifdef IM2IM_PP_TRACE
  if (tr_on && lane == 0) {
    unsigned long long* dst = reinterpret_cast<unsigned long long*>(a.bn_partial) + g * 2049;
    dst[0] = (unsigned long long)tr_n;
    for (int i = 0; i < tr_n && i < 2048; ++i) dst[1 + i] = *reinterpret_cast<unsigned long long*>(smem + tr_base + 8 * i);
  }
#endif

  // ---------------------------------------------------------------- epilogue (conv_mfma.hip's, per group)
  constexpr int EPR = WCOLS / EPP;
  constexpr int ROWS_PER_PASS = 64 / EPR;
  const int ncol = n0 + wn * WCOLS;
  const bool to_hi = a.y_hi != nullptr && ncol >= a.Co_lo;
  T* __restrict__ yg = reinterpret_cast<T*>(to_hi ? a.y_hi : a.y) + (to_hi ? ncol - a.Co_lo : ncol);
  const int ystride = a.y_hi == nullptr ? a.Co : (to_hi ? a.Co - a.Co_lo : a.Co_lo);
  constexpr bool want_stats = (EPI == 1);
  constexpr int PASSES = WROWS / ROWS_PER_PASS;
  __syncthreads();                                           // both groups are done with their operand buffers
  const int owb = gbase + wave * WBYTES;
  float st_n[NT], st_m[NT], st_q[NT];
  const bool tile_full = (b0 + TB <= aB) && (y0 + TH <= aH) && (x0 + TW <= aW);
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
          if (bb < aB && yy < aH && xx < aW) okmask |= 1ull << (mt * 16 + r);
        }
      cnt = (float)__popcll(okmask);
    }
  }
  auto convert_tile = [&](auto full_tag) __attribute__((always_inline)) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nl = (wn * NT + nt) * 32 + l31;
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
          *reinterpret_cast<T*>(smem + owb + row * WP + (nt * 32 + l31) * 2) = tv;
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
    const int pc = lane % EPR;
    const uint4 v = *reinterpret_cast<const uint4*>(smem + owb + row * WP + pc * 16);
    const int m = wm * WROWS + row;
    const int bb = b0 + m / MI, yy = y0 + (m % MI) / TW, xx = x0 + m % TW;
    if (!ghost && bb < aB && yy < aH && xx < aW) {
      const size_t off = (((size_t)bb * aH + yy) * aW + xx) * ystride + pc * EPP;
      __builtin_nontemporal_store(__builtin_bit_cast(u32x4, v), reinterpret_cast<u32x4*>(yg + off));
    }
  }
  if (want_stats) {
    __syncthreads();                                         // the stats scratch aliases wave 0's tile
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int nl = (wn * NT + nt) * 32 + l31;
      float n = st_n[nt], m = st_m[nt], q = st_q[nt];
      merge_moments_f32(n, m, q, __shfl_xor(n, 32, 64), __shfl_xor(m, 32, 64), __shfl_xor(q, 32, 64));
      if (half == 0) { lds_f32(gbase + 4 * ((wm * BN + nl) * 3 + 0)) = n; lds_f32(gbase + 4 * ((wm * BN + nl) * 3 + 1)) = m; lds_f32(gbase + 4 * ((wm * BN + nl) * 3 + 2)) = q; }
    }
    __syncthreads();
    if (t < BN && !ghost) {
      float n = lds_f32(gbase + 4 * (t * 3 + 0)), m = lds_f32(gbase + 4 * (t * 3 + 1)), q = lds_f32(gbase + 4 * (t * 3 + 2));
#pragma unroll
      for (int i = 1; i < WM; ++i)
        merge_moments_f32(n, m, q, lds_f32(gbase + 4 * ((i * BN + t) * 3 + 0)), lds_f32(gbase + 4 * ((i * BN + t) * 3 + 1)), lds_f32(gbase + 4 * ((i * BN + t) * 3 + 2)));
      float* st = a.stats + (size_t)tile_id * 3 * a.Co;
      st[n0 + t] = m;
      st[a.Co + n0 + t] = q;
      st[2 * a.Co + n0 + t] = n;
    }
  }
}

template <int TB, int TH, int TW, int BN, int WM, int WN, int EPI, bool LAZY>
int launch_pp_epi(const ConvArgs& a_in, hipStream_t stream) {
  ConvArgs a = a_in;
  a.tilesY = (int)cdiv(a.H, TH);
  a.tilesX = (int)cdiv(a.W, TW);
  constexpr int ROWB = 80;
  constexpr size_t main_fixed = (size_t)TB * (TH + 2) * ((TW + 2) * ROWB + 96) + (size_t)2 * BN * ROWB;
  constexpr size_t epi = (size_t)4 * (TB * TH * TW / WM) * ((BN / WN) * 2 + 16);
  static_assert(epi >= (size_t)WM * BN * 3 * 4, "stats scratch fits");
  constexpr size_t a_rounds = ((size_t)TB * (TH + 2) * (TW + 2) * 4 + 255) / 256;
  const size_t main_b = main_fixed + (LAZY ? (size_t)2 * a.Ci * sizeof(float) : 0) + a_rounds * 256 * 8 + 4096;
  size_t gb = main_b > epi ? main_b : epi;
  gb = (gb + 255) & ~(size_t)255;
  const size_t smem = 2 * gb;
  if (smem > 160 * 1024 || a.Ci % 64) return 1;               // very wide lazy inputs / odd chunk counts: the 4-wave kernel
  a.pp_group_bytes = (int)gb;
  a.pp_ntiles = (int)(cdiv(a.B, TB) * a.tilesY * a.tilesX);
  a.pp_flags = (conv_pp_mode() >> 2) & 1;
  auto kern = conv_pp_kernel<TB, TH, TW, BN, WM, WN, EPI, LAZY>;
  static size_t attr_set = 0;
  if (smem > attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = smem;
  }
  const int ncob = a.Co / BN;
  const int band = a.pp_ntiles >> 3;
  const int ppx = (band * ncob + 1) >> 1;
  const int left = (a.pp_ntiles - band * 8) * ncob;
  dim3 grid((unsigned)(ppx * 8 + ((left + 1) >> 1)));
  hipLaunchKernelGGL(kern, grid, dim3(512), smem, stream, a);
  return check_launch("conv_pp_kernel");
}

template <int TB, int TH, int TW, int BN, int WM, int WN>
int launch_pp(const ConvArgs& a, hipStream_t stream) {
  const bool lazy = a.in_ss || a.in_ss_hi;
  if (a.stats) return lazy ? launch_pp_epi<TB, TH, TW, BN, WM, WN, 1, true>(a, stream) : launch_pp_epi<TB, TH, TW, BN, WM, WN, 1, false>(a, stream);
  if (a.scale) return lazy ? launch_pp_epi<TB, TH, TW, BN, WM, WN, 2, true>(a, stream) : launch_pp_epi<TB, TH, TW, BN, WM, WN, 2, false>(a, stream);
  return lazy ? launch_pp_epi<TB, TH, TW, BN, WM, WN, 0, true>(a, stream) : launch_pp_epi<TB, TH, TW, BN, WM, WN, 0, false>(a, stream);
}

}  // namespace

namespace im2im {

// option word: bit 0 = use the ping-pong kernel for the 128-channel-wide tiles, bit 1 = also for the 64-wide ones
static int g_conv_pp = -1;
int conv_pp_mode() {
  if (g_conv_pp < 0) {
    const char* e = std::getenv("IM2IM_CONV_PP");
    g_conv_pp = e ? std::atoi(e) : 0;
  }
  return g_conv_pp;
}
void set_conv_pp_mode(int m) { g_conv_pp = m; }

int launch_conv_pp(const ConvArgs& a, hipStream_t stream) {
  const int mode = conv_pp_mode();
#ifndef IM2IM_PP_TRACE
  if (a.bn_partial) return 1;
#endif
  if (!mode || a.in_ss_img) return 1;
  const TileChoice tc = pick_tile(a.B, a.H, a.W, a.Co);
  if ((tc.tb == 1 && tc.th != 16) || (tc.tb != 1 && tc.tb != 4)) return 1;                       // a tile shape only the 4-wave kernel has (statistics rows follow it)
  if (tc.bn == 128 && (mode & 1)) {
    if (tc.tb == 1) return launch_pp<1, 16, 16, 128, 2, 2>(a, stream);
    return launch_pp<4, 8, 8, 128, 2, 2>(a, stream);
  }
  if (tc.bn == 64 && (mode & 2)) {
    if (tc.tb == 1) return launch_pp<1, 16, 16, 64, 4, 1>(a, stream);
    return launch_pp<4, 8, 8, 64, 4, 1>(a, stream);
  }
  return 1;
}

}  // namespace im2im
