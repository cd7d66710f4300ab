// Which ingredient of a real MFMA main loop costs the throughput?  Baseline = ldsprobe's loop at 6 ds_read_b128 per 8 MFMAs
// (conv_igemm_kernel's ratio); then one at a time, per 16 MFMAs (= one conv tap of a 128x64 wave tile):
//   SYNC : a workgroup barrier            LDSW : 2 ds_write_b128 (the weight tile)          GLD : 2 global 16-B loads, waited for
//   VALU : 48 fp32 FMAs (lazy BatchNorm+ReLU of a halo piece is ~32 per 9 taps -- this is a deliberate overdose)
// Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(2);} }while(0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

template <bool SYNC, bool LDSW, bool GLD, bool VALU>
__global__ __launch_bounds__(256, 2) void k_loop(const uint4* __restrict__ seed, const uint4* __restrict__ gbuf, float* __restrict__ out, int iters) {
  __shared__ uint4 lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed[i];
  __syncthreads();
  bf16x8_t f[6];
  for (int i = 0; i < 6; ++i) { uint4 v = lds[(threadIdx.x + 64 * i) & 4095]; __builtin_memcpy(&f[i], &v, 16); }
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int base = threadIdx.x & 63;
  uint4 g0 = make_uint4(0, 0, 0, 0), g1 = g0;
  float va[8] = {1.f, 2.f, 3.f, 4.f, 5.f, 6.f, 7.f, 8.f};
  const uint4* gp = gbuf + (size_t)blockIdx.x * 256 * 64 + threadIdx.x;
  for (int it = 0; it < iters; ++it) {
    if (GLD) { g0 = gp[(it & 31) * 512]; g1 = gp[(it & 31) * 512 + 256]; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const uint4 v = lds[(base + 64 * r + 448 * ((2 * it + half) & 7)) & 4095];
        __builtin_memcpy(&f[r], &v, 16);
      }
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], f[4], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], f[4], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], f[4], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[3], f[4], acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[0], f[5], acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[1], f[5], acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[2], f[5], acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[3], f[5], acc[7], 0, 0, 0);
    }
    if (VALU) {
#pragma unroll
      for (int k = 0; k < 48; ++k) va[k & 7] = __builtin_fmaf(va[k & 7], 1.0001f, 0.5f);
    }
    if (LDSW) {
      uint4 w0 = g0, w1 = g1;
      if (!GLD) { w0.x = it; w1.y = it; }
      lds[3584 + (threadIdx.x & 255)] = w0;                      // a region the reads above do not need to be ordered against
      lds[3840 + (threadIdx.x & 255)] = w1;
    }
    if (SYNC) __syncthreads();
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int k = 0; k < 8; ++k) s += va[k];
  s += (float)(g0.x + g1.y);
  if (s == 12345.678f) out[0] = s;
}

template <bool SYNC, bool LDSW, bool GLD, bool VALU> static void run(const uint4* seed, const uint4* gbuf, float* o, const char* what) {
  const int blocks = 512, iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_loop<SYNC, LDSW, GLD, VALU>), dim3(blocks), dim3(256), 0, 0, seed, gbuf, o, iters);
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((k_loop<SYNC, LDSW, GLD, VALU>), dim3(blocks), dim3(256), 0, 0, seed, gbuf, o, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  const double flop = (double)blocks * 4 * iters * 16 * 2.0 * 32 * 32 * 16;
  printf("%-44s %8.3f ms  %7.1f TFLOP/s\n", what, ms, flop / ms / 1e9);
}

int main() {
  std::vector<uint32_t> h(4096 * 4);
  srand(7);
  for (auto& x : h) { uint32_t lo = 0x3f00 | (rand() & 0x80ff), hi = 0x3f00 | (rand() & 0x80ff); x = lo | (hi << 16); }
  uint4 *seed, *gbuf; float* o;
  const size_t gn = (size_t)512 * 256 * 64;
  CK(hipMalloc(&seed, h.size() * 4)); CK(hipMalloc(&o, 64)); CK(hipMalloc(&gbuf, gn * 16));
  CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(gbuf, 0x3f, gn * 16));
  run<false, false, false, false>(seed, gbuf, o, "baseline (6 reads / 8 MFMA)");
  run<true, false, false, false>(seed, gbuf, o, "+ barrier per 16 MFMA");
  run<false, true, false, false>(seed, gbuf, o, "+ 2 ds_write_b128 per 16 MFMA");
  run<false, false, true, false>(seed, gbuf, o, "+ 2 global loads per 16 MFMA");
  run<false, false, false, true>(seed, gbuf, o, "+ 48 FMA per 16 MFMA");
  run<true, true, false, false>(seed, gbuf, o, "+ barrier + ds_write");
  run<true, true, true, false>(seed, gbuf, o, "+ barrier + ds_write + global loads");
  run<true, true, true, true>(seed, gbuf, o, "+ all four");
  run<false, false, false, false>(seed, gbuf, o, "baseline again");
  return 0;
}
