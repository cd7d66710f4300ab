// Does phase-locking two wave groups of one 512-thread workgroup pay on gfx950?  Synthetic loop with conv_igemm's operand
// traffic (6 ds_read_b128 per 8 MFMAs of 32x32x16 bf16, random operands), one workgroup per CU:
//   FREE   : 8 waves, no barriers, compiler-interleaved reads and MFMAs (the upper bound of the unsynchronised form)
//   LOCK   : every wave: [reads | barrier | MFMAs | barrier], both groups in the SAME phase
//   PINGPONG: the same, group 1 one barrier behind group 0 -> one wave per SIMD in MFMAs while its partner reads
// with PH = 8 or 16 MFMAs per phase.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(2);} }while(0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

__device__ __forceinline__ void bar() {
  __builtin_amdgcn_sched_barrier(0);
  __syncthreads();
  __builtin_amdgcn_sched_barrier(0);
}

// MODE 0 = FREE, 1 = LOCK, 2 = PINGPONG;  KS = k-steps (of 8 MFMAs / 6 reads) per phase;  GMAP: 0 = groups are waves 0-3 / 4-7, 1 = even / odd
template <int MODE, int KS, int GMAP>
__global__ __launch_bounds__(512, 2) void k_pp(const uint4* __restrict__ seed, float* __restrict__ out, int iters) {
  extern __shared__ uint4 lds[];
  for (int i = threadIdx.x; i < 8192; i += 512) lds[i] = seed[i & 4095];
  __syncthreads();
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int g = GMAP ? (wv & 1) : (wv >> 2);
  bf16x8_t f[KS][6];
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  const int base = (threadIdx.x & 63) + g * 4096;
  if (MODE == 2 && g == 1) bar();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int r = 0; r < 6; ++r) {
        const uint4 v = lds[g * 4096 + ((base + 64 * r + 448 * ((it * KS + ks) & 7)) & 4095)];
        __builtin_memcpy(&f[ks][r], &v, 16);
      }
    if (MODE) bar();
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][0], f[ks][4], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][1], f[ks][4], acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][2], f[ks][4], acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][3], f[ks][4], acc[3], 0, 0, 0);
      acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][0], f[ks][5], acc[4], 0, 0, 0);
      acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][1], f[ks][5], acc[5], 0, 0, 0);
      acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][2], f[ks][5], acc[6], 0, 0, 0);
      acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f[ks][3], f[ks][5], acc[7], 0, 0, 0);
    }
    if (MODE) bar();
  }
  if (MODE == 2 && g == 0) bar();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE, int KS, int GMAP> static void run(const uint4* seed, float* o, const char* what) {
  const int blocks = 256, iters = 4000 / KS;
  auto kern = k_pp<MODE, KS, GMAP>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 131072));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 131072, 0, seed, o, iters);
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 131072, 0, seed, o, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  const double flop = (double)blocks * 8 * iters * KS * 8 * 2.0 * 32 * 32 * 16;
  printf("%-64s %8.3f ms  %7.1f TFLOP/s\n", what, ms, flop / ms / 1e9);
}

int main() {
  std::vector<uint32_t> h(4096 * 4);
  srand(7);
  for (auto& x : h) { uint32_t lo = 0x3f00 | (rand() & 0x80ff), hi = 0x3f00 | (rand() & 0x80ff); x = lo | (hi << 16); }
  uint4* seed; float* o;
  CK(hipMalloc(&seed, h.size() * 4)); CK(hipMalloc(&o, 64));
  CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  run<0, 1, 0>(seed, o, "FREE: 8 waves, no barriers (6 reads / 8 MFMA)");
  run<1, 1, 0>(seed, o, "LOCK: both groups in phase, 8 MFMA per phase");
  run<2, 1, 0>(seed, o, "PINGPONG: groups = waves 0-3 / 4-7, 8 MFMA per phase");
  run<2, 1, 1>(seed, o, "PINGPONG: groups = even / odd waves, 8 MFMA per phase");
  run<1, 2, 0>(seed, o, "LOCK: 16 MFMA per phase");
  run<2, 2, 0>(seed, o, "PINGPONG: waves 0-3 / 4-7, 16 MFMA per phase");
  run<2, 2, 1>(seed, o, "PINGPONG: even / odd waves, 16 MFMA per phase");
  run<2, 4, 0>(seed, o, "PINGPONG: waves 0-3 / 4-7, 32 MFMA per phase");
  run<0, 1, 0>(seed, o, "FREE again");
  return 0;
}
