// What does operand delivery cost the matrix cores on gfx950?  One loop body = 8 x v_mfma_f32_32x32x16_bf16 (a 128x64 wave
// tile's k-step, as conv_igemm_kernel) fed by R ds_read_b128 of random bf16 data (R = 6 is what that kernel issues:
// 4 A fragments + 2 B fragments), 2 workgroups of 4 waves per CU like the product kernel.  TFLOP/s vs R separates "the
// MFMA pipe is power-limited by itself" from "the LDS operand traffic is what the power goes to".  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(2);} }while(0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

template <int R>
__global__ __launch_bounds__(256, 2) void k_feed(const uint4* __restrict__ seed, float* __restrict__ out, int iters) {
  __shared__ uint4 lds[4096];                                  // 64 KB of random bits
  for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = seed[i];
  __syncthreads();
  bf16x8_t f[6];
  for (int i = 0; i < 6; ++i) { uint4 v = lds[(threadIdx.x + 64 * i) & 4095]; __builtin_memcpy(&f[i], &v, 16); }
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int base = threadIdx.x & 63;                                 // consecutive lanes -> consecutive 16 B: conflict-free
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const uint4 v = lds[(base + 64 * r + 448 * (it & 7)) & 4095];
      if (r < 6) __builtin_memcpy(&f[r], &v, 16);
      else { uint4 o; __builtin_memcpy(&o, &f[r % 6], 16); o.x ^= v.x; o.y ^= v.y; o.z ^= v.z; o.w ^= v.w; __builtin_memcpy(&f[r % 6], &o, 16); }
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
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <int R> static void run(const uint4* seed, float* o, int blocks, int iters, const char* what) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_feed<R>, dim3(blocks), dim3(256), 0, 0, seed, o, iters);
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k_feed<R>, dim3(blocks), dim3(256), 0, 0, seed, o, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  const double flop = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
  printf("%-8s ds_read_b128 per 8 MFMA = %2d  (%.2f per MFMA): %8.3f ms  %7.1f TFLOP/s\n", what, R, R / 8.0, ms, flop / ms / 1e9);
}

int main() {
  std::vector<uint32_t> h(4096 * 4);
  srand(7);
  uint4* seed; float* o;
  CK(hipMalloc(&seed, h.size() * 4)); CK(hipMalloc(&o, 64));
  for (int pass = 0; pass < 2; ++pass) {
    // bf16 values with random sign / mantissa and exponents around 1.0 (no NaN / Inf), or all zero
    for (auto& x : h) { uint32_t lo = 0x3f00 | (rand() & 0x80ff), hi = 0x3f00 | (rand() & 0x80ff); x = pass == 0 ? (lo | (hi << 16)) : 0u; }
    CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    const char* what = pass == 0 ? "random" : "zeros";
    const int blocks = 512, iters = 4000;
    run<0>(seed, o, blocks, iters, what); run<2>(seed, o, blocks, iters, what); run<4>(seed, o, blocks, iters, what);
    run<6>(seed, o, blocks, iters, what); run<8>(seed, o, blocks, iters, what); run<12>(seed, o, blocks, iters, what);
    run<16>(seed, o, blocks, iters, what);
  }
  return 0;
}
