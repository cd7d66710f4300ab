// Would the conv main loop be faster with the WEIGHT fragments read straight from L2 into registers (fragment-major packed
// weights, 1 KiB per wave-load, prefetched one tap ahead) instead of staged through LDS behind a per-tap barrier?
// Both loops: per tap 16 MFMAs of a 128 px x 64 co wave tile, A fragments by ds_read_b128 from a halo image; per 9 taps one
// halo chunk (6 HBM loads + 6 ds_write_b128 per thread, two barriers).
//   V0 (conv_igemm_kernel today): + 2 weight loads per thread two taps ahead, 2 ds_write_b128, a barrier, 4 more ds_read_b128
//   V1: + 4 fragment loads per lane one tap ahead, nothing else
// Not part of the product.   hipcc --offload-arch=gfx950 -O3 -o directb_probe directb_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(2);} }while(0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
static __device__ __forceinline__ bf16x8_t asf(uint4 v) { bf16x8_t f; __builtin_memcpy(&f, &v, 16); return f; }

template <int V>
__global__ __launch_bounds__(256, 2) void k_loop(const uint4* __restrict__ seed, const uint4* __restrict__ wbuf, const uint4* __restrict__ xbuf,
                                                  float* __restrict__ out, int chunks) {
  __shared__ uint4 lds[3072];                      // 0..2047 halo image, 2048..3071 two weight buffers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave & 1;
  for (int i = tid; i < 3072; i += 256) lds[i] = seed[i];
  __syncthreads();
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  uint4 ra[6], rb[2][2];
  uint4 fb[2][4];                                  // V1: this tap's / the next tap's four weight fragments
  const uint4* xp = xbuf + (size_t)blockIdx.x * 6 * 256 * 8 + tid;          // HBM stream, private to the workgroup
  const uint4* wp = wbuf + tid;                    // V0: [tap][512 pieces]
  const uint4* wf = wbuf + wn * 2 * 64 + lane;     // V1: [tap][ks][co block 4][lane 64]
  for (int i = 0; i < 6; ++i) ra[i] = xp[i * 256];
  if (V == 0) { rb[0][0] = wp[0]; rb[0][1] = wp[256]; rb[1][0] = wp[512]; rb[1][1] = wp[768]; }
  else for (int j = 0; j < 4; ++j) fb[0][j] = wf[(j >> 1) * 256 + (j & 1) * 64];
  int t = 0;                                       // running tap index (weights wrap at 72 taps = 8 chunks)
  for (int c = 0; c < chunks; ++c) {
    if (c) __syncthreads();
#pragma unroll
    for (int i = 0; i < 6; ++i) lds[(tid + 256 * i) & 2047] = ra[i];
    if (V == 1) __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 9; ++tap, ++t) {
      const int set = tap & 1;
      const int tw = (t + (V == 0 ? 2 : 1)) % 72;
      if (V == 0) {
        lds[2048 + set * 512 + tid] = rb[set][0];
        lds[2048 + set * 512 + 256 + tid] = rb[set][1];
        __syncthreads();
        rb[set][0] = wp[tw * 512]; rb[set][1] = wp[tw * 512 + 256];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) fb[set ^ 1][j] = wf[tw * 512 + (j >> 1) * 256 + (j & 1) * 64];
      }
      if (tap == 6) {
#pragma unroll
        for (int i = 0; i < 6; ++i) ra[i] = xp[(((c + 1) & 7) * 6 + i) * 256];
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t fa[4], b0, b1;
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[m] = asf(lds[(lane + 80 * m + 5 * tap + 37 * ks + 320 * (wave >> 1)) & 2047]);
        if (V == 0) {
          b0 = asf(lds[2048 + set * 512 + ((lane + 64 * ks + 128 * wn) & 511)]);
          b1 = asf(lds[2048 + set * 512 + ((lane + 64 * ks + 128 * wn + 256) & 511)]);
        } else { b0 = asf(fb[set][ks * 2]); b1 = asf(fb[set][ks * 2 + 1]); }
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          acc[m * 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], b0, acc[m * 2], 0, 0, 0);
          acc[m * 2 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[m], b1, acc[m * 2 + 1], 0, 0, 0);
        }
      }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

template <int V> static void run(const uint4* seed, const uint4* w, const uint4* x, float* o, const char* what) {
  const int blocks = 512, chunks = 64;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((k_loop<V>), dim3(blocks), dim3(256), 0, 0, seed, w, x, o, chunks);
  CK(hipEventRecord(e0));
  for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL((k_loop<V>), dim3(blocks), dim3(256), 0, 0, seed, w, x, o, chunks);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
  const double flop = (double)blocks * 4 * chunks * 9 * 16 * 2.0 * 32 * 32 * 16;
  printf("%-60s %8.3f ms  %7.1f TFLOP/s\n", what, ms, flop / ms / 1e9);
}

int main() {
  std::vector<uint32_t> h((size_t)73 * 512 * 4);
  srand(7);
  for (auto& x : h) { uint32_t lo = 0x3f00 | (rand() & 0x80ff), hi = 0x3f00 | (rand() & 0x80ff); x = lo | (hi << 16); }
  uint4 *seed, *w, *x; float* o;
  const size_t xn = (size_t)512 * 6 * 256 * 8;
  CK(hipMalloc(&seed, 3072 * 16)); CK(hipMalloc(&o, 64)); CK(hipMalloc(&w, h.size() * 4)); CK(hipMalloc(&x, xn * 16));
  CK(hipMemcpy(seed, h.data(), 3072 * 16, hipMemcpyHostToDevice));
  CK(hipMemcpy(w, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  std::vector<uint32_t> hx(xn * 4);
  for (auto& v : hx) { uint32_t lo = 0x3f00 | (rand() & 0x80ff), hi = 0x3f00 | (rand() & 0x80ff); v = lo | (hi << 16); }
  CK(hipMemcpy(x, hx.data(), xn * 16, hipMemcpyHostToDevice));
  for (int rep = 0; rep < 2; ++rep) {
    run<0>(seed, w, x, o, "V0 weights through LDS (2 loads, 2 ds_write, barrier per tap)");
    run<1>(seed, w, x, o, "V1 weight fragments straight from L2 (4 loads per tap)");
  }
  return 0;
}
