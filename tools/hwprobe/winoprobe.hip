// Gate experiment for Winograd F(2x2, 3x3) on the 64 -> 64 channel full-resolution layer (320 x 320, batch 78).
// NOT a convolution: a TIMING SKELETON that moves the bytes and issues the instruction mix a fused Winograd kernel would --
//   halo (10 x 18 px x 64 ch bf16) -> LDS, input transform B^T d B in fp32 on the VALU (32 adds per tile and channel), V[16][32 tiles][64 ci]
//   bf16 -> LDS, [STAGE 2: the 16 batched GEMMs 32 tiles x 64 ci x 64 co on v_mfma_f32_32x32x16_bf16, A fragments from V in LDS,
//   the transformed weights U resident in registers (persistent workgroups), accumulators -> LDS fp32], output transform A^T M A
//   (24 adds per tile and channel), bf16 rows -> HBM --
// so that the question "can it reach 1,100 TF algorithmic = 0.535 ms for the 589 GFLOP launch?" has a measured upper bound before
// anyone writes the real kernel.  Results are not checked (the weights are random bits and the tap order is not a convolution's).
//   hipcc --offload-arch=gfx950 -O3 -o winoprobe winoprobe.hip && ./winoprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(2);} }while(0)
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;

__device__ __forceinline__ void unpack8(const uint4& v, float* f) {
  const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(w[i] << 16); f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
}
__device__ __forceinline__ uint4 pack8(const float* f) {
  uint32_t w[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const uint32_t a = __float_as_uint(f[2 * i]), b = __float_as_uint(f[2 * i + 1]);
    w[i] = ((a + 0x7fffu + ((a >> 16) & 1u)) >> 16) | ((b + 0x7fffu + ((b >> 16) & 1u)) & 0xffff0000u);
  }
  return make_uint4(w[0], w[1], w[2], w[3]);
}

constexpr int TH = 8, TW = 16, HP = TH + 2, WP = TW + 2, NT = (TH / 2) * (TW / 2);       // 32 Winograd tiles per block
constexpr int HALO_PIECES = HP * WP * 8;                                              // 16-byte pieces (8 channels each)

// STAGE: 1 = transforms + memory only, 2 = + the MFMAs (persistent workgroups, U in registers)
template <int STAGE>
__global__ __launch_bounds__(256, 1) void wino_skel(const uint4* __restrict__ x, uint4* __restrict__ y, const uint4* __restrict__ U,
                                                     int B, int H, int W, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  uint4* halo = reinterpret_cast<uint4*>(smem);                           // [HP][WP][8]              23,040 B
  uint4* V = halo + HALO_PIECES;                                          // [16][NT][8]              65,536 B
  float* M = reinterpret_cast<float*>(V + 16 * NT * 8);                   // [16][NT][32 co] fp32     65,536 B (one half of the channels per round)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tilesX = W / TW, tilesY = H / TH;
  // stage 2: this wave's transformed weights, 4 xi x 2 column blocks x 4 k-steps fragments of 16 B per lane = 128 registers
  bf16x8_t u[4][2][4];
  if constexpr (STAGE == 2) {
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) { const uint4 v = U[(((wave * 4 + a) * 2 + cb) * 4 + ks) * 64 + lane]; __builtin_memcpy(&u[a][cb][ks], &v, 16); }
  }
  constexpr int NPRE = (HALO_PIECES + 255) / 256;
  uint4 pre[NPRE];
  auto gload = [&](int tl) {
    int t = tl;
    const int bx = t % tilesX; t /= tilesX;
    const int by = t % tilesY;
    const int b = t / tilesY;
#pragma unroll
    for (int i = 0; i < NPRE; ++i) {
      const int p = tid + 256 * i;
      const int py = p / (WP * 8), r = p % (WP * 8), px = r / 8, c8 = r % 8;
      const int gy = by * TH - 1 + py, gx = bx * TW - 1 + px;
      pre[i] = make_uint4(0, 0, 0, 0);
      if (p < HALO_PIECES && gy >= 0 && gy < H && gx >= 0 && gx < W) pre[i] = x[(((size_t)b * H + gy) * W + gx) * 8 + c8];
    }
  };
  if ((int)blockIdx.x < ntiles) gload(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    int t = tile;
    const int bx = t % tilesX; t /= tilesX;
    const int by = t % tilesY;
    const int b = t / tilesY;
    const int y0 = by * TH, x0 = bx * TW;
    __syncthreads();                                                      // the previous tile's readers of halo / V / M are done
#pragma unroll
    for (int i = 0; i < NPRE; ++i) if (tid + 256 * i < HALO_PIECES) halo[tid + 256 * i] = pre[i];
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) gload(tile + gridDim.x);          // the next tile's halo travels under this tile's work
    {   // input transform: thread = (Winograd tile wt, 8-channel group g)
      const int wt = tid >> 3, g = tid & 7;
      const int ty = wt / (TW / 2), tx = wt % (TW / 2);
      float d[4][4][8];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) unpack8(halo[((2 * ty + i) * WP + 2 * tx + j) * 8 + g], d[i][j]);
#pragma unroll
      for (int j = 0; j < 4; ++j)                                          // B^T d: rows
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float a0 = d[0][j][k], a1 = d[1][j][k], a2 = d[2][j][k], a3 = d[3][j][k];
          d[0][j][k] = a0 - a2; d[1][j][k] = a1 + a2; d[2][j][k] = a2 - a1; d[3][j][k] = a1 - a3;
        }
#pragma unroll
      for (int i = 0; i < 4; ++i) {                                        // (.) B: columns, then to bf16 and into V[xi = 4 i + j][wt][g]
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float a0 = d[i][0][k], a1 = d[i][1][k], a2 = d[i][2][k], a3 = d[i][3][k];
          d[i][0][k] = a0 - a2; d[i][1][k] = a1 + a2; d[i][2][k] = a2 - a1; d[i][3][k] = a1 - a3;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) V[((4 * i + j) * NT + wt) * 8 + g] = pack8(d[i][j]);
      }
    }
    __syncthreads();
#pragma unroll
    for (int half = 0; half < 2; ++half) {                                 // output channels [32 half, 32 half + 32)
      if constexpr (STAGE == 2) {
        // this wave's 4 transform points: [32 tiles x 64 ci] x [64 ci x 32 co], 4 k-steps each, accumulators -> M
        const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const int xi = wave * 4 + a;
          f32x16_t acc;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) {
            const uint4 av = V[(xi * NT + l31) * 8 + ks * 2 + hi];
            bf16x8_t af; __builtin_memcpy(&af, &av, 16);
            acc = half == 0 ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, u[a][0][ks], acc, 0, 0, 0)
                            : __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, u[a][1][ks], acc, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 16; ++r) {                                   // C layout: row (tile) = (r & 3) + 8 (r >> 2) + 4 hi, column (co) = l31
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
            M[(xi * NT + row) * 32 + l31] = acc[r];
          }
        }
      } else {
        // stage 1: stand-in accumulators (the bf16 V values as floats), same LDS write volume
        for (int e = tid; e < 16 * NT * 32 / 4; e += 256) {
          float f[8];
          unpack8(V[e & (16 * NT * 8 - 1)], f);
          reinterpret_cast<float4*>(M)[e] = make_float4(f[0], f[1], f[2], f[3]);
        }
      }
      __syncthreads();
      {   // output transform: thread = (tile wt, 4-channel group q of this half's 32 channels): A^T M A -> 2 x 2 pixels
        const int wt = tid >> 3, q = tid & 7;
        const int ty = wt / (TW / 2), tx = wt % (TW / 2);
        float m[4][4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float4 v = reinterpret_cast<const float4*>(M)[((4 * i + j) * NT + wt) * 8 + q];
            m[i][j][0] = v.x; m[i][j][1] = v.y; m[i][j][2] = v.z; m[i][j][3] = v.w;
          }
        float o[2][2][4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float r0[4], r1[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) { r0[j] = m[0][j][k] + m[1][j][k] + m[2][j][k]; r1[j] = m[1][j][k] - m[2][j][k] - m[3][j][k]; }
          o[0][0][k] = r0[0] + r0[1] + r0[2]; o[0][1][k] = r0[1] - r0[2] - r0[3];
          o[1][0][k] = r1[0] + r1[1] + r1[2]; o[1][1][k] = r1[1] - r1[2] - r1[3];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const float f[8] = {o[a][c][0], o[a][c][1], o[a][c][2], o[a][c][3], 0.f, 0.f, 0.f, 0.f};
            const uint4 pk = pack8(f);
            const size_t px = ((size_t)b * H + y0 + 2 * ty + a) * W + x0 + 2 * tx + c;
            // 8 bytes per thread: channels [32 half + 4 q, + 4)
            reinterpret_cast<uint2*>(y)[px * 16 + half * 8 + q] = make_uint2(pk.x, pk.y);
          }
      }
      __syncthreads();
    }
  }
}

template <int STAGE> static float run(const uint4* x, uint4* y, const uint4* U, int B, int H, int W, int grid, int reps) {
  const int ntiles = B * (H / TH) * (W / TW);
  const size_t smem = (size_t)HALO_PIECES * 16 + 16 * NT * 8 * 16 + 16 * NT * 32 * 4;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(wino_skel<STAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < 2; ++r) hipLaunchKernelGGL(wino_skel<STAGE>, dim3(grid), dim3(256), smem, 0, x, y, U, B, H, W, ntiles);
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(wino_skel<STAGE>, dim3(grid), dim3(256), smem, 0, x, y, U, B, H, W, ntiles);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main() {
  const int B = 78, H = 320, W = 320, C = 64;
  const size_t n = (size_t)B * H * W * C;
  std::vector<uint16_t> h(n);
  srand(3);
  for (size_t i = 0; i < n; ++i) h[i] = (uint16_t)(0x3c00 + (rand() & 0x3ff)) ^ (uint16_t)((rand() & 1) << 15);     // random bf16 around +-1
  uint4 *x, *y, *U;
  CK(hipMalloc(&x, n * 2)); CK(hipMalloc(&y, n * 2)); CK(hipMalloc(&U, 16 * 2 * 4 * 64 * 16));
  CK(hipMemcpy(x, h.data(), n * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(U, h.data(), 16 * 2 * 4 * 64 * 16, hipMemcpyHostToDevice));
  const double gflop_direct = 2.0 * B * H * W * 64.0 * 64.0 * 9.0 / 1e9, gflop_wino = gflop_direct * 4.0 / 9.0;
  const double gb = (n * 2.0 * (double)(HP * WP) / (TH * TW) + n * 2.0) / 1e9;
  printf("layer 64 -> 64 @ %d x %d, batch %d: %.1f GFLOP direct, %.1f GFLOP as F(2x2,3x3); %.2f GB moved by this tiling (halo %.2fx)\n", H, W, B,
         gflop_direct, gflop_wino, gb, (double)(HP * WP) / (TH * TW));
  printf("go threshold: 1,100 TF algorithmic = %.3f ms;  conv_igemm_kernel today: 0.69 ms (848 TF)\n", gflop_direct / 1100.0);
  for (int grid : {256, 512}) {
    const float t1 = run<1>(x, y, U, B, H, W, grid, 5);
    const float t2 = run<2>(x, y, U, B, H, W, grid, 5);
    printf("grid %4d persistent workgroups (1 per CU by LDS):  stage 1 (transforms + memory) %.3f ms = %.2f TB/s   stage 2 (+ MFMA) %.3f ms = %.0f TF algorithmic, %.0f TF on the pipe\n",
           grid, t1, gb / t1, t2, gflop_direct / t2, gflop_wino / t2);
  }
  return 0;
}
