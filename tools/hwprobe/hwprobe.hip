// Hardware-facts probe for gfx950 (MI355X). Not part of the product path.
// Verifies the MFMA operand/result lane maps and the ds_read_b64_tr_b16
// gather semantics that im2im_uq_amd/csrc/*.hip rely on. Prints PASS/FAIL
// lines and raw dumps; run once on the GPU box, output kept in profiles/.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s4 lds_s4;

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(2);} }while(0)

// A: [32][16] row-major float (small ints), B: [16][32], D: [32][32]
__global__ void k_mfma_32x32x16(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  b8 a, b;
  for (int j = 0; j < 8; j++) {
    int k = (l >> 5) * 8 + j;
    a[j] = (__bf16)A[(l & 31) * 16 + k];
    b[j] = (__bf16)B[k * 32 + (l & 31)];
  }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[row * 32 + (l & 31)] = c[r];
  }
}
// A: [16][32], B: [32][16], D: [16][16]
__global__ void k_mfma_16x16x32(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  b8 a, b;
  for (int j = 0; j < 8; j++) {
    int k = (l >> 4) * 8 + j;
    a[j] = (__bf16)A[(l & 15) * 32 + k];
    b[j] = (__bf16)B[k * 16 + (l & 15)];
  }
  f4v c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}
// A: [32][2], B: [2][32]
__global__ void k_mfma_32x32x2f32(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  float a = A[(l & 31) * 2 + (l >> 5)];
  float b = B[(l >> 5) * 32 + (l & 31)];
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    D[row * 32 + (l & 31)] = c[r];
  }
}
// A: [16][4], B: [4][16]
__global__ void k_mfma_16x16x4f32(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)];
  float b = B[(l >> 4) * 16 + (l & 15)];
  f4v c = {0};
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; r++) D[((l >> 4) * 4 + r) * 16 + (l & 15)] = c[r];
}

// tr-read probe: lds[e] = e (u16). variant 0: addr(lane) = lane*8 bytes.
// variant 1: each 16-lane group g reads a 4x16 block with row stride RS bytes:
//   addr = g*BLK + ((lane&15)>>2)*RS + (lane&3)*8
// variant 2: transposed assignment: addr = g*BLK + (lane&3)*RS + ((lane&15)>>2)*8
__global__ void k_trread(short* out, int variant, int RS, int BLK) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x, g = l >> 4, q = l & 15;
  int addr;
  if (variant == 0) addr = l * 8;
  else if (variant == 1) addr = g * BLK + (q >> 2) * RS + (q & 3) * 8;
  else addr = g * BLK + (q & 3) * RS + (q >> 2) * 8;
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)((__attribute__((address_space(3))) char*)lds + addr));
  for (int j = 0; j < 4; j++) out[l * 4 + j] = v[j];
}

// global_load_lds 16B probe: each lane supplies gptr = src + perm(lane)*16B; LDS base uniform.
__global__ void k_glds(const int* src, int* out) {
  __shared__ __attribute__((aligned(16))) int lds[64 * 4 * 2];
  int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) lds[i] = -1;
  __syncthreads();
  const int* g = src + ((l * 7) & 63) * 4;  // permuted source
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = l; i < 512; i += 64) out[i] = lds[i];
}

// simple streaming-read bandwidth probe (float4 loads, sum)
__global__ void k_bw(const float4* __restrict__ p, size_t n, float* out) {
  size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  float acc = 0.f;
  for (; i < n; i += stride) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
  if (acc == 123.456f) out[0] = acc;
}

// pure-register MFMA throughput: 8 independent accumulators per wave, operands held in registers (random or zero bits),
// no LDS / global traffic inside the loop -- the ceiling a real kernel's data movement is added on top of.
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8_t;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16_t;
__global__ __launch_bounds__(256) void k_mfma_peak(const uint4* __restrict__ seed, float* __restrict__ out, int iters) {
  uint4 sa = seed[threadIdx.x], sb = seed[256 + threadIdx.x];
  bf16x8_t a0, a1, b0, b1;
  __builtin_memcpy(&a0, &sa, 16); __builtin_memcpy(&b0, &sb, 16);
  sa.x ^= 0x5a5a5a5a; sb.y ^= 0x3c3c3c3c;
  __builtin_memcpy(&a1, &sa, 16); __builtin_memcpy(&b1, &sb, 16);
  f32x16_t acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  for (int it = 0; it < iters; ++it) {
    acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0], 0, 0, 0);
    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1], 0, 0, 0);
    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[2], 0, 0, 0);
    acc[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[3], 0, 0, 0);
    acc[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[4], 0, 0, 0);
    acc[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[5], 0, 0, 0);
    acc[6] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[6], 0, 0, 0);
    acc[7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[7], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  if (s == 12345.678f) out[0] = s;
}

static void ref_mm(const float* A, const float* B, float* D, int M, int N, int K) {
  for (int i = 0; i < M; i++) for (int j = 0; j < N; j++) {
    float s = 0; for (int k = 0; k < K; k++) s += A[i * K + k] * B[k * N + j];
    D[i * N + j] = s;
  }
}
template <typename F>
static void test_mfma(const char* name, int M, int N, int K, F launch) {
  std::vector<float> A(M * K), B(K * N), D(M * N), R(M * N);
  srand(1234);
  for (auto& x : A) x = (float)(rand() % 9 - 4);
  for (auto& x : B) x = (float)(rand() % 7 - 3);
  float *dA, *dB, *dD;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dD, 0, D.size() * 4));
  launch(dA, dB, dD);
  CK(hipDeviceSynchronize());
  CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
  ref_mm(A.data(), B.data(), R.data(), M, N, K);
  int bad = 0;
  for (int i = 0; i < M * N; i++) if (D[i] != R[i]) bad++;
  printf("MFMA %-14s layout check: %s (%d/%d mismatches)\n", name, bad ? "FAIL" : "PASS", bad, M * N);
  hipFree(dA); hipFree(dB); hipFree(dD);
}

int main() {
  hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
  printf("device: %s arch=%s CUs=%d LDS/block=%zu clock=%d kHz\n", prop.name, prop.gcnArchName,
         prop.multiProcessorCount, prop.sharedMemPerBlock, prop.clockRate);
  test_mfma("32x32x16_bf16", 32, 32, 16, [](float* a, float* b, float* d) { k_mfma_32x32x16<<<1, 64>>>(a, b, d); });
  test_mfma("16x16x32_bf16", 16, 16, 32, [](float* a, float* b, float* d) { k_mfma_16x16x32<<<1, 64>>>(a, b, d); });
  test_mfma("32x32x2_f32", 32, 32, 2, [](float* a, float* b, float* d) { k_mfma_32x32x2f32<<<1, 64>>>(a, b, d); });
  test_mfma("16x16x4_f32", 16, 16, 4, [](float* a, float* b, float* d) { k_mfma_16x16x4f32<<<1, 64>>>(a, b, d); });

  short* dout; CK(hipMalloc(&dout, 64 * 4 * 2));
  std::vector<short> out(256);
  struct { int v, rs, blk; const char* what; } tv[] = {
      {0, 0, 0, "addr=lane*8"},
      {1, 32, 128, "rowmajor lanes (q>>2=row,q&3=quad) RS=32 BLK=128"},
      {1, 64, 1024, "rowmajor lanes RS=64 BLK=1024"},
      {2, 64, 1024, "colmajor lanes (q&3=row,q>>2=quad) RS=64 BLK=1024"},
  };
  for (auto& t : tv) {
    k_trread<<<1, 64>>>(dout, t.v, t.rs, t.blk);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost));
    printf("TRREAD variant [%s]\n", t.what);
    for (int l = 0; l < 64; l++) {
      printf("  lane %2d: %5d %5d %5d %5d\n", l, out[l * 4], out[l * 4 + 1], out[l * 4 + 2], out[l * 4 + 3]);
    }
    // check hypothesis H1: lane l elem j == element at block-row j, column (l&15) where
    // block rows are addressed by the lanes with (q>>2)==j: value = (g*BLK + j*RS)/2 + (l&15)
    if (t.v == 1) {
      int bad = 0;
      for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) {
        int expect = ((l >> 4) * t.blk + j * t.rs) / 2 + (l & 15);
        if (out[l * 4 + j] != (short)expect) bad++;
      }
      printf("  H1 (lane q supplies row q>>2, quad q&3; lane gets column l&15, elem j=row j): %s (%d bad)\n", bad ? "FAIL" : "PASS", bad);
    }
  }
  // glds
  {
    std::vector<int> src(256), o(512);
    for (int i = 0; i < 256; i++) src[i] = i;
    int *dsrc, *dOut2; CK(hipMalloc(&dsrc, 1024)); CK(hipMalloc(&dOut2, 2048));
    CK(hipMemcpy(dsrc, src.data(), 1024, hipMemcpyHostToDevice));
    k_glds<<<1, 64>>>(dsrc, dOut2);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(o.data(), dOut2, 2048, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; l++) for (int j = 0; j < 4; j++) if (o[l * 4 + j] != ((l * 7) & 63) * 4 + j) bad++;
    printf("GLDS16 (LDS dest = base + lane*16, per-lane global src): %s (%d bad) tail=%d\n", bad ? "FAIL" : "PASS", bad, o[256]);
  }
  // bandwidth
  {
    size_t bytes = (size_t)2 << 30; float4* p; float* o;
    CK(hipMalloc(&p, bytes)); CK(hipMalloc(&o, 4)); CK(hipMemset(p, 0, bytes));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; it++) {
      hipEventRecord(e0);
      k_bw<<<256 * 8, 256>>>(p, bytes / 16, o);
      hipEventRecord(e1); CK(hipEventSynchronize(e1));
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("BW read 2GiB float4: %.3f ms = %.2f TB/s\n", ms, bytes / ms / 1e9);
    }
  }
    {
    uint4* seed; float* o;
    CK(hipMalloc(&seed, 512 * sizeof(uint4))); CK(hipMalloc(&o, 4));
    std::vector<uint32_t> h(2048);
    for (int mode = 0; mode < 2; ++mode) {
      for (size_t i = 0; i < h.size(); ++i) {
        // random bf16 pairs with exponents near 1.0 (finite, no denormals); mode 1 = all zero bits
        uint32_t r = (uint32_t)(1103515245u * (uint32_t)(i * 2654435761u + 12345u) + 12345u);
        uint32_t lo = 0x3f00u | (r & 0x00ffu) | ((r >> 1) & 0x8000u), hi = 0x3f00u | ((r >> 8) & 0x00ffu) | ((r >> 9) & 0x8000u);
        h[i] = mode ? 0u : (lo | (hi << 16));
      }
      CK(hipMemcpy(seed, h.data(), h.size() * 4, hipMemcpyHostToDevice));
      const int iters = 4096, blocks = 256 * 8;                 // 2 workgroups of 4 waves per SIMD-quad -> 2 waves / SIMD
      hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
      for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, 0, seed, o, iters);
      CK(hipEventRecord(e0));
      for (int rep = 0; rep < 10; ++rep) hipLaunchKernelGGL(k_mfma_peak, dim3(blocks), dim3(256), 0, 0, seed, o, iters);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 10;
      const double flop = (double)blocks * 4 * iters * 8 * 2.0 * 32 * 32 * 16;
      printf("MFMA 32x32x16 bf16 register-only, %s operands: %.3f ms = %.0f TFLOP/s\n", mode ? "zero" : "random", ms, flop / ms / 1e9);
    }
  }
  return 0;
}
