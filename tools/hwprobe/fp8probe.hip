// fp8 (OCP e4m3) MFMA probe for gfx950: operand lane maps of v_mfma_f32_32x32x16_fp8_fp8 and of the block-scaled
// v_mfma_scale_f32_32x32x64_f8f6f4, the cvt_pk_fp8_f32 packing, and their issue rates.  Not part of the product path.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef int i8v __attribute__((ext_vector_type(8)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(2);} }while(0)

__device__ unsigned char to_fp8(float v) {
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false);
  return (unsigned char)(r & 0xff);
}
// A [32][K], B [K][32] floats holding small integers
__global__ void k_fp8_x16(const float* A, const float* B, float* D) {
  int l = threadIdx.x;
  unsigned long long a = 0, b = 0;
  for (int j = 0; j < 8; j++) {
    int k = (l >> 5) * 8 + j;
    a |= (unsigned long long)to_fp8(A[(l & 31) * 16 + k]) << (8 * j);
    b |= (unsigned long long)to_fp8(B[k * 32 + (l & 31)]) << (8 * j);
  }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8((long)a, (long)b, c, 0, 0, 0);
  for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
// variant: which k does byte j of lane l hold?  mode 0: k = 32*(l>>5) + j ; mode 1: k = 16*(l>>5) + (j&15) + 32*(j>>4)
__global__ void k_fp8_scale_x64(const float* A, const float* B, float* D, int mode, int scale_a, int scale_b) {
  int l = threadIdx.x;
  unsigned char ab[32], bb[32];
  for (int j = 0; j < 32; j++) {
    int k = mode == 0 ? 32 * (l >> 5) + j : 16 * (l >> 5) + (j & 15) + 32 * (j >> 4);
    ab[j] = to_fp8(A[(l & 31) * 64 + k]);
    bb[j] = to_fp8(B[k * 32 + (l & 31)]);
  }
  i8v a, b;
  for (int w = 0; w < 8; w++) {
    a[w] = ab[4 * w] | (ab[4 * w + 1] << 8) | (ab[4 * w + 2] << 16) | (ab[4 * w + 3] << 24);
    b[w] = bb[4 * w] | (bb[4 * w + 1] << 8) | (bb[4 * w + 2] << 16) | (bb[4 * w + 3] << 24);
  }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
  for (int r = 0; r < 16; r++) D[((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) * 32 + (l & 31)] = c[r];
}
__global__ void k_cvt(const float* in, unsigned* out) {
  int i = threadIdx.x;
  int r = __builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], 0, false);
  int r2 = __builtin_amdgcn_cvt_pk_fp8_f32(in[2 * i], in[2 * i + 1], 0x12345678, true);
  out[2 * i] = (unsigned)r; out[2 * i + 1] = (unsigned)r2;
}
template <int KIND>
__global__ __launch_bounds__(256) void k_rate(float* out, int iters, unsigned seed) {
  f16v c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
  unsigned s = seed + threadIdx.x * 2654435761u;
  i8v a, b;
  for (int w = 0; w < 8; w++) { s = s * 1664525u + 1013904223u; a[w] = (s & 0x77777777u); s = s * 1664525u + 1013904223u; b[w] = (s & 0x77777777u); }
  for (int i = 0; i < iters; i++) {
    if (KIND == 0) {
      long la = ((long)a[0] << 32) | (unsigned)a[1], lb = ((long)b[0] << 32) | (unsigned)b[1];
      c0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la, lb, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la, lb, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la, lb, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la, lb, c3, 0, 0, 0);
    } else {
      c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 127, 0, 127);
      c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c1, 0, 0, 0, 127, 0, 127);
      c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c2, 0, 0, 0, 127, 0, 127);
      c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c3, 0, 0, 0, 127, 0, 127);
    }
  }
  float acc = 0;
  for (int r = 0; r < 16; r++) acc += c0[r] + c1[r] + c2[r] + c3[r];
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const int K = 64;
  std::vector<float> A(32 * K), B(K * 32), D(32 * 32), R(32 * 32);
  srand(3);
  for (auto& v : A) v = (float)((rand() % 9) - 4);
  for (auto& v : B) v = (float)((rand() % 7) - 3);
  float *dA, *dB, *dD;
  CK(hipMalloc(&dA, A.size() * 4)); CK(hipMalloc(&dB, B.size() * 4)); CK(hipMalloc(&dD, D.size() * 4));
  // ---- non-scaled x16 (uses A[:, :16] with row pitch 16 -> repack)
  {
    std::vector<float> A16(32 * 16), B16(16 * 32);
    for (int i = 0; i < 32; i++) for (int k = 0; k < 16; k++) A16[i * 16 + k] = A[i * K + k];
    for (int k = 0; k < 16; k++) for (int j = 0; j < 32; j++) B16[k * 32 + j] = B[k * 32 + j];
    CK(hipMemcpy(dA, A16.data(), A16.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B16.data(), B16.size() * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_fp8_x16, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float r = 0; for (int k = 0; k < 16; k++) r += A16[i * 16 + k] * B16[k * 32 + j]; if (r != D[i * 32 + j]) bad++; }
    printf("mfma_f32_32x32x16_fp8_fp8 lane map (k = 8*(l>>5)+j, byte j): %s (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  }
  CK(hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice));
  for (int i = 0; i < 32; i++) for (int j = 0; j < 32; j++) { float r = 0; for (int k = 0; k < K; k++) r += A[i * K + k] * B[k * 32 + j]; R[i * 32 + j] = r; }
  for (int mode = 0; mode < 2; mode++) {
    hipLaunchKernelGGL(k_fp8_scale_x64, dim3(1), dim3(64), 0, 0, dA, dB, dD, mode, 127, 127);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    int bad = 0; for (int i = 0; i < 1024; i++) if (R[i] != D[i]) bad++;
    printf("mfma_scale_f32_32x32x64_f8f6f4 (fp8, scales 127) lane map mode %d: %s (%d mismatches)  D[0..3] %g %g %g %g  ref %g %g %g %g\n", mode,
           bad ? "FAIL" : "PASS", bad, D[0], D[1], D[2], D[3], R[0], R[1], R[2], R[3]);
  }
  for (int sa = 126; sa <= 129; sa++) {
    hipLaunchKernelGGL(k_fp8_scale_x64, dim3(1), dim3(64), 0, 0, dA, dB, dD, 0, sa, 127);
    CK(hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost));
    printf("scale_a = %d (scale_b 127): D[5]/ref[5] = %g\n", sa, R[5] != 0 ? D[5] / R[5] : -1.0f);
  }
  // ---- cvt packing
  {
    float h[8] = {1.0f, -2.5f, 0.3f, 448.f, 500.f, 0.001f, -0.0157f, 17.f};
    float* din; unsigned* dout; unsigned ho[8];
    CK(hipMalloc(&din, 32)); CK(hipMalloc(&dout, 32)); CK(hipMemcpy(din, h, 32, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(k_cvt, dim3(1), dim3(4), 0, 0, din, dout);
    CK(hipMemcpy(ho, dout, 32, hipMemcpyDeviceToHost));
    for (int i = 0; i < 4; i++) printf("cvt_pk_fp8_f32(%g, %g): lo-word form %08x  hi-word form (old 12345678) %08x\n", h[2 * i], h[2 * i + 1], ho[2 * i], ho[2 * i + 1]);
  }
  // ---- rates
  float* dout; CK(hipMalloc(&dout, 256 * 1024 * 4 * 4));
  for (int kind = 0; kind < 2; kind++) {
    const int iters = 4096, blocks = 256 * 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 2; rep++) {
      CK(hipEventRecord(e0));
      if (kind == 0) hipLaunchKernelGGL(k_rate<0>, dim3(blocks), dim3(256), 0, 0, dout, iters, 7u);
      else hipLaunchKernelGGL(k_rate<1>, dim3(blocks), dim3(256), 0, 0, dout, iters, 7u);
      CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    }
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double flops = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * (kind == 0 ? 16 : 64);
    printf("%s: %.1f TFLOP/s on random operands (register-only loop)\n", kind == 0 ? "mfma_f32_32x32x16_fp8_fp8" : "mfma_scale_f32_32x32x64_f8f6f4(fp8)", flops / ms / 1e9);
  }
  return 0;
}
