// Stand-alone probe of the packed-fp32 op_sel forms (profiles/r06_multiprocess_determinism.txt): the inner loop of
// smallconv_wgrad_vec_kernel (csrc/smallconv.hip) without its global memory traffic.  Every thread accumulates the same products twice --
// once through v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 with the op_sel form under test, once through scalar v_fmac_f32 / v_mul / v_add --
// and the two sets of sums must agree bit for bit.  A thread whose sums differ counts one mismatch.
//   hipcc --offload-arch=gfx950 -O3 -o pkfma_probe pkfma_probe.hip && ./pkfma_probe [launches] [iters]
// Run it alone, then beside other processes that keep the GPU busy (python tools/debug_victim.py aggressor 120 &).
// Forms (odd columns x of the halo row; even columns always use op_sel_hi:[1,0,1], the form that never failed):
//   0  v_pk_fma_f32 op_sel:[0,1,0]           low lane takes the HIGH register of src1      (the compiler's choice in the real kernel)
//   1  scalar                                 (control: packed only on even columns)
//   2  v_pk_fma_f32 op_sel:[1,0,0] with the operands swapped (the small-side pair as src0)
//   3  v_pk_mul_f32 op_sel:[0,1] + v_pk_add_f32                                            (not fused: compared with v_mul + v_add)
//   4  v_pk_fma_f32 op_sel:[0,1,0] with operands held in registers for the whole loop (no LDS reads inside the loop)
//   5  v_pk_fma_f32 op_sel:[0,0,1]: the THIRD source's high register into the low lane (sum = pair.hi + a*b, compared with v_fma)
//   6  v_pk_add_f32 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]  (a - b.hi: the fp32 conv's statistics epilogue) feeding a v_pk_fma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %s:%d\n",hipGetErrorString(e),__FILE__,__LINE__); exit(2);} }while(0)
using f2 = float __attribute__((ext_vector_type(2)));
constexpr int TS = 16, HS = 18, CL = 64, VL = 4, SW = HS + 2;

template <int FORM>
__global__ __launch_bounds__(256) void probe_kernel(const float* __restrict__ in, unsigned* __restrict__ mismatches, int iters) {
  __shared__ __attribute__((aligned(16))) float s_L[TS * TS][CL];
  __shared__ __attribute__((aligned(16))) float s_S[HS][SW];
  const int tid = threadIdx.x;
  const int lq = tid % 16, ty = tid / 16;
  for (int i = tid; i < TS * TS * CL; i += 256) s_L[i / CL][i % CL] = in[(blockIdx.x * 131 + i) % 65536];
  for (int i = tid; i < HS * SW; i += 256) s_S[i / SW][i % SW] = in[(blockIdx.x * 17 + i * 3) % 65536] * 1e-3f;
  __syncthreads();
  f2 accp[9][VL / 2];
  float accs[9][VL];
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int k = 0; k < VL; ++k) { accs[tp][k] = 0.f; if (k % 2 == 0) accp[tp][k / 2] = f2{0.f, 0.f}; }
  f2 pr[3][HS / 2];
  if (FORM == 4) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
      for (int j = 0; j < HS / 2; ++j) pr[dy][j] = *reinterpret_cast<const f2*>(&s_S[ty + dy][2 * j]);
  }
  for (int it = 0; it < iters; ++it) {
    if (FORM != 4) {
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int j = 0; j < HS / 2; ++j) pr[dy][j] = *reinterpret_cast<const volatile f2*>(&s_S[ty + dy][2 * j]);
    }
#pragma unroll
    for (int tx = 0; tx < TS; ++tx) {
      f2 lv2[VL / 2];
#pragma unroll
      for (int kp = 0; kp < VL / 2; ++kp) lv2[kp] = *reinterpret_cast<const volatile f2*>(&s_L[ty * TS + tx][lq * VL + 2 * kp]);
#pragma unroll
      for (int tp = 0; tp < 9; ++tp) {
        const int x = tx + tp % 3;
        const bool hi = x & 1;
        const f2 wp = pr[tp / 3][x / 2];
        const float wv = hi ? wp.y : wp.x;
#pragma unroll
        for (int kp = 0; kp < VL / 2; ++kp) {
          // reference: scalar
          if (FORM == 5 && hi) {          // acc = fma(lv, lv, w.hi)  -- the addend comes from the pair, the sum is NOT accumulated
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(accs[tp][2 * kp]) : "v"(lv2[kp].x), "v"(wv));
            asm volatile("v_fma_f32 %0, %1, %1, %2" : "=v"(accs[tp][2 * kp + 1]) : "v"(lv2[kp].y), "v"(wv));
          } else if (FORM == 6 && hi) {   // acc += (lv - w.hi)
            float m0, m1;
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(m0) : "v"(lv2[kp].x), "v"(wv));
            asm volatile("v_sub_f32 %0, %1, %2" : "=v"(m1) : "v"(lv2[kp].y), "v"(wv));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(accs[tp][2 * kp]) : "v"(m0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(accs[tp][2 * kp + 1]) : "v"(m1));
          } else if (FORM == 3 && hi) {
            float m0, m1;
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m0) : "v"(lv2[kp].x), "v"(wv));
            asm volatile("v_mul_f32 %0, %1, %2" : "=v"(m1) : "v"(lv2[kp].y), "v"(wv));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(accs[tp][2 * kp]) : "v"(m0));
            asm volatile("v_add_f32 %0, %0, %1" : "+v"(accs[tp][2 * kp + 1]) : "v"(m1));
          } else {
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(accs[tp][2 * kp]) : "v"(lv2[kp].x), "v"(wv));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(accs[tp][2 * kp + 1]) : "v"(lv2[kp].y), "v"(wv));
          }
          // under test
          f2& av = accp[tp][kp];
          if (!hi) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(av) : "v"(lv2[kp]), "v"(wp));
          else if (FORM == 0 || FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(av) : "v"(lv2[kp]), "v"(wp));
          else if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(av) : "v"(wp), "v"(lv2[kp]));
          else if (FORM == 3) {
            f2 m;
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(m) : "v"(lv2[kp]), "v"(wp));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(av) : "v"(m));
          } else if (FORM == 5) {
            asm volatile("v_pk_fma_f32 %0, %1, %1, %2 op_sel:[0,0,1]" : "=v"(av) : "v"(lv2[kp]), "v"(wp));
          } else if (FORM == 6) {
            f2 m;
            asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(m) : "v"(lv2[kp]), "v"(wp));
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(av) : "v"(m));
          } else {
            float a0 = av.x, a1 = av.y;
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a0) : "v"(lv2[kp].x), "v"(wv));
            asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a1) : "v"(lv2[kp].y), "v"(wv));
            av.x = a0; av.y = a1;
          }
        }
      }
    }
  }
  unsigned bad_lo = 0, bad_hi = 0;
#pragma unroll
  for (int tp = 0; tp < 9; ++tp)
#pragma unroll
    for (int kp = 0; kp < VL / 2; ++kp) {
      bad_lo += __float_as_uint(accp[tp][kp].x) != __float_as_uint(accs[tp][2 * kp]);
      bad_hi += __float_as_uint(accp[tp][kp].y) != __float_as_uint(accs[tp][2 * kp + 1]);
    }
  if (bad_lo) atomicAdd(&mismatches[0], 1u);
  if (bad_hi) atomicAdd(&mismatches[1], 1u);
  if (bad_lo | bad_hi) atomicAdd(&mismatches[2], bad_lo + bad_hi);
}

// What does the low lane return when it is wrong?  One v_pk_fma_f32 op_sel:[0,1,0] at a time, checked at once against
//   ref  = fma(a.lo, b.HI, c.lo)   (what the ISA says)      h1 = fma(a.lo, b.LO, c.lo)   (op_sel ignored for the low lane)
//   h2   = c.lo                    (low lane not executed)   h3 = fma(a.HI, b.HI, c.lo)   (src0 swizzled instead)
// counters: [0] checks, [1] wrong, [2] wrong == h1, [3] wrong == h2, [4] wrong == h3, [5] high lane wrong
__global__ __launch_bounds__(256) void classify_kernel(const float* __restrict__ in, unsigned* __restrict__ cnt, int iters) {
  const int tid = blockIdx.x * 256 + threadIdx.x;
  f2 a = {in[(tid * 4) % 65536], in[(tid * 4 + 1) % 65536]}, b = {in[(tid * 4 + 2) % 65536] * 1e-3f, in[(tid * 4 + 3) % 65536] * 1e-3f};
  f2 c = {0.f, 0.f};
  unsigned n = 0, wrong = 0, w1 = 0, w2 = 0, w3 = 0, whi = 0;
  for (int it = 0; it < iters; ++it) {
    f2 r = c;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0]" : "+v"(r) : "v"(a), "v"(b));
    float ref = c.x, h1 = c.x, h3 = c.x, rhi = c.y;
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(ref) : "v"(a.x), "v"(b.y));
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(h1) : "v"(a.x), "v"(b.x));
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(h3) : "v"(a.y), "v"(b.y));
    asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(rhi) : "v"(a.y), "v"(b.y));
    ++n;
    const unsigned got = __float_as_uint(r.x);
    if (got != __float_as_uint(ref)) {
      ++wrong;
      w1 += got == __float_as_uint(h1);
      w2 += got == __float_as_uint(c.x);
      w3 += got == __float_as_uint(h3);
    }
    whi += __float_as_uint(r.y) != __float_as_uint(rhi);
    c.x = ref; c.y = rhi;                                  // continue from the correct sums
    a.x = a.x * 0.9990234375f + 0.001f; b.y = b.y * 1.0009765625f;      // new operands every time (exact enough; both paths see the same values)
  }
  if (wrong | whi) {
    atomicAdd(&cnt[1], wrong); atomicAdd(&cnt[2], w1); atomicAdd(&cnt[3], w2); atomicAdd(&cnt[4], w3); atomicAdd(&cnt[5], whi);
  }
  if (threadIdx.x == 0) atomicAdd(&cnt[0], n * 256u);
}

template <int FORM> void run(const char* what, const float* d_in, unsigned* d_mis, int launches, int iters) {
  unsigned tot[3] = {0, 0, 0};
  int bad_launches = 0;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms_sum = 0.f;
  for (int l = 0; l < launches; ++l) {
    CK(hipMemset(d_mis, 0, 3 * sizeof(unsigned)));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(probe_kernel<FORM>, dim3(2048), dim3(256), 0, 0, d_in, d_mis, iters);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms_sum += ms;
    unsigned h[3];
    CK(hipMemcpy(h, d_mis, sizeof(h), hipMemcpyDeviceToHost));
    bad_launches += (h[0] | h[1]) != 0;
    for (int i = 0; i < 3; ++i) tot[i] += h[i];
  }
  printf("[pkfma form %d] %-72s launches with a mismatch: %d of %d; threads with a low-lane mismatch %u, high-lane %u, accumulators %u  (%.3f ms per launch)\n",
         FORM, what, bad_launches, launches, tot[0], tot[1], tot[2], ms_sum / launches);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int launches = argc > 1 ? atoi(argv[1]) : 100, iters = argc > 2 ? atoi(argv[2]) : 8;
  std::vector<float> h(65536);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  float* d_in; unsigned* d_mis;
  CK(hipMalloc(&d_in, h.size() * sizeof(float)));
  CK(hipMalloc(&d_mis, 3 * sizeof(unsigned)));
  CK(hipMemcpy(d_in, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
  run<1>("odd columns scalar (control)", d_in, d_mis, launches, iters);
  run<0>("odd columns v_pk_fma_f32 op_sel:[0,1,0]", d_in, d_mis, launches, iters);
  run<4>("the same, small side held in registers (no LDS reads of it in the loop)", d_in, d_mis, launches, iters);
  run<2>("odd columns v_pk_fma_f32 op_sel:[1,0,0] (small-side pair as src0)", d_in, d_mis, launches, iters);
  run<3>("odd columns v_pk_mul_f32 op_sel:[0,1] + v_pk_add_f32", d_in, d_mis, launches, iters);
  run<5>("odd columns v_pk_fma_f32 op_sel:[0,0,1] (third source)", d_in, d_mis, launches, iters);
  run<6>("odd columns v_pk_add_f32 op_sel:[0,1] neg (a - b.hi) + v_pk_add_f32", d_in, d_mis, launches, iters);
  {
    unsigned* d_cnt; CK(hipMalloc(&d_cnt, 6 * sizeof(unsigned))); CK(hipMemset(d_cnt, 0, 6 * sizeof(unsigned)));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL(classify_kernel, dim3(2048), dim3(256), 0, 0, d_in, d_cnt, 256);
    CK(hipDeviceSynchronize());
    unsigned h[6]; CK(hipMemcpy(h, d_cnt, sizeof(h), hipMemcpyDeviceToHost));
    printf("[pkfma classify] single v_pk_fma_f32 op_sel:[0,1,0], %u checks (x%d launches, counter wraps ignored): low lane wrong %u -- equal to fma(a.lo, b.LO, c.lo) [op_sel ignored] %u, "
           "equal to c.lo [not executed] %u, equal to fma(a.HI, b.HI, c.lo) %u; high lane wrong %u\n", h[0], launches, h[1], h[2], h[3], h[4], h[5]);
  }
  return 0;
}
