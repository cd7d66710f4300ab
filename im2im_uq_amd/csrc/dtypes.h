// Element types and vector typedefs shared by the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>

namespace im2im {

using bf16_t = __bf16;                                   // storage-compatible with torch.bfloat16
typedef short short8 __attribute__((ext_vector_type(8)));
typedef short short4v __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char lds_char;
typedef __attribute__((address_space(3))) short4v lds_short4;

__device__ __forceinline__ bf16x8 as_bf16x8(short8 v) { return __builtin_bit_cast(bf16x8, v); }

template <typename T> __device__ __forceinline__ T from_float(float v);
template <> __device__ __forceinline__ float from_float<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_float<bf16_t>(float v) { return (bf16_t)v; }   // RNE
__device__ __forceinline__ float to_float(float v) { return v; }
__device__ __forceinline__ float to_float(bf16_t v) { return (float)v; }

// 8 (bf16) or 4 (fp32) consecutive elements <-> floats, through one 16-byte access
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  static __device__ __forceinline__ void load(const float* p, float* out) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    out[0] = v.x; out[1] = v.y; out[2] = v.z; out[3] = v.w;
  }
  static __device__ __forceinline__ void store(float* p, const float* in) {
    *reinterpret_cast<float4*>(p) = make_float4(in[0], in[1], in[2], in[3]);
  }
  // streamed (non-temporal) form for tensors far larger than L2 that this kernel does not read again
  static __device__ __forceinline__ void store_nt(float* p, const float* in) {
    f32x4 v = {in[0], in[1], in[2], in[3]};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(p));
  }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  static __device__ __forceinline__ void load(const bf16_t* p, float* out) {
    const uint4 v = *reinterpret_cast<const uint4*>(p);
    const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      out[2 * i] = __uint_as_float(w[i] << 16);
      out[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
    }
  }
  static __device__ __forceinline__ void store(bf16_t* p, const float* in) {
    unsigned w[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t lo = (bf16_t)in[2 * i], hi = (bf16_t)in[2 * i + 1];
      w[i] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
    }
    *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
  }
  static __device__ __forceinline__ void store_nt(bf16_t* p, const float* in) {
    u32x4 w;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bf16_t lo = (bf16_t)in[2 * i], hi = (bf16_t)in[2 * i + 1];
      w[i] = (unsigned)__builtin_bit_cast(unsigned short, lo) | ((unsigned)__builtin_bit_cast(unsigned short, hi) << 16);
    }
    __builtin_nontemporal_store(w, reinterpret_cast<u32x4*>(p));
  }
};

}  // namespace im2im
