// Shared helpers for the gfx950 kernel library (error reporting, launch checks).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/im2im_uq.h"

namespace im2im {
void set_error(const char* fmt, ...);
inline int fail_invalid(const char* what) { set_error("invalid argument: %s", what); return IM2IM_ERR_INVALID; }
inline int check_launch(const char* kernel) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { set_error("launch of %s failed: %s", kernel, hipGetErrorString(e)); return IM2IM_ERR_HIP; }
  return IM2IM_OK;
}
inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
}  // namespace im2im

// [r6] a kernel compiled without packed fp32 instructions.  The compiler pairs scalar code into v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32
// and chooses their op_sel on its own; the forms that feed the HIGH register of the second / third source to the low lane drop that lane's
// write while another process shares the GPU (profiles/r06_multiprocess_determinism.txt).  Put on the (HBM-bound) kernels in which it chose
// such a form; tests/test_abi.py scans the built library so that none comes back unnoticed.
#if defined(__HIP_DEVICE_COMPILE__)
#define IM2IM_NO_PACKED_FP32 __attribute__((target("no-packed-fp32-ops")))
#else
#define IM2IM_NO_PACKED_FP32
#endif
#define IM2IM_REQUIRE(cond) do { if (!(cond)) return im2im::fail_invalid(#cond); } while (0)
#define IM2IM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
  im2im::set_error("%s failed: %s", #call, hipGetErrorString(e_)); return IM2IM_ERR_HIP; } } while (0)
