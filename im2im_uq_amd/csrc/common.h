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

#define IM2IM_REQUIRE(cond) do { if (!(cond)) return im2im::fail_invalid(#cond); } while (0)
#define IM2IM_HIP(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { \
  im2im::set_error("%s failed: %s", #call, hipGetErrorString(e_)); return IM2IM_ERR_HIP; } } while (0)
