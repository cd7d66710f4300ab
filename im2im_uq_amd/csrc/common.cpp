#include "common.h"
#include <cstring>

namespace im2im {
static thread_local char g_err[512] = "";
void set_error(const char* fmt, ...) {
  va_list ap; va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
const char* get_error() { return g_err; }
}  // namespace im2im

extern "C" int im2im_abi_version(void) { return IM2IM_ABI_VERSION; }
extern "C" const char* im2im_last_error(void) { return im2im::get_error(); }
