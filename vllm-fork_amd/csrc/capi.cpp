// Error plumbing and version of libmi355q (host only).
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>

#include "../../include/mi355q.h"

namespace mi355q {

static thread_local char g_err[512] = "";

int set_error(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

// hipGetLastError after a launch reports bad launch configurations without synchronising.
int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return set_error(MI355Q_EHIP, "%s: %s", what, hipGetErrorString(e));
  return MI355Q_OK;
}

}  // namespace mi355q

extern "C" const char* mi355q_last_error(void) { return mi355q::g_err; }
extern "C" int mi355q_version(void) { return 1; }
