// Error plumbing of the C ABI (thread-local message; no exceptions, no abort).
#include <stdarg.h>
#include <string.h>
#include "common.h"

namespace {
thread_local char g_err[512] = "";
}

void mi355_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int mi355_check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    mi355_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
    return MI355_ERR_HIP;
  }
  return MI355_OK;
}

extern "C" int mi355_version(void) { return 100; }
extern "C" const char* mi355_last_error(void) { return g_err; }
