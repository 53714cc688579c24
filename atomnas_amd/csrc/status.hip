// Error/status plumbing of the C-ABI library: every entry point returns 0 on success and a non-zero code otherwise;
// the message of the last failure (per host thread) is available through atomnas_last_error().
#include "common.h"
#include <cstdarg>
#include <cstdio>

namespace atomnas {
static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    set_error("%s: launch failed: %s", what, hipGetErrorString(e));
    return 2;
  }
  return 0;
}
}  // namespace atomnas

extern "C" const char* atomnas_last_error() { return atomnas::g_err; }

extern "C" int atomnas_abi_version() { return 1; }

extern "C" int atomnas_runtime_version() {
  int v = 0;
  if (hipRuntimeGetVersion(&v) != hipSuccess) return -1;
  return v;
}
