// Error/status plumbing of the C-ABI library: every entry point returns 0 on success and a non-zero code otherwise;
// the message of the last failure (per host thread) is available through atomnas_last_error().
#include "common.h"
#include <cstdarg>
#include <cstdio>
#include <mutex>
#include <unordered_map>

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

int num_cus() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t p;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&p, dev) == hipSuccess) n = p.multiProcessorCount;
    if (n <= 0) n = 256;
  }
  return n;
}

// LDS bytes one workgroup may use on the current device (gfx950: 160 KiB), queried once; 64 KiB -- what every CDNA part has -- when
// the query fails.  Shapes that need more take the tile kernels / are reported as unsupported instead of failing at launch.
size_t max_lds_bytes() {
  static size_t v = 0;
  if (v == 0) {
    int dev = 0, b = 0;
    if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&b, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && b > 0)
      v = (size_t)b;
    else {
      (void)hipGetLastError();
      v = 64 * 1024;
    }
  }
  return v;
}

int resident_per_cu_raw(const void* kern, int threads, size_t lds) {
  static std::mutex mu;
  static std::unordered_map<size_t, int> cache;
  static std::unordered_map<const void*, size_t> granted;
  std::lock_guard<std::mutex> lock(mu);
  if (lds > 64 * 1024) {  // dynamic LDS above 64 KiB must be opted into once per kernel
    size_t& have = granted[kern];
    if (lds > have) {
      (void)hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
      have = lds;
    }
  }
  const size_t key = ((size_t)kern * 1000003u + lds) * 31u + (size_t)threads;
  auto it = cache.find(key);
  if (it != cache.end()) return it->second;
  int nb = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kern, threads, lds) != hipSuccess || nb < 1) {
    (void)hipGetLastError();
    nb = (int)(max_lds_bytes() / (lds + 1024));
    if (nb > 2) nb = 2;
    if (nb < 1) nb = 1;
  }
  cache[key] = nb;
  return nb;
}
}  // namespace atomnas

extern "C" const char* atomnas_last_error() { return atomnas::g_err; }

extern "C" int atomnas_abi_version() { return 9; }   // ATOMNAS_ABI_VERSION in include/atomnas_hip.h

extern "C" int atomnas_runtime_version() {
  int v = 0;
  if (hipRuntimeGetVersion(&v) != hipSuccess) return -1;
  return v;
}
