// Error reporting and version entry points of libvts_hip.so.
#include <stdarg.h>
#include <stdio.h>

#include <hip/hip_runtime.h>

#include "vts.h"

static thread_local char g_err[512] = "";

void vts_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

extern "C" const char* vts_last_error(void) { return g_err; }

// name of the kernel instance the last dispatching call chose (profiling aid: matches rocprofv3's kernel names)
static thread_local char g_kernel[160] = "";
void vts_set_kernel(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_kernel, sizeof(g_kernel), fmt, ap);
  va_end(ap);
}
extern "C" const char* vts_last_kernel(void) { return g_kernel; }
extern "C" int vts_version(void) { return 1; }

// one {1, 0} constant per DEVICE (a host process that drives several GPUs gets the copy that lives on the current one), created on
// first use under a lock; device memory of a process is never freed here (process lifetime)
#include <mutex>

const float* vts_ident() {
  static float* dev[64] = {nullptr};
  static std::mutex mu;
  int d = 0;
  if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= 64) return nullptr;
  std::lock_guard<std::mutex> lock(mu);
  if (!dev[d]) {
    const float h[2] = {1.f, 0.f};
    float* p = nullptr;
    if (hipMalloc(&p, sizeof(h)) != hipSuccess) return nullptr;
    if (hipMemcpy(p, h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(p);
      return nullptr;
    }
    dev[d] = p;
  }
  return dev[d];
}
