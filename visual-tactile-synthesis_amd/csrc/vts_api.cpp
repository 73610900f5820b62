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

const float* vts_ident() {
  static float* dev = nullptr;
  if (!dev) {
    const float h[2] = {1.f, 0.f};
    if (hipMalloc(&dev, sizeof(h)) != hipSuccess) return nullptr;
    hipMemcpy(dev, h, sizeof(h), hipMemcpyHostToDevice);
  }
  return dev;
}
