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

// Nodes of the graph a stream is capturing into, counted at this point of the capture: what a replay of the step will launch
// (bench.py reports the replayed schedule's launch count beside the eager one).  Not capturing: both counts are 0.
extern "C" int vts_capture_node_count(void* stream, int* nodes, int* kernel_nodes) {
  if (!nodes || !kernel_nodes) {
    vts_set_error("vts_capture_node_count: null pointer");
    return VTS_ERR_ARG;
  }
  *nodes = *kernel_nodes = 0;
  hipStreamCaptureStatus status = hipStreamCaptureStatusNone;
  unsigned long long id = 0;
  hipGraph_t graph = nullptr;
  const hipGraphNode_t* deps = nullptr;
  size_t ndeps = 0;
  if (hipStreamGetCaptureInfo_v2((hipStream_t)stream, &status, &id, &graph, &deps, &ndeps) != hipSuccess) {
    (void)hipGetLastError();
    vts_set_error("vts_capture_node_count: hipStreamGetCaptureInfo_v2 failed");
    return VTS_ERR_LAUNCH;
  }
  if (status != hipStreamCaptureStatusActive || !graph) return VTS_OK;
  size_t n = 0;
  if (hipGraphGetNodes(graph, nullptr, &n) != hipSuccess) {
    (void)hipGetLastError();
    vts_set_error("vts_capture_node_count: hipGraphGetNodes failed");
    return VTS_ERR_LAUNCH;
  }
  *nodes = (int)n;
  if (n == 0) return VTS_OK;
  hipGraphNode_t* list = new hipGraphNode_t[n];
  int k = 0;
  if (hipGraphGetNodes(graph, list, &n) == hipSuccess) {
    for (size_t i = 0; i < n; ++i) {
      hipGraphNodeType t;
      if (hipGraphNodeGetType(list[i], &t) == hipSuccess && t == hipGraphNodeTypeKernel) ++k;
    }
  } else {
    (void)hipGetLastError();
  }
  delete[] list;
  *kernel_nodes = k;
  return VTS_OK;
}

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
