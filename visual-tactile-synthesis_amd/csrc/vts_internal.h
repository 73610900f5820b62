// Internal helpers shared by the HIP translation units of libvts_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "vts.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

void vts_set_error(const char* fmt, ...);
void vts_set_kernel(const char* fmt, ...);
// device pointer to the two floats {1, 0}: identity scale / shift for operands without an affine, so the
// kernels can fetch scale/shift unconditionally (no data-dependent branch around a load)
const float* vts_ident();
// small-map (flattened-batch) path of vts_conv4x4; VTS_ERR_UNSUPPORTED means "use the tiled kernel"
int vts_conv_small_try(const vts_conv_desc* d, hipStream_t st);
// thin (Cout <= 16) stride-2 transposed layers on full-size maps: direct packed-FMA kernel; VTS_ERR_UNSUPPORTED otherwise
int vts_conv_thin_try(const vts_conv_desc* d, hipStream_t st);
// thin (Cout <= 20) stride-2 convolutions on full-size maps: lane = pixel on the 4x4x1 MFMA (vts_conv_px.hip); VTS_ERR_UNSUPPORTED otherwise
int vts_conv_px_try(const vts_conv_desc* d, hipStream_t st, float* stat_part, float* bsum_part, int64_t part_floats, int* stat_spl);
// single-output-channel stride-1 layers (PatchGAN prediction heads) on full-size maps: LDS-tiled vector-ALU kernel; VTS_ERR_UNSUPPORTED otherwise
int vts_conv_head_try(const vts_conv_desc* d, hipStream_t st);
// stride-1 3x3 weight gradient in Winograd F(3x3, 2x2) form (vts_conv3x3_wino.hip): partials [KS][Cout][Cin][9] with KS <= max_ks; returns KS, 0 = shape not taken
int vts_wgrad3x3_wino_try(const float* dout, const float* in, float* part, int N, int Cin, int Cout, int H, int W, int max_ks, hipStream_t st);
// PatchNCE on the MFMA path (vts_patchnce.hip): P, D <= 256
bool vts_patchnce_mfma_ok(int P, int D);
int vts_patchnce_mfma(const float* q, const float* k, int B, int P, int D, float T, float gscale, float* loss, float* dq, hipStream_t st);
// LeakyReLU slope that expresses the activation codes as  t > 0 ? t : slope * t
// Tuning / experiment switches (tile tables, thresholds, "run the other kernel" knobs: ~50 VTS_* names across the library) exist only in
// the instrumented build (make PROFILING=1 -> libvts_hip_prof.so, loaded through VTS_LIB_PATH by tools/): there they are read from the
// environment.  The PRODUCTION library reads exactly one environment variable (VTS_RCCL_LIB, the RCCL library to dlopen, vts_comm.cpp);
// every dispatch decision in it is the measured default, a compile-time constant -- no switch can change what a user's run computes.
#ifdef VTS_PROFILING
#include <stdlib.h>
static inline int vts_tune(const char* name, int dflt) {
  const char* e = getenv(name);
  return e ? atoi(e) : dflt;
}
static inline bool vts_tune_set(const char* name) { return getenv(name) != nullptr; }
static inline const char* vts_tune_str(const char* name) { return getenv(name); }
#else
static inline constexpr int vts_tune(const char*, int dflt) { return dflt; }
static inline constexpr bool vts_tune_set(const char*) { return false; }
static inline constexpr const char* vts_tune_str(const char*) { return nullptr; }
#endif

static inline float vts_slope(int act) { return act == VTS_ACT_LRELU ? 0.2f : (act == VTS_ACT_RELU ? 0.f : 1.f); }

#define VTS_CHECK_ARG(cond, ...)     \
  do {                               \
    if (!(cond)) {                   \
      vts_set_error(__VA_ARGS__);    \
      return VTS_ERR_ARG;            \
    }                                \
  } while (0)

#define VTS_CHECK_LAUNCH(name)                                             \
  do {                                                                     \
    hipError_t e__ = hipGetLastError();                                    \
    if (e__ != hipSuccess) {                                               \
      vts_set_error("%s: launch failed: %s", name, hipGetErrorString(e__)); \
      return VTS_ERR_LAUNCH;                                               \
    }                                                                      \
  } while (0)

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__device__ __forceinline__ float vts_act(float v, int act) {
  if (act == VTS_ACT_LRELU) return v > 0.f ? v : 0.2f * v;
  if (act == VTS_ACT_RELU) return v > 0.f ? v : 0.f;
  return v;
}
__device__ __forceinline__ float vts_act_grad(float v, int act) {
  if (act == VTS_ACT_LRELU) return v > 0.f ? 1.f : 0.2f;
  if (act == VTS_ACT_RELU) return v > 0.f ? 1.f : 0.f;
  return 1.f;
}

// wave64 sum via DPP-free shuffles
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// block (<=1024 threads) sum; result valid in every thread. `red` must hold >= 16 floats.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if (lane == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
