// Probe: do VALU instructions issue in the shadow of a running v_mfma_f32_16x16x4_f32 (32 cycles), from the same wave and from
// other waves of the same SIMD?   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_valu_overlap.hip -o tools/probes/mfma_valu_overlap.bin
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// every wave: per iteration NM MFMAs and NV independent v_fma_f32 (interleaved NV/NM after each MFMA)
template <int NM, int NV>
__global__ __launch_bounds__(256) void same_wave(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = lane * 0.01f, b = 1.f + lane * 0.001f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane + i;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < (NM > 0 ? NM : 1); ++j) {
      if (NM > 0) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV / (NM > 0 ? NM : 1); ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(a), "v"(b));
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  f32x4 t = acc[0] + acc[1] + acc[2] + acc[3];
  if (s + t[0] + t[1] + t[2] + t[3] == 123.456f) out[0] = 1.f;
}

// waves of even workgroup-local index run MFMAs only, odd ones VALU only (each NI instructions per iteration)
template <int NI>
__global__ __launch_bounds__(512) void split_waves(float* out, int iters, int mode) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;   // 8 waves: 2 per SIMD
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = lane * 0.01f, b = 1.f + lane * 0.001f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = lane + i;
  const bool do_m = (wave < 4) ? (mode & 1) : false, do_v = (wave >= 4) ? (mode & 2) : false;
  if (do_m) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < NI; ++j) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j & 3], 0, 0, 0);
  }
  if (do_v) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int j = 0; j < NI * 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j & 7]) : "v"(a), "v"(b));
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += v[i];
  f32x4 t = acc[0] + acc[1] + acc[2] + acc[3];
  if (s + t[0] + t[1] + t[2] + t[3] == 123.456f) out[0] = 1.f;
}

template <typename F>
float timeit(F f) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    f();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  float* out;
  hipMalloc(&out, 4);
  const int iters = 20000, wgs = 256;   // one workgroup per CU
#define SW(NM, NV) printf("same wave, 1 wave/SIMD: %2d MFMA + %3d VALU per iter: %7.3f ms  (MFMA alone %.3f ms at 32 cyc, VALU alone %.3f ms at 4 cyc, 2.4 GHz)\n", NM, NV, \
                          timeit([&] { hipLaunchKernelGGL((same_wave<NM, NV>), dim3(wgs), dim3(256), 0, 0, out, iters); }), iters * NM * 32 / 2.4e6, iters * NV * 4 / 2.4e6)
  SW(4, 0);
  SW(0, 32);
  SW(4, 8);
  SW(4, 16);
  SW(4, 24);
  SW(4, 32);
  SW(4, 48);
  for (int mode : {1, 2, 3})
    printf("split waves (4 MFMA waves + 4 VALU waves per CU), mode %d (1 MFMA only, 2 VALU only, 3 both): %7.3f ms\n", mode,
           timeit([&] { hipLaunchKernelGGL((split_waves<8>), dim3(wgs), dim3(512), 0, 0, out, iters, mode); }));
  return 0;
}
