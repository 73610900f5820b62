// Probe: what does v_mfma_f32_16x16x4_f32 sustain on this part, alone and fed from LDS the way conv4x4 feeds it?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_peak.hip -o tools/probes/mfma_peak.bin && tools/probes/mfma_peak.bin
// Modes: 0 registers only; 1 every A operand is a fresh ds_read_b32 consumed right away (the compiler's conv4x4 schedule);
//        2 A operands of the next group are read while the current group multiplies (one group = 4 MFMAs ahead).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE, int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = lane * 0.01f, b = 1.f + lane * 0.001f;
  const float* pp = lds + (lane & 15) * 2 + (lane >> 4) + (tid >> 6) * 1024;
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[j % NACC], 0, 0, 0);
    }
  } else if (MODE == 1) {
    for (int it = 0; it < iters; ++it) {
      const float* q = pp + (it & 3) * 67;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        float av = q[j * 33];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, b, acc[j % NACC], 0, 0, 0);
      }
    }
  } else {
    float nx[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) nx[j] = pp[j * 33];
    for (int it = 0; it < iters; ++it) {
      const float* q = pp + (it & 3) * 67;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float cur[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) cur[j] = nx[j];
#pragma unroll
        for (int j = 0; j < 4; ++j) nx[j] = q[((g + 1) * 4 + j) * 33];
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[(g * 4 + j) % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur[j], b, acc[(g * 4 + j) % NACC], 0, 0, 0);
      }
    }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}

template <int MODE, int NACC>
void run(int wgs, int iters) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double fl = (double)wgs * 4 * iters * 16 * 2048.0;
  printf("mode %d nacc %d wgs %5d (%.0f waves/SIMD): %8.3f ms  %7.2f TF\n", MODE, NACC, wgs, wgs / 256.0, ms, fl / ms / 1e9);
  hipFree(out);
}

int main() {
  const int iters = 20000;
  for (int w : {256, 512, 1024, 2048}) run<0, 4>(w, iters);
  for (int w : {256, 1024}) run<0, 2>(w, iters);
  for (int w : {256, 1024}) run<0, 1>(w, iters);
  for (int w : {256, 512, 1024, 2048}) run<1, 4>(w, iters);
  for (int w : {256, 512, 1024, 2048}) run<2, 4>(w, iters);
  return 0;
}
