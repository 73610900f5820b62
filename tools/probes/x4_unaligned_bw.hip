// Probe: streaming bandwidth of raw-buffer dwordx4 loads at 16-byte-aligned vs dword-aligned (+4 B) addresses, and of dword loads,
// on gfx950.  Rows of W floats (W = 1024 aligned / 1025 unaligned pitch) are read by one wave instruction per 256 floats.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

template <int MODE>   // 0: x4 aligned pitch, 1: x4 with byte offset +4 (every load dword-aligned only), 2: dword loads
__global__ __launch_bounds__(256) void rd(const float* src, long long nfloats, float* out) {
  const rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)(nfloats * 4 > 0x7fffffffLL ? 0x7fffffff : nfloats * 4), 0x00020000);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  const long long per_wg = nfloats / gridDim.x;
  const long long base = per_wg * blockIdx.x;
  if (MODE < 2) {
    for (long long i = threadIdx.x * 4; i + 4 <= per_wg - 4; i += 256 * 4 * 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)((base + i + u * 1024) * 4 + (MODE == 1 ? 4 : 0)), 0, 0));
#pragma unroll
      for (int u = 0; u < 4; ++u) acc += v[u];
    }
  } else {
    for (long long i = threadIdx.x; i < per_wg; i += 256 * 16) {
      float v[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) v[u] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)((base + i + u * 256) * 4), 0, 0));
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[u & 3] += v[u];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) out[0] = 1.f;
}

int main() {
  const long long n = 96ll << 20;   // 384 MB
  float *d, *o;
  hipMalloc(&d, n * 4 + 64); hipMalloc(&o, 64);
  hipMemset(d, 0, n * 4 + 64);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      for (int k = 0; k < 5; ++k) {
        if (mode == 0) rd<0><<<4096, 256>>>(d, n, o);
        else if (mode == 1) rd<1><<<4096, 256>>>(d, n, o);
        else rd<2><<<4096, 256>>>(d, n, o);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep) printf("mode %d (%s): %.1f us per 384 MB = %.2f TB/s\n", mode, mode == 0 ? "x4 aligned" : mode == 1 ? "x4 +4B" : "dword", ms / 5 * 1e3, n * 4.0 / (ms / 5 * 1e-3) / 1e12);
    }
  }
  return 0;
}
