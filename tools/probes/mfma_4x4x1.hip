// Probe: v_mfma_f32_4x4x1_16b_f32 on gfx950 -- operand / result lane layout, the CBSZ / ABID broadcast of the A operand, and the rate it sustains.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_4x4x1.hip -o tools/probes/mfma_4x4x1.bin && tools/probes/mfma_4x4x1.bin
// Hypothesis (ISA, 16 blocks of 4x4x1): D[lane j+4b][reg i] = A[lane i+4b] * B[lane j+4b]; with CBSZ = 4 every block takes the A of block ABID.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ABID is an immediate: a compile-time loop hands every step its own constant
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>());
    static_for<I + 1, N>(f);
  }
}

template <int CBSZ, int ABID>
__global__ void sem(const float* a, const float* b, float* d) {
  const int lane = threadIdx.x;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  acc = __builtin_amdgcn_mfma_f32_4x4x1f32(a[lane], b[lane], acc, CBSZ, ABID, 0);
  for (int i = 0; i < 4; ++i) d[lane * 4 + i] = acc[i];
}

template <int NACC, int BC>
__global__ __launch_bounds__(256) void rate(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = lane * 0.01f, b = 1.f + lane * 0.001f;
  for (int it = 0; it < iters; ++it) {
    static_for<0, 16>([&](auto jc) {
      constexpr int j = decltype(jc)::value;
      if (BC) acc[j % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j % NACC], 4, j, 0);
      else acc[j % NACC] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[j % NACC], 0, 0, 0);
    });
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}

// the same with one fresh LDS operand per MFMA (B = 64 pixel values), read one group of 4 ahead
template <int NACC>
__global__ __launch_bounds__(256) void rate_lds(float* out, int iters) {
  __shared__ float lds[8192];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 8192; i += 256) lds[i] = (float)(i & 7) * 0.125f;
  __syncthreads();
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float a = lane * 0.01f;
  const float* pp = lds + lane + (tid >> 6) * 1024;
  float nx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) nx[j] = pp[j * 65];
  for (int it = 0; it < iters; ++it) {
    const float* q = pp + (it & 3) * 67;
    static_for<0, 4>([&](auto gc) {
      constexpr int g = decltype(gc)::value;
      float cur[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) cur[j] = nx[j];
#pragma unroll
      for (int j = 0; j < 4; ++j) nx[j] = q[((g + 1) * 4 + j) * 65];
      static_for<0, 4>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int c = 0; c < NACC; ++c) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, cur[j], acc[c], 4, (g * 4 + j) & 15, 0);
      });
    });
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
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

template <int CBSZ, int ABID>
void check(const float* da, const float* db, float* dd, const float* ha, const float* hb) {
  float hd[256];
  hipLaunchKernelGGL((sem<CBSZ, ABID>), dim3(1), dim3(64), 0, 0, da, db, dd);
  hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int i = 0; i < 4; ++i) {
      const int b = lane >> 2;
      const int src = CBSZ == 4 ? ABID : (CBSZ == 0 ? b : ((b >> CBSZ) << CBSZ) + ABID);
      const float want = ha[i + 4 * src] * hb[lane];
      if (hd[lane * 4 + i] != want) {
        if (bad < 4) printf("  cbsz %d abid %d lane %d reg %d: got %g want %g\n", CBSZ, ABID, lane, i, hd[lane * 4 + i], want);
        ++bad;
      }
    }
  printf("semantics cbsz %d abid %2d: %s (%d mismatches)\n", CBSZ, ABID, bad ? "DIFFERENT" : "as hypothesised", bad);
}

int main() {
  float ha[64], hb[64], *da, *db, *dd;
  for (int i = 0; i < 64; ++i) {
    ha[i] = 1.f + i;
    hb[i] = 100.f + 3.f * i;
  }
  hipMalloc(&da, 256);
  hipMalloc(&db, 256);
  hipMalloc(&dd, 1024);
  hipMemcpy(da, ha, 256, hipMemcpyHostToDevice);
  hipMemcpy(db, hb, 256, hipMemcpyHostToDevice);
  check<0, 0>(da, db, dd, ha, hb);
  check<4, 0>(da, db, dd, ha, hb);
  check<4, 5>(da, db, dd, ha, hb);
  check<4, 15>(da, db, dd, ha, hb);
  check<2, 1>(da, db, dd, ha, hb);
  float* out;
  hipMalloc(&out, 4);
  const int iters = 4096;
  for (int wgs : {256, 512, 1024}) {
    const double fl = (double)wgs * 4 * iters * 16 * 512.0;
    float ms;
    ms = timeit([&] { hipLaunchKernelGGL((rate<1, 0>), dim3(wgs), dim3(256), 0, 0, out, iters); });
    printf("4x4x1 dependent chain      wgs %4d: %7.3f ms %7.2f TF\n", wgs, ms, fl / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((rate<4, 0>), dim3(wgs), dim3(256), 0, 0, out, iters); });
    printf("4x4x1 4 accumulators       wgs %4d: %7.3f ms %7.2f TF\n", wgs, ms, fl / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((rate<8, 0>), dim3(wgs), dim3(256), 0, 0, out, iters); });
    printf("4x4x1 8 accumulators       wgs %4d: %7.3f ms %7.2f TF\n", wgs, ms, fl / ms / 1e9);
    ms = timeit([&] { hipLaunchKernelGGL((rate<8, 1>), dim3(wgs), dim3(256), 0, 0, out, iters); });
    printf("4x4x1 8 acc, cbsz 4        wgs %4d: %7.3f ms %7.2f TF\n", wgs, ms, fl / ms / 1e9);
    {
      const double fl3 = (double)wgs * 4 * iters * 16 * 3 * 512.0;
      ms = timeit([&] { hipLaunchKernelGGL((rate_lds<3>), dim3(wgs), dim3(256), 0, 0, out, iters); });
      printf("4x4x1 LDS operand, 3 blocks wgs %4d: %7.3f ms %7.2f TF\n", wgs, ms, fl3 / ms / 1e9);
      const double fl5 = (double)wgs * 4 * iters * 16 * 5 * 512.0;
      ms = timeit([&] { hipLaunchKernelGGL((rate_lds<5>), dim3(wgs), dim3(256), 0, 0, out, iters); });
      printf("4x4x1 LDS operand, 5 blocks wgs %4d: %7.3f ms %7.2f TF\n", wgs, ms, fl5 / ms / 1e9);
    }
  }
  return 0;
}
