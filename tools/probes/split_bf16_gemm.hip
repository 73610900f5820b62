// Probe (round 6; VERDICT r5 item 7, as a MEASURED EXPERIMENT outside the product): the fp32 contraction of the convolution kernels emulated
// by bf16 MFMAs on split operands, against the exact-fp32 instruction the product uses (v_mfma_f32_16x16x4_f32).
//   x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)          (24 mantissa bits in three bf16 values)
//   3 products:  a1 b1 + a1 b2 + a2 b1                                                    (error ~ 2^-16 per product)
//   6 products:  a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)                        (error ~ 2^-24: fp32 level)
// on v_mfma_f32_16x16x32_bf16 (gfx950: 8 bf16 per lane and operand, K = 32 per instruction), fp32 accumulation, small terms first.
// Part 1: error of a 16 x 16 x K tile against float64 for the three forms (K = 512, 4608; operands with a 2^12 dynamic range).
// Part 2: MFMA issue time per K = 32 of contraction, register-resident operands, 1 .. 4 waves per SIMD.
// Part 3: vector-ALU cost of the split per element (what a kernel pays once per STAGED element, not per MFMA).
//   hipcc --offload-arch=gfx950 -O3 tools/probes/split_bf16_gemm.hip -o tools/probes/bin/split_bf16_gemm && tools/probes/bin/split_bf16_gemm
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float bf16_round(float x) {      // round to nearest even onto the bf16 grid, returned as float
  uint32_t u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return __uint_as_float(u & 0xffff0000u);
}
__device__ __forceinline__ unsigned short bf16_bits(float on_grid) { return (unsigned short)(__float_as_uint(on_grid) >> 16); }

struct Split3 {
  bf16x8 p[3];
};
__device__ __forceinline__ Split3 split8(const float (&x)[8]) {
  Split3 s;
  unsigned short b[3][8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x1 = bf16_round(x[i]);
    const float r1 = x[i] - x1;             // exact
    const float x2 = bf16_round(r1);
    const float x3 = bf16_round(r1 - x2);   // r1 - x2 exact
    b[0][i] = bf16_bits(x1); b[1][i] = bf16_bits(x2); b[2][i] = bf16_bits(x3);
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) s.p[k] = __builtin_bit_cast(bf16x8, b[k]);
  return s;
}

// ---- part 1: one wave, one 16 x 16 output tile, A [16][K] row-major, B [K][16] row-major ----
template <int MODE>
__global__ __launch_bounds__(64) void tile_kernel(const float* __restrict__ A, const float* __restrict__ B, int K, float* __restrict__ Cout) {
  const int lane = threadIdx.x, m = lane & 15, kq = lane >> 4;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, acc2 = {0.f, 0.f, 0.f, 0.f}, acc3 = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0) {
    for (int k0 = 0; k0 < K; k0 += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[m * K + k0 + kq], B[(k0 + kq) * 16 + m], acc, 0, 0, 0);
  } else {
    for (int k0 = 0; k0 < K; k0 += 32) {
      float a[8], b[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        a[i] = A[m * K + k0 + kq * 8 + i];
        b[i] = B[(k0 + kq * 8 + i) * 16 + m];
      }
      const Split3 sa = split8(a), sb = split8(b);
      // orders of magnitude apart: each order has its own accumulator, summed small to large at the end
      acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[0], sb.p[0], acc, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[0], sb.p[1], acc2, 0, 0, 0);
      acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[1], sb.p[0], acc2, 0, 0, 0);
      if (MODE == 2) {
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[0], sb.p[2], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[1], sb.p[1], acc3, 0, 0, 0);
        acc3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[2], sb.p[0], acc3, 0, 0, 0);
      }
    }
    acc = acc + (acc2 + acc3);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) Cout[(kq * 4 + r) * 16 + m] = acc[r];     // C/D: col = lane & 15, row = (lane >> 4) * 4 + reg
}

// ---- part 2: MFMA issue time per K = 32, register-resident ----
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
  const int lane = threadIdx.x & 63;
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float af = 0.001f * lane, bfv = 1.f + 0.002f * lane;
  float a8[8], b8[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { a8[i] = af + i; b8[i] = bfv - i; }
  const Split3 sa = split8(a8), sb = split8(b8);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < NACC; ++t) {       // NACC independent output tiles, K = 32 each per iteration
      if (MODE == 0) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a8[j], b8[j], acc[t], 0, 0, 0);
      } else {
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[0], sb.p[0], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[0], sb.p[1], acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[1], sb.p[0], acc[t], 0, 0, 0);
        if (MODE == 2) {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[0], sb.p[2], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[1], sb.p[1], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(sa.p[2], sb.p[0], acc[t], 0, 0, 0);
        }
      }
    }
  }
  f32x4 s = acc[0];
#pragma unroll
  for (int i = 1; i < NACC; ++i) s += acc[i];
  if (s[0] + s[1] + s[2] + s[3] == 123.456f) out[0] = 1.f;
}

// ---- part 3: the split itself, per element (8 elements per thread and iteration, results kept live) ----
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ x, unsigned* __restrict__ out, int iters) {
  float v[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = x[threadIdx.x * 8 + i];
  unsigned h = 0;
  for (int it = 0; it < iters; ++it) {
    const Split3 s = split8(v);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const uint4 w = __builtin_bit_cast(uint4, s.p[k]);
      h ^= w.x ^ w.y ^ w.z ^ w.w;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] += 1.0009765625f;      // fresh values every iteration
  }
  out[blockIdx.x * 256 + threadIdx.x] = h;
}

static double now_ms(hipEvent_t e0, hipEvent_t e1) {
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

template <int MODE>
static void error_case(int K) {
  std::vector<float> A(16 * K), B(K * 16);
  std::vector<double> ref(256, 0.0);
  uint64_t st = 88172645463325252ull + K;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
  for (auto& v : A) v = (float)((rnd() - 0.5) * exp2(12.0 * rnd() - 6.0));      // magnitudes over 2^12, both signs
  for (auto& v : B) v = (float)((rnd() - 0.5) * exp2(12.0 * rnd() - 6.0));
  for (int m = 0; m < 16; ++m)
    for (int n = 0; n < 16; ++n) {
      double s = 0.0;
      for (int k = 0; k < K; ++k) s += (double)A[m * K + k] * (double)B[k * 16 + n];
      ref[m * 16 + n] = s;
    }
  float *dA, *dB, *dC;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 256 * 4);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL((tile_kernel<MODE>), dim3(1), dim3(64), 0, 0, dA, dB, K, dC);
  std::vector<float> C(256);
  hipMemcpy(C.data(), dC, 256 * 4, hipMemcpyDeviceToHost);
  double num = 0, den = 0, worst = 0;
  for (int i = 0; i < 256; ++i) {
    const double e = C[i] - ref[i];
    num += e * e; den += ref[i] * ref[i];
    worst = fmax(worst, fabs(e) / fmax(fabs(ref[i]), 1e-30));
  }
  const char* name[] = {"fp32 MFMA 16x16x4            ", "bf16 split, 3 products       ", "bf16 split, 6 products       "};
  printf("  K %5d  %s rel-L2 %.3e   worst element %.3e\n", K, name[MODE], sqrt(num / den), worst);
  hipFree(dA); hipFree(dB); hipFree(dC);
}

template <int MODE, int NACC>
static void rate_case(int wgs) {
  float* out;
  hipMalloc(&out, 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((rate_kernel<MODE, NACC>), dim3(wgs), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
  }
  const double ms = now_ms(e0, e1);
  // fp32-equivalent flops: every (tile, K = 32) unit is 16 * 16 * 32 * 2 flops whatever the number of products
  const double units = (double)wgs * 4 * iters * NACC;
  const char* name[] = {"fp32 MFMA 16x16x4 (8 / unit)", "bf16 split, 3 products      ", "bf16 split, 6 products      "};
  printf("  %s  %d tiles/wave, %4d workgroups (%.0f waves/SIMD): %7.3f ms  %7.1f ns per 1e6 units  %7.1f fp32-equivalent TFLOP/s\n", name[MODE], NACC, wgs,
         wgs / 256.0, ms, ms * 1e6 / (units / 1e6) * 1e-3, units * 16384.0 / ms / 1e9);
  hipFree(out);
}

int main() {
  printf("part 1: error of a 16 x 16 x K tile against float64\n");
  for (int K : {512, 4608}) {
    error_case<0>(K);
    error_case<1>(K);
    error_case<2>(K);
  }
  printf("part 2: MFMA time per K = 32 unit of a 16 x 16 tile (register-resident operands)\n");
  for (int w : {256, 512, 1024}) {
    rate_case<0, 4>(w);
    rate_case<1, 4>(w);
    rate_case<2, 4>(w);
  }
  printf("part 3: the three-way split itself\n");
  {
    float* x;
    unsigned* out;
    const int wgs = 2048, iters = 2000;
    hipMalloc(&x, 256 * 8 * 4);
    hipMalloc(&out, (size_t)wgs * 256 * 4);
    hipMemset(x, 0x3f, 256 * 8 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(split_kernel, dim3(wgs), dim3(256), 0, 0, x, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
    }
    const double ms = now_ms(e0, e1), elems = (double)wgs * 256 * 8 * iters;
    printf("  %.3f ms for %.3g elements: %.2f T elements/s on the chip = %.1f lane-cycles per element at 2.4 GHz x 256 CUs x 4 SIMDs x 64 lanes / 4\n", ms, elems,
           elems / ms / 1e9, ms * 1e-3 * 2.4e9 * 256 * 4 * 16 / elems);
    hipFree(x); hipFree(out);
  }
  return 0;
}
