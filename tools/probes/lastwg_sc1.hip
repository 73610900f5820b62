// Probe (round 4): what does "the last workgroup merges the partials" cost on a multi-XCD part, and can it be had without the
// device-scope fence (buffer_wbl2 / buffer_inv of a whole L2) that made VTS_FUSE_FINALIZE=1 three times slower than a second launch?
//   A  two launches: producer (streams `bytes` per workgroup, writes one partial) + a one-workgroup merge kernel
//   B  one launch, __threadfence() + atomicAdd + __threadfence() in the last workgroup (the classic pattern)
//   C  one launch, partial written by an agent-scope relaxed atomic store (sc1: write-through past the XCD's L2), s_waitcnt, agent-scope
//      atomicAdd; the last workgroup reads the partials with agent-scope relaxed atomic loads -- no fence anywhere
// Every variant checks the merged sum on the host over many repetitions (stale or missing partials show up as a wrong sum).
// hipcc --offload-arch=gfx950 -O3 -o bin/lastwg_sc1 lastwg_sc1.hip && ./bin/lastwg_sc1
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ float work(const float* __restrict__ in, float* __restrict__ out, int floats, int rep) {
  // stream: read `floats`, write `floats` (dirty lines in this XCD's L2, like a convolution's output tile)
  const int64_t base = (int64_t)blockIdx.x * floats;
  float acc = 0.f;
  for (int i = threadIdx.x; i < floats; i += 256) {
    const float v = in[base + i] + (float)rep;
    out[base + i] = v;
    acc += v;
  }
  return acc;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
  for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}

// the partial a workgroup publishes: a small integer that depends on (workgroup, repetition) -- exact in any summation order -- tied to the
// streamed data so that the compiler keeps the work
__device__ __forceinline__ float partial_of(float acc, int rep) { return (float)((blockIdx.x + rep) % 13) + (acc == -1.f ? 1.f : 0.f); }

__global__ __launch_bounds__(256) void producer_kernel(const float* in, float* out, int floats, int rep, float* part) {
  __shared__ float red[4];
  const float s = partial_of(block_sum(work(in, out, floats, rep), red), rep);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void merge_kernel(const float* part, int n, float* result) {
  __shared__ float red[4];
  float a = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) a += part[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) *result = a;
}

__global__ __launch_bounds__(256) void fence_kernel(const float* in, float* out, int floats, int rep, float* part, int* counter, float* result) {
  __shared__ float red[4];
  __shared__ int flag;
  const float s = partial_of(block_sum(work(in, out, floats, rep), red), rep);
  if (threadIdx.x == 0) part[blockIdx.x] = s;
  __threadfence();
  __syncthreads();
  if (threadIdx.x == 0) {
    const int t = atomicAdd(counter, 1);
    flag = t == (int)gridDim.x - 1;
    if (flag) *counter = 0;
  }
  __syncthreads();
  if (!flag) return;
  __threadfence();
  float a = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) a += ((const volatile float*)part)[i];
  a = block_sum(a, red);
  if (threadIdx.x == 0) *result = a;
}

__global__ __launch_bounds__(256) void sc1_kernel(const float* in, float* out, int floats, int rep, float* part, int* counter, float* result) {
  __shared__ float red[4];
  __shared__ int flag;
  const float s = partial_of(block_sum(work(in, out, floats, rep), red), rep);
  if (threadIdx.x == 0) {
    __hip_atomic_store(&part[blockIdx.x], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_s_waitcnt(0);      // the write-through store is acknowledged before the counter moves
    const int t = __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    flag = t == (int)gridDim.x - 1;
    if (flag) __hip_atomic_store(counter, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (!flag) return;
  float a = 0.f;
  for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) a += __hip_atomic_load(&part[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  a = block_sum(a, red);
  if (threadIdx.x == 0) *result = a;
}

int main() {
  const int reps = 200;
  for (int wgs : {256, 1024, 4096}) {
    for (int floats : {1024, 16384}) {
      const int64_t n = (int64_t)wgs * floats;
      float *in, *out, *part, *result;
      int* counter;
      CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4)); CK(hipMalloc(&part, wgs * 4)); CK(hipMalloc(&result, 4 * reps)); CK(hipMalloc(&counter, 4));
      CK(hipMemset(in, 0, n * 4)); CK(hipMemset(counter, 0, 4)); CK(hipMemset(part, 0, wgs * 4));
      std::vector<float> h(reps);
      for (int variant = 0; variant < 3; ++variant) {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipMemset(result, 0, 4 * reps));
        for (int pass = 0; pass < 2; ++pass) {     // pass 0 warms up
          CK(hipEventRecord(e0));
          for (int r = 0; r < reps; ++r) {
            if (variant == 0) {
              hipLaunchKernelGGL(producer_kernel, dim3(wgs), dim3(256), 0, 0, in, out, floats, r, part);
              hipLaunchKernelGGL(merge_kernel, dim3(1), dim3(256), 0, 0, part, wgs, result + r);
            } else if (variant == 1) {
              hipLaunchKernelGGL(fence_kernel, dim3(wgs), dim3(256), 0, 0, in, out, floats, r, part, counter, result + r);
            } else {
              hipLaunchKernelGGL(sc1_kernel, dim3(wgs), dim3(256), 0, 0, in, out, floats, r, part, counter, result + r);
            }
          }
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
        }
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        CK(hipMemcpy(h.data(), result, 4 * reps, hipMemcpyDeviceToHost));
        int bad = 0;
        for (int r = 0; r < reps; ++r) {
          long want = 0;
          for (int b = 0; b < wgs; ++b) want += (b + r) % 13;
          bad += h[r] != (float)want;
        }
        printf("wgs %5d x %6d floats  %-28s %8.2f us / iteration   wrong sums %d of %d\n", wgs, floats,
               variant == 0 ? "A two launches" : variant == 1 ? "B one launch, __threadfence" : "C one launch, sc1 store/load", ms * 1000.f / reps, bad, reps);
      }
      CK(hipFree(in)); CK(hipFree(out)); CK(hipFree(part)); CK(hipFree(result)); CK(hipFree(counter));
    }
  }
  return 0;
}
