// Probe: semantics of raw buffer loads on gfx950 -- per-dword range check of dwordx4, unaligned (dword-aligned) dwordx4,
// whether soffset takes part in the range check.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* src, int nbytes, float* out) {
  auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
  const int lane = threadIdx.x;
  // case A: dwordx4 at byte offset 4*lane (unaligned for most lanes), crossing num_records for the last lanes
  f32x4 a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, 4 * lane, 0, 0));
  out[lane * 4 + 0] = a[0]; out[lane * 4 + 1] = a[1]; out[lane * 4 + 2] = a[2]; out[lane * 4 + 3] = a[3];
  // case B: soffset pushes the address beyond num_records while voffset stays in range
  float b = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * lane, nbytes, 0));
  out[256 + lane] = b;
  // case C: huge voffset
  float c = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, lane & 1 ? 0x80000000u : 4 * lane, 0, 0));
  out[320 + lane] = c;
}

int main() {
  const int n = 64;                 // 64 floats in the descriptor; the allocation is larger so over-reads are harmless here
  float h[256];
  for (int i = 0; i < 256; ++i) h[i] = 100.f + i;
  float *d, *o;
  hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4096);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipMemset(o, 0, 4096);
  probe<<<1, 64>>>(d, n * 4, o);
  float r[1024];
  hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
  printf("A (dwordx4 at dword offset lane; valid floats 0..63 = 100..163):\n");
  for (int l = 56; l < 64; ++l) printf("  lane %d: %g %g %g %g\n", l, r[l * 4], r[l * 4 + 1], r[l * 4 + 2], r[l * 4 + 3]);
  printf("  lane 1 (unaligned): %g %g %g %g\n", r[4], r[5], r[6], r[7]);
  printf("B (soffset = num_records, voffset in range): lane0 %g lane5 %g  (164/169 = soffset not range-checked)\n", r[256], r[261]);
  printf("C (voffset 0x80000000 on odd lanes): lane0 %g lane1 %g lane2 %g lane3 %g\n", r[320], r[321], r[322], r[323]);
  return 0;
}
