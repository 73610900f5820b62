// Probe: direction and edge behaviour of the whole-wave DPP shifts on gfx950 (wave_shl:1 = 0x130, wave_shr:1 = 0x138; bound_ctrl on).
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void k(int* out) {
  const int lane = threadIdx.x;
  out[lane] = __builtin_amdgcn_update_dpp(-1, lane + 100, 0x130, 0xF, 0xF, true);
  out[64 + lane] = __builtin_amdgcn_update_dpp(-1, lane + 100, 0x138, 0xF, 0xF, true);
  out[128 + lane] = __builtin_amdgcn_update_dpp(-1, lane + 100, 0x130, 0xF, 0xF, false);
}
int main() {
  int *d, h[192];
  hipMalloc(&d, sizeof h);
  k<<<1, 64>>>(d);
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  printf("wave_shl:1 bound_ctrl: lane0 %d lane1 %d lane15 %d lane16 %d lane62 %d lane63 %d  (lane i <- lane i+1 shows 101 102 116 117 163 0)\n", h[0], h[1], h[15], h[16], h[62], h[63]);
  printf("wave_shr:1 bound_ctrl: lane0 %d lane1 %d lane15 %d lane16 %d lane62 %d lane63 %d\n", h[64], h[65], h[79], h[80], h[126], h[127]);
  printf("wave_shl:1 no bound_ctrl (old = -1): lane62 %d lane63 %d\n", h[190], h[191]);
  return 0;
}
