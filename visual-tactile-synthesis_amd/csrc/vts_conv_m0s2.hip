// conv4x4_kernel instances of one operator of the family: MODE, stride = 0, 2 (vts_conv_kernel.h; dispatch table as in round 2)
#include "vts_conv_kernel.h"

int vts_conv_full_m0s2(const ConvK& k, int nr, int N, hipStream_t st) {
  if (nr == 1) {   // tile of the thin forward layers: 8x32 outputs (measured best of 8x64 / 4x64 / 4x32 / 8x32); VTS_TILE01=rw*10+mt
    static const int tile01 = vts_tune("VTS_TILE01", 22);
    if (tile01 == 14) return launch<0, 2, 1, 1, 4, 4>(k, N, st);
    if (tile01 == 12) return launch<0, 2, 1, 1, 2, 4>(k, N, st);
    if (tile01 == 22) return launch<0, 2, 1, 2, 2, 4>(k, N, st);
  }
  switch (nr) {
    case 1: return launch<0, 2, 1, 2, 4, 4>(k, N, st);
    case 2: return vts_prefer_mt3(k, false, 4) ? launch<0, 2, 2, 1, 3, 4>(k, N, st) : launch<0, 2, 2, 1, 4, 4>(k, N, st);
    case 3: return launch<0, 2, 3, 1, 4, 4>(k, N, st);
    case 4: return launch<0, 2, 4, 1, 2, 4>(k, N, st);
    default: return launch<0, 2, 5, 1, 2, 4>(k, N, st);
  }
}

// small grids: one 16-channel output group per workgroup (CG groups) and, if asked, KS slices of the input-channel loop
int vts_conv_split_m0s2(const ConvK& k, int N, hipStream_t st, int CG, int KS, int ck) {
  return ck == 8 ? launch<0, 2, 1, 1, 2, 8>(k, N, st, CG, KS) : launch<0, 2, 1, 1, 2, 4>(k, N, st, CG, KS);
}
