// conv4x4_kernel instances of one operator of the family: MODE, stride = 1, 2 (vts_conv_kernel.h; dispatch table as in round 2)
#include "vts_conv_kernel.h"

int vts_conv_full_m1s2(const ConvK& k, int nr, int N, hipStream_t st) {
  switch (nr) {
    case 1: return vts_prefer_mt3(k, true, 4) ? launch<1, 2, 1, 1, 3, 4>(k, N, st) : launch<1, 2, 1, 1, 4, 4>(k, N, st);
    case 2: return launch<1, 2, 2, 1, 2, 4>(k, N, st);
    case 3: return launch<1, 2, 3, 1, 2, 4>(k, N, st);
    case 4: return launch<1, 2, 4, 1, 1, 4>(k, N, st);
    default: return launch<1, 2, 5, 1, 1, 4>(k, N, st);
  }
}

// small grids: one 16-channel output group per workgroup (CG groups) and, if asked, KS slices of the input-channel loop
int vts_conv_split_m1s2(const ConvK& k, int N, hipStream_t st, int CG, int KS, int ck) {
  return ck == 8 ? launch<1, 2, 1, 1, 2, 8>(k, N, st, CG, KS) : launch<1, 2, 1, 1, 2, 4>(k, N, st, CG, KS);
}
