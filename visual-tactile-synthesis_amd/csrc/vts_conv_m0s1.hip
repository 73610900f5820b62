// conv4x4_kernel instances of one operator of the family: MODE, stride = 0, 1 (vts_conv_kernel.h; dispatch table as in round 2)
#include "vts_conv_kernel.h"

int vts_conv_full_m0s1(const ConvK& k, int nr, int N, hipStream_t st) {
  switch (nr) {
    case 1: return launch<0, 1, 1, 2, 4, 4>(k, N, st);
    case 2: return launch<0, 1, 2, 1, 4, 4>(k, N, st);
    case 3: return launch<0, 1, 3, 1, 4, 4>(k, N, st);
    case 4: return launch<0, 1, 4, 1, 2, 4>(k, N, st);   // (MT = 3 measured neutral here: 32 -> 64 at 130^2 111 vs 113 us)
    default: return launch<0, 1, 5, 1, 2, 4>(k, N, st);
  }
}

// small grids: one 16-channel output group per workgroup (CG groups) and, if asked, KS slices of the input-channel loop
int vts_conv_split_m0s1(const ConvK& k, int N, hipStream_t st, int CG, int KS, int ck) {
  return ck == 8 ? launch<0, 1, 1, 1, 2, 8>(k, N, st, CG, KS) : launch<0, 1, 1, 1, 2, 4>(k, N, st, CG, KS);
}
