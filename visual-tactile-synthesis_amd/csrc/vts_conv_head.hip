// Single-output-channel member of the vts_conv4x4 family: the prediction heads of the PatchGAN discriminators
// (Conv2d(ndf*8, 1, 4, stride 1, pad 2), reference models/networks.py:1739-1741) on full-size maps.
//
// On the 16 x 16 x 4 MFMA tiles of conv4x4_kernel one of the sixteen output-channel columns carries work: the 64 -> 1 head at
// 131 x 131 took 78 us for 8 images, against 4.3 us of HBM time for its 34.6 MB input.  This layer is a 1024-term dot product per
// output pixel, so it runs on the vector ALUs instead: a workgroup owns a 32 x 32 output tile, stages the 35 x 35 input patch of 8
// channels at a time in LDS (normalise + LeakyReLU applied on the way, zero padding resolved there; the loads of the next chunk are
// in flight in registers while the current one is multiplied), and a thread computes four horizontally adjacent outputs: per
// channel and tap row two 16-byte LDS reads feed 16 FMAs with the four taps as a wave-uniform 16-byte broadcast read.
// A 131 x 131 map has only 25 tiles per image, i.e. ~1 workgroup per CU: to have several waves per SIMD (LDS latency is otherwise
// fully exposed: measured 65 us with 256 threads) the workgroup has 1024 threads -- four groups of 256 share the staged patch and
// split the channels of a chunk; their partial sums are combined through LDS in group order at the end.
// Algorithmic bytes = 4 (in + out + w).  Exact fp32; the summation order over (channel, ky, kx) is fixed.
#include <stdlib.h>

#include "vts_internal.h"

namespace {

struct HeadK {
  const float* x;
  const float *sc, *sh;
  int64_t ns;
  int C, IH, IW, OH, OW, pad, padx;
  const float* w;
  int ws_ci;
  const float* bias;
  float* out;
  int64_t ons;
  float slope;
  const float* ident;
};

constexpr int HK_TY = 32, HK_TX = 32, HK_PR = HK_TY + 3, HK_PC = HK_TX + 3, HK_PCP = 36;
constexpr int HK_MAXC = 512;

template <int CK>
__global__ __launch_bounds__(1024) void conv_head_kernel(const HeadK p) {
  constexpr int BLOCK = 1024, NG = BLOCK / 256, CPG = CK / NG;   // thread groups, channels of a chunk per group
  static_assert(CK % NG == 0, "a chunk splits evenly over the thread groups");
  constexpr int NEL = CK * HK_PR * HK_PC;            // staged elements per chunk
  constexpr int NLD = (NEL + BLOCK - 1) / BLOCK;     // ... per thread
  __shared__ __attribute__((aligned(16))) float tile[CK][HK_PR][HK_PCP];
  __shared__ __attribute__((aligned(16))) float wl[CK][16];
  __shared__ float ssc[HK_MAXC], ssh[HK_MAXC];
  const int tid = threadIdx.x, n = blockIdx.z;
  const int grp = __builtin_amdgcn_readfirstlane(tid >> 8), t256 = tid & 255;
  const int tx4 = t256 & 7, ty = t256 >> 3;
  const int oy0 = blockIdx.y * HK_TY, ox0 = blockIdx.x * HK_TX;
  const int iy0 = oy0 - p.pad, ix0 = ox0 - p.padx;
  for (int c = tid; c < p.C; c += BLOCK) {
    ssc[c] = p.sc ? p.sc[n * p.C + c] : 1.f;
    ssh[c] = p.sh ? p.sh[n * p.C + c] : 0.f;
  }
  // per-thread staging slots: element e = i * 256 + tid of the chunk -> (channel, row, column); fixed for all chunks
  int soff[NLD], sdst[NLD];   // offset inside the chunk's planes (or -1: padding / beyond the patch), LDS word
  const int64_t plane = (int64_t)p.IH * p.IW;
#pragma unroll
  for (int i = 0; i < NLD; ++i) {
    const int e = i * BLOCK + tid;
    const int c = e / (HK_PR * HK_PC), rem = e - c * (HK_PR * HK_PC);
    const int r = rem / HK_PC, col = rem - r * HK_PC;
    const int iy = iy0 + r, ix = ix0 + col;
    const bool in = e < NEL && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
    soff[i] = in ? iy * p.IW + ix : -1;
    sdst[i] = e < NEL ? (c * HK_PR + r) * HK_PCP + col : -1;
  }
  const float* xb = p.x + n * p.ns;
  float pv[NLD];
  // chunk-local channel of slot i: BLOCK consecutive elements cross at most one channel boundary (a plane has 1225 of them)
  static_assert(BLOCK <= HK_PR * HK_PC, "one boundary per slot");
  auto chan = [&](int i) {
    const int b = (i * BLOCK) / (HK_PR * HK_PC);
    return b + (tid >= (b + 1) * (HK_PR * HK_PC) - i * BLOCK ? 1 : 0);
  };
  auto load_chunk = [&](int c0) {
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = c0 + chan(i);
      pv[i] = (soff[i] >= 0 && c < p.C) ? xb[c * plane + soff[i]] : 0.f;
    }
  };
  auto store_chunk = [&](int c0) {
    float* t = &tile[0][0][0];
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      const int c = c0 + chan(i);
      if (sdst[i] >= 0) {
        const bool ok = soff[i] >= 0 && c < p.C;
        const int cc = min(c, p.C - 1);
        const float u = fmaf(pv[i], ssc[cc], ssh[cc]);
        t[sdst[i]] = ok ? fmaxf(u, 0.f) + p.slope * fminf(u, 0.f) : 0.f;
      }
    }
    if (tid < CK * 16) {
      const int c = c0 + (tid >> 4);
      wl[tid >> 4][tid & 15] = c < p.C ? p.w[(int64_t)c * p.ws_ci + (tid & 15)] : 0.f;
    }
  };
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  __syncthreads();   // ssc / ssh
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  for (int c0 = 0; c0 < p.C; c0 += CK) {
    const bool more = c0 + CK < p.C;
    if (more) load_chunk(c0 + CK);
#pragma unroll
    for (int cg = 0; cg < CPG; ++cg) {
      const int c = grp * CPG + cg;
#pragma unroll
      for (int ky = 0; ky < 4; ++ky) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(&tile[c][ty + ky][tx4 * 4]);
        const f32x4 b = *reinterpret_cast<const f32x4*>(&tile[c][ty + ky][tx4 * 4 + 4]);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&wl[c][ky * 4]);
        const float v[8] = {a[0], a[1], a[2], a[3], b[0], b[1], b[2], b[3]};
#pragma unroll
        for (int kx = 0; kx < 4; ++kx)
#pragma unroll
          for (int j = 0; j < 4; ++j) acc[j] = fmaf(v[j + kx], wv[kx], acc[j]);
      }
    }
    __syncthreads();
    if (more) {
      store_chunk(c0 + CK);
      __syncthreads();
    }
  }
  // partial sums of the thread groups, combined in group order (the tile memory is free after the last barrier)
  f32x4* red = reinterpret_cast<f32x4*>(&tile[0][0][0]);
  if (grp > 0) red[(grp - 1) * 256 + t256] = (f32x4){acc[0], acc[1], acc[2], acc[3]};
  __syncthreads();
  if (grp > 0) return;
#pragma unroll
  for (int g = 1; g < NG; ++g) {
    const f32x4 v = red[(g - 1) * 256 + t256];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] += v[j];
  }
  const float bias = p.bias ? p.bias[0] : 0.f;
  const int y = oy0 + ty;
  if (y < p.OH) {
    float* o = p.out + n * p.ons + (int64_t)y * p.OW;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int x = ox0 + tx4 * 4 + j;
      if (x < p.OW) o[x] = acc[j] + bias;
    }
  }
}

}  // namespace

int vts_conv_head_try(const vts_conv_desc* d, hipStream_t st) {
  static const int off = getenv("VTS_NO_HEAD") ? 1 : 0;
  if (off || d->transposed || d->stride != 1 || d->Cout != 1 || d->in1.data || d->dmask.data || d->accumulate || d->act_out != VTS_ACT_NONE)
    return VTS_ERR_UNSUPPORTED;
  if (d->in0.C > HK_MAXC || d->in0.C < 8 || (int64_t)d->OH * d->OW < 1024 || (int64_t)d->IH * d->IW * d->in0.C >= (1ll << 31))
    return VTS_ERR_UNSUPPORTED;
  HeadK k;
  k.x = d->in0.data; k.sc = d->in0.scale; k.sh = d->in0.shift; k.ns = d->in0.nstride; k.C = d->in0.C;
  k.IH = d->IH; k.IW = d->IW; k.OH = d->OH; k.OW = d->OW; k.pad = d->pad; k.padx = d->pad + d->pad_dx;
  k.w = d->w; k.ws_ci = d->ws_ci; k.bias = d->bias; k.out = d->out; k.ons = d->out_nstride;
  k.slope = vts_slope(d->act_in);
  k.ident = nullptr;
  dim3 grid(cdiv(d->OW, HK_TX), cdiv(d->OH, HK_TY), d->N);
  hipLaunchKernelGGL((conv_head_kernel<8>), grid, dim3(1024), 0, st, k);
  vts_set_kernel("conv_head_kernel<8>");
  VTS_CHECK_LAUNCH("vts_conv4x4 (head)");
  return VTS_OK;
}
