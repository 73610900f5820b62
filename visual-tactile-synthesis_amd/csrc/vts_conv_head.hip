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
#include <limits.h>
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

// ---- small maps (the same heads on the D2 patch passes: 640 maps of 6x6 / 4x4 / 3x3 -> 7x7 / 5x5 / 4x4): one thread per output of
// IPB whole images per workgroup (IPB * OH * OW <= 256), the zero-haloed input planes of 16 channels at a time in LDS
// (normalise + LeakyReLU on load), taps as wave-uniform LDS broadcast reads.  The flattened-batch MFMA kernel took 73 us for the
// 640-patch pass (one of sixteen output-channel columns carries work); this is ~1 MFLOP per workgroup of plain FMAs.
struct HeadSK {
  const float* x;
  const float *sc, *sh;
  int64_t ns;
  int C, N, IH, IW, OH, OW, pad, padx;
  const float* w;
  int ws_ci;
  const float* bias;
  float* out;
  int64_t ons;
  float slope;
  int IPB, PH, PW, CK;   // CK: channels staged per pass (all of them when two images' worth fits in LDS: one pass, one latency chain)
};

__global__ __launch_bounds__(256) void conv_head_small_kernel(const HeadSK p) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* tile = smem;                                   // [IPB][p.CK][PH * PW]
  float* wl = smem + p.IPB * p.CK * p.PH * p.PW;       // [CK][16]
  const int tid = threadIdx.x;
  const int n0 = blockIdx.x * p.IPB;
  const int nimg = min(p.IPB, p.N - n0);
  const int plane = p.PH * p.PW, ohw = p.OH * p.OW, ihw = p.IH * p.IW;
  const int img = tid / ohw, r = tid - img * ohw;
  const int oy = r / p.OW, ox = r - oy * p.OW;
  const bool active = img < nimg;
  const int base = img * p.CK * plane + oy * p.PW + ox;          // window origin of this output inside channel 0 of its image
  for (int i = tid; i < p.IPB * p.CK * plane; i += 256) tile[i] = 0.f;
  float acc = 0.f;
  for (int c0 = 0; c0 < p.C; c0 += p.CK) {
    __syncthreads();
    // batches of 8 elements per thread: all global loads of a batch are issued before the first LDS store (a load -> store loop
    // exposes the full memory latency per element: measured 52 us for this kernel)
    const int total = nimg * p.CK * ihw;
    for (int e0 = tid; e0 < total; e0 += 256 * 8) {
      float raw[8], fsc[8], fsh[8];
      int dst[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int e = e0 + j * 256;
        const int ee = min(e, total - 1);
        const int ic = ee / ihw, q = ee - ic * ihw;
        const int im = ic / p.CK, c = ic - im * p.CK;
        const int y = q / p.IW, x = q - y * p.IW;
        const int cc = min(c0 + c, p.C - 1), n = n0 + im;
        raw[j] = p.x[n * p.ns + (int64_t)cc * ihw + q];
        fsc[j] = p.sc ? p.sc[n * p.C + cc] : 1.f;
        fsh[j] = p.sh ? p.sh[n * p.C + cc] : 0.f;
        dst[j] = (e < total && c0 + c < p.C) ? (im * p.CK + c) * plane + (y + p.pad) * p.PW + x + p.padx : -1 - ((im * p.CK + c) * plane + (y + p.pad) * p.PW + x + p.padx);
        if (e >= total) dst[j] = INT_MIN;
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float t = fmaf(raw[j], fsc[j], fsh[j]);
        if (dst[j] >= 0) tile[dst[j]] = fmaxf(t, 0.f) + p.slope * fminf(t, 0.f);
        else if (dst[j] != INT_MIN) tile[-1 - dst[j]] = 0.f;      // channel beyond C inside the last chunk
      }
    }
    for (int e = tid; e < p.CK * 16; e += 256) wl[e] = (c0 + (e >> 4) < p.C) ? p.w[(int64_t)(c0 + (e >> 4)) * p.ws_ci + (e & 15)] : 0.f;
    __syncthreads();
    if (active) {
#pragma unroll 4
      for (int c = 0; c < p.CK; ++c) {
        const float* t = tile + base + c * plane;
#pragma unroll
        for (int ky = 0; ky < 4; ++ky)
#pragma unroll
          for (int kx = 0; kx < 4; ++kx) acc = fmaf(t[ky * p.PW + kx], wl[c * 16 + ky * 4 + kx], acc);
      }
    }
  }
  if (active) p.out[(n0 + img) * p.ons + r] = acc + (p.bias ? p.bias[0] : 0.f);
}

}  // namespace

int vts_conv_head_try(const vts_conv_desc* d, hipStream_t st) {
  static const int off = vts_tune_set("VTS_NO_HEAD") ? 1 : 0;
  if (off || d->transposed || d->stride != 1 || d->Cout != 1 || d->in1.data || d->dmask.data || d->accumulate || d->act_out != VTS_ACT_NONE)
    return VTS_ERR_UNSUPPORTED;
  if (d->in0.C > HK_MAXC || d->in0.C < 8 || (int64_t)d->IH * d->IW * d->in0.C >= (1ll << 31)) return VTS_ERR_UNSUPPORTED;
  if ((int64_t)d->OH * d->OW < 1024) {
    // small maps with a large batch (D2 patch passes)
    const int padx = d->pad + d->pad_dx;
    if (d->N < 32 || d->OH * d->OW > 256 || d->pad < 0 || padx < 0 || d->IH + d->pad < d->OH + 3 - d->pad || d->IW + padx < d->OW + 3 - padx)
      return VTS_ERR_UNSUPPORTED;
    HeadSK q;
    q.x = d->in0.data; q.sc = d->in0.scale; q.sh = d->in0.shift; q.ns = d->in0.nstride; q.C = d->in0.C; q.N = d->N;
    q.IH = d->IH; q.IW = d->IW; q.OH = d->OH; q.OW = d->OW; q.pad = d->pad; q.padx = padx;
    q.w = d->w; q.ws_ci = d->ws_ci; q.bias = d->bias; q.out = d->out; q.ons = d->out_nstride; q.slope = vts_slope(d->act_in);
    q.PH = d->IH + 2 * d->pad; q.PW = d->IW + 2 * padx;
    q.IPB = 256 / (d->OH * d->OW);
    q.CK = d->in0.C;
    auto bytes = [&]() { return (int64_t)(q.IPB * q.CK * q.PH * q.PW + q.CK * 16) * 4; };
    while (q.IPB > 2 && bytes() > 60 * 1024) --q.IPB;
    if (bytes() > 60 * 1024) {      // not even two images with all channels: 16 channels per pass
      q.CK = 16;
      q.IPB = 256 / (d->OH * d->OW);
      while (q.IPB > 1 && bytes() > 60 * 1024) --q.IPB;
    }
    const int lds = (int)bytes();
    if (lds > 64 * 1024) return VTS_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(conv_head_small_kernel, dim3(cdiv(d->N, q.IPB)), dim3(256), lds, st, q);
    vts_set_kernel("conv_head_small_kernel");
    VTS_CHECK_LAUNCH("vts_conv4x4 (head, small maps)");
    return VTS_OK;
  }
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
