// Small-map variant of the 4x4 convolution family (maps up to ~34x34 in, ~18x18 out): the D2
// discriminator runs on hundreds of 32x32 patches whose activations shrink to 2x2..17x17, where one
// workgroup per spatial tile of one image wastes >90% of every MFMA tile and pays the full
// per-chunk barrier cost per image.  Here the GEMM M dimension is the FLATTENED (image, y, x) pixel
// index of `IPB` whole images per workgroup, so MFMA tiles are dense regardless of the map size.
//
// Same arithmetic contract as conv4x4_kernel (vts_conv.hip): normalise + activate + concat on load,
// bias / tanh / derivative mask / accumulate epilogue, exact fp32 MFMA (v_mfma_f32_16x16x4_f32).
// LDS holds the zero-haloed (halo = 2) input planes of IPB images x CK channels; per-lane A-fragment
// bases are looked up per unit, everything else is uniform.
#include <stdlib.h>

#include "vts_internal.h"

namespace {

constexpr int HALO = 2;

struct SmallK {
  const float *s0, *s1, *sc0, *sh0, *sc1, *sh1;
  int64_t ns0, ns1;
  int C0, C1, Cin;
  int N, IH, IW, OH, OW, Cout, pad;
  const float* w;
  int ws_co, ws_ci;
  const float* bias;
  float* out;
  int64_t ons;
  int act_out;
  const float *dm, *dmsc, *dmsh;
  int64_t dmns;
  int dm_act, dmC;
  int accumulate;
  const float* ident;
  float slope_in;
  int IPB, PWi, plane;  // images per block, padded row pitch, padded plane size (floats)
  int GH, GW, PPI, MTP; // phase-grid dims, pixels per image per phase, M-tiles per phase
  int ablate;           // profiling only (env VTS_ABLATE)
  int wbytes;           // extent of the weight view in bytes (buffer descriptor)
  int fast;             // 1: pipelined flat staging (single source, Cin % 8 == 0, contiguous samples: see the kernel)
  int inbytes;          // extent of the input tensor in bytes (buffer descriptor of the flat loads)
};

template <int MODE, int S, int NR, int UMAX, int CK>
__global__ __launch_bounds__(256) void conv_small_kernel(const SmallK p) {
  constexpr int P = (MODE == 1 && S == 2) ? 4 : 1;
  constexpr int COP = (NR % 2 == 1) ? NR * 16 : NR * 16 + 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* patch = smem;                                   // [IPB][CK][plane]
  float* lds_w = smem + p.IPB * CK * p.plane;            // [CK][16][COP]
  int* pix_tab = reinterpret_cast<int*>(lds_w + CK * 16 * COP);   // pixel id -> img<<16 | y<<8 | x
  int* line_tab = pix_tab + p.MTP * 16;                           // staging line -> img<<16 | c<<8 | y

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m16 = lane & 15, kq = lane >> 4;
  const int n0 = blockIdx.x * p.IPB;
  const int co0 = blockIdx.y * NR * 16;
  const int imgstride = CK * p.plane;
  const int npix = p.IPB * p.PPI;
  const int nunits = p.MTP * P;

  // zero the whole patch once: halos (and rows of absent images / channels) stay zero
  for (int i = tid; i < p.IPB * CK * p.plane; i += 256) patch[i] = 0.f;
  // decode tables (integer divisions by run-time sizes happen once per entry, not per use)
  for (int q = tid; q < p.MTP * 16; q += 256) {
    const int img = q / p.PPI, rem = q - img * p.PPI;
    const int y = rem / p.GW, x = rem - y * p.GW;
    pix_tab[q] = q < npix ? (img << 16) | (y << 8) | x : -1;
  }
  for (int l = tid; l < p.IPB * CK * p.IH; l += 256) {
    const int img = l / (CK * p.IH), rem = l - img * (CK * p.IH);
    const int c = rem / p.IH, y = rem - c * p.IH;
    line_tab[l] = (img << 16) | (c << 8) | y;
  }
  __syncthreads();

  // per-lane A-fragment base of every unit this wave owns (unit = one 16-pixel M-tile of one phase)
  int abase[UMAX], uph[UMAX];
#pragma unroll
  for (int k = 0; k < UMAX; ++k) {
    const int uu = wave + 4 * k;
    const int ph = min(uu / p.MTP, P - 1), j = uu - (uu / p.MTP) * p.MTP;
    uph[k] = ph;
    const int q = 16 * j + m16;
    int off = 0;
    const int pt = uu < nunits ? pix_tab[q] : -1;
    if (pt >= 0) {
      const int img = pt >> 16, y = (pt >> 8) & 255, x = pt & 255;
      if (MODE == 0) {
        off = img * imgstride + (y * S - p.pad + HALO) * p.PWi + (x * S - p.pad + HALO) + kq;
      } else if (S == 2) {
        const int py = ph >> 1, px = ph & 1;
        off = img * imgstride + (y + ((py + p.pad) >> 1) - (kq >> 1) + HALO) * p.PWi + (x + ((px + p.pad) >> 1) - (kq & 1) + HALO);
      } else {
        off = img * imgstride + (y + p.pad + HALO) * p.PWi + (x + p.pad + HALO - kq);
      }
    }
    abase[k] = off;
  }

  f32x4 acc[UMAX][NR];
#pragma unroll
  for (int k = 0; k < UMAX; ++k)
#pragma unroll
    for (int nr = 0; nr < NR; ++nr) acc[k][nr] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int64_t iplane = (int64_t)p.IH * p.IW;
  const int nchunks = (p.Cin + CK - 1) / CK;
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc((void*)p.w, 0, p.wbytes, 0x00020000);
  const int half = lane >> 5, xl = lane & 31;
  const int nlines = p.IPB * CK * p.IH;   // (image, channel, row) lines of the inner region
  __syncthreads();

  // ---- MFMA accumulate of the staged chunk ----
  auto mfma_phase = [&](int cbase) {
    const int cvalid = (p.ablate & 2) ? 0 : min(CK, p.Cin - cbase);
    for (int c = 0; c < cvalid; ++c) {
      const float* pp = patch + c * p.plane;
      const float* ww = lds_w + c * 16 * COP + kq * COP + m16;
      if (MODE == 1 && S == 2) {
        // every unit slot computes (absent units read offset 0 and are discarded in the epilogue):
        // a per-unit branch costs more than the spare MFMA
#pragma unroll
        for (int k = 0; k < UMAX; ++k) {
          const float a = pp[abase[k]];
#pragma unroll
          for (int nr = 0; nr < NR; ++nr)
            acc[k][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, ww[uph[k] * 4 * COP + nr * 16], acc[k][nr], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float b[NR];
#pragma unroll
          for (int nr = 0; nr < NR; ++nr) b[nr] = ww[g * 4 * COP + nr * 16];
          const int goff = (MODE == 0) ? g * p.PWi : -g * p.PWi;
#pragma unroll
          for (int k = 0; k < UMAX; ++k) {
            const float a = pp[abase[k] + goff];
#pragma unroll
            for (int nr = 0; nr < NR; ++nr) acc[k][nr] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nr], acc[k][nr], 0, 0, 0);
          }
        }
      }
    }
  };

  if (p.fast) {
    // ---- pipelined flat staging (round 3).  The line-wise staging below spends one wave instruction per 2 map rows -- 5 - 9 useful lanes
    // of 32 on these maps -- and waits for every chunk's loads before it multiplies: 33 of 65 us on the 32 -> 64 layer of 640 5 x 5 maps
    // (tools/mb_small.py, VTS_ABLATE).  A chunk of 8 channels of one sample is 8 * IH * IW CONTIGUOUS floats: threads load 16-byte
    // quads of that run (fully used lanes), the quad -> (channel, y, x) decode is done once (it is the same for every chunk), and the
    // quads and the weight slice of chunk k + 1 are in flight during the MFMA phase of chunk k.
    constexpr int NQ = 3;                                   // quads per thread (the host side checks IPB * 2 * IH * IW <= 768)
    constexpr int NCO = NR * 16;
    constexpr int NWQ = CK * 4 * NCO / 256;                 // weight quads per thread
    float* aff_sc = reinterpret_cast<float*>(line_tab + p.IPB * CK * p.IH);   // [IPB][Cin] scale, then shift
    float* aff_sh = aff_sc + p.IPB * p.Cin;
    const int HW = p.IH * p.IW, QPI = 2 * HW, nq = p.IPB * QPI;
    for (int i = tid; i < p.IPB * p.Cin; i += 256) {
      const int img = i / p.Cin, ci = i - img * p.Cin;
      const int nn = min(n0 + img, p.N - 1);
      aff_sc[i] = p.sc0 ? p.sc0[nn * p.C0 + ci] : 1.f;
      aff_sh[i] = p.sh0 ? p.sh0[nn * p.C0 + ci] : 0.f;
    }
    unsigned qoff[NQ];
    int qpk[NQ][4];                                         // (affine index << 16) | patch offset of the four elements
#pragma unroll
    for (int j = 0; j < NQ; ++j) {
      const int q = tid + 256 * j;
      const int img = q / QPI, r = q - img * QPI;
      const bool ok = q < nq && n0 + img < p.N;
      qoff[j] = ok ? (unsigned)(((int64_t)(n0 + img) * p.ns0 + 4 * r) * 4) : 0x40000000u;
#pragma unroll
      for (int e4 = 0; e4 < 4; ++e4) {
        const int e = 4 * r + e4;
        const int cc = e / HW, pos = e - cc * HW;
        const int y = pos / p.IW, x = pos - y * p.IW;
        qpk[j][e4] = ok ? ((img * p.Cin + cc) << 16) | (img * imgstride + cc * p.plane + (y + HALO) * p.PWi + x + HALO) : -1;
      }
    }
    const __amdgpu_buffer_rsrc_t irs = __builtin_amdgcn_make_buffer_rsrc((void*)p.s0, 0, p.inbytes, 0x00020000);
    unsigned wvo[NWQ];
    int wdst[NWQ];
#pragma unroll
    for (int e = 0; e < NWQ; ++e) {
      const int u = tid + e * 256;
      const int co = u % NCO, rest = u / NCO;
      const int ky = rest & 3, cw = rest >> 2;
      wvo[e] = co0 + co < p.Cout ? (unsigned)((co0 + co) * p.ws_co + cw * p.ws_ci + ky * 4) * 4u : 0x40000000u;
      wdst[e] = (cw * 16) * COP + co;                        // + slot * COP per tap
    }
    f32x4 qv[NQ], wq[NWQ];
    auto load_chunk = [&](int cbase) {
      if (p.ablate & 1) return;
#pragma unroll
      for (int j = 0; j < NQ; ++j) qv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(irs, (int)qoff[j], cbase * HW * 4, 0));
#pragma unroll
      for (int e = 0; e < NWQ; ++e) wq[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)wvo[e], cbase * p.ws_ci * 4, 0));
    };
    auto store_chunk = [&](int cbase) {
#pragma unroll
      for (int j = 0; j < NQ; ++j) {
        const f32x4 v = qv[j];
#pragma unroll
        for (int e4 = 0; e4 < 4; ++e4) {
          const int pk = qpk[j][e4];
          if (pk >= 0) {
            const int ai = (pk >> 16) + cbase;
            const float t = fmaf(v[e4], aff_sc[ai], aff_sh[ai]);
            patch[pk & 0xFFFF] = fmaxf(t, 0.f) + p.slope_in * fminf(t, 0.f);
          }
        }
      }
#pragma unroll
      for (int e = 0; e < NWQ; ++e) {
        const f32x4 v = wq[e];
        const int u = tid + e * 256;
        const int ky = (u / NCO) & 3;
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          int slot = ky * 4 + kx;
          if (MODE == 1 && S == 2) slot = ((((ky + p.pad) & 1) * 2 + ((kx + p.pad) & 1)) * 4) + (ky >> 1) * 2 + (kx >> 1);
          lds_w[wdst[e] + slot * COP] = v[kx];
        }
      }
    };
#pragma unroll
    for (int j = 0; j < NQ; ++j) qv[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < NWQ; ++e) wq[e] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_chunk(0);
    __syncthreads();          // affine table complete
    store_chunk(0);
    __syncthreads();
    for (int chunk = 0; chunk < nchunks; ++chunk) {
      if (chunk + 1 < nchunks) load_chunk((chunk + 1) * CK);
      mfma_phase(chunk * CK);
      __syncthreads();
      if (chunk + 1 < nchunks) {
        store_chunk((chunk + 1) * CK);
        __syncthreads();
      }
    }
  } else
  for (int chunk = 0; chunk < nchunks; ++chunk) {
    const int cbase = chunk * CK;
    // ---- stage input lines: unconditional clamped loads first, branch-free finish afterwards ----
    if (!(p.ablate & 1))
    for (int l0 = 0; l0 < nlines; l0 += 64) {   // 8 lines per wave-instruction group x 8 batches
      float raw[8], rsc[8], rsh[8];
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int line = min(l0 + (wave * 2 + half) + 8 * b, nlines - 1);
        const int lt = line_tab[line];
        const int img = lt >> 16, c = (lt >> 8) & 255, y = lt & 255;
        const int nn = min(n0 + img, p.N - 1), cic = min(cbase + c, p.Cin - 1);
        const bool first = cic < p.C0;
        const int cl = first ? cic : cic - p.C0;
        const float* base = first ? p.s0 + nn * p.ns0 : p.s1 + nn * p.ns1;
        raw[b] = base[cl * iplane + (int64_t)y * p.IW + min(xl, p.IW - 1)];
        const float* scp = first ? p.sc0 : p.sc1;
        const float* shp = first ? p.sh0 : p.sh1;
        const int aidx = nn * (first ? p.C0 : p.C1) + cl;
        const bool hsc = scp != nullptr, hsh = shp != nullptr;
        rsc[b] = (hsc ? scp : p.ident)[hsc ? aidx : 0];
        rsh[b] = (hsh ? shp : p.ident)[hsh ? aidx : 1];
      }
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const int line = l0 + (wave * 2 + half) + 8 * b;
        const int lt = line_tab[min(line, nlines - 1)];
        const int img = lt >> 16, c = (lt >> 8) & 255, y = lt & 255;
        const bool ok = line < nlines && n0 + img < p.N && cbase + c < p.Cin && xl < p.IW;
        const float t = fmaf(raw[b], rsc[b], rsh[b]);
        const float a = fmaxf(t, 0.f) + p.slope_in * fminf(t, 0.f);
        if (line < nlines && xl < p.IW) patch[img * imgstride + c * p.plane + (y + HALO) * p.PWi + xl + HALO] = ok ? a : 0.f;
      }
      if (p.IW > 32) {  // columns 32.. of wide rows (only the 32x32 first layer of a 34-wide case)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
          const int line = l0 + (wave * 2 + half) + 8 * b;
          const int lt = line_tab[min(line, nlines - 1)];
          const int img = lt >> 16, c = (lt >> 8) & 255, y = lt & 255;
          const int x = 32 + xl;
          const int nn = min(n0 + img, p.N - 1), cic = min(cbase + c, p.Cin - 1);
          const bool first = cic < p.C0;
          const int cl = first ? cic : cic - p.C0;
          const float* base = first ? p.s0 + nn * p.ns0 : p.s1 + nn * p.ns1;
          const float v = base[cl * iplane + (int64_t)y * p.IW + min(x, p.IW - 1)];
          const bool ok = line < nlines && n0 + img < p.N && cbase + c < p.Cin && x < p.IW;
          const float t = fmaf(v, rsc[b], rsh[b]);
          const float a = fmaxf(t, 0.f) + p.slope_in * fminf(t, 0.f);
          if (line < nlines && x < p.IW) patch[img * imgstride + c * p.plane + (y + HALO) * p.PWi + x + HALO] = ok ? a : 0.f;
        }
      }
    }
    // ---- stage the weight slice: lds_w[c][slot][co].  A thread loads the four kx taps of one (cout, channel, ky) as one 16-byte
    // buffer load (out-of-range couts / channels carry the OOB offset: the hardware returns 0); threads of a wave differ in cout, so
    // the four LDS stores are bank-conflict-free (the former tap-fastest mapping was 16-way conflicted).
    {
      constexpr int NCO = NR * 16;
      constexpr int UNITS = CK * 4 * NCO;
#pragma unroll
      for (int e = 0; e < UNITS / 256; ++e) {
        const int u = tid + e * 256;
        const int co = u % NCO, rest = u / NCO;
        const int ky = rest & 3, c = rest >> 2;
        const bool ok = co0 + co < p.Cout && cbase + c < p.Cin;
        const unsigned vo = ok ? (unsigned)((co0 + co) * p.ws_co + (cbase + c) * p.ws_ci + ky * 4) * 4u : 0x40000000u;
        const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(wrs, (int)vo, 0, 0));
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
          int slot = ky * 4 + kx;
          if (MODE == 1 && S == 2) slot = ((((ky + p.pad) & 1) * 2 + ((kx + p.pad) & 1)) * 4) + (ky >> 1) * 2 + (kx >> 1);
          lds_w[(c * 16 + slot) * COP + co] = v[kx];
        }
      }
    }
    __syncthreads();
    mfma_phase(cbase);
    __syncthreads();
  }

  // ---- epilogue: D layout col (cout) = lane&15, row (pixel of the M-tile) = (lane>>4)*4 + reg ----
  const int64_t oplane = (int64_t)p.OH * p.OW;
#pragma unroll
  for (int k = 0; k < UMAX; ++k) {
    const int uu = wave + 4 * k;
    if (uu >= nunits) continue;
    const int ph = uu / p.MTP, j = uu - ph * p.MTP;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int q = 16 * j + kq * 4 + r;
      const int pt = pix_tab[q];
      if (pt < 0) continue;
      const int img = pt >> 16, gy = (pt >> 8) & 255, gx = pt & 255;
      const int y = (P == 4) ? gy * 2 + (ph >> 1) : gy, x = (P == 4) ? gx * 2 + (ph & 1) : gx;
      const int n = n0 + img;
      if (n >= p.N || y >= p.OH || x >= p.OW) continue;
#pragma unroll
      for (int nr = 0; nr < NR; ++nr) {
        const int co = co0 + nr * 16 + m16;
        if (co >= p.Cout) continue;
        float v = acc[k][nr][r] + (p.bias ? p.bias[co] : 0.f);
        if (p.act_out == VTS_ACT_TANH) v = tanhf(v);
        const int64_t o = co * oplane + (int64_t)y * p.OW + x;
        if (p.dm) {
          const float dsc = p.dmsc ? p.dmsc[n * p.dmC + co] : 1.f, dsh = p.dmsh ? p.dmsh[n * p.dmC + co] : 0.f;
          v *= vts_act_grad(p.dm[n * p.dmns + o] * dsc + dsh, p.dm_act);
        }
        float* ob = p.out + n * p.ons + o;
        *ob = p.accumulate ? *ob + v : v;
      }
    }
  }
}

template <int MODE, int S, int NR, int UMAX, int CK>
int launch_small(const SmallK& k, size_t lds_bytes, hipStream_t st) {
  dim3 grid(cdiv(k.N, k.IPB), cdiv(k.Cout, NR * 16));
  hipLaunchKernelGGL((conv_small_kernel<MODE, S, NR, UMAX, CK>), grid, dim3(256), lds_bytes, st, k);
  vts_set_kernel("conv_small_kernel<%d, %d, %d, %d, %d>%s", MODE, S, NR, UMAX, CK, k.fast ? "+flat" : "");
  VTS_CHECK_LAUNCH("vts_conv4x4 (small maps)");
  return VTS_OK;
}

}  // namespace

// Returns VTS_ERR_UNSUPPORTED when the shape is not a small-map case (caller falls back to the tiled kernel).
int vts_conv_small_try(const vts_conv_desc* d, hipStream_t st) {
  // only the standard geometry (square, non-negative padding; output size tied to the input size)
  if (d->pad_dx != 0 || d->pad < 0) return VTS_ERR_UNSUPPORTED;
  if (!d->transposed && (d->OH != (d->IH + 2 * d->pad - 4) / d->stride + 1 || d->OW != (d->IW + 2 * d->pad - 4) / d->stride + 1)) return VTS_ERR_UNSUPPORTED;
  if (d->transposed && ((d->OH + 2 * d->pad - 4) / d->stride + 1 != d->IH || (d->OW + 2 * d->pad - 4) / d->stride + 1 != d->IW)) return VTS_ERR_UNSUPPORTED;
  if (d->IH > 34 || d->IW > 34 || d->OH > 34 || d->OW > 34 || d->IW > 64 || d->N < 8) return VTS_ERR_UNSUPPORTED;
  const bool ph4 = d->transposed && d->stride == 2;
  const int P = ph4 ? 4 : 1;
  SmallK k;
  k.s0 = d->in0.data; k.sc0 = d->in0.scale; k.sh0 = d->in0.shift; k.ns0 = d->in0.nstride; k.C0 = d->in0.C;
  k.s1 = d->in1.data; k.sc1 = d->in1.scale; k.sh1 = d->in1.shift; k.ns1 = d->in1.nstride;
  k.C1 = d->in1.data ? d->in1.C : 0;
  k.Cin = k.C0 + k.C1;
  k.N = d->N; k.IH = d->IH; k.IW = d->IW; k.OH = d->OH; k.OW = d->OW; k.Cout = d->Cout; k.pad = d->pad;
  k.w = d->w; k.ws_co = d->ws_co; k.ws_ci = d->ws_ci; k.bias = d->bias;
  k.out = d->out; k.ons = d->out_nstride; k.act_out = d->act_out;
  k.dm = d->dmask.data; k.dmsc = d->dmask.scale; k.dmsh = d->dmask.shift; k.dmns = d->dmask.nstride;
  k.dm_act = d->dmask_act; k.dmC = d->dmask.C;
  k.accumulate = d->accumulate;
  k.ident = vts_ident();
  if (!k.ident) return VTS_ERR_UNSUPPORTED;
  k.slope_in = vts_slope(d->act_in);
  static const int ablate = vts_tune("VTS_ABLATE", 0);
  k.ablate = ablate;
  k.wbytes = (int)(((int64_t)(d->Cout - 1) * d->ws_co + (int64_t)(k.Cin - 1) * d->ws_ci + 16) * 4);
  k.PWi = d->IW + 2 * HALO + 1;
  k.plane = (d->IH + 2 * HALO) * k.PWi;
  k.GH = ph4 ? (d->OH + 1) / 2 : d->OH;
  k.GW = ph4 ? (d->OW + 1) / 2 : d->OW;
  k.PPI = k.GH * k.GW;
  // Work decomposition: these layers are latency-bound (tiny maps, up to 64 channels), so the goal is
  // >= ~256 workgroups with few K-chunks each: at most 2 cout tiles per workgroup (more cout groups),
  // 8 input channels per chunk, only as many images per block as needed, and a kernel instance whose
  // unit count per wave (1/2/4/8) just covers the block so that no unit slot needs a branch.
  const int nrt = (d->Cout + 15) / 16;
  const int NR = nrt >= 2 ? 2 : 1;
  constexpr int CK = 8;
  const int COP = (NR % 2 == 1) ? NR * 16 : NR * 16 + 16;
  const int groups = cdiv(d->Cout, NR * 16);
  int ipb = (4 * 8 * 16) / (k.PPI * P);                          // fills 8 units per wave
  const int lds_cap = (48 * 1024 / 4 - CK * 16 * COP) / (CK * k.plane);
  static const int small_wgs = vts_tune("VTS_SMALL_WGS", 256);
  int ipb_par = (d->N * groups) / small_wgs;                     // keeps >= ~256 workgroups
  if (ipb_par < 1) ipb_par = 1;
  if (ipb > lds_cap) ipb = lds_cap;
  if (ipb > ipb_par) ipb = ipb_par;
  if (ipb > d->N) ipb = d->N;
  if (ipb < 1) return VTS_ERR_UNSUPPORTED;
  k.IPB = ipb;
  k.MTP = cdiv(ipb * k.PPI, 16);
  const int upw = cdiv(k.MTP * P, 4);                            // units per wave
  if (upw > 8) return VTS_ERR_UNSUPPORTED;
  const int U = upw <= 1 ? 1 : (upw <= 2 ? 2 : (upw <= 4 ? 4 : 8));
  static const int fast_on = vts_tune("VTS_SMALL_FLAT", 1);
  k.inbytes = (int)((int64_t)d->N * d->in0.nstride * 4 < (int64_t)0x40000000 ? (int64_t)d->N * d->in0.nstride * 4 : 0);
  k.fast = fast_on && !d->in1.data && k.Cin % CK == 0 && d->in0.nstride == (int64_t)k.Cin * d->IH * d->IW && (reinterpret_cast<uintptr_t>(d->in0.data) & 15) == 0 &&
           ipb * 2 * d->IH * d->IW <= 768 && k.inbytes > 0 && (reinterpret_cast<uintptr_t>(d->w) & 15) == 0 && ((d->ws_co | d->ws_ci) & 3) == 0;
  const size_t lds = (size_t)(ipb * CK * k.plane + CK * 16 * COP + k.MTP * 16 + ipb * CK * d->IH + (k.fast ? 2 * ipb * k.Cin : 0)) * sizeof(float);
#define SMALL_U(MODE, S, NRV)                                                  \
  switch (U) {                                                                 \
    case 1: return launch_small<MODE, S, NRV, 1, CK>(k, lds, st);              \
    case 2: return launch_small<MODE, S, NRV, 2, CK>(k, lds, st);              \
    case 4: return launch_small<MODE, S, NRV, 4, CK>(k, lds, st);              \
    default: return launch_small<MODE, S, NRV, 8, CK>(k, lds, st);             \
  }
#define SMALL_CASE(MODE, S)             \
  if (NR == 1) { SMALL_U(MODE, S, 1) }  \
  SMALL_U(MODE, S, 2)
  if (!d->transposed) {
    if (d->stride == 2) { SMALL_CASE(0, 2) }
    SMALL_CASE(0, 1)
  }
  if (d->stride == 2) { SMALL_CASE(1, 2) }
  SMALL_CASE(1, 1)
#undef SMALL_U
#undef SMALL_CASE
}
