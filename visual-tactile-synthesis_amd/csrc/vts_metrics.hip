// Evaluation metrics on device (reference models/model_utils.py:431-561 compute_evaluation_metric): the ones that need no
// pretrained network -- I_PSNR (:496, on images min-max normalised with the REAL image's range, the fake one clamped to
// [0,1], :481-485), T_MSE (:557) and T_AE, the mean angle in degrees between the surface normals of the real and the fake
// tactile patches (:531-536, compute_normal :408-428 with scale_nz = 1, normal_losses.py:10-33 mode 'evaluate'); the fake
// tactile patches are clamped to [0,1] first, as the reference does (:521).  Streaming reductions: per-workgroup partials
// and a fixed-order final sum (deterministic).
#include "vts_internal.h"

namespace {

constexpr int MB = 512;   // workgroups of the first stage

__global__ __launch_bounds__(256) void minmax_partial_kernel(const float* __restrict__ x, int64_t n, float* __restrict__ part) {
  __shared__ float smin[4], smax[4];
  float lo = INFINITY, hi = -INFINITY;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    lo = fminf(lo, x[i]);
    hi = fmaxf(hi, x[i]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    smin[threadIdx.x >> 6] = lo;
    smax[threadIdx.x >> 6] = hi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    part[blockIdx.x * 2] = fminf(fminf(smin[0], smin[1]), fminf(smin[2], smin[3]));
    part[blockIdx.x * 2 + 1] = fmaxf(fmaxf(smax[0], smax[1]), fmaxf(smax[2], smax[3]));
  }
}

__global__ __launch_bounds__(64) void minmax_final_kernel(const float* __restrict__ part, int nb, float* __restrict__ out) {
  float lo = INFINITY, hi = -INFINITY;
  for (int i = threadIdx.x; i < nb; i += 64) {
    lo = fminf(lo, part[2 * i]);
    hi = fmaxf(hi, part[2 * i + 1]);
  }
  for (int o = 32; o > 0; o >>= 1) {
    lo = fminf(lo, __shfl_xor(lo, o, 64));
    hi = fmaxf(hi, __shfl_xor(hi, o, 64));
  }
  if (threadIdx.x == 0) {
    out[0] = lo;
    out[1] = hi;
  }
}

// mode 0: sum (clamp((b - lo) / (hi - lo), 0, 1) - (a - lo) / (hi - lo))^2   (lo, hi = range[0], range[1]; PSNR)
// mode 1: sum (a - clamp(b, 0, 1))^2                                           (T_MSE)
__global__ __launch_bounds__(256) void sqdiff_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n, int mode,
                                                              const float* __restrict__ range, float* __restrict__ part) {
  __shared__ float red[16];
  const float lo = mode == 0 ? range[0] : 0.f, inv = mode == 0 ? 1.f / (range[1] - range[0]) : 1.f;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const float ra = (a[i] - lo) * inv;
    const float fb = fminf(fmaxf((b[i] - lo) * inv, 0.f), 1.f);
    const float d = ra - fb;
    acc += d * d;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// real / fake tactile patches [P, 2, HW]: angle between normalize(gx, gy, 1) of the real and of the clamped fake patch
__global__ __launch_bounds__(256) void angle_partial_kernel(const float* __restrict__ real, const float* __restrict__ fake, int64_t P, int HW,
                                                             float* __restrict__ part) {
  __shared__ float red[16];
  const int64_t n = P * HW;
  float acc = 0.f;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
    const int64_t p = i / HW, o = i - p * HW;
    const float rx = real[(p * 2) * HW + o], ry = real[(p * 2 + 1) * HW + o];
    const float fx = fminf(fmaxf(fake[(p * 2) * HW + o], 0.f), 1.f), fy = fminf(fmaxf(fake[(p * 2 + 1) * HW + o], 0.f), 1.f);
    // F.normalize(eps 1e-12) then cosine_similarity(eps 1e-6): the normals have unit length, so cos = dot of the unit vectors
    const float rn = fmaxf(sqrtf(rx * rx + ry * ry + 1.f), 1e-12f), fn = fmaxf(sqrtf(fx * fx + fy * fy + 1.f), 1e-12f);
    const float ux = rx / rn, uy = ry / rn, uz = 1.f / rn, vx = fx / fn, vy = fy / fn, vz = 1.f / fn;
    const float nu = fmaxf(sqrtf(ux * ux + uy * uy + uz * uz), 1e-6f), nv = fmaxf(sqrtf(vx * vx + vy * vy + vz * vz), 1e-6f);
    float c = (ux * vx + uy * vy + uz * vz) / (nu * nv);
    c = fminf(fmaxf(c, -1.f), 1.f);
    acc += acosf(c) * 57.29577951308232f;
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc;
}

// out[0] = scale * sum(part)  (mode 0), or 10 log10(1 / (scale * sum))  (mode 1: PSNR with data_range 1)
__global__ __launch_bounds__(64) void metric_final_kernel(const float* __restrict__ part, int nb, float scale, int mode, float* __restrict__ out) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nb; i += 64) s += part[i];
  s = wave_sum(s);
  if (threadIdx.x == 0) out[0] = mode == 1 ? 10.f * log10f(1.f / (s * scale)) : s * scale;
}

// I_SSIM (:498-499): torchmetrics' structural_similarity_index_measure(data_range = 1) defaults -- 11 x 11 Gaussian window
// (sigma 1.5), k1 = 0.01, k2 = 0.03, reflect padding by 5 and the padded border cropped again, i.e. the mean of the SSIM map over the
// positions whose window lies inside the image -- on the same normalised images as the PSNR.  One output position per thread,
// 16 x 16 positions per tile from a 26 x 26 LDS tile of both images, workgroups walk the tiles (fixed order: deterministic).
struct SsimK {
  const float *a, *b, *range;
  int NC, H, W;
  float g[11], c1, c2;
  float* part;
};

__global__ __launch_bounds__(256) void ssim_partial_kernel(const SsimK p) {
  __shared__ float ta[26][27], tb[26][27];
  __shared__ float red[16];
  const float lo = p.range[0], inv = 1.f / (p.range[1] - p.range[0]);
  const int VH = p.H - 10, VW = p.W - 10;
  const int tiles_x = (VW + 15) / 16, tiles_y = (VH + 15) / 16;
  const int64_t ntiles = (int64_t)p.NC * tiles_y * tiles_x;
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  float acc = 0.f;
  for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
    const int64_t nc = t / (tiles_y * tiles_x);
    const int r = (int)(t - nc * tiles_y * tiles_x), y0 = (r / tiles_x) * 16, x0 = (r % tiles_x) * 16;
    const float* pa = p.a + nc * p.H * p.W;
    const float* pb = p.b + nc * p.H * p.W;
    __syncthreads();
    for (int e = threadIdx.x; e < 26 * 26; e += 256) {
      const int yy = e / 26, xx = e - yy * 26;
      const int y = min(y0 + yy, p.H - 1), x = min(x0 + xx, p.W - 1);
      ta[yy][xx] = (pa[(int64_t)y * p.W + x] - lo) * inv;
      tb[yy][xx] = fminf(fmaxf((pb[(int64_t)y * p.W + x] - lo) * inv, 0.f), 1.f);
    }
    __syncthreads();
    float ma = 0.f, mb = 0.f, saa = 0.f, sbb = 0.f, sab = 0.f;
    for (int dy = 0; dy < 11; ++dy)
#pragma unroll
      for (int dx = 0; dx < 11; ++dx) {
        const float w = p.g[dy] * p.g[dx], va = ta[ty + dy][tx + dx], vb = tb[ty + dy][tx + dx];
        ma += w * va; mb += w * vb; saa += w * va * va; sbb += w * vb * vb; sab += w * va * vb;
      }
    if (y0 + ty < VH && x0 + tx < VW) {
      const float va = fmaxf(saa - ma * ma, 0.f), vb = fmaxf(sbb - mb * mb, 0.f), cab = sab - ma * mb;
      acc += ((2.f * ma * mb + p.c1) * (2.f * cab + p.c2)) / ((ma * ma + mb * mb + p.c1) * (va + vb + p.c2));
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) p.part[blockIdx.x] = acc;
}

// ---- Fréchet distance between two feature sets (SIFID glue, reference models/sifid.py:102-176: calculate_activation_statistics
// = np.mean / np.cov(rowvar=False) in float64, calculate_frechet_distance = |mu1-mu2|^2 + tr(S1) + tr(S2) - 2 tr(sqrtm(S1 S2)) with
// scipy's sqrtm).  Features are channel-major [D, P] (one image's NCHW feature map, D <= 64 channels, P positions).  Moments are
// accumulated in float64 by FD_BLOCKS workgroups (fixed order: deterministic); tr(sqrtm(S1 S2)) by a coupled Newton-Schulz iteration in
// float64 on the 64 x 64 matrices in LDS (one workgroup): Y <- Y (3I - ZY) / 2, Z <- (3I - ZY) Z / 2 from Y = A / |A|_F, Z = I, so that
// Y -> sqrtm(A / |A|_F); 60 iterations cover condition numbers up to ~2^60.
constexpr int FD_D = 64, FD_BLOCKS = 128, FD_CHUNK = 64;

__global__ __launch_bounds__(256) void fd_moments_partial_kernel(const float* __restrict__ f, int D, int64_t P, double* __restrict__ part) {
  __shared__ float tile[FD_D][FD_CHUNK + 1];
  const int tid = threadIdx.x;
  double sx = 0.0, sxx[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) sxx[i] = 0.0;
  const int d0 = tid >> 2, e0 = (tid & 3) * 16;      // this thread's 16 (d, e) pairs: d = d0, e = e0 .. e0 + 15
  for (int64_t base = (int64_t)blockIdx.x * FD_CHUNK; base < P; base += (int64_t)gridDim.x * FD_CHUNK) {
    __syncthreads();
    for (int e = tid; e < FD_D * FD_CHUNK; e += 256) {
      const int d = e / FD_CHUNK, j = e - d * FD_CHUNK;
      tile[d][j] = (d < D && base + j < P) ? f[(int64_t)d * P + base + j] : 0.f;
    }
    __syncthreads();
    for (int j = 0; j < FD_CHUNK; ++j) {
      const double a = tile[d0][j];
      if ((tid & 3) == 0) sx += a;
#pragma unroll
      for (int i = 0; i < 16; ++i) sxx[i] += a * (double)tile[e0 + i][j];
    }
  }
  double* o = part + (int64_t)blockIdx.x * (FD_D + FD_D * FD_D);
  if ((tid & 3) == 0) o[d0] = sx;
#pragma unroll
  for (int i = 0; i < 16; ++i) o[FD_D + d0 * FD_D + e0 + i] = sxx[i];
}

// mean [64] and unbiased covariance [64][64] (zero rows / columns beyond D) from the partial raw moments
__global__ __launch_bounds__(256) void fd_moments_final_kernel(const double* __restrict__ part, int nb, int D, int64_t P, double* __restrict__ stat) {
  const int tid = threadIdx.x;
  __shared__ double mu[FD_D];
  if (tid < FD_D) {
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[(int64_t)b * (FD_D + FD_D * FD_D) + tid];
    mu[tid] = s / (double)P;
    stat[tid] = mu[tid];
  }
  __syncthreads();
  for (int e = tid; e < FD_D * FD_D; e += 256) {
    double s = 0.0;
    for (int b = 0; b < nb; ++b) s += part[(int64_t)b * (FD_D + FD_D * FD_D) + FD_D + e];
    const int d = e / FD_D, c = e - d * FD_D;
    stat[FD_D + e] = (d < D && c < D) ? (s - (double)P * mu[d] * mu[c]) / (double)(P - 1) : 0.0;
  }
}

// 64 x 64 float64 matrix steps of the Newton-Schulz iteration, one workgroup each (a workgroup may hold at most 64 KB of LDS -- two
// such matrices -- so the iteration runs as a chain of small launches on global scratch; kernel boundaries order them)
__global__ __launch_bounds__(256) void fd_init_kernel(const double* __restrict__ st1, const double* __restrict__ st2, double* __restrict__ Y,
                                                      double* __restrict__ Z, double* __restrict__ nrm_out) {
  __shared__ double red[256];
  const int tid = threadIdx.x;
  const double *S1 = st1 + FD_D, *S2 = st2 + FD_D;
  double a[16], fro = 0.0;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = tid + i * 256, r = e / FD_D, c = e - r * FD_D;
    double s = 0.0;
    for (int k = 0; k < FD_D; ++k) s += S1[r * FD_D + k] * S2[k * FD_D + c];
    a[i] = s;
    fro += s * s;
  }
  red[tid] = fro;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (tid < o) red[tid] += red[tid + o];
    __syncthreads();
  }
  const double nrm = sqrt(red[0]);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int e = tid + i * 256, r = e / FD_D, c = e - r * FD_D;
    Y[e] = nrm > 0.0 ? a[i] / nrm : 0.0;
    Z[e] = r == c ? 1.0 : 0.0;
  }
  if (tid == 0) nrm_out[0] = nrm;
}

// T = 3I - Z Y
__global__ __launch_bounds__(256) void fd_t_kernel(const double* __restrict__ Y, const double* __restrict__ Z, double* __restrict__ T) {
  const int tid = threadIdx.x;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int e = tid + i * 256, r = e / FD_D, c = e - r * FD_D;
    double s = 0.0;
    for (int k = 0; k < FD_D; ++k) s += Z[r * FD_D + k] * Y[k * FD_D + c];
    T[e] = (r == c ? 3.0 : 0.0) - s;
  }
}

// Y2 = Y T / 2, Z2 = T Z / 2
__global__ __launch_bounds__(256) void fd_yz_kernel(const double* __restrict__ Y, const double* __restrict__ Z, const double* __restrict__ T,
                                                    double* __restrict__ Y2, double* __restrict__ Z2) {
  const int tid = threadIdx.x;
#pragma unroll 4
  for (int i = 0; i < 16; ++i) {
    const int e = tid + i * 256, r = e / FD_D, c = e - r * FD_D;
    double a = 0.0, b = 0.0;
    for (int k = 0; k < FD_D; ++k) {
      a += Y[r * FD_D + k] * T[k * FD_D + c];
      b += T[r * FD_D + k] * Z[k * FD_D + c];
    }
    Y2[e] = 0.5 * a;
    Z2[e] = 0.5 * b;
  }
}

__global__ __launch_bounds__(64) void fd_final_kernel(const double* __restrict__ st1, const double* __restrict__ st2, const double* __restrict__ Y,
                                                      const double* __restrict__ nrm, int D, float* __restrict__ out) {
  if (threadIdx.x == 0) {
    const double *S1 = st1 + FD_D, *S2 = st2 + FD_D;
    double tr = 0.0, t1 = 0.0, t2 = 0.0, dm = 0.0;
    for (int d = 0; d < D; ++d) {
      tr += Y[d * FD_D + d];
      t1 += S1[d * FD_D + d];
      t2 += S2[d * FD_D + d];
      const double df = st1[d] - st2[d];
      dm += df * df;
    }
    out[0] = (float)(dm + t1 + t2 - 2.0 * sqrt(nrm[0]) * tr);
  }
}

inline int nblocks(int64_t n) {
  const int64_t b = (n + 255) / 256;
  return (int)(b < 1 ? 1 : (b > MB ? MB : b));
}

}  // namespace

extern "C" int64_t vts_metric_ws_floats(void) { return 2 * MB; }

extern "C" int vts_minmax(const float* x, int64_t n, float* out2, float* ws, void* stream) {
  VTS_CHECK_ARG(x && out2 && ws && n >= 1, "vts_minmax: bad args");
  const int nb = nblocks(n);
  hipLaunchKernelGGL(minmax_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, x, n, ws);
  hipLaunchKernelGGL(minmax_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, nb, out2);
  VTS_CHECK_LAUNCH("vts_minmax");
  return VTS_OK;
}

extern "C" int vts_metric_psnr(const float* real, const float* fake, int64_t n, const float* range2, float* out, float* ws, void* stream) {
  VTS_CHECK_ARG(real && fake && range2 && out && ws && n >= 1, "vts_metric_psnr: bad args");
  const int nb = nblocks(n);
  hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, real, fake, n, 0, range2, ws);
  hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, nb, 1.f / (float)n, 1, out);
  VTS_CHECK_LAUNCH("vts_metric_psnr");
  return VTS_OK;
}

extern "C" int vts_metric_tactile(const float* real_T, const float* fake_T, int64_t P, int HW, float* out_ae, float* out_mse, float* ws,
                                  void* stream) {
  VTS_CHECK_ARG(real_T && fake_T && out_ae && out_mse && ws && P >= 1 && HW >= 1, "vts_metric_tactile: bad args");
  const int64_t n = P * HW;
  const int nb = nblocks(n), nb2 = nblocks(2 * n);
  hipLaunchKernelGGL(angle_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, real_T, fake_T, P, HW, ws);
  hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, nb, 1.f / (float)n, 0, out_ae);
  hipLaunchKernelGGL(sqdiff_partial_kernel, dim3(nb2), dim3(256), 0, (hipStream_t)stream, real_T, fake_T, 2 * n, 1, (const float*)nullptr, ws + MB);
  hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws + MB, nb2, 1.f / (float)(2 * n), 0, out_mse);
  VTS_CHECK_LAUNCH("vts_metric_tactile");
  return VTS_OK;
}

extern "C" int vts_metric_ssim(const float* real, const float* fake, int NC, int H, int W, const float* range2, float* out, float* ws, void* stream) {
  VTS_CHECK_ARG(real && fake && range2 && out && ws && NC >= 1 && H >= 11 && W >= 11, "vts_metric_ssim: bad args (the 11 x 11 window needs H, W >= 11)");
  SsimK k;
  k.a = real; k.b = fake; k.range = range2; k.NC = NC; k.H = H; k.W = W; k.part = ws;
  float sum = 0.f;
  for (int i = 0; i < 11; ++i) { const float d = (float)(i - 5) / 1.5f; k.g[i] = expf(-0.5f * d * d); sum += k.g[i]; }
  for (int i = 0; i < 11; ++i) k.g[i] /= sum;
  k.c1 = 0.01f * 0.01f; k.c2 = 0.03f * 0.03f;
  const int64_t ntiles = (int64_t)NC * ((H - 10 + 15) / 16) * ((W - 10 + 15) / 16);
  const int nb = (int)(ntiles < MB ? ntiles : MB);
  hipLaunchKernelGGL(ssim_partial_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, k);
  hipLaunchKernelGGL(metric_final_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, ws, nb, 1.f / ((float)NC * (float)(H - 10) * (float)(W - 10)), 0, out);
  VTS_CHECK_LAUNCH("vts_metric_ssim");
  return VTS_OK;
}

extern "C" int64_t vts_frechet_ws_floats(void) { return 2 * ((int64_t)2 * (FD_BLOCKS + 1) * (FD_D + FD_D * FD_D) + 5 * FD_D * FD_D + 2); }

extern "C" int vts_frechet_distance(const float* feat1, const float* feat2, int D, int64_t P1, int64_t P2, float* out, float* ws, void* stream) {
  VTS_CHECK_ARG(feat1 && feat2 && out && ws && D >= 1 && D <= FD_D && P1 >= 2 && P2 >= 2, "vts_frechet_distance: bad args (D <= 64, P >= 2)");
  VTS_CHECK_ARG(((uintptr_t)ws & 7) == 0, "vts_frechet_distance: ws must be 8-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  const int64_t per = FD_D + FD_D * FD_D, MM = FD_D * FD_D;
  double* w = reinterpret_cast<double*>(ws);
  double* part[2] = {w, w + (FD_BLOCKS + 1) * per};
  double* stat[2] = {part[0] + FD_BLOCKS * per, part[1] + FD_BLOCKS * per};
  double* mats = w + 2 * (FD_BLOCKS + 1) * per;
  double *Y = mats, *Z = mats + MM, *T = mats + 2 * MM, *Y2 = mats + 3 * MM, *Z2 = mats + 4 * MM, *nrm = mats + 5 * MM;
  const float* f[2] = {feat1, feat2};
  const int64_t P[2] = {P1, P2};
  for (int i = 0; i < 2; ++i) {
    const int64_t chunks = (P[i] + FD_CHUNK - 1) / FD_CHUNK;
    const int nb = (int)(chunks < FD_BLOCKS ? chunks : FD_BLOCKS);
    hipLaunchKernelGGL(fd_moments_partial_kernel, dim3(nb), dim3(256), 0, st, f[i], D, P[i], part[i]);
    hipLaunchKernelGGL(fd_moments_final_kernel, dim3(1), dim3(256), 0, st, part[i], nb, D, P[i], stat[i]);
  }
  hipLaunchKernelGGL(fd_init_kernel, dim3(1), dim3(256), 0, st, stat[0], stat[1], Y, Z, nrm);
  for (int it = 0; it < 60; ++it) {      // 60 iterations: condition numbers up to ~2^60
    hipLaunchKernelGGL(fd_t_kernel, dim3(1), dim3(256), 0, st, Y, Z, T);
    hipLaunchKernelGGL(fd_yz_kernel, dim3(1), dim3(256), 0, st, Y, Z, T, Y2, Z2);
    double* t = Y; Y = Y2; Y2 = t;
    t = Z; Z = Z2; Z2 = t;
  }
  hipLaunchKernelGGL(fd_final_kernel, dim3(1), dim3(64), 0, st, stat[0], stat[1], Y, nrm, D, out);
  VTS_CHECK_LAUNCH("vts_frechet_distance");
  return VTS_OK;
}


// ---- SIFID input preparation (models/model_utils.py:481-488 for images, :541-549 for tactile patches): three-channel network
// input from C = 3 channels (or one channel tiled three times) of `src`, optional {lo, hi} min-max normalisation to (0, 1) with the
// fake image's clamp, optional clamp to (0, 1) of raw values, nearest-neighbour resize (F.interpolate default: source index =
// floor(dst * in / out) in float, as PyTorch's legacy nearest), and the 2x - 1 scaling of InceptionV3.forward (models/inception.py:135).
namespace {
__global__ __launch_bounds__(256) void sifid_input_kernel(const float* __restrict__ src, int64_t nstride, int c0, int C, int IH, int IW,
                                                          const float* __restrict__ lohi, int clamp01, float* __restrict__ out, int OH, int OW) {
  const int n = blockIdx.z, k = blockIdx.y;
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= OH * OW) return;
  const int y = o / OW, x = o - y * OW;
  const float sy = (float)IH / (float)OH, sx = (float)IW / (float)OW;
  const int iy = min((int)floorf((float)y * sy), IH - 1), ix = min((int)floorf((float)x * sx), IW - 1);
  float v = src[n * nstride + (int64_t)(c0 + (C == 1 ? 0 : k)) * IH * IW + (int64_t)iy * IW + ix];
  if (lohi) {
    v = (v - lohi[0]) / (lohi[1] - lohi[0]);
    if (clamp01) v = fminf(fmaxf(v, 0.f), 1.f);
    v = 2.f * v - 1.f;
  } else if (clamp01) {
    v = fminf(fmaxf(v, 0.f), 1.f);
  }
  out[((int64_t)n * 3 + k) * OH * OW + o] = v;
}
}  // namespace

extern "C" int vts_sifid_input(const float* src, int64_t nstride, int N, int c0, int C, int IH, int IW, const float* lohi, int clamp01,
                               float* out, int OH, int OW, void* stream) {
  VTS_CHECK_ARG(src && out && N >= 1 && (C == 1 || C == 3) && c0 >= 0 && IH >= 1 && IW >= 1 && OH >= 1 && OW >= 1, "vts_sifid_input: bad args");
  hipLaunchKernelGGL(sifid_input_kernel, dim3((unsigned)cdiv64((int64_t)OH * OW, 256), 3, N), dim3(256), 0, (hipStream_t)stream, src, nstride, c0, C,
                     IH, IW, lohi, clamp01, out, OH, OW);
  VTS_CHECK_LAUNCH("vts_sifid_input");
  return VTS_OK;
}
