// A host that is NOT Python running the discriminator's training-mode forward on libvts_hip.so (the network-level
// entry vts_msd_forward of include/vts.h).  This is `netD2(fake_concat)` of a reference training step (models/sinskitG_model.py:1490-1501,
// 1781 -> MultiscaleDiscriminator.forward, models/networks.py:1649-1691): num_D PatchGANs over an average-pooled pyramid, BatchNorm with
// batch statistics, running statistics advanced -- without torch:
//   msd_forward_host <in.bin> <out.bin>
// in.bin  (little endian; written by tests/test_network_abi_gpu.py from a discriminator's state dict):
//   int32  magic 0x4453544d ("MTSD"), N, C, H, W, num_D, n_convs
//   int32  cout[n_convs], stride[n_convs], has_bn[n_convs]
//   float  input [N][C][H][W]
//   float  per scale s (full resolution first = the reference's layer<num_D-1-s>), per convolution j: weight [cout][cin][4][4], bias [cout],
//          and where has_bn[j]: gamma, beta, running_mean, running_var [cout]
// out.bin: per scale: float prediction [N][1][h][w]; then per scale and BatchNorm layer: running_mean [cout], running_var [cout]
// Build: hipcc -O2 -I include examples/msd_forward_host.cpp -L visual-tactile-synthesis_amd -lvts_hip -Wl,-rpath,'$ORIGIN/..' -o <bin>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#include "vts.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)

static bool read_exact(FILE* f, void* p, size_t bytes) { return fread(p, 1, bytes, f) == bytes; }

int main(int argc, char** argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s <in.bin> <out.bin>\n", argv[0]); return 1; }
  FILE* f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  int32_t hdr[7];
  if (!read_exact(f, hdr, sizeof hdr) || hdr[0] != 0x4453544d) { fprintf(stderr, "bad header\n"); return 1; }
  const int N = hdr[1], C = hdr[2], H = hdr[3], W = hdr[4], num_D = hdr[5], nc = hdr[6];
  if (num_D < 1 || num_D > VTS_MSD_MAX_SCALES || nc < 2 || nc > VTS_PATCHGAN_MAX_CONVS) { fprintf(stderr, "num_D %d / n_convs %d\n", num_D, nc); return 1; }
  int32_t meta[3][VTS_PATCHGAN_MAX_CONVS];
  for (int k = 0; k < 3; ++k)
    if (!read_exact(f, meta[k], sizeof(int32_t) * nc)) { fprintf(stderr, "short file\n"); return 1; }
  std::vector<float*> owned;
  auto upload = [&](int64_t n) -> float* {        // next n floats of the file -> device memory
    std::vector<float> h((size_t)n);
    if (!read_exact(f, h.data(), sizeof(float) * (size_t)n)) { fprintf(stderr, "short file\n"); exit(1); }
    float* p = nullptr;
    if (hipMalloc(&p, sizeof(float) * (size_t)n) != hipSuccess || hipMemcpy(p, h.data(), sizeof(float) * (size_t)n, hipMemcpyHostToDevice) != hipSuccess) {
      fprintf(stderr, "device upload failed\n");
      exit(2);
    }
    owned.push_back(p);
    return p;
  };
  vts_msd_desc d{};
  d.num_D = num_D;
  const int64_t HW = (int64_t)H * W;
  float* x = upload((int64_t)N * C * HW);
  std::vector<int64_t*> counters;
  std::vector<int> ph(num_D), pw(num_D);
  for (int s = 0, h = H, w = W; s < num_D; ++s, h = (h + 1) / 2, w = (w + 1) / 2) {
    vts_patchgan_desc& p = d.scale[s];
    p.N = N; p.H = h; p.W = w; p.n_convs = nc; p.eps = 1e-5f; p.momentum = 0.1f; p.run_head = 1;
    int cin = C, oh = h, ow = w;
    for (int j = 0; j < nc; ++j) {
      p.cout[j] = meta[0][j]; p.stride[j] = meta[1][j];
      p.w[j] = upload((int64_t)p.cout[j] * cin * 16);
      p.b[j] = upload(p.cout[j]);
      if (meta[2][j]) {
        p.gamma[j] = upload(p.cout[j]); p.beta[j] = upload(p.cout[j]);
        p.running_mean[j] = upload(p.cout[j]); p.running_var[j] = upload(p.cout[j]);
        int64_t* nbt = nullptr;
        HIP_OK(hipMalloc(&nbt, sizeof(int64_t)));
        HIP_OK(hipMemset(nbt, 0, sizeof(int64_t)));
        p.num_batches_tracked[j] = nbt;
        counters.push_back(nbt);
      }
      cin = p.cout[j];
      oh = oh / p.stride[j] + 1; ow = ow / p.stride[j] + 1;     // Conv2d(4, stride, padding 2)
    }
    ph[s] = oh; pw[s] = ow;
    float* pred = nullptr;
    HIP_OK(hipMalloc(&pred, sizeof(float) * (size_t)N * oh * ow));
    owned.push_back(pred);
    p.pred = pred;
  }
  fclose(f);
  d.scale[0].in0 = vts_operand{x, nullptr, nullptr, C, C * HW};
  const int64_t need = vts_msd_forward_ws_floats(&d);
  if (need < 0) { fprintf(stderr, "vts_msd_forward_ws_floats: %s\n", vts_last_error()); return 3; }
  float* ws = nullptr;
  HIP_OK(hipMalloc(&ws, sizeof(float) * (size_t)need));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));
  if (vts_msd_forward(&d, ws, need, st) != VTS_OK) { fprintf(stderr, "vts_msd_forward: %s\n", vts_last_error()); return 3; }
  HIP_OK(hipStreamSynchronize(st));
  FILE* g = fopen(argv[2], "wb");
  if (!g) { perror(argv[2]); return 1; }
  auto dump = [&](const float* p, int64_t n) {
    std::vector<float> h((size_t)n);
    if (hipMemcpy(h.data(), p, sizeof(float) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess || fwrite(h.data(), sizeof(float), h.size(), g) != h.size()) {
      fprintf(stderr, "write failed\n");
      exit(1);
    }
  };
  for (int s = 0; s < num_D; ++s) dump(d.scale[s].pred, (int64_t)N * ph[s] * pw[s]);
  for (int s = 0; s < num_D; ++s)
    for (int j = 0; j < nc; ++j)
      if (meta[2][j]) { dump(d.scale[s].running_mean[j], d.scale[s].cout[j]); dump(d.scale[s].running_var[j], d.scale[s].cout[j]); }
  fclose(g);
  int64_t tracked = -1;
  if (!counters.empty()) HIP_OK(hipMemcpy(&tracked, counters[0], sizeof(int64_t), hipMemcpyDeviceToHost));
  printf("vts_msd_forward: N %d, %d channels, %d x %d, %d scales of %d convolutions: ok (%lld scratch floats, num_batches_tracked %lld)\n", N, C, H, W, num_D, nc,
         (long long)need, (long long)tracked);
  for (float* p : owned) (void)hipFree(p);
  for (int64_t* p : counters) (void)hipFree(p);
  (void)hipFree(ws);
  return 0;
}
