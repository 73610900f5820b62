// A host that is NOT Python running the generator's inference forward on libvts_hip.so (the network-level entry
// vts_unet_forward of include/vts.h).  This is what the hot loop of the reference's test.py does per item (test.py:62-74 ->
// SinSKITGModel.forward, models/sinskitG_model.py:1309-1319: CustomUnetGenerator on cat(sketch, positional grid)), without torch:
//   unet_infer_host <in.bin> <out.bin>
// in.bin  (little endian; written by tests/test_network_abi_gpu.py from a generator's state dict):
//   int32  magic 0x55535456 ("VTSU"), N, H, W, num_downs, num_layer_separate, in0_C, in1_C, style_C
//   int32  channels[num_downs], up_cout[num_downs], upT_cout[num_downs]
//   float  in0 [N][in0_C][H][W], in1 [N][in1_C][H][W], style [N][style_C][H >> nd][W >> nd]
//   float  per layer i: down_w, down_b, up_w, up_b, and for i < num_layer_separate upT_w, upT_b   (shapes as in include/vts.h)
// out.bin: float [N][up_cout[0] + upT_cout[0]][H][W]
// Build: hipcc -O2 -I include examples/unet_infer_host.cpp -L visual-tactile-synthesis_amd -lvts_hip -Wl,-rpath,'$ORIGIN/..' -o <bin>
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
  int32_t hdr[9];
  if (!read_exact(f, hdr, sizeof hdr) || hdr[0] != 0x55535456) { fprintf(stderr, "bad header\n"); return 1; }
  vts_unet_desc d{};
  d.N = hdr[1]; d.H = hdr[2]; d.W = hdr[3]; d.num_downs = hdr[4]; d.num_layer_separate = hdr[5];
  const int c0 = hdr[6], c1 = hdr[7], cs = hdr[8], nd = d.num_downs, nls = d.num_layer_separate;
  if (nd < 2 || nd > VTS_UNET_MAX_DOWNS) { fprintf(stderr, "num_downs %d\n", nd); return 1; }
  int32_t chan[3][VTS_UNET_MAX_DOWNS];
  for (int k = 0; k < 3; ++k)
    if (!read_exact(f, chan[k], sizeof(int32_t) * nd)) { fprintf(stderr, "short file\n"); return 1; }
  for (int i = 0; i < nd; ++i) { d.channels[i] = chan[0][i]; d.up_cout[i] = chan[1][i]; d.upT_cout[i] = chan[2][i]; }

  std::vector<float*> owned;
  auto upload = [&](int64_t n) -> float* {        // next n floats of the file -> device memory
    if (n == 0) return nullptr;
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
  const int64_t HW = (int64_t)d.H * d.W, hw_in = (int64_t)(d.H >> nd) * (d.W >> nd);
  d.in0 = vts_operand{upload(d.N * c0 * HW), nullptr, nullptr, c0, c0 * HW};
  d.in1 = vts_operand{upload(d.N * c1 * HW), nullptr, nullptr, c1, c1 * HW};
  d.style = vts_operand{upload(d.N * cs * hw_in), nullptr, nullptr, cs, cs * hw_in};
  for (int i = 0; i < nd; ++i) {
    const int cin_down = i == 0 ? c0 + c1 : d.channels[i - 1];
    // up_i reads cat(x, skip_i): x has channels[i] channels, skip_i as many again; the innermost block reads the encoder's last output
    // [+ style], the outermost has no skip
    const int cin_up = i == nd - 1 ? d.channels[i] + cs : (i == 0 ? d.channels[0] : 2 * d.channels[i]);
    d.down_w[i] = upload((int64_t)d.channels[i] * cin_down * 16);
    d.down_b[i] = upload(d.channels[i]);
    d.up_w[i] = upload((int64_t)cin_up * d.up_cout[i] * 16);
    d.up_b[i] = upload(d.up_cout[i]);
    if (i < nls) {
      d.upT_w[i] = upload((int64_t)cin_up * d.upT_cout[i] * 16);
      d.upT_b[i] = upload(d.upT_cout[i]);
    }
  }
  fclose(f);
  const int oc = d.up_cout[0] + (nls > 0 ? d.upT_cout[0] : 0);
  float* out = nullptr;
  HIP_OK(hipMalloc(&out, sizeof(float) * d.N * oc * HW));
  d.out = out;
  const int64_t need = vts_unet_forward_ws_floats(&d);
  if (need < 0) { fprintf(stderr, "vts_unet_forward_ws_floats: %s\n", vts_last_error()); return 3; }
  float* ws = nullptr;
  HIP_OK(hipMalloc(&ws, sizeof(float) * (size_t)need));
  hipStream_t st, side;
  HIP_OK(hipStreamCreate(&st));
  HIP_OK(hipStreamCreate(&side));
  d.side_stream = side;       // the tactile decoder branch runs beside the visual one
  hipEvent_t e0, e1;
  HIP_OK(hipEventCreate(&e0));
  HIP_OK(hipEventCreate(&e1));
  const int reps = 20;
  float ms = 0.f;
  for (int r = 0; r <= reps; ++r) {          // repetition 0 warms up; the forward is deterministic, every repetition writes the same output
    if (r == 1) HIP_OK(hipEventRecord(e0, st));
    const int rc = vts_unet_forward(&d, ws, need, st);
    if (rc != VTS_OK) { fprintf(stderr, "vts_unet_forward: %s\n", vts_last_error()); return 3; }
  }
  HIP_OK(hipEventRecord(e1, st));
  HIP_OK(hipStreamSynchronize(st));
  HIP_OK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<float> h((size_t)d.N * oc * HW);
  HIP_OK(hipMemcpy(h.data(), out, sizeof(float) * h.size(), hipMemcpyDeviceToHost));
  FILE* g = fopen(argv[2], "wb");
  if (!g || fwrite(h.data(), sizeof(float), h.size(), g) != h.size()) { perror(argv[2]); return 1; }
  fclose(g);
  printf("vts_unet_forward: N %d, %d x %d, %d down blocks: %.3f ms per forward = %.3f ms per image (%lld scratch floats)\n", d.N, d.H, d.W, nd, ms / reps,
         ms / reps / d.N, (long long)need);
  for (float* p : owned) (void)hipFree(p);
  (void)hipFree(out);
  (void)hipFree(ws);
  return 0;
}
