"""Forward / forward+backward time of pix2pixHD's GlobalGenerator (ngf 64, 4 downsamplings, 9 blocks: 1024 channels)
on the HIP path, with HIP events.  python tools/probes/bench_global.py [H W [N]]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

from models import networks  # noqa: E402
from vts import engine  # noqa: E402
from vts.optim import FlatParams  # noqa: E402


def main():
    h, w = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (256, 512)
    n = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    dev = torch.device("cuda:0")
    G = networks.define_G(1, 5, 64, "global", "batch", gpu_ids=[0])
    FlatParams(G)
    G.train()
    x = torch.randn(n, 1, h, w, device=dev)

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def fwd():
        return engine.resnet_forward(G, x, keep=False)

    def fwd_bwd():
        y, ctx = engine.resnet_forward(G, x)
        engine.resnet_backward(G, ctx, torch.ones_like(y))

    # conv MACs: 2 * sum over layers (counted at the true 3x3 / 7x7 tap counts)
    flops = 0.0
    c, hh, ww = 64, h, w
    flops += 2 * 49 * 1 * 64 * h * w
    for _ in range(4):
        hh, ww = hh // 2, ww // 2
        flops += 2 * 9 * c * 2 * c * hh * ww
        c *= 2
    flops += 9 * 2 * 2 * 9 * c * c * hh * ww
    for _ in range(4):
        flops += 2 * 9 * c * (c // 2) * hh * ww      # transposed: each input pixel meets 9 taps
        hh, ww, c = hh * 2, ww * 2, c // 2
    flops += 2 * 49 * 64 * 5 * h * w
    flops *= n
    tf = timed(fwd)
    tb = timed(fwd_bwd)
    print("GlobalGenerator N%d %dx%d: forward %.1f ms (%.1f TFLOP/s useful), forward+backward %.1f ms (%.1f TFLOP/s useful); "
          "%.1f GFLOP forward" % (n, h, w, tf, flops / tf / 1e9, tb, 3 * flops / tb / 1e9, flops / 1e9))


if __name__ == "__main__":
    main()
