"""Is the step time a property of the PROCESS or of the graph instantiation?  Captures the step's HIP graphs several times in one
process and times each instantiation (process-to-process the same tree shows two modes ~0.2 ms apart: tools/probes/bimodal.sh)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def timed(model, steps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3


def main():
    model, opt = bench.build_model(1024, 4, "skitG")
    batch = bench.make_batch(1024, 4, 0, opt.style_code_dim if getattr(opt, "use_style_code", False) else 0)
    model.set_input(batch, phase="train")
    for _ in range(4):
        model.optimize_parameters(epoch=1)
    for trial in range(int(os.environ.get("TRIALS", "6"))):
        a, b = timed(model, 60), timed(model, 60)
        print("instantiation %d: %.3f ms  %.3f ms" % (trial, a, b), flush=True)
        model._drop_graphs()
        model.optimize_parameters(epoch=1)     # captures again
        model.optimize_parameters(epoch=1)


if __name__ == "__main__":
    main()
