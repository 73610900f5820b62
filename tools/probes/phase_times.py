"""Time the captured segments of the skitG step (HIP events around the graph replays).  python tools/probes/phase_times.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
import torch  # noqa: E402

import bench  # noqa: E402


def main():
    model, opt = bench.build_model(1024, 4, "skitG")
    batch = bench.make_batch(1024, 4, 0, opt.style_code_dim if getattr(opt, "use_style_code", False) else 0)
    model.set_input(batch, phase="train")
    for _ in range(4):
        model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    assert model._graphs is not None
    names = [s[0].__name__ for s in model._segments()]
    acc = [0.0] * len(names)
    reps = 20
    for _ in range(reps):
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
        evs[0].record()
        for i, g in enumerate(model._graphs):
            g.replay()
            evs[i + 1].record()
        torch.cuda.synchronize()
        for i in range(len(names)):
            acc[i] += evs[i].elapsed_time(evs[i + 1])
    for n, t in zip(names, acc):
        print("%-20s %.2f ms" % (n, t / reps))


if __name__ == "__main__":
    main()
