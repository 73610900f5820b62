"""GPU timeline of the fresh-batch loop from a rocprofv3 kernel + memory-copy trace: per iteration, how long the device runs the step's
kernels, set_input's kernels, and nothing at all.
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/ft -o ft -- python tools/probes/fresh_trace.py run
  python tools/probes/fresh_trace.py report gpurun_out/ft"""
import csv, glob, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
    import torch, bench
    model, opt = bench.build_model(1024, 4, "skitG")
    sd = opt.style_code_dim
    b = [bench.make_batch(1024, 4, r, sd, quantize8=True) for r in (0, 1)]
    b = [{k: (v.pin_memory() if torch.is_tensor(v) else v) for k, v in x.items()} for x in b]
    for i in range(8):
        model.set_input(b[i % 2], phase="train"); model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()
    time.sleep(0.05)
    for i in range(12):
        model.set_input(b[i % 2], phase="train"); model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
    # the last 12 iterations: split at step_begin kernels
    idx = [i for i, e in enumerate(ev) if "step_begin" in e[2]]
    idx = idx[-12:]
    starts = [ev[i][0] for i in idx]
    print("iteration period (step_begin to step_begin): " + " ".join("%.2f" % ((b - a) / 1e6) for a, b in zip(starts, starts[1:])))
    SI = ("u8_expand", "mask_mul", "avgpool_rows4", "copy_words", "mask_cand", "mask_rowcount", "mask_prefix", "copyBuffer", "input_images", "elementwise")
    for k in range(len(idx) - 1):
        seg = [e for e in ev if starts[k] <= e[0] < starts[k + 1]]
        # union of busy intervals
        iv = sorted((a, b) for a, b, _ in seg)
        busy, cur_a, cur_b, gaps = 0, iv[0][0], iv[0][1], []
        for a, b in iv[1:]:
            if a > cur_b:
                busy += cur_b - cur_a
                gaps.append((a - cur_b, cur_b))
                cur_a, cur_b = a, b
            else:
                cur_b = max(cur_b, b)
        busy += cur_b - cur_a
        period = starts[k + 1] - starts[k]
        si = [e for e in seg if any(s in e[2] for s in SI)]
        si_busy = sum(b - a for a, b, _ in si)
        si_span = (max(b for a, b, _ in si) - min(a for a, b, _ in si)) if si else 0
        big = sorted(gaps, reverse=True)[:3]
        print("it %2d: period %.3f ms, device idle %.3f ms, set_input kernels %d busy %.3f span %.3f; largest gaps (us): %s" % (
            k, period / 1e6, (period - busy) / 1e6, len(si), si_busy / 1e6, si_span / 1e6, " ".join("%.0f@%.2f" % (g / 1e3, (t - starts[k]) / 1e6) for g, t in big)))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
