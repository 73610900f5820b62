"""Which hardware queue every kernel of the replayed step runs on, and when (rocprofv3 --kernel-trace of a few replayed steps).
  run:      rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qt -o run -- python tools/probes/queue_trace.py run
  analyse:  python tools/probes/queue_trace.py report gpurun_out/qt > gpurun_out/queue_trace.txt
The report takes the LAST step in the trace (from one step_begin_kernel to the next), and prints per queue: launches, busy time, first
start / last end, and the kernel sequence with start offsets -- the lane-to-queue mapping the graph replay actually used."""
import csv
import glob
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    sys.path.insert(0, os.path.join(ROOT, "visual-tactile-synthesis_amd"))
    sys.path.insert(0, ROOT)
    import torch

    import bench

    model, opt = bench.build_model(1024, 4, "skitG")
    opt.use_hip_graph = True
    data = bench.make_batch(1024, 4, 0, opt.style_code_dim)
    model.set_input(data, phase="train")
    for _ in range(8):
        model.optimize_parameters(epoch=1)
    torch.cuda.synchronize()


def short(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9:]+)(<[^>]*>)?", name)
    base = m.group(1)
    t = m.group(2) or ""
    return base.replace("_kernel", "") + t.replace(" ", "")


def report(d):
    f = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    begins = [i for i, r in enumerate(rows) if "step_begin_kernel" in r["Kernel_Name"]]
    a, b = begins[-2], begins[-1]
    step = rows[a:b]
    t0 = int(step[0]["Start_Timestamp"])
    t1 = max(int(r["End_Timestamp"]) for r in step)
    print("step: %d kernels, %.3f ms from the first start to the last end" % (len(step), (t1 - t0) / 1e6))
    queues = {}
    for r in step:
        queues.setdefault(r["Queue_Id"], []).append(r)
    for q, rs in sorted(queues.items(), key=lambda kv: int(kv[0])):
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs)
        print("\nqueue %s: %d launches, busy %.3f ms, first start +%.3f ms, last end +%.3f ms" % (
            q, len(rs), busy / 1e6, (int(rs[0]["Start_Timestamp"]) - t0) / 1e6, (max(int(r["End_Timestamp"]) for r in rs) - t0) / 1e6))
        line = []
        for r in rs:
            line.append("+%.0f %s (%.0f)" % ((int(r["Start_Timestamp"]) - t0) / 1e3, short(r["Kernel_Name"]), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3))
        print("  " + " | ".join(line))
    # concurrency profile: how many queues are busy over time (100 us bins)
    nb = int((t1 - t0) / 1e5) + 1
    act = [set() for _ in range(nb)]
    for r in step:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        for k in range(int(s / 1e5), int(e / 1e5) + 1):
            act[k].add(r["Queue_Id"])
    print("\nbusy queues per 100 us bin: " + " ".join(str(len(x)) for x in act))


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else report(sys.argv[2])
