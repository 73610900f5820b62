"""Timeline analysis of one training step from a rocprofv3 --kernel-trace CSV (graph replay, concurrent lanes).

    python tools/probes/trace_timeline.py <run_kernel_trace.csv> [--step K] [--dump out.txt]

Steps are delimited by the generator's Adam launch (the last kernel of a step).  Prints the span of the step, the time during
which 0 / 1 / 2 / ... kernels run, and which kernels account for the time at concurrency 1 (the critical path candidates)."""
import collections
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def main():
    path = sys.argv[1]
    step_k = int(sys.argv[sys.argv.index("--step") + 1]) if "--step" in sys.argv else -2
    dump = sys.argv[sys.argv.index("--dump") + 1] if "--dump" in sys.argv else None
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Stream_Id", "0"), r.get("Queue_Id", "0"),
                     int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]) // max(1, int(r["Workgroup_Size_X"]))))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if r[2].startswith("adam")]
    # a step ends with three Adam launches at most; group consecutive adam launches (D, D2 ... G): the step boundary is after the LAST adam
    # of each group of kernels; here: boundaries at adam launches that are followed by a gap (next kernel starts a new step's forward)
    bounds = []
    for j, i in enumerate(ends):
        nxt = rows[i + 1][0] if i + 1 < len(rows) else None
        if nxt is None or nxt - rows[i][1] > 20000 or (j + 1 < len(ends) and ends[j + 1] - i > 400):
            bounds.append(i)
    if len(bounds) < 3:
        print("could not find step boundaries (%d adam launches)" % len(ends))
        return
    b0, b1 = bounds[step_k - 1], bounds[step_k]
    step = rows[b0 + 1:b1 + 1]
    t0 = min(r[0] for r in step)
    t1 = max(r[1] for r in step)
    print("step: %d kernels, span %.3f ms, sum of kernel time %.3f ms" % (len(step), (t1 - t0) / 1e6, sum(r[1] - r[0] for r in step) / 1e6))
    ev = []
    for i, r in enumerate(step):
        ev.append((r[0], 1, i))
        ev.append((r[1], -1, i))
    ev.sort()
    conc = collections.Counter()
    alone = collections.Counter()
    active = set()
    last = t0
    for t, d, i in ev:
        dt = t - last
        if dt > 0:
            conc[len(active)] += dt
            if len(active) == 1:
                alone[step[next(iter(active))][2]] += dt
        last = t
        if d > 0:
            active.add(i)
        else:
            active.discard(i)
    print("time at concurrency k (ms): " + "  ".join("%d:%.3f" % (k, v / 1e6) for k, v in sorted(conc.items())))
    print("kernels running ALONE (top 25, ms):")
    for k, v in alone.most_common(25):
        print("  %8.3f  %s" % (v / 1e6, k))
    streams = collections.Counter((r[3], r[4]) for r in step)
    print("streams (stream, queue): launches  " + "  ".join("%s/%s:%d" % (s[0], s[1], c) for s, c in sorted(streams.items())))
    if dump:
        with open(dump, "w") as f:
            f.write("# start_us  dur_us  stream queue  workgroups  kernel\n")
            for r in step:
                f.write("%9.1f %7.1f  %3s %3s  %7d  %s\n" % ((r[0] - t0) / 1e3, (r[1] - r[0]) / 1e3, r[3], r[4], r[5], r[2]))


if __name__ == "__main__":
    main()
