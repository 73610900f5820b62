"""Summarise a rocprofv3 PMC pass (SQ_VALU_MFMA_BUSY_CYCLES, GRBM_GUI_ACTIVE) into MFMA pipe utilisation per kernel instance:
    mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)    (the gfx94x MfmaUtil formula, as a fraction;
                rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs -- cross-checked against the kernel-trace durations: the
                summary also holds busy / (duration_ns * 2.4 GHz * 1024) as mfma_util_by_time)
and `step_mfma_util` = the same ratio over the sums of all kernels of the run.  python tools/pmc_mfma.py <dir> <out.json> [note]"""
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import csrc_sha16, kernel_key  # noqa: E402


def main(d, dst, note=""):
    rows = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            a = rows.setdefault((r["Dispatch_Id"], kernel_key(r["Kernel_Name"])), {})
            a[r["Counter_Name"]] = a.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    dur = {}
    for path in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(path, newline="")):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    acc = {}
    for (did, k), c in rows.items():
        a = acc.setdefault(k, [0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[2] += c.get("GRBM_GUI_ACTIVE", 0.0)
        a[3] += dur.get(did, 0)
    simd_cycles = lambda g: g / 8.0 * 1024.0          # noqa: E731
    kernels = {k: {"launches_sampled": n, "mfma_busy_cycles_per_launch": b / n, "gui_active_cycles_per_launch": g / n, "avg_ns": t / n,
                   "mfma_util": (b / simd_cycles(g)) if g else None, "mfma_util_by_time": (b / (t * 2.4 * 1024.0)) if t else None}
               for k, (n, b, g, t) in sorted(acc.items())}
    tb, tg = sum(v[1] for v in acc.values()), sum(v[2] for v in acc.values())
    out = {"how": "rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE over `python bench.py --steps 2 --warmup 1 --no_cpu_baseline "
                  "--no_graph`; mfma_util = busy / (gui_active / 8 XCDs * 256 CUs * 4 SIMDs); fp32 MFMA: 32 busy cycles per v_mfma_f32_16x16x4_f32",
           "note": note, "csrc_sha16": csrc_sha16(), "step_mfma_util": tb / simd_cycles(tg) if tg else None, "kernels": kernels}
    json.dump(out, open(dst, "w"), indent=1)
    print("wrote %s (%d kernels), step mfma util %.3f" % (dst, len(kernels), out["step_mfma_util"] or 0.0))


if __name__ == "__main__":
    main(*sys.argv[1:4])
