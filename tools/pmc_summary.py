"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE) into per-kernel HBM bytes per launch.

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE profiles/r01e_traffic_pmc.json "note"

Keys are kernel instance names exactly as `vts_last_kernel()` / bench.py's breakdown print them
(`conv4x4_kernel<1, 2, 1, 1, 4, 4>`, `norm_bwd_fused_kernel`, ...).  Bytes follow MI355X_MICROARCH.md's
HBM section: both counters are in KiB; gfx950 FETCH_SIZE under-counts wide reads by 2x, so
bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.
"""
import csv
import glob
import json
import os
import re
import sys


def kernel_key(name):
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    m = re.match(r"([A-Za-z_0-9:]+(<[^>]*>)?)", name)
    return m.group(1) if m else name


def collect(d, counter, by_grid=False):
    """{instance: [launches, sum]}; by_grid: {(instance, "grid x wg"): [launches, sum]} -- one instance serves several shapes of a step"""
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                if row["Counter_Name"] != counter:
                    continue
                k = kernel_key(row["Kernel_Name"])
                if by_grid:
                    k = (k, "%s/%s" % (row.get("Grid_Size", "?"), row.get("Workgroup_Size", "?")))
                a = acc.setdefault(k, [0, 0.0])
                a[0] += 1
                a[1] += float(row["Counter_Value"])
    return acc


def csrc_sha16():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    return bench.csrc_sha16()


def main(fetch_dir, write_dir, dst, note=""):
    fe, wr = collect(fetch_dir, "FETCH_SIZE"), collect(write_dir, "WRITE_SIZE")
    kernels = {}
    for k in sorted(set(fe) | set(wr)):
        fc, fv = fe.get(k, [0, 0.0])
        wc, wv = wr.get(k, [0, 0.0])
        f_kb = fv / fc if fc else 0.0
        w_kb = wv / wc if wc else 0.0
        kernels[k] = {"launches_sampled": max(fc, wc), "FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
                      "hbm_bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0}
    feg, wrg = collect(fetch_dir, "FETCH_SIZE", True), collect(write_dir, "WRITE_SIZE", True)
    for (k, g) in sorted(set(feg) | set(wrg)):
        fc, fv = feg.get((k, g), [0, 0.0])
        wc, wv = wrg.get((k, g), [0, 0.0])
        kernels[k].setdefault("by_grid", {})[g] = {"launches_sampled": max(fc, wc),
                                                   "hbm_bytes_per_launch": (2.0 * (fv / fc if fc else 0.0) + (wv / wc if wc else 0.0)) * 1024.0}
    out = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) over "
                  "`python bench.py --steps 2 --warmup 1 --train_only --no_graph`; mean per launch of each kernel "
                  "instance (train steps only: launches_sampled = launches per step x 3; by_grid splits an instance by its grid size / workgroup size); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE correction, MI355X_MICROARCH.md)",
           "note": note, "csrc_sha16": csrc_sha16(), "kernels": kernels}
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote %s (%d kernels)" % (dst, len(kernels)))


if __name__ == "__main__":
    main(*sys.argv[1:5])
