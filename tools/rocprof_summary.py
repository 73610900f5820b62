"""Turn a rocprofv3 result (rocpd sqlite .db or *_kernel_stats.csv) into the per-kernel summary kept under profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r1 profiles/r01_kernel_stats.csv
"""
import csv
import glob
import os
import sqlite3
import sys


def from_db(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    return [dict(name=r[0], calls=r[1], total_us=r[2], avg_us=r[3], pct=r[4]) for r in rows]


def main(src, dst):
    dbs = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
    rows = []
    for d in dbs:
        rows += from_db(d)
    rows.sort(key=lambda r: -r["total_us"])
    with open(dst, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=["name", "calls", "total_us", "avg_us", "pct"])
        w.writeheader()
        for r in rows:
            w.writerow(r)
    print("wrote %s (%d kernels, %.1f ms total)" % (dst, len(rows), sum(r["total_us"] for r in rows) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
