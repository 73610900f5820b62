"""Reads the per-workgroup phase stamps written by vts_conv4x4 under VTS_CONV_TRACE=<file> and prints, per launch block:
phase durations (median / p90, microseconds; s_memrealtime ticks are 10 ns), workgroups resident per CU, and how much of the
launch's span the chip spent with workgroups in each phase (the overlap picture: do loads and MFMA phases coexist?)."""
import sys
from collections import defaultdict

import numpy as np

PH = ["-", "setup", "issue loads", "wait+LDS store", "MFMA(first unit)", "rest of loop", "epilogue"]


def blocks(path):
    head, rows = None, []
    for line in open(path):
        if line.startswith("#"):
            if head is not None:
                yield head, rows
            head, rows = line[1:].strip(), []
        else:
            f = line.split()
            rows.append([int(f[0]), int(f[1], 16)] + [int(x) for x in f[2:]])
    if head is not None:
        yield head, rows


def main():
    for head, rows in blocks(sys.argv[1]):
        a = np.array(rows, dtype=np.int64)
        hw = a[:, 1]
        t = a[:, 2:].astype(np.float64)   # t1..t7
        t7 = np.where(t[:, 6] > 0, t[:, 6], t[:, 5])
        t[:, 6] = t7
        t0 = t[:, 0].min()
        t = (t - t0) * 0.01   # us
        print("==", head)
        print("   span %.1f us, %d workgroups" % (t[:, 6].max(), len(a)))
        for i in range(6):
            d = t[:, i + 1] - t[:, i]
            print("   %-18s median %6.2f  p90 %6.2f us" % (PH[i + 1], np.median(d), np.percentile(d, 90)))
        life = t[:, 6] - t[:, 0]
        print("   %-18s median %6.2f  p90 %6.2f us" % ("lifetime", np.median(life), np.percentile(life, 90)))
        cu = ((hw >> 32) & 0xF) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xF)
        per = defaultdict(list)
        for i, c in enumerate(cu):
            per[int(c)].append((t[i, 0], t[i, 6]))
        conc = []
        for c, iv in per.items():
            ev = sorted([(s, 1) for s, e in iv] + [(e, -1) for s, e in iv])
            cur = mx = 0
            for _, d in ev:
                cur += d
                mx = max(mx, cur)
            conc.append(mx)
        print("   CUs seen %d, workgroups per CU %.1f, max resident per CU: median %d max %d" % (len(per), len(a) / len(per), np.median(conc), max(conc)))
        # chip-wide phase occupancy over time
        grid = np.linspace(0, t[:, 6].max(), 400)
        occ = np.zeros((6, len(grid)))
        for i in range(6):
            s, e = t[:, i], t[:, i + 1]
            occ[i] = ((s[None, :] <= grid[:, None]) & (grid[:, None] < e[None, :])).sum(1)
        tot = occ.sum(0)
        print("   mean workgroups in flight %.0f; share by phase: %s" % (tot.mean(), ", ".join("%s %.0f%%" % (PH[i + 1], 100 * occ[i].sum() / tot.sum()) for i in range(6))))


if __name__ == "__main__":
    main()
