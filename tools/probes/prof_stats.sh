# rocprofv3 kernel stats of the headline bench:  bash tools/probes/prof_stats.sh <tag>   -> gpurun_out/stats_<tag>.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-x}; O=gpurun_out/stats_$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/log.txt 2>&1
cp $O/run_kernel_stats.csv gpurun_out/stats_$T.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("gpurun_out/stats_$T.csv")))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print("total kernel ms per step (29 steps incl. eager/warm-up): %.3f, launches/step %.0f" % (tot / 29 / 1e6, sum(int(r["Calls"]) for r in rows) / 29))
for r in rows[:28]:
    print("%6s calls %8.1f us avg %6.3f ms/step  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, int(r["TotalDurationNs"]) / 29 / 1e6, r["Name"][:100]))
PY
rm -rf $O
