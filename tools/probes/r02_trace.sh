cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_trace; rm -rf $O; mkdir -p $O
for b in 4 1; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b$b -o run -- python bench.py --batch $b --steps 10 --warmup 3 --no_cpu_baseline > $O/b$b.log 2>&1
python tools/probes/trace_timeline.py $O/b$b/run_kernel_trace.csv --dump $O/timeline_b$b.txt | head -8
python - <<PY
import csv, collections
c = collections.Counter()
rows = list(csv.DictReader(open("$O/b$b/run_kernel_stats.csv")))
tot = sum(int(r["Calls"]) for r in rows)
print("batch $b: kernel launches total", tot)
for r in rows[:12]: print("   %6s calls %8.1f us avg  %5s%%  %s" % (r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"], r["Name"][:90]))
PY
done
