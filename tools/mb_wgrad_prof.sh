# kernel-only durations of the weight-gradient microbenchmark (rocprofv3 kernel trace), optionally under VTS_ABLATE values:
#   bash tools/mb_wgrad_prof.sh "0 3 4 7 11"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTS_MB=quick VTS_MB_NOCHECK=1 VTS_MB_EAGER=1
for a in ${1:-0}; do
  O=gpurun_out/mbw_$a; rm -rf $O; mkdir -p $O
  VTS_ABLATE=$a rocprofv3 --kernel-trace --stats --output-format csv -d $O -o run -- python tools/mb_wgrad.py > $O/log.txt 2>&1
  echo "== ABLATE $a"
  python - <<PY
import csv,glob
f=glob.glob("$O/**/run_kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    n=r["Name"].replace("void (anonymous namespace)::","")
    if "wgrad" in n: print("%8.1f us x%-4s %s"%(float(r["AverageNs"])/1e3, r["Calls"], n[:90]))
PY
done
