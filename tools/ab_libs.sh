# same-box A/B of builds of the library on the conv micro-benchmark:  bash tools/ab_libs.sh <base.so> <other.so> [<other2.so> ...]
# (paths relative to visual-tactile-synthesis_amd/; prints us per shape and the ratio to the base, whose run is repeated at the end)
cd $GRAFT_REPO_ROOT
P=$PWD/visual-tactile-synthesis_amd
i=0
for L in "$@" "$1"; do
  VTS_LIB_PATH=$P/$L python tools/mb_conv_ab.py $L > /tmp/ab_$i.txt 2>/dev/null
  i=$((i+1))
done
python - $i "$@" <<'PY' | tee gpurun_out/ab_libs.txt
import sys
n=int(sys.argv[1]); names=sys.argv[2:]+[sys.argv[2]+" (again)"]
runs=[[l for l in open('/tmp/ab_%d.txt'%k) if not l.startswith('#')] for k in range(n)]
print("%-34s" % "shape" + "".join(" %14s" % nm.replace("libvts_hip","").replace(".so","")[-14:] for nm in names))
tot=[0.0]*n
for r in range(len(runs[0])):
    us=[float(runs[k][r][34:].split()[0]) for k in range(n)]
    base=min(us[0],us[-1])
    for k in range(n): tot[k]+=us[k]
    print("%-34s" % runs[0][r][:34] + "".join(" %7.1f %6.3f" % (u,u/base) for u in us))
base=min(tot[0],tot[-1])
print("%-34s" % "sum" + "".join(" %7.1f %6.3f" % (t,t/base) for t in tot))
PY
