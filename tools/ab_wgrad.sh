# same-box A/B of library builds on the weight-gradient micro-benchmark:  bash tools/ab_wgrad.sh <base.so> <other.so> ...   (paths under visual-tactile-synthesis_amd/)
cd $GRAFT_REPO_ROOT
P=$PWD/visual-tactile-synthesis_amd
i=0
for L in "$@" "$1"; do
  VTS_LIB_PATH=$P/$L python tools/mb_wgrad.py 2>/dev/null | grep "^wgrad" > /tmp/abw_$i.txt
  i=$((i+1))
done
python - $i "$@" <<'PY' | tee gpurun_out/ab_wgrad.txt
import sys,re
n=int(sys.argv[1]); names=sys.argv[2:]+[sys.argv[2]+" (again)"]
runs=[open('/tmp/abw_%d.txt'%k).read().splitlines() for k in range(n)]
def us(l): return float(re.search(r":\s+([0-9.]+) us", l).group(1))
print("%-44s" % "shape" + "".join(" %14s" % nm.replace("libvts_hip","").replace(".so","")[-14:] for nm in names))
tot=[0.0]*n
for r in range(len(runs[0])):
    u=[us(runs[k][r]) for k in range(n)]
    base=min(u[0],u[-1])
    for k in range(n): tot[k]+=u[k]
    print("%-44s" % runs[0][r].split(":")[0][:44] + "".join(" %7.1f %6.3f" % (x,x/base) for x in u))
base=min(tot[0],tot[-1])
print("%-44s" % "sum" + "".join(" %7.1f %6.3f" % (t,t/base) for t in tot))
PY
