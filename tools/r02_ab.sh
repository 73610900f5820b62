cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "VTS_WGRAD_CAP_MB=1" "VTS_WGRAD_CAP_MB=8" "VTS_WGRAD_CAP_MB=32" "VTS_WGRAD_CAP_MB=64"; do
  echo "== $v"
  for b in 4 1; do echo -n "   batch $b: "; env $v python bench.py --batch $b --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'])"; done
done
