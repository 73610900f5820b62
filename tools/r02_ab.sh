cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "VTS_WGRAD_DEFER=0" "VTS_WGRAD_DEFER=1" "VTS_WGRAD_DEFER=1 VTS_WGRAD_FLUSH_MB=32" "VTS_WGRAD_DEFER=1 VTS_WGRAD_FLUSH_MB=4000" "VTS_WGRAD_DEFER=0 VTS_WGRAD_NS_WGS=256" "VTS_WGRAD_DEFER=0 VTS_WGRAD_NS_WGS=768"; do
  for b in 4 1; do
    echo -n "$v batch $b: "; env $v python bench.py --batch $b --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'])"
  done
done
