#!/bin/bash
# how stable is the step time from process to process on one box?  (same tree, same flags)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for i in $(seq ${1:-8}); do
  env $2 python bench.py --no_cpu_baseline --steps 150 --warmup 10 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['ms_per_step_spread']; print('run $i %.3f ms  min %.3f med %.3f p90 %.3f max %.3f' % (d['ms_per_step'], s['min'], s['median'], s['p90'], s['max']))"
done
