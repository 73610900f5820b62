#!/bin/bash
# A/B of two whole source trees on the same GPU box (python side included): tools/ab_tree.sh <dirA> <dirB> [rounds] [bench flags...]
# Each dir holds a full copy of the repo (bench.py, package, built libvts_hip.so), e.g. `git worktree` exports under gpurun_ab/.
A=$1; B=$2; R=${3:-3}; shift 3
cd /tmp && export TMPDIR=/tmp
for i in $(seq $R); do
  for v in $A $B; do
    (cd $GRAFT_REPO_ROOT/$v && python bench.py --no_cpu_baseline --steps 150 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v %.3f ms  fresh %s' % (d['ms_per_step'], d.get('ms_per_step_fresh_input')))")
  done
done
