#!/bin/bash
# same-box A/B of environment knobs on the current tree: tools/ab_env.sh <rounds> "<ENV=.. ENV=..>" "<ENV=..>" ...   (use "A=0" for the default)
R=$1; shift
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
# experiment switches are honoured only by the instrumented library (make -C visual-tactile-synthesis_amd/csrc PROFILING=1) and under VTS_TUNING=1
export VTS_TUNING=1
[ -f visual-tactile-synthesis_amd/libvts_hip_prof.so ] && export VTS_LIB_PATH=$PWD/visual-tactile-synthesis_amd/libvts_hip_prof.so
for i in $(seq $R); do
  for v in "$@"; do
    env $v python bench.py --no_cpu_baseline --steps 150 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-40s %.3f ms' % ('$v', d['ms_per_step']))"
  done
done
