cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab_env.sh 2 "A=0" "VTS_LANE_WGRAD_SIDE=1" 2>&1 | tail -6
VTS_LANE_WGRAD_SIDE=1 timeout 300 python -m pytest tests/test_step_gpu.py -x -q 2>&1 | tail -3
