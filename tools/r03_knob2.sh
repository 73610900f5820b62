cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py tests/test_ddp_step_gpu.py tests/test_rccl_gpu.py -x -q 2>&1 | tail -3
bash tools/ab_env.sh 3 "A=0" "VTS_LANE_ACC=0" 2>&1 | tail -6
