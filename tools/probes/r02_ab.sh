cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_step_gpu.py -m gpu -x -q 2>&1 | tail -2
VTS_MB=small python tools/microbench_conv.py 2>&1 | grep "^conv" | cut -c1-100
for i in 1 2; do python bench.py --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'])"; done
