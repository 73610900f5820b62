cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/probes/mb_norm.py 2>&1 | grep "640"
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "norm or batchnorm" 2>&1 | tail -2
for i in 1 2; do python bench.py --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'])"; done
