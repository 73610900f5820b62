cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py tests/test_resnet_gpu.py -m gpu -x -q 2>&1 | tail -3
for v in 0 1; do echo "== DIRECT_EPI $v"; VTS_DIRECT_EPI=$v VTS_MB=top python tools/microbench_conv.py 2>&1 | grep "^conv" | cut -c1-100; VTS_DIRECT_EPI=$v python tools/microbench_conv.py 2>&1 | grep "^conv" | head -3 | cut -c1-100; 
for b in 4 1; do echo -n "   batch $b: "; VTS_DIRECT_EPI=$v python bench.py --batch $b --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'])"; done; done
