# round 6: the decoder lanes' skip-feature gradients on the side queue (VTS_DEFER_SKIP), same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06m
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py tests/test_ddp_step_gpu.py -m gpu -x -q > gpurun_out/r06m/tests.txt 2>&1
tail -3 gpurun_out/r06m/tests.txt
run() { timeout 300 python bench.py --train_only --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
export VTS_TUNING=1
for rep in 1 2 3; do
echo -n "default (deferred): "; run
echo -n "VTS_DEFER_SKIP=0: "; VTS_DEFER_SKIP=0 run
done
