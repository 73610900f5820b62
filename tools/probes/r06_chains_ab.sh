# round 6: the chained step schedule against the joined one (same box, bench.py --train_only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06c
timeout 900 python -m pytest tests/test_step_gpu.py -m gpu -x -q -s -k "chains or graph_replay" > gpurun_out/r06c/step.txt 2>&1
grep -n "passed\|failed\|Error\|error\|Segmentation" gpurun_out/r06c/step.txt | tail -8
run() { timeout 300 python bench.py --train_only --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
export VTS_TUNING=1
echo -n "joined schedule (VTS_D_CHAINS=0 VTS_LAZY_PYRAMID=0): "; VTS_D_CHAINS=0 VTS_LAZY_PYRAMID=0 run
echo -n "joined schedule, lazy pyramids: "; VTS_D_CHAINS=0 run
echo -n "chains, D2 update as two lanes (default): "; run
echo -n "chains, one side queue for the weight gradients: "; VTS_SIDE_QUEUES=1 run
echo -n "chains, whole D2 chain as one lane: "; VTS_D2_CHAIN=serial run
echo -n "chains, without D1 real early: "; VTS_D1_REAL_EARLY=0 run
