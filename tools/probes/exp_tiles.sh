cd $GRAFT_REPO_ROOT
export VTS_TUNING=1
for i in 1 2 3; do for v in 1 0; do echo -n "one_graph=$v: "; VTS_ONE_GRAPH=$v python bench.py --train_only --steps 250 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['ms_per_step_spread'])"; done; done
python -m pytest tests/test_step_gpu.py tests/test_fullsize_gpu.py tests/test_rccl_gpu.py -x -q -m gpu 2>&1 | tail -2
