# round 6: D2's generator-step forward as a lane under the backward in the joined / data-parallel schedule (forced one-rank RCCL group), same box
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_ddp_step_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q 2>&1 | tail -2
export VTS_TUNING=1 VTS_DDP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29590
run() { timeout 300 python bench.py --no_cpu_baseline --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for rep in 1 2 3; do echo -n "ddp forced 1 rank, tail lane: "; run; echo -n "ddp forced 1 rank, VTS_D2_TAIL_LANE=0: "; VTS_D2_TAIL_LANE=0 run; done
