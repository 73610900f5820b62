# round 6: older scheduling knobs re-measured under the chained schedule (same box)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTS_TUNING=1
run() { timeout 300 python bench.py --train_only --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
echo -n "default: "; run
for v in "VTS_WGRAD_FLUSH_MB=32" "VTS_WGRAD_FLUSH_MB=256" "VTS_WGRAD_FLUSH_MB=4096" "VTS_SIDE_QUEUES=1" "VTS_SIDE_QUEUES=3" "VTS_BWD_SUMS=0" "VTS_MSD_C=0" "VTS_G_PRE_LANE=0" "VTS_WGRAD_DEFER=0"; do echo -n "$v: "; env $v bash -c "$(declare -f run); run"; done
echo -n "default: "; run
