cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTS_TUNING=1
run() { timeout 300 python bench.py --train_only --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for rep in 1 2; do echo -n "default: "; run; echo -n "VTS_D2_JOIN_LATE=1: "; VTS_D2_JOIN_LATE=1 run; done
