# round 6: where D2's forward of the generator step runs in the chained schedule: one lane under the backward (default) / three lanes in front of it
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTS_TUNING=1
run() { timeout 300 python bench.py --train_only --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for rep in 1 2 3; do echo -n "lane under the backward (default): "; run; echo -n "VTS_D2_TAIL=front: "; VTS_D2_TAIL=front run; done
