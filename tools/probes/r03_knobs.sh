# round-3 knob sweep (timing only): bash tools/probes/r03_knobs.sh  -> gpurun_out/r03_knobs.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_knobs.txt; : > $O
run() { echo "== $*" >> $O; env "$@" python bench.py --steps 150 --warmup 10 --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), d['ms_per_step_spread'])" >> $O; }
run A=1
run VTS_WGRAD_NS_WGS=256
run VTS_WGRAD_NS_WGS=384
run VTS_WGRAD_FLUSH_MB=32
run VTS_WGRAD_FLUSH_MB=512
run VTS_WGRAD_CAP_MB=4
run VTS_FUSE_STATS=0
run A=2
cat $O
