# round 6: LDS footprint of the small-map weight-gradient kernel (D2 patch stacks) on the chained schedule; instrumented library (make PROFILING=1)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTS_TUNING=1 VTS_LIB_PATH=$PWD/visual-tactile-synthesis_amd/libvts_hip_prof.so
run() { timeout 300 python bench.py --train_only --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for rep in 1 2 3; do
for k in 150 48 32 24 16 12 8; do echo -n "VTS_WGRAD_SMALL_LDS_KB=$k: "; VTS_WGRAD_SMALL_LDS_KB=$k run; done
done
