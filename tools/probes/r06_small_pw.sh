# round 6: partial copies (= persistent workgroups) of wgrad_small_kernel; instrumented library
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTS_TUNING=1 VTS_LIB_PATH=$PWD/visual-tactile-synthesis_amd/libvts_hip_prof.so
run() { timeout 300 python bench.py --train_only --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
for rep in 1 2; do for w in 256 128 64 32; do echo -n "VTS_WGRAD_SMALL_PW=$w: "; VTS_WGRAD_SMALL_PW=$w run; done; done
