cd $GRAFT_REPO_ROOT
P=$PWD/visual-tactile-synthesis_amd
for L in libvts_hip_base.so libvts_hip.so libvts_hip_base.so; do echo "== $L"; VTS_LIB_PATH=$P/$L python tools/mb_px.py 2>/dev/null | grep " us"; done
for i in 1 2; do for L in libvts_hip_base.so libvts_hip.so; do echo -n "step $L: "; VTS_LIB_PATH=$P/$L python bench.py --train_only --steps 200 --warmup 10 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; done; done
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
