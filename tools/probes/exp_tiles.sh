cd $GRAFT_REPO_ROOT
P=$PWD/visual-tactile-synthesis_amd
export VTS_LIB_PATH=$P/libvts_hip_ck16.so
echo "== off"; python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
echo "== ck16>=64"; VTS_SPLIT_CK16=64 python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
echo "== ck16>=128"; VTS_SPLIT_CK16=128 python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
echo "== off again"; python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
export VTS_TUNING=1
for v in 0 64 128; do echo -n "step ck16=$v: "; VTS_SPLIT_CK16=$v python bench.py --train_only --steps 150 --warmup 10 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; done
for v in 0 64 128; do echo -n "step ck16=$v: "; VTS_SPLIT_CK16=$v python bench.py --train_only --steps 150 --warmup 10 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],3))"; done
