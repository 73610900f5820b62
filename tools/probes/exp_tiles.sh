cd $GRAFT_REPO_ROOT
export VTS_LIB_PATH=$PWD/visual-tactile-synthesis_amd/libvts_hip_skew.so
echo "== off"; python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
for m in 1 2 3; do for d in 2 4; do echo "== m${m}d${d}"; VTS_SKEW=$m VTS_SKEW_DIV=$d python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"; done; done
VTS_SKEW=2 VTS_SKEW_DIV=4 python -m pytest tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -x -q -m gpu 2>&1 | tail -2
