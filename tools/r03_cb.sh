cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cb; mkdir -p $O
P=$GRAFT_REPO_ROOT/visual-tactile-synthesis_amd
VTS_LIB_PATH=$P/libvts_hip_pf.so timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or backward_data" 2>&1 | tail -2
VTS_MB_INNER=1 python tools/mb_conv_big.py 2>&1 | grep -v amdgpu.ids > $O/i0.txt
VTS_MB_INNER=1 VTS_LIB_PATH=$P/libvts_hip_pf.so python tools/mb_conv_big.py 2>&1 | grep -v amdgpu.ids > $O/i1.txt
paste -d'|' <(cut -c1-62 $O/i0.txt) <(awk -F: '{print $2}' $O/i1.txt | cut -c1-12)
bash tools/ab_env.sh 3 "A=0" "VTS_LIB_PATH=$P/libvts_hip_pf.so" 2>&1 | tail -6
