cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cb; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or backward_data" 2>&1 | tail -2
VTS_MT3=0 python tools/mb_conv_big.py 2>&1 | grep -v amdgpu.ids | tail -5 > $O/m0.txt
python tools/mb_conv_big.py 2>&1 | grep -v amdgpu.ids | tail -5 > $O/m1.txt
paste -d'|' <(cut -c1-62 $O/m0.txt) <(awk -F: '{print $2}' $O/m1.txt | cut -c1-12)
bash tools/ab_env.sh 3 "A=0" "VTS_MT3=0" 2>&1 | tail -6
