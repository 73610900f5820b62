cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/sm; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or small" 2>&1 | tail -3
python tools/mb_small.py 2>&1 | grep -v amdgpu.ids | tee $O/flat.txt | cut -c1-160
VTS_SMALL_FLAT=0 python tools/mb_small.py 2>&1 | grep -v amdgpu.ids | tee $O/noflat.txt | cut -c1-160
bash tools/ab_env.sh 2 "A=0" "VTS_SMALL_FLAT=0"
