cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/cb; mkdir -p $O
P=$GRAFT_REPO_ROOT/visual-tactile-synthesis_amd
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -k "conv or backward_data" 2>&1 | tail -3
VTS_LIB_PATH=$P/libvts_hip_pf1.so python tools/mb_conv_big.py 2>&1 | grep -v amdgpu.ids > $O/pf1.txt
python tools/mb_conv_big.py 2>&1 | grep -v amdgpu.ids > $O/db.txt
paste -d'|' <(cut -c1-62 $O/pf1.txt) <(awk -F: '{print $2}' $O/db.txt | cut -c1-12)
for i in 1 2; do
  VTS_LIB_PATH=$P/libvts_hip_pf1.so python bench.py --no_cpu_baseline --steps 150 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('pf1 %.3f ms' % d['ms_per_step'])"
  python bench.py --no_cpu_baseline --steps 150 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('db %.3f ms' % d['ms_per_step'])"
done
