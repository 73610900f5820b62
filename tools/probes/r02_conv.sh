cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_conv; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -5
VTS_MB=top python tools/microbench_conv.py 2>&1 | grep "^conv" > $O/top_new.txt; cat $O/top_new.txt
python tools/microbench_conv.py 2>&1 | grep "^conv" >> $O/top_new.txt; tail -7 $O/top_new.txt
