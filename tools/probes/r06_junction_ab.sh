# round 6: the last pyramid-merge level inside g_out_grad, same box (the second junction change measured with this script -- DiffAugment of the real
# image in a lane beside the forward -- was slower and its switch is gone: profiles/r06a_experiments.md)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r06h
timeout 900 python -m pytest tests/test_step_gpu.py tests/test_kernels_gpu.py tests/test_fullsize_gpu.py -m gpu -x -q > gpurun_out/r06h/tests.txt 2>&1
tail -3 gpurun_out/r06h/tests.txt
run() { timeout 300 python bench.py --train_only --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
export VTS_TUNING=1
for rep in 1 2; do
echo -n "default: "; run
echo -n "VTS_FUSE_MERGE=0: "; VTS_FUSE_MERGE=0 run
done
