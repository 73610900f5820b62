cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/px; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -5 $O/pytest.log
python tools/mb_px.py 2>&1 | grep -v amdgpu.ids > $O/mb_px.txt
python bench.py --detail $O/kernel_shape_table.txt --no_cpu_baseline > $O/bench.json 2>$O/bench.err; head -c 400 $O/bench.json; echo
VTS_NO_PX=1 VTS_NO_PXT=1 python bench.py --no_cpu_baseline > $O/bench_nopx.json 2>/dev/null; head -c 400 $O/bench_nopx.json; echo
python bench.py --no_cpu_baseline > $O/bench2.json 2>/dev/null; head -c 400 $O/bench2.json; echo
VTS_NO_PX=1 VTS_NO_PXT=1 python bench.py --no_cpu_baseline > $O/bench_nopx2.json 2>/dev/null; head -c 400 $O/bench_nopx2.json; echo
