# baseline pass of a re-entered session: GPU tests, bench line with the per-shape table, rocprof kernel stats
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/base_${1:-r03d}; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --detail $O/kernel_shape_table.txt --no_cpu_baseline > $O/bench.json 2>$O/bench.err; head -c 600 $O/bench.json; echo
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/stats.log 2>&1
cp $O/stats/run_kernel_stats.csv $O/kernel_stats.csv; rm -rf $O/stats
head -40 $O/kernel_stats.csv
