cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03_trace; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/b4 -o run -- python bench.py --steps 6 --warmup 3 --no_cpu_baseline > $O/b4.log 2>&1
python tools/probes/trace_timeline.py $O/b4/run_kernel_trace.csv --dump $O/timeline_b4.txt | head -3
cp $O/b4/run_kernel_trace.csv $O/kernel_trace.csv; rm -rf $O/b4
ls -la $O
