# round-2 baseline: headline bench + per-shape single-stream table, batch-1 latency, launch count
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_base; rm -rf $O; mkdir -p $O
python bench.py --detail $O/detail.txt --no_cpu_baseline > $O/bench.json 2>$O/bench.err
python bench.py --batch 1 --no_cpu_baseline > $O/bench_b1.json 2>$O/bench_b1.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/stats.log 2>&1
cat $O/bench.json | head -c 600; echo; cat $O/bench_b1.json | head -c 300
