cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r1i; rm -rf $O; mkdir -p $O
python bench.py --detail $O/detail.txt > $O/bench.json 2>$O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python bench.py --steps 10 --warmup 3 --no_cpu_baseline > $O/stats.log 2>&1
tail -c 400 $O/bench.json; ls $O/stats
