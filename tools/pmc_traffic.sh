# HBM traffic per kernel instance: two separate PMC passes over an eager (no graph) bench run, then tools/pmc_summary.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc; rm -rf $O; mkdir -p $O
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python bench.py --steps 2 --warmup 1 --train_only --no_graph > $O/$c.log 2>&1
done
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic_pmc.json "${1:-state}"
ls -la $O/traffic_pmc.json
