# round-2 GPU check: parity tests + headline bench (+ batch 1) + detail table
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_gpu; rm -rf $O; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/pytest.txt 2>&1; tail -5 $O/pytest.txt
python bench.py --detail $O/detail.txt --no_cpu_baseline > $O/bench.json 2>$O/bench.err; head -c 400 $O/bench.json; echo
python bench.py --batch 1 --no_cpu_baseline > $O/bench_b1.json 2>$O/bench_b1.err; head -c 300 $O/bench_b1.json; echo
