cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in 24 14 12 22; do echo "== TILE01 $v"; VTS_TILE01=$v VTS_MB=top python tools/microbench_conv.py 2>&1 | grep "^conv" | head -2 | cut -c1-100; VTS_TILE01=$v python tools/microbench_conv.py 2>&1 | grep "^conv" | head -1 | cut -c1-100;
echo -n "   batch 4: "; VTS_TILE01=$v python bench.py --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],3))"; done
