cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q 2>&1 | tail -2
for v in 0 1; do echo "== XCD_SWIZZLE $v"; VTS_XCD_SWIZZLE=$v VTS_MB=top python tools/microbench_conv.py 2>&1 | grep "^conv" | cut -c1-100; VTS_XCD_SWIZZLE=$v python tools/microbench_conv.py 2>&1 | grep "^conv" | head -3 | cut -c1-100;
echo -n "   batch 4: "; VTS_XCD_SWIZZLE=$v python bench.py --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'], d['roofline']['kernel'], round(d['roofline']['frac'],3))"; done
O=gpurun_out/xcd; rm -rf $O; mkdir -p $O
for v in 0 1; do VTS_XCD_SWIZZLE=$v VTS_MB=one:4,4,1024,1024,8,2,2,0 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/f$v -o run -- python tools/microbench_conv.py > $O/f$v.log 2>&1
python - <<PY
import csv,glob
f=glob.glob('$O/f$v/**/*counter_collection.csv', recursive=True)[0]
v=[float(r['Counter_Value']) for r in csv.DictReader(open(f)) if 'conv4x4_kernel' in r['Kernel_Name'] and r['Counter_Name']=='FETCH_SIZE']
import collections
print("swizzle $v: FETCH_SIZE KB per launch (sum over rows / launches)", sum(v)/23, "algorithmic input 65536 KB -> x2 correction:", 2*sum(v)/23/65536)
PY
done
