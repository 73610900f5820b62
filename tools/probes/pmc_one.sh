cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcone; rm -rf $O; mkdir -p $O
i=0
for c in 4,4,1024,1024,8,2,2,0 4,16,257,257,8,2,2,1; do
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS" "SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_FLAT SQ_INSTS_SMEM"; do
i=$((i+1))
VTS_MB=one:$c rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o run -- python tools/microbench_conv.py > $O/p$i.log 2>&1
done; done
python - <<'PY'
import csv,glob,collections
for i in range(1,7):
    f=glob.glob('gpurun_out/pmcone/p%d/**/*counter_collection.csv'%i, recursive=True)
    if not f: print('missing',i); continue
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f[0])):
        if 'conv4x4_kernel' in r['Kernel_Name']:
            a=acc[r['Counter_Name']]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    print('pass',i, {k: round(v[1]/v[0]) for k,v in acc.items()})
PY
