# SQ counters per kernel instance over one eager bench run (separate PMC passes; no trace domains besides kernel-trace)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcsq; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32"; do
i=$((i+1))
rocprofv3 --kernel-trace --pmc $set --output-format csv -d $O/p$i -o run -- python bench.py --steps 2 --warmup 1 --train_only --no_graph > $O/p$i.log 2>&1
done
python - <<'PY'
import csv,glob,collections,re,json
def key(name):
    name=name.replace("void ","").replace("(anonymous namespace)::","")
    m=re.match(r"([A-Za-z_0-9:]+(<[^>]*>)?)",name); return m.group(1) if m else name
out=collections.defaultdict(dict)
for i in (1,2):
    f=glob.glob('gpurun_out/pmcsq/p%d/**/*counter_collection.csv'%i, recursive=True)
    acc=collections.defaultdict(lambda:[0,0.0])
    for r in csv.DictReader(open(f[0])):
        a=acc[(key(r['Kernel_Name']),r['Counter_Name'])]; a[0]+=1; a[1]+=float(r['Counter_Value'])
    for (k,c),(n,v) in acc.items(): out[k][c]=v/n
json.dump(out,open('gpurun_out/pmcsq/sq.json','w'),indent=1)
print(len(out))
PY
