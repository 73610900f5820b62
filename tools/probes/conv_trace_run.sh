#!/bin/bash
# usage: tools/probes/conv_trace_run.sh <kernel "MODE,S,NR,RW,MT"> <tag> [env assignments...]   (on the GPU box)
K=$1; TAG=$2; shift 2
rm -f /tmp/tr.txt
env "$@" VTS_CONV_TRACE_KERNEL=$K VTS_CONV_TRACE=/tmp/tr.txt VTS_MB=top python tools/microbench_conv.py > /dev/null 2>&1
python - <<PY
seen={}
for b in open("/tmp/tr.txt").read().split("# ")[1:]:
    seen[b.split("\n",1)[0]]=b
open("/tmp/tr_last.txt","w").write("".join("# "+b for b in seen.values()))
PY
echo "#### $TAG $@"
python tools/probes/conv_trace.py /tmp/tr_last.txt
mkdir -p gpurun_out/conv_trace && cp /tmp/tr_last.txt gpurun_out/conv_trace/$TAG.txt
