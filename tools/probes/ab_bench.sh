#!/bin/bash
# A/B of two builds of libvts_hip.so on the same GPU box: tools/probes/ab_bench.sh gpurun_ab/libvts_hip_base.so gpurun_ab/libvts_hip_new.so [rounds]
A=$1; B=$2; R=${3:-3}
L=visual-tactile-synthesis_amd/libvts_hip.so
cp $L /tmp/keep.so
for i in $(seq $R); do
  for v in $A $B; do
    cp $v $L
    python bench.py --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v %.3f ms' % d['ms_per_step'])"
  done
done
cp /tmp/keep.so $L
