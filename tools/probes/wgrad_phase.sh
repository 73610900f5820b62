#!/bin/bash
# one gpurun call: the weight-gradient kernel alone under its ablation / grid knobs
out=gpurun_out/wgrad_phase; mkdir -p $out
run() { echo "== $*"; env "$@" python tools/probes/wgrad_phase.py 2>&1 | grep -v amdgpu.ids; }
{
run A=0
run VTS_ABLATE=1
run VTS_ABLATE=2
run VTS_ABLATE=4
run VTS_ABLATE=3
run VTS_ABLATE=7
run VTS_WGRAD_NS_WGS=256
run VTS_WGRAD_NS_WGS=1024
run VTS_WGRAD_CAP_MB=64
run VTS_WGRAD_NS_WGS=256 VTS_WGRAD_CAP_MB=64
} > $out/phase.txt 2>&1
tail -5 $out/phase.txt
