cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_wide; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python tools/probes/bench_global.py 1024 2048 1 > $O/stats.log 2>&1
VTS_MB=wide rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcF -o run -- python tools/microbench_conv.py > $O/pmcF.log 2>&1
VTS_MB=wide rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcW -o run -- python tools/microbench_conv.py > $O/pmcW.log 2>&1
VTS_MB=wide rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmcM -o run -- python tools/microbench_conv.py > $O/pmcM.log 2>&1
tail -2 $O/stats.log; ls $O/*
