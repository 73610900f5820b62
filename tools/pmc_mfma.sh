# MFMA pipe utilisation per kernel instance: SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE over one eager bench run (one PMC pass, kernel-trace only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcmfma; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/p -o run -- python bench.py --steps 2 --warmup 1 --train_only --no_graph > $O/p.log 2>&1
python tools/pmc_mfma.py $O/p $O/mfma_util.json "${1:-state}"
