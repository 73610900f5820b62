# end-of-round measurement pass (run through gpurun): headline bench + per-shape table + rocprof kernel stats + PMC traffic / MFMA utilisation
# + per-layer generator roofline table + secondary workloads.   bash tools/prof_round.sh [tag]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=${1:-r06}
WHAT=${2:-all}     # core: the headline step only (bench line, rocprof stats, PMC traffic / MFMA / SQ tables, per-layer table); all: + secondary workloads
O=gpurun_out/prof_$T; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python bench.py --steps 10 --warmup 3 --train_only > $O/stats.log 2>&1
cp $O/stats/run_kernel_stats.csv $O/kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python bench.py --steps 2 --warmup 1 --train_only --no_graph > $O/$c.log 2>&1
done
python tools/pmc_summary.py $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/traffic_pmc.json "$T"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_mfma -o run -- python bench.py --steps 2 --warmup 1 --train_only --no_graph > $O/mfma.log 2>&1
python tools/pmc_mfma.py $O/pmc_mfma $O/mfma_util.json "$T"
# the bench line reads the two summaries from profiles/: install them first so that the line and the summaries belong together
cp $O/traffic_pmc.json profiles/${T}_traffic_pmc.json; cp $O/mfma_util.json profiles/${T}_mfma_util.json
# SQ counters per kernel instance (two passes of 8 counters: instructions, issue / LDS waits, MFMA busy, LDS bank conflicts)
bash tools/pmc_sq.sh > $O/pmc_sq.log 2>&1; cp gpurun_out/pmcsq/sq.json $O/sq_counters.json; rm -rf gpurun_out/pmcsq
python tools/sq_table.py $O/sq_counters.json $O/mfma_util.json > $O/sq_table.md
[ -x tools/probes/bin/lastwg_sc1 ] && timeout 120 tools/probes/bin/lastwg_sc1 > $O/lastwg_probe.txt 2>&1
python bench.py --detail $O/kernel_shape_table.txt > $O/bench.json 2>$O/bench.err
python tools/g_layer_table.py $O/g_layer_table.md > /dev/null 2>&1
if [ "$WHAT" = core ]; then head -c 2500 $O/bench.json; echo; exit 0; fi
VTS_DDP_FORCE=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29590 python bench.py --no_cpu_baseline > $O/bench_ddp_forced_1rank.json 2>/dev/null
python bench.py --batch 1 --no_cpu_baseline > $O/bench_batch1.json 2>/dev/null
python bench.py --no_viz --no_cpu_baseline > $O/bench_no_viz.json 2>/dev/null
python bench.py --infer > $O/infer_bench.json 2>/dev/null
python tools/mb_wgrad.py > $O/wgrad_microbench.txt 2>&1
VTS_MB=top python tools/microbench_conv.py > $O/conv_microbench.txt 2>&1
python bench.py --model pix2pixHD --batch 32 --no_cpu_baseline > $O/pix2pixHD_patch_bench.json 2>/dev/null
python bench.py --model pix2pixHD --p2p_size 1024 --batch 2 --steps 5 --warmup 3 --no_cpu_baseline > $O/pix2pixHD_2x1024_bench.json 2>/dev/null
python bench.py --model pix2pixHD --p2p_h 1024 --p2p_w 2048 --batch 1 --steps 5 --warmup 3 --no_cpu_baseline > $O/pix2pixHD_2048x1024_bench.json 2>/dev/null
python bench.py --model pix2pixHD --p2p_h 1024 --p2p_w 2048 --batch 1 --steps 5 --warmup 3 --no_cpu_baseline --p2p_vgg > $O/pix2pixHD_2048x1024_vgg_bench.json 2>/dev/null
python bench.py --model sinskitG --netG resnet_9blocks --no_cpu_baseline > $O/resnet9_bench.json 2>/dev/null
python bench.py --lpips --detail $O/lpips_kernel_shape_table.txt > $O/bench_lpips.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_lpips -o run -- python bench.py --lpips --steps 6 --warmup 3 > $O/stats_lpips.log 2>&1
cp $O/stats_lpips/run_kernel_stats.csv $O/lpips_kernel_stats.csv; rm -rf $O/stats_lpips
rm -rf $O/stats $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/pmc_mfma
head -c 1500 $O/bench.json; echo; for f in bench_batch1 bench_no_viz infer_bench pix2pixHD_patch_bench pix2pixHD_2x1024_bench pix2pixHD_2048x1024_bench pix2pixHD_2048x1024_vgg_bench resnet9_bench bench_lpips bench_ddp_forced_1rank; do python -c "import json,sys; d=json.load(open('$O/$f.json')); print('$f', d['metric'], round(d['value'],2), d['unit'], round(d['ms_per_step'],3), 'ms')"; done
