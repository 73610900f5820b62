# knock-out timings on the replayed step, round 6 (chained schedule): a knocked-out launch is skipped -- results WRONG by construction, timing only.
# VTS_KNOCKOUT matches launch LABELS of vts/ops.py:_run (conv4x4 / wgrad4x4 / patch_conv4x4 / patch_wgrad4x4 / norm_* / norm_patch_* /
# channel_sum / wgrad_reduce_batch ...).  Round 5's list named KERNELS ("conv_small,wgrad_small,conv_head_small", "conv_head", "chsum"):
# those matched no label and measured nothing.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTS_TUNING=1 VTS_KO_LANES_ACK=timing-only
run() { python bench.py --train_only --steps 150 --warmup 10 $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'])"; }
echo -n "baseline (chained schedule): "; run
for l in 0 "1,2" "0,1,2" 3 "4,5" "3,4,5" "0,1,2,3,4,5"; do echo -n "KO lanes [$l]: "; VTS_KO_LANES=$l run; done
for v in "patch_" "patch_conv4x4" "patch_wgrad4x4" "norm_patch" "norm_" "norm_bwd" "norm_finalize,norm_from_partials,norm_stats" "channel_sum" "wgrad_reduce" "wgrad4x4" "wgrad4x4,patch_wgrad4x4,wgrad_reduce" "conv4x4" "conv4x4,patch_conv4x4"; do echo -n "knockout [$v]: "; VTS_KNOCKOUT=$v run; done
echo -n "without the D2 visualisation pass (--no_viz): "; EXTRA=--no_viz run
echo -n "baseline again: "; run
echo "# the JOINED schedule of round 5 on this box (VTS_D_CHAINS=0 VTS_LAZY_PYRAMID=0 VTS_FUSE_MERGE=0) and its phases (phase_times.py cuts THAT schedule into seven graphs)"
echo -n "joined schedule: "; VTS_D_CHAINS=0 VTS_LAZY_PYRAMID=0 VTS_FUSE_MERGE=0 run
VTS_LAZY_PYRAMID=0 VTS_FUSE_MERGE=0 python tools/probes/phase_times.py 2>/dev/null | tail -8
