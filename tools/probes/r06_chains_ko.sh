# round 6: label knock-outs on the chained schedule (results of these runs are WRONG by construction: timing only).  The patch-stack
# launches have their own label class now (patch_conv4x4 / patch_wgrad4x4 / norm_patch_*): round 5's "conv_small,wgrad_small,conv_head_small"
# matched no label.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --train_only --steps 150 $EXTRA 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
export VTS_TUNING=1 VTS_KO_LANES_ACK=timing-only
echo -n "chains default: "; run
for v in "patch_" "patch_conv4x4" "patch_wgrad4x4" "patch_,norm_patch" "norm_patch" "channel_sum" "ganloss" "avgpool" "norm_finalize,norm_from_partials,norm_stats" "norm_bwd" "wgrad_reduce" "wgrad4x4" "conv4x4" "l1,patch_scatter,patch_jobs,mask_select"; do echo -n "knockout [$v]: "; VTS_KNOCKOUT=$v run; done
echo -n "chains default again: "; run
