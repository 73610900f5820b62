# knock-out timing experiments on the replayed step (results of these runs are WRONG by construction: timing only)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export VTS_TUNING=1 VTS_KO_LANES_ACK=timing-only
run() { python bench.py --train_only --steps 150 --warmup 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'])"; }
echo -n "baseline: "; run
for l in 0 1 2 3 4 5 "0,1,2" "3,4,5" "0,1,2,3,4,5"; do echo -n "KO lanes [$l]: "; VTS_KO_LANES=$l run; done
for v in "conv_small,wgrad_small,conv_head_small" "norm_" "channel_sum,chsum" "wgrad4x4,wgrad_small,wgrad_head,wgrad_reduce" "conv4x4" "conv_head"; do echo -n "knockout [$v]: "; VTS_KNOCKOUT=$v run; done
python tools/probes/phase_times.py 2>/dev/null | tail -8
