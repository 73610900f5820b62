cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in "" "wgrad4x4" "norm_stats" "norm_bwd" "norm_" "channel_sum" "conv4x4" "wgrad_reduce_batch" "conv4x4,wgrad4x4" "conv4x4,wgrad4x4,norm_,channel_sum"; do
    echo -n "knockout [$v]: "; VTS_KNOCKOUT=$v python bench.py --no_cpu_baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f ms'%d['ms_per_step'])"
done
python tools/probes/phase_times.py 2>/dev/null | tail -4
