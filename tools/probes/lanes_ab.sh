run() { python bench.py --no_cpu_baseline --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), d['ms_per_step_spread']['median'])"; }
run "default (LPT both phases)"
for g in "0|1,5|3|2,4" "0|1|2,4|3,5" "0|1,2|3|4,5" "0|3|1,5|2,4" "0,5|1|2,4|3" "0|1|2|3|4|5" "0|1,2|3,4,5" "0,3|1,4|2,5" "0|1,3|2,4,5" "0|1|2,3|4,5" "0|1,4|2,5|3"; do
VTS_LANE_GROUPS_G="$g" run "G: $g"
done
run "default (LPT both phases)"
