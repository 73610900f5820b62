# Same-box A/B of lane-to-stream packings and hardware queue counts.  bash tools/probes/lanes_ab.sh (through gpurun)
run() { python bench.py --no_cpu_baseline --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['ms_per_step'],3), d['ms_per_step_spread']['median'], round(d['ms_per_step_fresh_input'],3))"; }
for rep in 1 2; do
run "packed into four streams, L1 terms as a lane (default)"
VTS_LANE_STREAMS=0 VTS_G_PRE_LANE=0 run "one stream per lane, L1 terms serial (round 3)"
VTS_G_PRE_LANE=0 run "packed, L1 terms serial"
VTS_LANE_STREAMS=3 run "three streams"
VTS_LANE_STREAMS=5 run "five streams"
VTS_LANE_GROUPS="0|1,2,3,4,5" run "two streams"
GPU_MAX_HW_QUEUES=5 run "GPU_MAX_HW_QUEUES=5"
GPU_MAX_HW_QUEUES=8 run "GPU_MAX_HW_QUEUES=8"
done
