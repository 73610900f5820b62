cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { timeout 300 python bench.py --train_only --steps 150 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3))"; }
export VTS_TUNING=1 VTS_KO_LANES_ACK=timing-only
echo -n "chains default: "; run
echo -n "D2 lanes 0|1|2: "; VTS_D2_LANES="0|1|2" run
echo -n "D2 lanes 0|1|2, D1 serial on main: "; VTS_D2_LANES="0|1|2" VTS_D1_SERIAL=1 run
echo -n "D2 lanes 0|1,2, D1 serial on main: "; VTS_D1_SERIAL=1 run
echo -n "D2 lanes 0,2|1: "; VTS_D2_LANES="0,2|1" run
for l in 0 "1,2" "0,1,2" 3 "4,5" "3,4,5" "0,1,2,3,4,5"; do echo -n "KO lanes [$l]: "; VTS_KO_LANES=$l run; done
for v in "norm_" "wgrad4x4,wgrad_small,wgrad_head,wgrad_reduce" "conv4x4"; do echo -n "knockout [$v]: "; VTS_KNOCKOUT=$v run; done
