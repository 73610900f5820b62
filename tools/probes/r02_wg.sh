cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_wg; mkdir -p $O
python tools/mb_wgrad.py > $O/new2.txt 2>&1; grep "^wgrad\|MAX" $O/new2.txt | cut -c1-175
