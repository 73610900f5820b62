cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_wg; mkdir -p $O
python tools/mb_wgrad.py > $O/new.txt 2>&1; cat $O/new.txt | cut -c1-175
for a in 1 2 4 7; do echo "ABLATE $a"; VTS_ABLATE=$a VTS_MB=quick python tools/mb_wgrad.py 2>&1 | grep "^wgrad" | cut -c1-90; done > $O/ablate.txt
cat $O/ablate.txt
