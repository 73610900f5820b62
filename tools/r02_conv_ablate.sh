cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r02_conv; mkdir -p $O
for a in 0 1 2 4 3 5 6 7; do echo "ABLATE $a"; VTS_ABLATE=$a VTS_MB=top python tools/microbench_conv.py 2>&1 | grep "^conv" ; done > $O/ablate.txt
cat $O/ablate.txt
