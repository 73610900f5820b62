cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for a in 0 1 2 4 3 6 7; do echo "ABLATE $a"; VTS_ABLATE=$a VTS_MB=top python tools/microbench_conv.py 2>&1 | grep "^conv" | head -2; VTS_ABLATE=$a python tools/microbench_conv.py 2>&1 | grep "^conv" | head -2; done
