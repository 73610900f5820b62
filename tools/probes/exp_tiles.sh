cd $GRAFT_REPO_ROOT
export VTS_LIB_PATH=$PWD/visual-tactile-synthesis_amd/libvts_hip_exp.so
export VTS_MB_SEL="G up3 fwd"
for ab in 0 1 5 7 3; do echo "ABLATE=$ab default: $(VTS_ABLATE=$ab python tools/mb_conv_ab.py 2>/dev/null | grep -v '^#')   MT1: $(VTS_ABLATE=$ab VTS_EXP_MT1=1 python tools/mb_conv_ab.py 2>/dev/null | grep -v '^#')"; done
