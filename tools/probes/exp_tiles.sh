cd $GRAFT_REPO_ROOT
P=$PWD/visual-tactile-synthesis_amd
echo "== run1 default"; VTS_LIB_PATH=$P/libvts_hip_run1.so python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
echo "== run3 default"; VTS_LIB_PATH=$P/libvts_hip_run3.so python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
echo "== run3 RUN=2"; VTS_TILE_RUN=2 VTS_LIB_PATH=$P/libvts_hip_run3.so python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
echo "== run3 RUN=4"; VTS_TILE_RUN=4 VTS_LIB_PATH=$P/libvts_hip_run3.so python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
echo "== run3 chunks<=10"; VTS_TILE_RUN_CHUNKS=10 VTS_LIB_PATH=$P/libvts_hip_run3.so python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
echo "== run3 chunks<=10 wgs1024"; VTS_TILE_RUN_CHUNKS=10 VTS_TILE_RUN_WGS=1024 VTS_LIB_PATH=$P/libvts_hip_run3.so python tools/mb_conv_ab.py 2>/dev/null | grep -v "^#"
