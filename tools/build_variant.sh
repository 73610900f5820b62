# build a variant of the library next to the production one (same-box A/B with tools/ab_libs.sh):
#   bash tools/build_variant.sh <name> "<extra hipcc flags, e.g. -DVTS_X=1>"   ->  visual-tactile-synthesis_amd/libvts_hip_<name>.so
set -e
R=$(cd $(dirname $0)/.. && pwd)
D=/tmp/csrc_$1
rm -rf $D && mkdir -p $D && cp $R/visual-tactile-synthesis_amd/csrc/*.hip $R/visual-tactile-synthesis_amd/csrc/*.h $R/visual-tactile-synthesis_amd/csrc/*.cpp $R/visual-tactile-synthesis_amd/csrc/Makefile $D/
cd $D
sed -i "s#-I../../include#-I$R/include#; s#\.\./\.\./include/vts.h#$R/include/vts.h#g; s#\.\./libvts_hip.so#$R/visual-tactile-synthesis_amd/libvts_hip_$1.so#" Makefile
make -j8 FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$R/include -I. -Wno-unused-result $2" 2>&1 | grep -v "hipcc" | tail -3
ls -la $R/visual-tactile-synthesis_amd/libvts_hip_$1.so
