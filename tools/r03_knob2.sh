cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/ab_env.sh 2 "A=0" "VTS_SMALL_WGS=384" "VTS_SMALL_WGS=512" "VTS_SMALL_WGS=768" "VTS_SMALL_WGS=1024"
