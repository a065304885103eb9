cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 tools/trace_timeline.py $ARGS 2>&1 | grep -v amdgpu.ids; }
ARGS="1 1920 1080 8 64"
run A=1
run PTAMD_BM=128 PTAMD_BS=16
run PTAMD_BM=64 PTAMD_BS=17
run PTAMD_BM=32 PTAMD_BS=18
ARGS="1 1920 1080 8 64 8 0"
run A=1
run PTAMD_BM=128
run PTAMD_BM=64
run PTAMD_BM=32
