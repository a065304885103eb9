cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/pathtrace-on-cuda_amd/build
PTAMD_WG=1 timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02_wg3_tests.log 2>&1 || { tail -30 gpurun_out/r02_wg3_tests.log; exit 1; }
tail -2 gpurun_out/r02_wg3_tests.log
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'iters', r['bounce_iterations'])"; }
ARGS=""
runb PTAMD_WG=0
runb PTAMD_WG=1
runb PTAMD_WG=1 PTAMD_WGR=1
runb PTAMD_WG=1 PTAMD_WGR=64
runb PTAMD_WG=1 PTAMD_LIB=$B/libptamd_wg6.so PTAMD_WGB=1536
runb PTAMD_WG=1 PTAMD_LIB=$B/libptamd_wg5.so PTAMD_WGB=1280
ARGS="--emulate-world 8 --rank 0"
runb PTAMD_WG=0
runb PTAMD_WG=1
