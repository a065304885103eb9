cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/pathtrace-on-cuda_amd/build
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
ARGS=""
runb A=1
runb PTAMD_LIB=$B/libptamd_addr.so
runb PTAMD_LIB=$B/libptamd_stop.so
runb PTAMD_LIB=$B/libptamd_both.so
runb PTAMD_LIB=$B/libptamd_bothc.so
runb A=1
