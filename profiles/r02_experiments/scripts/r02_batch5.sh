cd $GRAFT_REPO_ROOT
NOTOP=$GRAFT_REPO_ROOT/pathtrace-on-cuda_amd/build/libptamd_notop.so
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
run PTAMD_LIB=$NOTOP PTAMD_SPLIT=0 PTAMD_BFS=1
run PTAMD_LIB=$NOTOP PTAMD_SPLIT=0 PTAMD_BFS=1024
run PTAMD_LIB=$NOTOP PTAMD_SPLIT=0 PTAMD_BFS=100000
run PTAMD_LIB=$NOTOP PTAMD_SPLIT=1 PTAMD_BFS=1
run PTAMD_SPLIT=0 PTAMD_BFS=1024 PTAMD_TOP=76
run PTAMD_SPLIT=0 PTAMD_BFS=1 PTAMD_TOP=1
