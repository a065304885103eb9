cd $GRAFT_REPO_ROOT
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
ARGS=""
runb A=1
runb PTAMD_ST=1024
runb PTAMD_ST=768
runb PTAMD_ST=640
runb PTAMD_ST=576
ARGS="--emulate-world 8 --rank 0"
runb A=1
runb PTAMD_SW=3 PTAMD_ST=256
