cd $GRAFT_REPO_ROOT
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
ARGS="--emulate-world 8 --rank 0"
runb A=1
runb PTAMD_BR=64
runb PTAMD_BR=64 PTAMD_CS=14
runb PTAMD_BR=32 PTAMD_CS=14
runb PTAMD_BR=128 PTAMD_CS=13
runb PTAMD_BR=16 PTAMD_CS=15
ARGS=""
runb A=1
runb PTAMD_BR=64 PTAMD_CS=14
ARGS="--config 1"
runb A=1
runb PTAMD_BR=64 PTAMD_CS=14
