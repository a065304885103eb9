cd $GRAFT_REPO_ROOT
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
ARGS=""
for i in 1 2; do
runb PTAMD_SW=3
runb PTAMD_SW=4 PTAMD_ST=512
runb PTAMD_SW=4 PTAMD_ST=384
runb PTAMD_SW=4 PTAMD_ST=448
runb PTAMD_SW=3 PTAMD_ST=384
done
ARGS="--config 1"
runb PTAMD_SW=3
runb PTAMD_SW=4 PTAMD_ST=512
ARGS="--config 3 --steps 4"
runb PTAMD_SW=3
runb PTAMD_SW=4 PTAMD_ST=512
