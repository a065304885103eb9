cd $GRAFT_REPO_ROOT
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'iters', r['bounce_iterations'])"; }
ARGS=""
for r in 1 16 32 48 64 96 128; do runb PTAMD_WG=1 PTAMD_WGR=$r; done
