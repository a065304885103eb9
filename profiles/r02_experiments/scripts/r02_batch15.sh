cd $GRAFT_REPO_ROOT
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
ARGS="--emulate-world 8 --rank 0"
runb A=1
for cs in 13 14 15 16; do runb PTAMD_CS=$cs; done
for gs in 6 7 8 10; do runb PTAMD_GS=$gs; done
runb PTAMD_CS=15 PTAMD_GS=7
runb PTAMD_CS=16 PTAMD_GS=6
runb PTAMD_RF=16
runb PTAMD_RF=32
runb PTAMD_RF=40
ARGS=""
runb A=1
runb PTAMD_CS=14
runb PTAMD_GS=7
