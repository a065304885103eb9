cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_t7_tests.log 2>&1 || { tail -30 gpurun_out/r02_t7_tests.log; exit 1; }
tail -2 gpurun_out/r02_t7_tests.log
run() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
ARGS=""
run PTAMD_TR=1
run PTAMD_TR=0
run PTAMD_TR=1
run PTAMD_TR=0
run PTAMD_TR=-1
ARGS="--config 1"
run PTAMD_TR=1
run PTAMD_TR=0
ARGS="--config 3 --steps 4"
run PTAMD_TR=1
run PTAMD_TR=0
ARGS="--emulate-world 8 --rank 0"
run PTAMD_TR=1
run PTAMD_TR=0
ARGS="--emulate-world 2 --rank 0"
run PTAMD_TR=1
run PTAMD_TR=0
