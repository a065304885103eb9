cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_b11_tests.log 2>&1 || { tail -30 gpurun_out/r02_b11_tests.log; exit 1; }
tail -2 gpurun_out/r02_b11_tests.log
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'], 'frac', r.get('frac'))"; }
ARGS=""
runb A=1
runb A=2
ARGS="--config 1"
runb A=1
ARGS="--config 3 --steps 4"
runb A=1
