cd $GRAFT_REPO_ROOT
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
ARGS=""
runb PTAMD_OCT=0
runb PTAMD_OCT=1
ARGS="--config 1"
runb PTAMD_OCT=0
runb PTAMD_OCT=1
for o in 0 1; do echo "== stat OCT=$o"; PTAMD_OCT=$o timeout -k 10 200 python3 tools/trace_stat.py 1 1920 1080 4 32 2>&1 | grep -v amdgpu.ids | head -3; done
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "image or split" 2>&1 | tail -2
