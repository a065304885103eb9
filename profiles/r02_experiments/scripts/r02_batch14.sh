cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02_b14_tests.log 2>&1 || { tail -30 gpurun_out/r02_b14_tests.log; exit 1; }
tail -2 gpurun_out/r02_b14_tests.log
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 tools/trace_timeline.py $ARGS 2>&1 | grep -v amdgpu.ids | head -1; }
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
for ARGS in "1 1920 1080 8 64" "1 1920 1080 8 64 8 0"; do
  echo "#### $ARGS"
  for dt in 0 4 8 16 32; do run PTAMD_DT=$dt; done
done
ARGS=""
for dt in 0 4 8 16; do runb PTAMD_DT=$dt; done
ARGS="--emulate-world 8 --rank 0"
for dt in 0 4 8 16; do runb PTAMD_DT=$dt; done
runb PTAMD_DT=8 PTAMD_CB=512
