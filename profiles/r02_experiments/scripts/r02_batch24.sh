cd $GRAFT_REPO_ROOT
B=$GRAFT_REPO_ROOT/pathtrace-on-cuda_amd/build
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 tools/trace_timeline.py $ARGS 2>&1 | grep -v amdgpu.ids | grep "render" ; }
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"; }
for ARGS in "1 1920 1080 8 64" "1 1920 1080 8 64 8 0"; do
  echo "#### $ARGS"
  run A=1
  for t in 24 48 96; do run PTAMD_LIB=$B/libptamd_prio$t.so; done
done
ARGS=""
runb A=1
for t in 24 48 96; do runb PTAMD_LIB=$B/libptamd_prio$t.so; done
ARGS="--emulate-world 8 --rank 0"
runb A=1
for t in 24 48 96; do runb PTAMD_LIB=$B/libptamd_prio$t.so; done
