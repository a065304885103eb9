cd $GRAFT_REPO_ROOT
CONTIG=$GRAFT_REPO_ROOT/pathtrace-on-cuda_amd/build/libptamd_contig.so
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 tools/trace_timeline.py $ARGS 2>&1 | grep -v amdgpu.ids | head -1; }
for ARGS in "1 1920 1080 8 64" "1 1920 1080 8 64 8 0"; do
  echo "#### $ARGS"
  run PTAMD_LIB=$CONTIG
  run A=1
done
runb() { echo "== $*"; env "$@" timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes $ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'], 'frac', r.get('frac'), 'traffic', r.get('traffic'))"; }
ARGS=""
runb PTAMD_LIB=$CONTIG
runb A=1
