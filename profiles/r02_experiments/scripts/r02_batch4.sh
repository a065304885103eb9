cd $GRAFT_REPO_ROOT
set -e
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_b4_tests.log 2>&1 || { tail -30 gpurun_out/r02_b4_tests.log; exit 1; }
tail -2 gpurun_out/r02_b4_tests.log
for v in "PTAMD_SPLIT=0 PTAMD_TOP=0" "PTAMD_SPLIT=0 PTAMD_TOP=76" "PTAMD_SPLIT=1 PTAMD_TOP=0" "PTAMD_SPLIT=1 PTAMD_TOP=76" "PTAMD_SPLIT=1 PTAMD_TOP=21" "PTAMD_SPLIT=1 PTAMD_TOP=76 PTAMD_SW=2"; do
  echo "== $v"
  env $v timeout -k 10 200 python3 bench.py --no-cpu-baseline --no-probes 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(round(d['value'],1), 'ms/step', round(d['ms_per_step'],1), 'trace sum', round(r['kernel_ms_sum'],1), 'iters', r['bounce_iterations'])"
done 2>&1 | tee gpurun_out/r02_b4.log
