# quick K=8 / K=1 bench lines (value, ms/step, sum of wf_trace ms, iterations); extra env is passed through
cd $GRAFT_REPO_ROOT
for k in 8 1; do
  timeout -k 10 150 python bench.py --no-cpu-baseline --steps $k 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('K=$k', round(d['value'],1), round(d['ms_per_step'],1), round(r['kernel_ms_sum'],1), r['bounce_iterations'])" || exit 1
done
