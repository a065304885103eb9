cd $GRAFT_REPO_ROOT
timeout -k 10 150 python bench.py --no-cpu-baseline --steps 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print('$1', round(d['value'],1), round(d['ms_per_step'],1), round(r['kernel_ms_sum'],1), r['bounce_iterations'])"
