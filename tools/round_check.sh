# GPU-box check used during development: parity suite, PMC summary at spp 64, refill-threshold sweep
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
PMC_SET_TIMEOUT=200 timeout -k 10 450 python3 tools/pmc_pass.py gpurun_out/pmc_traffic_raw.json tools/pmc_sets/traffic.txt -- python3 bench.py --no-cpu-baseline --spp 64 2>&1 | tee gpurun_out/pmc_traffic.log | grep -E "^set .* rc=|TIMED"
python3 tools/pmc_traffic.py gpurun_out/pmc_traffic_raw.json gpurun_out/r01_pmc_traffic.json --spp 64 --steps 8
for rf in 24; do
  echo "RF=$rf K=8" >> gpurun_out/rf.log
  PTAMD_RF=$rf timeout -k 10 120 python bench.py --no-cpu-baseline --steps 8 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); r=d['roofline']; print(d['value'], d['ms_per_step'], r['kernel_ms_sum'], r['bounce_iterations'])" >> gpurun_out/rf.log || exit 1
done
cat gpurun_out/rf.log
