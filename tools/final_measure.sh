set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || { tail -20 gpurun_out/final_tests.log; exit 1; }
tail -2 gpurun_out/final_tests.log
timeout -k 10 300 python bench.py 2>/dev/null > gpurun_out/bench_r01.json
cut -c1-300 gpurun_out/bench_r01.json
export TMPDIR=/tmp
rm -rf /tmp/prof && (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01_kernel_stats.csv \;
head -8 gpurun_out/r01_kernel_stats.csv | cut -c1-200
PMC_SET_TIMEOUT=400 timeout -k 10 900 python3 tools/pmc_pass.py gpurun_out/pmc_traffic_raw.json tools/pmc_sets/traffic.txt -- python3 bench.py --no-cpu-baseline > gpurun_out/pmc_traffic.log 2>&1
grep -E "^set|TIMED|^   [0-9]" gpurun_out/pmc_traffic.log | cut -c1-80
python3 tools/pmc_traffic.py gpurun_out/pmc_traffic_raw.json gpurun_out/r01_pmc_traffic.json --spp 256 --steps 8
