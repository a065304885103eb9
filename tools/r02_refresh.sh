# refresh of the judged measurement files after a kernel change: counters for configs 1 and 3, then the bench lines (which read profiles/r02_pmc_config<C>.json)
cd $GRAFT_REPO_ROOT
bash tools/r02_pmc_config.sh 1 64 8 > gpurun_out/r02_pmc1.log 2>&1 && cp gpurun_out/r02_pmc_config1.json profiles/r02_pmc_config1.json && echo pmc1 ok
bash tools/r02_pmc_config.sh 3 64 4 > gpurun_out/r02_pmc3.log 2>&1 && cp gpurun_out/r02_pmc_config3.json profiles/r02_pmc_config3.json && echo pmc3 ok
timeout -k 10 300 python3 bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_line.err && cut -c1-200 gpurun_out/r02_bench_line.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_line_steps20.json 2>/dev/null && cut -c1-200 gpurun_out/r02_bench_line_steps20.json
for c in "1 8" "3 8" "4 2"; do set -- $c; timeout -k 10 400 python3 bench.py --config $1 --steps $2 > gpurun_out/r02_bench_config$1.json 2>/dev/null; cut -c1-200 gpurun_out/r02_bench_config$1.json; done
