# PMC summary of the default bench call at 64 spp per pass (run on the GPU box from the repo root) -> gpurun_out/r01_pmc_traffic.json
cd $GRAFT_REPO_ROOT
PMC_SET_TIMEOUT=200 timeout -k 10 600 python3 tools/pmc_pass.py gpurun_out/pmc_traffic_raw.json tools/pmc_sets/traffic.txt -- python3 bench.py --no-cpu-baseline --spp 64 2>&1 | tee gpurun_out/pmc_traffic.log | grep -E "^set .* rc=|TIMED"
python3 tools/pmc_traffic.py gpurun_out/pmc_traffic_raw.json gpurun_out/r01_pmc_traffic.json --spp 64 --steps 8 | grep -E "wf_|triad"
timeout -k 10 100 python3 tools/trace_stat.py 1 1920 1080 8 32 2>&1 | grep -v amdgpu.ids | head -4
