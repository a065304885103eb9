cd $GRAFT_REPO_ROOT
export PTAMD_LIB=$GRAFT_REPO_ROOT/pathtrace-on-cuda_amd/build/libptamd_notop.so PTAMD_SPLIT=0
PMC_SET_TIMEOUT=150 timeout -k 10 1100 python3 tools/pmc_pass.py gpurun_out/r02_pmc_deep_raw.json tools/pmc_sets/deep.txt -- python3 bench.py --no-cpu-baseline --no-probes --steps 4 --spp 32 2>&1 | grep -E "^set .* rc=|TIMED|s$" | tail -30
