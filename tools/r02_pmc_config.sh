# PMC per-sample summary of one config: tools/r02_pmc_config.sh CONFIG SPP STEPS  -> gpurun_out/r02_pmc_config<C>.json (copy to profiles/)
cd $GRAFT_REPO_ROOT
C=$1; SPP=$2; STEPS=$3
PMC_SET_TIMEOUT=${PMC_SET_TIMEOUT:-240} timeout -k 10 1100 python3 tools/pmc_pass.py gpurun_out/r02_pmc_config${C}_raw.json tools/pmc_sets/traffic.txt -- python3 bench.py --no-cpu-baseline --no-probes --config $C --spp $SPP --steps $STEPS --warmup 1 2>&1 | grep -E "^set .* rc=|TIMED"
python3 - $C $SPP $STEPS <<'PY'
import sys, subprocess
sys.path.insert(0, '.')
import bench
c, spp, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
cfg = bench.CONFIGS[c]
samples = cfg["W"] * cfg["H"] * spp * (steps + 1)
subprocess.run([sys.executable, "tools/pmc_summary.py", "gpurun_out/r02_pmc_config%d_raw.json" % c, "gpurun_out/r02_pmc_config%d.json" % c, "--samples", str(samples),
                "--note", "collected at %d spp per pass, %d + 1 passes; per-sample figures" % (spp, steps)], check=True)
PY
