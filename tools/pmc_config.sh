# PMC per-sample summary of one config: tools/pmc_config.sh ROUND CONFIG SPP STEPS  -> gpurun_out/<ROUND>_pmc_config<C>.json (copy to profiles/)
# One rocprofv3 --pmc pass per line of tools/pmc_sets/traffic.txt (--kernel-trace only), then tools/pmc_summary.py.
cd $GRAFT_REPO_ROOT
R=$1; C=$2; SPP=$3; STEPS=$4
PMC_SET_TIMEOUT=${PMC_SET_TIMEOUT:-240} timeout -k 10 1100 python3 tools/pmc_pass.py gpurun_out/${R}_pmc_config${C}_raw.json tools/pmc_sets/traffic.txt -- python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra --config $C --spp $SPP --steps $STEPS --warmup 1 2>&1 | grep -E "^set .* rc=|TIMED" || exit 1
python3 - $R $C $SPP $STEPS <<'PY'
import sys, subprocess
sys.path.insert(0, '.')
import bench
r, c, spp, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
cfg = bench.CONFIGS[c]
samples = cfg["W"] * cfg["H"] * spp * (steps + 1)
subprocess.run([sys.executable, "tools/pmc_summary.py", "gpurun_out/%s_pmc_config%d_raw.json" % (r, c), "gpurun_out/%s_pmc_config%d.json" % (r, c), "--samples", str(samples),
                "--note", "collected at %d spp per pass, %d + 1 passes; per-sample figures" % (spp, steps)], check=True)
PY
