# round 3, batch 2: counters for the lobe-sort experiment, the TA counter passes that timed out in round 2 (once), the new bench line
cd $GRAFT_REPO_ROOT
B="python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra --config 2 --spp 32 --steps 4 --warmup 1"
PMC_SET_TIMEOUT=150 timeout -k 10 400 python3 tools/pmc_pass.py gpurun_out/r03_b2_pmc_main.json tools/pmc_sets/basic.txt -- $B > gpurun_out/r03_b2_pmc_main.log 2>&1 || { tail -20 gpurun_out/r03_b2_pmc_main.log; exit 1; }
PTAMD_SORT=1 PMC_SET_TIMEOUT=150 timeout -k 10 400 python3 tools/pmc_pass.py gpurun_out/r03_b2_pmc_sort.json tools/pmc_sets/basic.txt -- $B > gpurun_out/r03_b2_pmc_sort.log 2>&1 || { tail -20 gpurun_out/r03_b2_pmc_sort.log; exit 1; }
python3 - <<'PY'
import json
for tag in ("main", "sort"):
    d = json.load(open("gpurun_out/r03_b2_pmc_%s.json" % tag))["kernels"]
    for k, c in d.items():
        if "wf_shade" in k or "wf_trace" in k:
            print(tag, k.split("(")[0][:40], "launches", c.get("launches"), "VALU insts %.3e" % c["SQ_INSTS_VALU"], "lane util %.3f" % (c["SQ_THREAD_CYCLES_VALU"] / (64.0 * c["SQ_ACTIVE_INST_VALU"])),
                  "wait %.3f" % (c["SQ_WAIT_ANY"] / c["SQ_WAVE_CYCLES"]), "wave cycles %.3e" % c["SQ_WAVE_CYCLES"], "LDS insts %.3e" % c.get("SQ_INSTS_LDS", 0), "busy cycles %.3e" % c.get("GRBM_GUI_ACTIVE", 0))
PY
# the TA passes: ONE run, time limit 150 s per pass; on a timeout the launcher kills the whole group, keeps the child's log and stops
PMC_SET_TIMEOUT=150 timeout -k 10 400 python3 tools/pmc_pass.py gpurun_out/r03_b2_pmc_ta.json tools/pmc_sets/ta_only.txt -- python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra --steps 2 --warmup 1 --spp 32 > gpurun_out/r03_b2_pmc_ta.log 2>&1; echo "TA pass exit $?"; grep -E "^set|TIMED" gpurun_out/r03_b2_pmc_ta.log
timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > gpurun_out/r03_b2_bench_driver.json 2> gpurun_out/r03_b2_bench_driver.err || { tail -20 gpurun_out/r03_b2_bench_driver.err; exit 1; }
cat gpurun_out/r03_b2_bench_driver.json
