# round-2 batch 1: VALU-rate probe, baseline bench lines, per-rank times of the 8-way split, cohort experiments at that size
cd $GRAFT_REPO_ROOT
set -e
timeout -k 10 120 python3 tools/valu_probe.py > gpurun_out/r02_valu_probe.json 2> gpurun_out/r02_valu_probe.err
echo "probe done"
timeout -k 10 200 python3 bench.py --no-cpu-baseline > gpurun_out/r02_b1_bench8.json 2> gpurun_out/r02_b1_bench8.err
cut -c1-200 gpurun_out/r02_b1_bench8.json
timeout -k 10 200 python3 tools/emulate_world.py --worlds 1,8 --ranks 0,3 > gpurun_out/r02_b1_emul.log 2>&1
tail -1 gpurun_out/r02_b1_emul.log | cut -c1-600
for v in "PTAMD_COHORTS=2" "PTAMD_COHORTS=2 PTAMD_TB=896" "PTAMD_TB=896" "PTAMD_TB=1280"; do
  echo "== $v"
  env $v timeout -k 10 120 python3 tools/emulate_world.py --worlds 8 --ranks 0 2>&1 | grep "^world"
done > gpurun_out/r02_b1_cohorts.log 2>&1
cat gpurun_out/r02_b1_cohorts.log
