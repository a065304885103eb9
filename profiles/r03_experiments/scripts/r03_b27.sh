#!/bin/bash
# b27: wave-by-wave dump of one wf_trace launch with the per-trip log; and what the timestamped build costs (sum of launch durations)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
: > gpurun_out/r03_b27.log
for args in "0 1920 1080 8 64 8 0" "1 1920 1080 8 256 8 0"; do
  timeout -k 10 300 python3 tools/trace_sum.py $args >> gpurun_out/r03_b27.log 2>&1
  PTAMD_TSTAT=2 timeout -k 10 300 python3 tools/trace_sum.py $args >> gpurun_out/r03_b27.log 2>&1
done
for args in "4 0 1920 1080 8 64 8 0" "4 1 1920 1080 8 256 8 0" "4 1 1920 1080 8 256" ; do
  echo "== wave_dump $args" >> gpurun_out/r03_b27.log
  timeout -k 10 300 python3 tools/wave_dump.py $args >> gpurun_out/r03_b27.log 2>&1
done
grep TSTAT gpurun_out/r03_b27.log
