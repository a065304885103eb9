#!/bin/bash
# b28: the launch timeline once more, now that the timestamped build no longer stretches the tail it measures (striped timeline words, no
# pooled histograms): cost of the build (trace_sum), then timeline of an 8-way rank (bunny, Cornell) and of the full frame
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
: > gpurun_out/r03_b28.log
for args in "0 1920 1080 8 64 8 0" "1 1920 1080 8 256 8 0" "1 1920 1080 8 256"; do
  timeout -k 10 300 python3 tools/trace_sum.py $args >> gpurun_out/r03_b28.log 2>&1
  PTAMD_TSTAT=2 timeout -k 10 300 python3 tools/trace_sum.py $args >> gpurun_out/r03_b28.log 2>&1
done
for args in "0 1920 1080 8 64 8 0" "1 1920 1080 8 256 8 0" "1 1920 1080 8 256" "1 1920 1080 8 256 4 0" "1 1920 1080 8 256 2 0"; do
  echo "== trace_timeline $args" >> gpurun_out/r03_b28.log
  timeout -k 10 300 python3 tools/trace_timeline.py $args >> gpurun_out/r03_b28.log 2>&1
done
grep -E "TSTAT|^render|^==" gpurun_out/r03_b28.log
