#!/bin/bash
# b26: block_append with the per-wave start positions computed once per list (main) vs summed by every thread (oldscan);
# then the wave-by-wave dump of one wf_trace launch (tools/wave_dump.py): 8-way rank of bunny / Cornell, and the full frame
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
bash tools/ab.sh r03_b26 main oldscan
for args in "4 1 1920 1080 8 256 8 0" "40 1 1920 1080 8 256 8 0" "4 0 1920 1080 8 64 8 0" "4 1 1920 1080 8 256" ; do
  echo "== wave_dump $args" >> gpurun_out/r03_b26.log
  timeout -k 10 300 python3 tools/wave_dump.py $args >> gpurun_out/r03_b26.log 2>&1
done
tail -5 gpurun_out/r03_b26.log
