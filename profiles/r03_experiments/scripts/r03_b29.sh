#!/bin/bash
# b29: hand the last live streams of a render to wf_drain (PTAMD_DRAIN = live-stream threshold) instead of hundreds of sparse iterations
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b29.log; : > $L
for d in 0 500 2000 8000 30000 100000; do
  for cfg in "--emulate-world 8 --rank 0" "--config 2" "--config 1"; do
    echo "== PTAMD_DRAIN=$d $cfg" >> $L
    PTAMD_DRAIN=$d timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $cfg 2>> $L | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
  done
done
cat $L
