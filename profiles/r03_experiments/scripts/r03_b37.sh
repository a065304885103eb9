#!/bin/bash
# b37: one-at-a-time re-sweep of the scheduling knobs at the end-of-round state (class order, 4 shards, wf_drain), configs[2] and an 8-way rank
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b37.log; : > $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; s = d['roofline_shade']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' shade sum', round(s.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for e in "X=1" "PTAMD_CS=11" "PTAMD_CS=13" "PTAMD_GS=8" "PTAMD_GS=10" "PTAMD_RF=16" "PTAMD_RF=32" "PTAMD_TT=48" "PTAMD_TT=32" "PTAMD_TB=2048" "PTAMD_TB=1536" "PTAMD_ST=256" "PTAMD_ST=1024" "PTAMD_TRS=2000000" "PTAMD_TRS=8000000" "PTAMD_HELP=2" "PTAMD_HELP=8" "X=2"; do
  run "$e" "--config 2"
done
for e in "X=1" "PTAMD_CS=11" "PTAMD_CS=13" "PTAMD_GS=8" "PTAMD_GS=10" "PTAMD_RF=16" "PTAMD_RF=32" "PTAMD_TT=48" "PTAMD_EST=128" "PTAMD_EARLY=3000000" "PTAMD_HELP=2" "PTAMD_HELP=8" "PTAMD_TB=1536" "X=2"; do
  run "$e" "--emulate-world 8 --rank 0"
done
paste - - < $L | cut -c1-200
