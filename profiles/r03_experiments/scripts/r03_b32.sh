#!/bin/bash
# b32: wf_drain hand-over threshold, finer sweep (8-way ranks 0 and 5, a 4-way rank, configs 2 and 4)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b32.log; : > $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for rep in 1 2; do
for e in "PTAMD_DRAIN=0" "PTAMD_DRAIN=40000" "PTAMD_DRAIN=50000" "PTAMD_DRAIN=60000" "PTAMD_DRAIN=80000" "PTAMD_DRAIN=48000 PTAMD_DSPREAD=2"; do
  run "$e" "--emulate-world 8 --rank 0"
done
done
for e in "PTAMD_DRAIN=0" "PTAMD_DRAIN=40000" "PTAMD_DRAIN=60000"; do
  run "$e" "--emulate-world 8 --rank 5"; run "$e" "--emulate-world 4 --rank 1"; run "$e" "--config 2"; run "$e" "--config 4 --steps 2"
done
paste - - < $L | cut -c1-190
