#!/bin/bash
# b36: early shade for a 4-way rank too?  (PTAMD_EARLY = largest render, in streams, that uses it; default 2.5 M = an 8-way rank of 1080p x 8 passes)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b36.log; : > $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for rep in 1 2; do
for e in "PTAMD_EARLY=2500000" "PTAMD_EARLY=4200000"; do
  run "$e" "--emulate-world 4 --rank 1"; run "$e" "--emulate-world 4 --rank 2"
done
done
for e in "PTAMD_EARLY=2500000" "PTAMD_EARLY=4200000" "PTAMD_EARLY=8400000"; do run "$e" "--emulate-world 2 --rank 0"; done
paste - - < $L | cut -c1-170
