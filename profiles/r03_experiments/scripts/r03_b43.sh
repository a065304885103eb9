#!/bin/bash
# b43: confirmation of b42 — PTAMD_LB = 64 against 0, interleaved and repeated (the chip's two states flip between processes)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b43.log; : > $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; s = d['roofline_shade']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' shade sum', round(s.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for c in "--emulate-world 4 --rank 1" "--emulate-world 4 --rank 2" "--emulate-world 8 --rank 3" "--emulate-world 8 --rank 6" "--emulate-world 2 --rank 0" "--config 2" "--config 3" "--config 4 --steps 2"; do
  for rep in 1 2; do for e in "PTAMD_LB=0" "PTAMD_LB=64"; do run "$e" "$c"; done; done
done
paste - - < $L | cut -c1-200
