#!/bin/bash
# b38: the passes of a tile side by side in the stream order (PTAMD_UNITORDER=1: unit = tile * passes + pass) instead of pass by pass —
# the 8 streams of a pixel then sit in neighbouring waves of the same XCD.  Parity suite with it on, then A/B.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b38.log; : > $L
PTAMD_UNITORDER=1 timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_b38_tests.log 2>&1 || { tail -30 gpurun_out/r03_b38_tests.log; exit 1; }
tail -1 gpurun_out/r03_b38_tests.log | tee -a $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; s = d['roofline_shade']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' shade sum', round(s.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for rep in 1 2; do for e in "PTAMD_UNITORDER=0" "PTAMD_UNITORDER=1"; do run "$e" "--config 2"; done; done
for c in "--config 1" "--config 3" "--config 4 --steps 2" "--emulate-world 8 --rank 0" "--emulate-world 4 --rank 1"; do for e in "PTAMD_UNITORDER=0" "PTAMD_UNITORDER=1"; do run "$e" "$c"; done; done
paste - - < <(tail -n +2 $L) | cut -c1-200
