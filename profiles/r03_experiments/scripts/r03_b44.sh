#!/bin/bash
# b44: late budget — how empty must the wave be (PTAMD_LBL idle lanes: 62 = at most two rays left) and how many steps (PTAMD_LB)
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b44.log; : > $L
PTAMD_LB=32 PTAMD_LBL=32 timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_b44_tests.log 2>&1 || { tail -30 gpurun_out/r03_b44_tests.log; exit 1; }
tail -1 gpurun_out/r03_b44_tests.log | tee -a $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; s = d['roofline_shade']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' shade sum', round(s.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for c in "--emulate-world 4 --rank 1" "--emulate-world 8 --rank 6" "--config 2"; do
  for e in "PTAMD_LB=64" "PTAMD_LB=64 PTAMD_LBL=56" "PTAMD_LB=64 PTAMD_LBL=48" "PTAMD_LB=64 PTAMD_LBL=32" "PTAMD_LB=96 PTAMD_LBL=48" "PTAMD_LB=96 PTAMD_LBL=32" "PTAMD_LB=128 PTAMD_LBL=32" "PTAMD_LB=64"; do run "$e" "$c"; done
done
paste - - < <(tail -n +2 $L) | cut -c1-200
