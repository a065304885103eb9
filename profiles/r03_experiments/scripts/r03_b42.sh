#!/bin/bash
# b42: a ray that has done PTAMD_LB node steps is suspended once the queue is dry and its wave holds at most two rays (the launch no longer
# waits for a lone long ray; tools/straggler_cost.py, r03_b41.log: an 8-way rank's big launches wait 15 us on average for their latest
# stripe of waves, 20 % of them more than 25 us).  Parity suite with LB = 32 first.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b42.log; : > $L
PTAMD_LB=32 timeout -k 10 800 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_b42_tests.log 2>&1 || { tail -30 gpurun_out/r03_b42_tests.log; exit 1; }
tail -1 gpurun_out/r03_b42_tests.log | tee -a $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; s = d['roofline_shade']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' shade sum', round(s.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for rep in 1 2; do for e in "PTAMD_LB=0" "PTAMD_LB=32" "PTAMD_LB=48" "PTAMD_LB=64" "PTAMD_LB=96" "PTAMD_LB=128"; do run "$e" "--emulate-world 8 --rank 0"; done; done
for c in "--emulate-world 8 --rank 5" "--emulate-world 4 --rank 1" "--config 2" "--config 3" "--config 1"; do for e in "PTAMD_LB=0" "PTAMD_LB=48" "PTAMD_LB=96"; do run "$e" "$c"; done; done
paste - - < <(tail -n +2 $L) | cut -c1-200
