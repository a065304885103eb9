#!/bin/bash
# b45: issue priority for the waves that carry a long ray (PTAMD_PRIO = node steps from which a ray counts as long; s_setprio 2 while the wave holds one) —
# round 2's experiment (f) once more, now that the launch tail can be measured
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b45.log; : > $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; s = d['roofline_shade']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' shade sum', round(s.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for c in "--emulate-world 4 --rank 1" "--emulate-world 8 --rank 6" "--config 2"; do
  for e in "PTAMD_PRIO=0" "PTAMD_PRIO=16" "PTAMD_PRIO=24" "PTAMD_PRIO=32" "PTAMD_PRIO=48" "PTAMD_PRIO=0"; do run "$e" "$c"; done
done
paste - - < $L | cut -c1-200
