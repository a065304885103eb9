#!/bin/bash
# b30: node budget that shrinks with the launch (PTAMD_BS shift, PTAMD_BM floor): small budgets only for the sparse launches of a render's tail; with and without wf_drain
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b30.log; : > $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for e in "X=1" "PTAMD_BM=128" "PTAMD_BM=64" "PTAMD_BM=32" "PTAMD_BM=16" "PTAMD_BS=13 PTAMD_BM=32" "PTAMD_BS=12 PTAMD_BM=32" "PTAMD_BM=32 PTAMD_DRAIN=30000" "PTAMD_BM=64 PTAMD_DRAIN=30000" "PTAMD_DRAIN=30000" "PTAMD_DRAIN=50000" "PTAMD_DRAIN=20000"; do
  run "$e" "--emulate-world 8 --rank 0"
done
for e in "X=1" "PTAMD_BM=64" "PTAMD_BM=32" "PTAMD_BM=32 PTAMD_DRAIN=30000"; do
  run "$e" "--config 2"; run "$e" "--config 3"
done
paste - - < $L | cut -c1-190
