#!/bin/bash
# b31: wf_drain for the last live streams, with the streams spread over more waves (PTAMD_DSPREAD) and a finer poll near the hand-over
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b31.log; : > $L
PTAMD_DRAIN=30000 timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_b31_tests.log 2>&1 || { tail -30 gpurun_out/r03_b31_tests.log; exit 1; }
tail -1 gpurun_out/r03_b31_tests.log | tee -a $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
for e in "PTAMD_DRAIN=0" "PTAMD_DRAIN=30000 PTAMD_DSPREAD=0" "PTAMD_DRAIN=30000 PTAMD_DSPREAD=2" "PTAMD_DRAIN=30000" "PTAMD_DRAIN=60000" "PTAMD_DRAIN=100000" "PTAMD_DRAIN=150000 PTAMD_DSPREAD=1" "PTAMD_DRAIN=15000" "PTAMD_DRAIN=0"; do
  run "$e" "--emulate-world 8 --rank 0"
done
for e in "PTAMD_DRAIN=0" "PTAMD_DRAIN=30000" "PTAMD_DRAIN=100000"; do
  run "$e" "--config 2"; run "$e" "--config 3"; run "$e" "--config 1"
done
paste - - < <(tail -n +2 $L) | cut -c1-190
