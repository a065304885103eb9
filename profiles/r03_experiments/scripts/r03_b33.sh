#!/bin/bash
# b33: wf_drain on the 4-wide tree, the rays of a bounce in one flat loop, shadow rays stopping at any occluder (main) against the binary-tree
# walk (PTAMD_DQUAD=0); drain3 = the same at 168 VGPRs (3 waves/SIMD, 16 spilled).  Parity suite first, also with everything drained.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b33.log; : > $L
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_b33_tests.log 2>&1 || { tail -30 gpurun_out/r03_b33_tests.log; exit 1; }
tail -1 gpurun_out/r03_b33_tests.log | tee -a $L
PTAMD_DRAIN=1000000000 timeout -k 10 900 python3 -m pytest tests -m gpu -x -q -k "oracle or golden or image or parity or bit" > gpurun_out/r03_b33_tests_alldrain.log 2>&1 || { tail -30 gpurun_out/r03_b33_tests_alldrain.log; exit 1; }
tail -1 gpurun_out/r03_b33_tests_alldrain.log | tee -a $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" >> $L
}
V=$PWD/pathtrace-on-cuda_amd/build
for rep in 1; do
for e in "PTAMD_DRAIN=0" "PTAMD_DQUAD=0" "X=1" "PTAMD_LIB=$V/libptamd_drain3.so" "PTAMD_DRAIN=80000" "PTAMD_DRAIN=120000" "PTAMD_DRAIN=80000 PTAMD_LIB=$V/libptamd_drain3.so"; do
  run "$e" "--emulate-world 8 --rank 0"
done
done
for e in "PTAMD_DQUAD=0" "X=1" "PTAMD_DRAIN=120000"; do
  run "$e" "--emulate-world 4 --rank 1"; run "$e" "--config 2"; run "$e" "--config 3"; run "$e" "--config 1"
done
paste - - < <(tail -n +3 $L) | sed "s#$V/##" | cut -c1-190
bash tools/r03_b34.sh
