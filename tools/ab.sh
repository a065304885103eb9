#!/bin/bash
# A/B bench on the GPU box.  Stops at the first leg that fails: a comparison with a missing leg is no comparison.
#   tools/ab.sh TAG [--no-tests] LEG [LEG ...]
# LEG = name[:ENV=VAL[,ENV=VAL...]]   name "main" = the library in tree, anything else = pathtrace-on-cuda_amd/build/libptamd_<name>.so
#       (tools/build_variant.sh); the ENV=VAL pairs are exported for that leg only (PTAMD_* scheduling knobs).
# Runs the GPU parity suite first (unless --no-tests), then per leg: configs[2] twice, configs[1], configs[3] and one rank of an 8-way split once.
# Everything (stderr included) goes to gpurun_out/<TAG>.log; a leg that yields no JSON line aborts the run with a non-zero status.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
TAG=$1; shift
TESTS=1; if [ "${1:-}" = "--no-tests" ]; then TESTS=0; shift; fi
LOG=gpurun_out/${TAG}.log
mkdir -p gpurun_out; : > "$LOG"
V=$PWD/pathtrace-on-cuda_amd/build
if [ $TESTS = 1 ]; then
  timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_tests.log 2>&1 || { tail -30 gpurun_out/${TAG}_tests.log; echo "ab.sh: GPU tests failed" | tee -a "$LOG"; exit 1; }
  tail -2 gpurun_out/${TAG}_tests.log | tee -a "$LOG"
fi
run() {
  local leg=$1; shift
  local name=${leg%%:*} envs=""
  [ "$leg" != "$name" ] && envs=${leg#*:}
  local -a E=()
  if [ "$name" != main ]; then
    [ -f "$V/libptamd_$name.so" ] || { echo "ab.sh: variant library $V/libptamd_$name.so missing" | tee -a "$LOG"; exit 2; }
    E+=("PTAMD_LIB=$V/libptamd_$name.so")
  fi
  if [ -n "$envs" ]; then IFS=, read -ra kv <<< "$envs"; E+=("${kv[@]}"); fi
  echo "== $leg $*" | tee -a "$LOG"
  local out
  out=$(env "${E[@]}" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes "$@" 2>> "$LOG") || { echo "ab.sh: leg '$leg $*' failed (see $LOG)" | tee -a "$LOG"; exit 3; }
  echo "$out" >> "$LOG"
  echo "$out" | python3 -c "
import sys, json
line = sys.stdin.readline()
if not line.strip().startswith('{'): sys.exit('no JSON line')
d = json.loads(line); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 1), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' iters', r.get('bounce_iterations'))" | tee -a "$LOG" || { echo "ab.sh: leg '$leg $*' produced no bench line" | tee -a "$LOG"; exit 4; }
}
for rep in 1 2; do for n in "$@"; do run "$n"; done; done
for n in "$@"; do run "$n" --config 1; done
for n in "$@"; do run "$n" --config 3; done
for n in "$@"; do run "$n" --emulate-world 8 --rank 0; done
