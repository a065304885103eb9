#!/bin/bash
# b40: are the chip's two run-to-run states (configs[2] 1,750 / 1,805) a matter of how the stream-state arrays lie relative to each other?
# PTAMD_N16PAD adds elements to every stream-indexed array (spacing 265,420,800 B = 2^17 x 2025 at 1080p x 8 passes); repeated runs per value
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b40.log; : > $L
run() {
  echo "== $1 $2" >> $L
  env $1 timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $2 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']; s = d['roofline_shade']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 2), ' trace sum', round(r.get('kernel_ms_sum', 0), 1), ' shade sum', round(s.get('kernel_ms_sum', 0), 1))" >> $L
}
for rep in 1 2 3 4; do
  for p in 0 260 4100 65540 1048580; do run "PTAMD_N16PAD=$p" "--config 2"; done
done
paste - - < $L | sort -k2,2 -s | cut -c1-160
