#!/bin/bash
# b34: how long does wf_drain itself run?  rocprofv3 kernel statistics of an 8-way rank, 4-wide walk and binary walk
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
export TMPDIR=/tmp
L=$PWD/gpurun_out/r03_b34.log; : > $L
for q in 1 0; do
  export PTAMD_DQUAD=$q
  rm -rf /tmp/b34_$q
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/b34_$q -- python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra --emulate-world 8 --rank 0 > /tmp/b34_$q.out 2>&1 || { tail -20 /tmp/b34_$q.out; exit 1; }
  echo "== PTAMD_DQUAD=$q" >> $L
  f=$(find /tmp/b34_$q -name "*kernel_stats.csv" | head -1)
  python3 - "$f" >> $L <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if any(k in n for k in ("wf_drain", "wf_trace", "wf_shade", "wf_init")):
        print("%-60s calls %6s total %10.3f ms  avg %9.1f us  max %9.1f us" % (n[:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
done
cat $L
