# rocprofv3 kernel-trace stats of one bench command: tools/prof_stats.sh TAG [bench args...]  -> gpurun_out/TAG_kernel_stats.csv
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-probes "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench.log 2>&1) || { tail -5 gpurun_out/${TAG}_bench.log; exit 1; }
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} gpurun_out/${TAG}_kernel_stats.csv \;
python3 - gpurun_out/${TAG}_kernel_stats.csv <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    print("%-60s calls %6s  total %10.2f ms  avg %9.1f us  %5s%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
