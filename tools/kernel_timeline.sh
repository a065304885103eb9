# rocprofv3 kernel trace of one bench command, printed as a timeline: tools/kernel_timeline.sh TAG [bench args...]   (environment passes through)
# -> per-kernel totals, a 3 ms window of launches from the middle of the run (start, duration, gap to the previous kernel's end, queue)
#    and the time during which two kernels were running at once.
cd $GRAFT_REPO_ROOT
TAG=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
(cd /tmp && timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_$TAG -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_bench.log 2>&1) || { tail -5 gpurun_out/${TAG}_bench.log; exit 1; }
tail -1 gpurun_out/${TAG}_bench.log | cut -c1-200
KT=$(find /tmp/prof_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$KT" <<'PY'
import csv, sys, re
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wf_" in r["Kernel_Name"]]
def short(n):
    n = n.split("(")[0]
    return re.sub(r"^void ptd::|^ptd::", "", n)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "?")) for r in rows)
tot = {}
for s, e, n, q in ev:
    a = tot.setdefault(n, [0, 0]); a[0] += 1; a[1] += e - s
for n, (c, t) in sorted(tot.items(), key=lambda x: -x[1][1]):
    print("%-34s calls %6d  total %9.2f ms  avg %8.1f us" % (n, c, t / 1e6, t / c / 1e3))
t0 = ev[len(ev) // 2][0]
mid = [e for e in ev if t0 <= e[0] < t0 + 3_000_000]
last_end = None
for s, e, n, q in mid[:36]:
    print("%9.1f us  +%7.1f us  %-30s queue %s" % ((s - t0) / 1e3, (e - s) / 1e3, n, q))
pts = sorted([(s, 1) for s, e, n, q in ev] + [(e, -1) for s, e, n, q in ev])
cur = 0; last = pts[0][0]; t1 = t2 = 0
for t, d in pts:
    if cur >= 1: t1 += t - last
    if cur >= 2: t2 += t - last
    cur += d; last = t
print("wall span of the wf_* kernels: %.1f ms; time with >= 1 kernel running: %.1f ms, with >= 2: %.1f ms" % ((pts[-1][0] - pts[0][0]) / 1e6, t1 / 1e6, t2 / 1e6))
PY
