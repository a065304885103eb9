# Do the kernels of two stream cohorts really overlap?  rocprofv3 kernel trace of one rank of an 8-way split with PTAMD_COHORTS=2.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PTAMD_COHORTS=${1:-2}
rm -rf /tmp/prof_co && (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_co -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-probes --emulate-world 8 --rank 0 --steps 8 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/co_bench.log 2>&1)
tail -1 gpurun_out/co_bench.log | cut -c1-160
KT=$(find /tmp/prof_co -name "*kernel_trace.csv" | head -1)
python3 - "$KT" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "wf_" in r["Kernel_Name"]]
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][-12:], r.get("Queue_Id", "?")) for r in rows)
t0 = ev[len(ev) // 2][0]
mid = [e for e in ev if t0 <= e[0] < t0 + 3_000_000]      # 3 ms window in the middle of the run
for s, e, n, q in mid[:40]:
    print("%9.1f us  +%7.1f us  %-12s queue %s" % ((s - t0) / 1e3, (e - s) / 1e3, n, q))
# overlap statistics: fraction of time with >= 2 kernels running
pts = sorted([(s, 1) for s, e, n, q in ev] + [(e, -1) for s, e, n, q in ev])
cur = 0; last = pts[0][0]; t1 = t2 = 0
for t, d in pts:
    if cur >= 1: t1 += t - last
    if cur >= 2: t2 += t - last
    cur += d; last = t
print("time with >=1 kernel: %.1f ms, with >=2 kernels: %.1f ms" % (t1 / 1e6, t2 / 1e6))
PY
