# What produced profiles/ (run on the GPU box from the repo root): parity suite, the default bench line,
# the rocprofv3 kernel trace of the same command with a per-render split.  (PMC summaries: tools/round_check.sh.)
set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python -m pytest tests -m gpu -x -q > gpurun_out/final_tests.log 2>&1 || { tail -20 gpurun_out/final_tests.log; exit 1; }
tail -2 gpurun_out/final_tests.log
timeout -k 10 300 python bench.py 2>/dev/null > gpurun_out/bench_r01.json
cut -c1-260 gpurun_out/bench_r01.json
export TMPDIR=/tmp
rm -rf /tmp/prof && (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01_kernel_stats.csv \;
KT=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python3 - "$KT" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
out = {}
inits = sorted(int(r["Start_Timestamp"]) for r in rows if "wf_init" in r["Kernel_Name"])
for key in ("wf_trace", "wf_shade"):
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if key in r["Kernel_Name"])
    timed = [x[1] for x in d if x[0] > inits[-1]]      # launches after the wf_init of the second (timed) render
    out[key] = {"launches_total": len(d), "launches_timed_render": len(timed), "avg_ns_timed_render": sum(timed) / max(1, len(timed)),
                "sum_ms_timed_render": sum(timed) / 1e6, "avg_ns_all": sum(x[1] for x in d) / max(1, len(d))}
out["command"] = "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline"
out["note"] = "bench.py renders a 1-pass warm-up before the timed 8-pass call; its roofline.kernel_ms_avg (HIP events) covers the timed call only"
json.dump(out, open("gpurun_out/r01_rocprofv3_timed_render_breakdown.json", "w"), indent=1)
print(json.dumps(out["wf_trace"]))
PY
python3 -c "
import json; d=json.load(open('gpurun_out/bench_r01.json')); r=d['roofline']; print('HIP events: wf_trace avg ms', r['kernel_ms_avg'], 'launches', r['launches_timed'], 'value', d['value'], 'triad', r['peak_measured_triad'], 'traffic', r['traffic'])"
