# rocprofv3 kernel trace of the default bench command + the PMC traffic pass (run on the GPU box from the repo root)
set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf /tmp/prof && (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.log 2>&1)
find /tmp/prof -name "*kernel_stats.csv" -exec cp {} gpurun_out/r01_kernel_stats.csv \;
KT=$(find /tmp/prof -name "*kernel_trace.csv" | head -1)
python3 - "$KT" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
out = {}
for key in ("wf_trace", "wf_shade"):
    d = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if key in r["Kernel_Name"]]
    d.sort()
    # the bench command renders twice: 1 warm-up pass, then the 8 timed passes; the timed render is the longer, later run of launches
    # split at the wf_init of the second render
    inits = sorted(int(r["Start_Timestamp"]) for r in rows if "wf_init" in r["Kernel_Name"])
    timed = [x[1] for x in d if x[0] > inits[-1]]
    out[key] = {"launches_total": len(d), "launches_timed_render": len(timed), "avg_ns_timed_render": sum(timed) / max(1, len(timed)),
                "sum_ms_timed_render": sum(timed) / 1e6, "avg_ns_all": sum(x[1] for x in d) / max(1, len(d))}
out["command"] = "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --no-cpu-baseline"
out["note"] = "bench.py renders a 1-pass warm-up before the timed 8-pass call; its roofline.kernel_ms_avg covers the timed call only"
json.dump(out, open("gpurun_out/r01_rocprofv3_timed_render_breakdown.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
PMC_SET_TIMEOUT=900 timeout -k 10 950 python3 tools/pmc_pass.py gpurun_out/pmc_traffic_raw.json tools/pmc_sets/traffic_only.txt -- python3 bench.py --no-cpu-baseline > gpurun_out/pmc_traffic.log 2>&1 || true
grep -E "^set|TIMED|^   [0-9]" gpurun_out/pmc_traffic.log | cut -c1-80
python3 tools/pmc_traffic.py gpurun_out/pmc_traffic_raw.json gpurun_out/r01_pmc_traffic.json --spp 256 --steps 8 || true
