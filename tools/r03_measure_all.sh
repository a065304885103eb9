# Round-3 measurement set (run on the GPU box from the repo root; about 12 minutes).  Everything lands in gpurun_out/; the summaries that
# are judged are copied into profiles/ afterwards (tools/r03_collect.sh).
cd $GRAFT_REPO_ROOT
set -e
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_final_tests.log 2>&1 || { tail -30 gpurun_out/r03_final_tests.log; exit 1; }
tail -2 gpurun_out/r03_final_tests.log
# counters first: the bench lines read profiles/r03_pmc_config<C>.json
for c in "2 64 8" "1 64 8" "3 64 4" "4 32 1"; do set -- $c
  bash tools/pmc_config.sh r03 $1 $2 $3 > gpurun_out/r03_pmc$1.log 2>&1 && cp gpurun_out/r03_pmc_config$1.json profiles/r03_pmc_config$1.json && echo "pmc config $1 ok" || { tail -5 gpurun_out/r03_pmc$1.log; exit 1; }
done
# the driver's command line, the default line (the config's own 8 passes), the other configs
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench_driver.json 2> gpurun_out/r03_bench_driver.err
cut -c1-200 gpurun_out/r03_bench_driver.json
timeout -k 10 300 python3 bench.py > gpurun_out/r03_bench_line.json 2> gpurun_out/r03_bench_line.err
cut -c1-200 gpurun_out/r03_bench_line.json
for c in "1 8" "3 8" "4 2"; do set -- $c; timeout -k 10 400 python3 bench.py --config $1 --steps $2 > gpurun_out/r03_bench_config$1.json 2>/dev/null; cut -c1-200 gpurun_out/r03_bench_config$1.json; done
# per-rank times of the tile split, every rank, config 2, with the defaults (early shade on for the 8-way ranks) and with early shade off
timeout -k 10 400 python3 tools/emulate_world.py --worlds 1,2,4,8 > gpurun_out/r03_emulate_world.log 2>&1
tail -1 gpurun_out/r03_emulate_world.log > gpurun_out/r03_emulated_world.json
PTAMD_EARLY=0 timeout -k 10 300 python3 tools/emulate_world.py --worlds 8 > gpurun_out/r03_emulate_world_noearly.log 2>&1
tail -1 gpurun_out/r03_emulate_world_noearly.log > gpurun_out/r03_emulated_world_noearly.json
python3 -c "
import json
d=json.load(open('gpurun_out/r03_emulated_world.json'))
for w,v in d['worlds'].items(): print('world',w,'slowest %.4f s'%v['slowest_s'],'implied speed-up %.2f'%v.get('implied_speedup_vs_1',1.0))
e=json.load(open('gpurun_out/r03_emulated_world_noearly.json'))
print('world 8 without early shade: slowest %.4f s' % e['worlds']['8']['slowest_s'])"
# rocprofv3 kernel trace + stats of the EXACT driver command (no counters in this run)
export TMPDIR=/tmp
rm -rf /tmp/prof_final && (cd /tmp && timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_final -- python3 $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $GRAFT_REPO_ROOT/gpurun_out/r03_prof_bench.log 2>&1)
find /tmp/prof_final -name "*kernel_stats.csv" -exec cp {} gpurun_out/r03_rocprofv3_kernel_stats.csv \;
KT=$(find /tmp/prof_final -name "*kernel_trace.csv" | head -1)
python3 - "$KT" <<'PY'
import csv, json, sys
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
out = {}
inits = sorted(int(r["Start_Timestamp"]) for r in rows if "wf_init" in r["Kernel_Name"])
# bench.py --steps 20 --warmup 5: render calls = warm-up (5 passes), then the timed 8 + 8 + 4, then the all-in-flight extra (20 passes)
first_timed, after_timed = inits[1], inits[4] if len(inits) > 4 else 1 << 62
for key in ("wf_trace", "wf_shade"):
    d = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in rows if key in r["Kernel_Name"])
    timed = [x[1] for x in d if first_timed < x[0] < after_timed]
    out[key] = {"launches_total": len(d), "launches_timed_renders": len(timed), "avg_ns_timed_renders": sum(timed) / max(1, len(timed)),
                "sum_ms_timed_renders": sum(timed) / 1e6, "avg_ns_all": sum(x[1] for x in d) / max(1, len(d))}
out["wf_init_launches"] = len(inits)
out["command"] = "rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --gpus 1 --steps 20 --warmup 5"
out["note"] = "render calls in this command: warm-up (5 passes), the timed 8 + 8 + 4 passes, the all-in-flight extra (20 passes); bench.py's roofline.kernel_ms_avg / roofline_shade.kernel_ms_avg (HIP events) cover the three timed calls only — compare with avg_ns_timed_renders"
json.dump(out, open("gpurun_out/r03_rocprofv3_timed_render_breakdown.json", "w"), indent=1)
print(json.dumps(out))
PY
python3 - <<'PY'
import json
b = json.loads([l for l in open("gpurun_out/r03_prof_bench.log").read().splitlines() if l.startswith('{"metric"')][-1])
print("profiled run: kernel_ms_avg trace %.4f shade %.4f" % (b["roofline"]["kernel_ms_avg"], b["roofline_shade"]["kernel_ms_avg"]))
PY
