# copy the judged summaries of tools/r03_measure_all.sh from gpurun_out/ into profiles/
cd "$(dirname "$(dirname "$(readlink -f "$0")")")"
cp gpurun_out/r03_rocprofv3_kernel_stats.csv gpurun_out/r03_rocprofv3_timed_render_breakdown.json gpurun_out/r03_emulated_world.json gpurun_out/r03_emulated_world_noearly.json profiles/
python3 - <<'PY'
import json
out = {}
for tag, f in (("driver_steps20_warmup5", "r03_bench_driver"), ("config2_steps8", "r03_bench_line"), ("config1_steps8", "r03_bench_config1"), ("config3_steps8", "r03_bench_config3"), ("config4_steps2", "r03_bench_config4")):
    try:
        out[tag] = json.loads(open("gpurun_out/%s.json" % f).readline())
    except Exception as e:
        out[tag] = {"error": str(e)}
try:
    out["driver_command_under_rocprofv3"] = json.loads([l for l in open("gpurun_out/r03_prof_bench.log").read().splitlines() if l.startswith('{"metric"')][-1])
except Exception as e:
    out["driver_command_under_rocprofv3"] = {"error": str(e)}
json.dump(out, open("profiles/r03_bench_lines.json", "w"), indent=1)
print({k: (round(v.get("value", 0), 1) if isinstance(v, dict) else v) for k, v in out.items()})
PY
