# Round-2 measurement set (run on the GPU box from the repo root).  Everything lands in gpurun_out/; the summaries that are
# judged are copied into profiles/ afterwards.
cd $GRAFT_REPO_ROOT
set -e
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02_final_tests.log 2>&1 || { tail -30 gpurun_out/r02_final_tests.log; exit 1; }
tail -2 gpurun_out/r02_final_tests.log
timeout -k 10 120 python3 tools/valu_probe.py > gpurun_out/r02_valu_probe.json 2>/dev/null
echo "probe ok"
# the default bench line (config 2, 8 steps) and the driver's command line
timeout -k 10 300 python3 bench.py > gpurun_out/r02_bench_line.json 2> gpurun_out/r02_bench_line.err
cut -c1-220 gpurun_out/r02_bench_line.json
timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02_bench_line_steps20.json 2>/dev/null
cut -c1-220 gpurun_out/r02_bench_line_steps20.json
# the other configs (configs[1], [3], [4]; [4] at 2 of its 8 passes to keep the run short)
for c in "1 8" "3 8" "4 2"; do set -- $c; timeout -k 10 400 python3 bench.py --config $1 --steps $2 > gpurun_out/r02_bench_config$1.json 2>/dev/null; cut -c1-200 gpurun_out/r02_bench_config$1.json; done
# per-rank times of the 8-way split, every rank, config 2
timeout -k 10 300 python3 tools/emulate_world.py --worlds 1,2,4,8 > gpurun_out/r02_emulate_world.log 2>&1
tail -1 gpurun_out/r02_emulate_world.log > gpurun_out/r02_emulated_world.json
python3 -c "
import json; d=json.load(open('gpurun_out/r02_emulated_world.json'))
for w,v in d['worlds'].items(): print('world',w,'slowest %.4f s'%v['slowest_s'],'implied speed-up %.2f'%v.get('implied_speedup_vs_1',1.0))"
