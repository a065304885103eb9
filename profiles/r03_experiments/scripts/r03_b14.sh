# round 3, batch 14: after the knob refactor (phase-1 workgroups 64 threads by default): parity suite, emulated split, a single full-frame pass with / without early shade
cd $GRAFT_REPO_ROOT
timeout -k 10 700 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_final_tests.log 2>&1 || { tail -30 gpurun_out/r03_final_tests.log; exit 1; }
tail -2 gpurun_out/r03_final_tests.log
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
for e in 0 2500000; do for rep in 1 2; do echo "== one full-frame pass of 256 spp, PTAMD_EARLY=$e"; PTAMD_EARLY=$e timeout -k 10 200 python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-probes 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 1), ' trace sum', round(d['roofline']['kernel_ms_sum'], 1))"; done; done
