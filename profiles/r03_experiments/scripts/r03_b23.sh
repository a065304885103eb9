# round 3, batch 23: fair shares of the SIMD — a wave lowers its issue priority as it gets ahead (PTAMD_FAIR = shift: one level every max(8, rays >> shift) trips)
cd $GRAFT_REPO_ROOT
PTAMD_FAIR=16 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "image_matches or early or tile_split" > gpurun_out/r03_b23_tests.log 2>&1 || { tail -30 gpurun_out/r03_b23_tests.log; exit 1; }
tail -2 gpurun_out/r03_b23_tests.log
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra $ARGS 2>>gpurun_out/r03_b23.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  ms/step', round(d['ms_per_step'], 1), ' trace sum', round(r['kernel_ms_sum'], 1), ' iters', r['bounce_iterations'])" || exit 1; }
ARGS="--emulate-world 8 --rank 0"
echo "#### 8-way rank, early shade off"
run PTAMD_EARLY=0
for f in 14 15 16 17 18; do run PTAMD_EARLY=0 PTAMD_FAIR=$f; done
echo "#### 8-way rank, defaults (early shade on)"
run X=1
for f in 15 16 17; do run PTAMD_FAIR=$f; done
ARGS=""
echo "#### full frame"
run X=1
for f in 15 16 17 18; do run PTAMD_FAIR=$f; done
ARGS="--config 1"
echo "#### configs[1]"
run X=1
for f in 16 17; do run PTAMD_FAIR=$f; done
