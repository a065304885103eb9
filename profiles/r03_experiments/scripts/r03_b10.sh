# round 3, batch 10: cache-policy bits on the state stores (sc1 / sc0 sc1 / sc0) under early shade, 8-way rank
cd $GRAFT_REPO_ROOT
V=$PWD/pathtrace-on-cuda_amd/build
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra --emulate-world 8 --rank 0 2>>gpurun_out/r03_b10.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  seconds', round(d['emulated']['seconds'], 4), ' trace sum', round(r['kernel_ms_sum'], 1), ' shade(rest) sum', round(d['roofline_shade']['kernel_ms_sum'], 1), ' iters', r['bounce_iterations'])" || exit 1; }
run X=1
run PTAMD_EARLY=1000000000 PTAMD_EST=64
for m in 1 2 3; do
  run PTAMD_LIB=$V/libptamd_stm$m.so
  run PTAMD_LIB=$V/libptamd_stm$m.so PTAMD_EARLY=1000000000 PTAMD_EST=64
done
