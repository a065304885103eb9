# round 3, batch 9: early shade + non-temporal state stores — which kernel gets slow (timeline); top of the tree in LDS under early shade
cd $GRAFT_REPO_ROOT
V=$PWD/pathtrace-on-cuda_amd/build
PTAMD_LIB=$V/libptamd_ntst.so PTAMD_EARLY=1000000000 PTAMD_EST=64 bash tools/kernel_timeline.sh r03_b9_tl_ntst --emulate-world 8 --rank 0 --steps 8 --warmup 1 > gpurun_out/r03_b9_timeline_ntst.log 2>&1 || { tail gpurun_out/r03_b9_timeline_ntst.log; exit 1; }
cat gpurun_out/r03_b9_timeline_ntst.log
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra --emulate-world 8 --rank 0 2>>gpurun_out/r03_b9.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  seconds', round(d['emulated']['seconds'], 4), ' trace sum', round(r['kernel_ms_sum'], 1), ' shade(rest) sum', round(d['roofline_shade']['kernel_ms_sum'], 1), ' iters', r['bounce_iterations'])" || exit 1; }
run X=1
run PTAMD_LIB=$V/libptamd_top76.so PTAMD_TOP=76
run PTAMD_EARLY=1000000000 PTAMD_EST=64
run PTAMD_LIB=$V/libptamd_top76.so PTAMD_TOP=76 PTAMD_EARLY=1000000000 PTAMD_EST=64
