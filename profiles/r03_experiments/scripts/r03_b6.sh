# round 3, batch 6: early shade — what slows the traversal kernel down (device-scope stores or the shading waves beside it), issue priority
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra --emulate-world $W --rank 0 2>>gpurun_out/r03_b6.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  seconds', round(d['emulated']['seconds'], 4), ' trace sum', round(r['kernel_ms_sum'], 1), ' shade(rest) sum', round(d['roofline_shade']['kernel_ms_sum'], 1), ' iters', r['bounce_iterations'])" || exit 1; }
W=8
run X=1
run PTAMD_EARLY=1000000000 PTAMD_EPUB=1
for p in 0 1 2 3; do run PTAMD_EARLY=1000000000 PTAMD_EST=64 PTAMD_EPRIO=$p; done
run PTAMD_EARLY=1000000000 PTAMD_EST=128 PTAMD_EPRIO=3
run PTAMD_EARLY=1000000000 PTAMD_EST=64 PTAMD_EPRIO=3 PTAMD_TR=0
W=4
run X=1
run PTAMD_EARLY=1000000000 PTAMD_EST=64 PTAMD_EPRIO=3
