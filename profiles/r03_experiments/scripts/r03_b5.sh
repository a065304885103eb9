# round 3, batch 5: early shade, workgroup size of phase 1, at the 8-way size (and 2-, 4-way)
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra --emulate-world $W --rank 0 2>>gpurun_out/r03_b5.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  seconds', round(d['emulated']['seconds'], 4), ' trace sum', round(r['kernel_ms_sum'], 1), ' shade(rest) sum', round(d['roofline_shade']['kernel_ms_sum'], 1), ' iters', r['bounce_iterations'])" || exit 1; }
for W in 8; do
  run X=1
  for t in 512 256 128 64; do run PTAMD_EARLY=1000000000 PTAMD_EST=$t; done
  run X=1
  run PTAMD_EARLY=1000000000 PTAMD_EST=256
  run PTAMD_EARLY=1000000000 PTAMD_EST=256 PTAMD_TB=1536
  run PTAMD_EARLY=1000000000 PTAMD_EST=256 PTAMD_TB=1280
done
for W in 4 2; do run X=1; run PTAMD_EARLY=1000000000 PTAMD_EST=256; done
