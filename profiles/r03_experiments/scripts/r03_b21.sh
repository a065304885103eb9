# round 3, batch 21: a wave's trips get slower after the queue is dry and waves leave one after the other (oldest first): chunk size / guided shrink / refill threshold at the 8-way size
cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env PTAMD_HELP=4 PTAMD_EARLY=0 "$@" timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-probes --no-all-in-flight-extra --emulate-world 8 --rank 0 --config $CFG 2>>gpurun_out/r03_b21.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(round(d['value'], 1), 'Msamples/s  seconds', round(d['emulated']['seconds'], 4), ' trace sum', round(r['kernel_ms_sum'], 1), ' iters', r['bounce_iterations'])" || exit 1; }
for CFG in 2 1; do
  echo "#### config $CFG"
  run X=1
  for cs in 13 14 15 16 17; do run PTAMD_CS=$cs; done
  for gs in 7 8 10 11; do run PTAMD_GS=$gs; done
  for rf in 8 16 32 48; do run PTAMD_RF=$rf; done
  run PTAMD_TB=1536
  run PTAMD_TB=1280
  run X=1
done
