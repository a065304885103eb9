#!/bin/bash
# b35: the arrangement of the ranks inside a group of `world` tiles (PTAMD_TILEMAP: 0 = t % world = vertical stripes at 1080p, 1 = rotated by tile row,
# 2 = rotated by a hash of the group): per-rank times of the emulated 8- and 4-way split
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-$(dirname "$(dirname "$(readlink -f "$0")")")}"
L=gpurun_out/r03_b35.log; : > $L
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_b35_tests.log 2>&1 || { tail -30 gpurun_out/r03_b35_tests.log; exit 1; }
tail -1 gpurun_out/r03_b35_tests.log | tee -a $L
for m in 0 1 2 0 2; do
  echo "== PTAMD_TILEMAP=$m" >> $L
  PTAMD_TILEMAP=$m timeout -k 10 400 python3 tools/emulate_world.py --worlds 8,4 2>/dev/null | tail -1 | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
for w, v in d['worlds'].items():
    t = [r['seconds'] for r in v['per_rank']]
    print('world', w, 'slowest %.4f mean %.4f spread %.1f %%' % (max(t), sum(t) / len(t), 100 * (max(t) - min(t)) / (sum(t) / len(t))), ' '.join('%.4f' % x for x in t))" >> $L
done
cat $L
