# round 3, batch 25: guided chunks sized by what the wave expects to be left at its NEXT grab (head movement between its last two grabs) — PTAMD_GS=-9 is the old rule
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "image_matches or early or tile_split or config" > gpurun_out/r03_b25_tests.log 2>&1 || { tail -30 gpurun_out/r03_b25_tests.log; exit 1; }
tail -2 gpurun_out/r03_b25_tests.log
for g in 9 -9; do
  echo "== PTAMD_GS=$g: full frame; 8-way rank (no early shade)"
  PTAMD_GS=$g timeout -k 10 300 python3 tools/trace_timeline.py 1 1920 1080 8 32 2>/dev/null | head -1
  PTAMD_EARLY=0 PTAMD_GS=$g timeout -k 10 300 python3 tools/trace_timeline.py 1 1920 1080 8 64 8 0 2>/dev/null | head -1
done
bash tools/ab.sh r03_b25 --no-tests main main:PTAMD_GS=-9 main:PTAMD_GS=8 main:PTAMD_GS=10
