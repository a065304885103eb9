# round 3, batch 18: queue order by ray class — rays whose segment misses the box around the scene's small triangles are short and go LAST
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests -m gpu -x -q > gpurun_out/r03_b18_tests.log 2>&1 || { tail -30 gpurun_out/r03_b18_tests.log; exit 1; }
tail -2 gpurun_out/r03_b18_tests.log
for c in 1 0; do
  echo "== PTAMD_CLASS=$c, full frame, 8 x 32 spp"
  PTAMD_CLASS=$c timeout -k 10 300 python3 tools/trace_timeline.py 1 1920 1080 8 32 2>/dev/null | grep -v "^  *[0-9]*:"
  echo "== PTAMD_CLASS=$c, one rank of an 8-way split, 8 x 64 spp, no early shade"
  PTAMD_EARLY=0 PTAMD_CLASS=$c timeout -k 10 300 python3 tools/trace_timeline.py 1 1920 1080 8 64 8 0 2>/dev/null | grep -v "^  *[0-9]*:"
done
bash tools/ab.sh r03_b18 --no-tests main main:PTAMD_CLASS=0
