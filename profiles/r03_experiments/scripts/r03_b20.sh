# round 3, batch 20: the drain is there with 12 triangles too (every ray <= 10 trips) — is it the 16 returning atomics every wave spends on finding the queue dry?
cd $GRAFT_REPO_ROOT
for h in 16 4 2 1; do
  echo "== PTAMD_HELP=$h: Cornell room only, one rank of an 8-way split; bunny scene full frame; bunny scene 8-way rank"
  PTAMD_EARLY=0 PTAMD_HELP=$h timeout -k 10 300 python3 tools/trace_timeline.py 0 1920 1080 8 64 8 0 2>/dev/null | head -1
  PTAMD_HELP=$h timeout -k 10 300 python3 tools/trace_timeline.py 1 1920 1080 8 32 2>/dev/null | head -1
  PTAMD_EARLY=0 PTAMD_HELP=$h timeout -k 10 300 python3 tools/trace_timeline.py 1 1920 1080 8 64 8 0 2>/dev/null | head -1
done
PTAMD_HELP=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_b20_tests_help1.log 2>&1 || { tail -30 gpurun_out/r03_b20_tests_help1.log; exit 1; }
tail -2 gpurun_out/r03_b20_tests_help1.log
bash tools/ab.sh r03_b20 --no-tests main main:PTAMD_HELP=4 main:PTAMD_HELP=2 main:PTAMD_HELP=1
