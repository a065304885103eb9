cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests -m gpu -x -q > gpurun_out/r02_b8_tests.log 2>&1 || { tail -30 gpurun_out/r02_b8_tests.log; exit 1; }
tail -2 gpurun_out/r02_b8_tests.log
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 tools/trace_timeline.py $ARGS 2>&1 | grep -v amdgpu.ids | head -1; }
for ARGS in "1 1920 1080 8 64" "1 1920 1080 8 64 8 0"; do
  echo "#### $ARGS"
  for dt in 1000000 64 32 24 16 8; do run PTAMD_DT=$dt; done
done
