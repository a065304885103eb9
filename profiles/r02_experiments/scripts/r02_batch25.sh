cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" timeout -k 10 120 python3 tools/trace_timeline.py $ARGS 2>&1 | grep -v amdgpu.ids | grep "render\|launches  *[0-9]* *- *[0-9]*:" | head -3; }
for ARGS in "1 1920 1080 8 64 8 0" "1 1920 1080 8 64"; do
  echo "#### $ARGS"
  run A=1
  for cs in 14 15 16 17; do run PTAMD_CS=$cs; done
  run PTAMD_CS=16 PTAMD_GS=12
done
