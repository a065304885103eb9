# round 3, batch 19: what is the drain made of?  the Cornell room alone (every ray <= 10 trips), and trip statistics with / without the class order
cd $GRAFT_REPO_ROOT
echo "== Cornell room only (12 triangles), full frame, 8 x 32 spp"
timeout -k 10 300 python3 tools/trace_timeline.py 0 1920 1080 8 32 2>/dev/null | grep -v "^  *[0-9]*:"
echo "== Cornell room only, one rank of an 8-way split, 8 x 64 spp"
PTAMD_EARLY=0 timeout -k 10 300 python3 tools/trace_timeline.py 0 1920 1080 8 64 8 0 2>/dev/null | grep -v "^  *[0-9]*:"
for c in 1 0; do
  echo "== trip statistics, PTAMD_CLASS=$c, full frame 4 x 32 spp"
  PTAMD_CLASS=$c timeout -k 10 300 python3 tools/trace_stat.py 2>/dev/null | head -12
done
