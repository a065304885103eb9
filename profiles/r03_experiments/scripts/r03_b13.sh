# round 3, batch 13: the drain of wf_trace (queue dry -> last wave out) with and without quad_tail, production code path with timestamps
cd $GRAFT_REPO_ROOT
for q in 0 1; do
  echo "== PTAMD_QUAD=$q, one rank of an 8-way split, 8 x 64 spp"
  PTAMD_EARLY=0 PTAMD_QUAD=$q timeout -k 10 300 python3 tools/trace_timeline.py 1 1920 1080 8 64 8 0 2>/dev/null | grep -v "^  *[0-9]*:" 
  echo "== PTAMD_QUAD=$q, full frame, 8 x 32 spp"
  PTAMD_EARLY=0 PTAMD_QUAD=$q timeout -k 10 300 python3 tools/trace_timeline.py 1 1920 1080 8 32 2>/dev/null | grep -v "^  *[0-9]*:"
done
