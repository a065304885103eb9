# round 3, batch 8: non-temporal hint on the stream-state LOADS only / STORES only
cd $GRAFT_REPO_ROOT
bash tools/ab.sh r03_b8 --no-tests main ntld ntst main:PTAMD_EARLY=1000000000,PTAMD_EST=64 ntld:PTAMD_EARLY=1000000000,PTAMD_EST=64 ntst:PTAMD_EARLY=1000000000,PTAMD_EST=64
