# round 3, batch 22: with 4 shards tried by default — is the queue order by ray class worth its cost in wf_shade?  (and the old default, 16 shards, for reference)
cd $GRAFT_REPO_ROOT
bash tools/ab.sh r03_b22 main main:PTAMD_CLASS=0 main:PTAMD_CLASS=0,PTAMD_HELP=16
