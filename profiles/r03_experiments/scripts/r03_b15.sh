# round 3, batch 15: wf_shade takes stream idx for list position idx while every stream is alive (one dependent fetch level less)
cd $GRAFT_REPO_ROOT
bash tools/ab.sh r03_b15 main nodense
