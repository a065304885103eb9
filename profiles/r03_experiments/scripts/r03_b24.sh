# round 3, batch 24: repeat of the fair-share A/B on all configs (alternating legs)
cd $GRAFT_REPO_ROOT
bash tools/ab.sh r03_b24 --no-tests main main:PTAMD_FAIR=16 main main:PTAMD_FAIR=16
