cd $GRAFT_REPO_ROOT
bash tools/r02_pmc_config.sh 3 32 4 2>&1 | tail -3
bash tools/r02_pmc_config.sh 1 64 8 2>&1 | tail -3
