# The whole judged measurement set in one call (about 11 minutes of box time): counters first (the bench lines read profiles/r02_pmc_config<C>.json),
# then tools/r02_measure_all.sh (parity suite, probes, bench lines of all configs, emulated split, rocprofv3 kernel trace).
cd $GRAFT_REPO_ROOT
for c in "2 64 8" "1 64 8" "3 64 4" "4 32 1"; do set -- $c
  bash tools/r02_pmc_config.sh $1 $2 $3 > gpurun_out/r02_pmc$1.log 2>&1 && cp gpurun_out/r02_pmc_config$1.json profiles/r02_pmc_config$1.json && echo "pmc config $1 ok"
done
bash tools/r02_measure_all.sh
