cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rm -rf gpurun_out/pcs; mkdir -p gpurun_out/pcs
timeout -k 10 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-method stochastic --pc-sampling-unit cycles --pc-sampling-interval 1048576 --output-format csv -d gpurun_out/pcs -- python3 bench.py --config 2 --spp 16 --steps 1 --warmup 0 --no-cpu-baseline --no-probes > gpurun_out/pcs/run.log 2>&1
echo rc=$?
tail -5 gpurun_out/pcs/run.log
find gpurun_out/pcs -type f | head; du -sh gpurun_out/pcs
