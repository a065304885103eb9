# round 3, batch 7: non-temporal loads / stores for the stream state (does shade's state traffic push the tree out of the L2s?), alone and with early shade
cd $GRAFT_REPO_ROOT
V=$PWD/pathtrace-on-cuda_amd/build
PTAMD_LIB=$V/libptamd_nt.so timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_b7_tests_nt.log 2>&1 || { tail -30 gpurun_out/r03_b7_tests_nt.log; exit 1; }
tail -2 gpurun_out/r03_b7_tests_nt.log
bash tools/ab.sh r03_b7 --no-tests main nt main:PTAMD_EARLY=1000000000,PTAMD_EST=64 nt:PTAMD_EARLY=1000000000,PTAMD_EST=64
