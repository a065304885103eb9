# round 3, batch 12: quad_tail — a wave that is down to <= 16 rays after the queue has run dry finishes them four lanes per ray
cd $GRAFT_REPO_ROOT
PTAMD_QUAD=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_b12_tests_quad.log 2>&1 || { tail -30 gpurun_out/r03_b12_tests_quad.log; exit 1; }
tail -2 gpurun_out/r03_b12_tests_quad.log
PTAMD_QUAD=1 PTAMD_EARLY=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_b12_tests_quad_noearly.log 2>&1 || { tail -30 gpurun_out/r03_b12_tests_quad_noearly.log; exit 1; }
tail -2 gpurun_out/r03_b12_tests_quad_noearly.log
bash tools/ab.sh r03_b12 --no-tests main main:PTAMD_QUAD=1 main:PTAMD_EARLY=0 main:PTAMD_EARLY=0,PTAMD_QUAD=1
