# round 3, batch 3: early shade (wf_shade beside the draining wf_trace) — parity, then A/B at the sizes that matter
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "early_shade or image_matches or wavefront_pipeline" > gpurun_out/r03_b3_tests.log 2>&1 || { tail -30 gpurun_out/r03_b3_tests.log; exit 1; }
tail -2 gpurun_out/r03_b3_tests.log
PTAMD_EARLY=1000000000 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_b3_tests_early.log 2>&1 || { tail -30 gpurun_out/r03_b3_tests_early.log; exit 1; }
tail -2 gpurun_out/r03_b3_tests_early.log
bash tools/ab.sh r03_b3 --no-tests main main:PTAMD_EARLY=1000000000 main:PTAMD_EARLY=4000000
