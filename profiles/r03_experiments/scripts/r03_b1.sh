# round 3, batch 1: in-workgroup lobe sort of wf_shade (PTAMD_SORT=1), with both shading schedules and two workgroup sizes
cd $GRAFT_REPO_ROOT
PTAMD_SORT=1 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_b1_tests_sort.log 2>&1 || { tail -30 gpurun_out/r03_b1_tests_sort.log; exit 1; }
tail -2 gpurun_out/r03_b1_tests_sort.log
PTAMD_SORT=1 PTAMD_TR=0 timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/r03_b1_tests_sort_tr0.log 2>&1 || { tail -30 gpurun_out/r03_b1_tests_sort_tr0.log; exit 1; }
tail -2 gpurun_out/r03_b1_tests_sort_tr0.log
bash tools/ab.sh r03_b1 --no-tests main main:PTAMD_SORT=1 main:PTAMD_TR=0 main:PTAMD_TR=0,PTAMD_SORT=1 main:PTAMD_TR=0,PTAMD_SORT=1,PTAMD_ST=1024 main:PTAMD_SORT=1,PTAMD_ST=1024
