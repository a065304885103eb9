# round 3, batch 4: where does an iteration of an 8-way rank go with and without early shade (kernel timeline), and the node step without the 4-way sort
cd $GRAFT_REPO_ROOT
bash tools/kernel_timeline.sh r03_b4_tl_plain --emulate-world 8 --rank 0 --steps 8 --warmup 1 > gpurun_out/r03_b4_timeline_plain.log 2>&1 || { tail gpurun_out/r03_b4_timeline_plain.log; exit 1; }
PTAMD_EARLY=1000000000 bash tools/kernel_timeline.sh r03_b4_tl_early --emulate-world 8 --rank 0 --steps 8 --warmup 1 > gpurun_out/r03_b4_timeline_early.log 2>&1 || { tail gpurun_out/r03_b4_timeline_early.log; exit 1; }
cat gpurun_out/r03_b4_timeline_plain.log gpurun_out/r03_b4_timeline_early.log
bash tools/ab.sh r03_b4 main nosort
