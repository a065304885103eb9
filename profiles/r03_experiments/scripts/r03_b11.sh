# round 3, batch 11: rehearsal of the N > 1 bench path on the one-GPU box (gloo, ranks share the device) — checks the new bench flow, not speed
cd $GRAFT_REPO_ROOT
for n in 2 4; do
BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2951$n bench.py --gpus $n --steps 8 --warmup 1 > gpurun_out/r03_b11_gloo$n.json 2> gpurun_out/r03_b11_gloo$n.err || { tail -20 gpurun_out/r03_b11_gloo$n.err; exit 1; }
cut -c1-600 gpurun_out/r03_b11_gloo$n.json
done
BENCH_BACKEND=gloo timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29519 bench.py --gpus 2 --steps 20 --warmup 5 > gpurun_out/r03_b11_gloo2_s20.json 2> gpurun_out/r03_b11_gloo2_s20.err || { tail -20 gpurun_out/r03_b11_gloo2_s20.err; exit 1; }
cut -c1-600 gpurun_out/r03_b11_gloo2_s20.json
