#!/usr/bin/env python3
"""bench.py — Msamples/s (pixels x spp) of the radiance integrator on MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload = BASELINE.json configs[2], the configuration the metric is quoted on: Cornell room
+ 69,564-triangle "bunny" stand-in with the specular-reflection BRDF (69,576 triangles,
45,075 BVH nodes), 1920x1080, NEE on, the reference's camera.  Scene data is synthetic
(the reference ships no assets).

A STEP is one full-frame PASS of the hot path: every pixel of the frame runs
`spp_per_pass` = 256 camera paths (2048 spp = 8 passes x 256, as SURVEY.md §8d maps config 3).
With N GPUs the frame's 8x8 tiles are dealt round-robin to the ranks (total work fixed ->
"strong" scaling); after the K timed steps the finished tiles are gathered to rank 0 with
ONE collective (RCCL over xGMI) and de-interleaved — that exchange is inside the timed
region.  value = W*H*spp_per_pass*K / seconds / 1e6, whole job, inputs resident in HBM.

Also on the JSON line:
  roofline     — for the dominant kernel (wf_trace, the traversal kernel of the pipeline):
                 achieved = algorithmic bytes per launch / average launch duration (HIP event
                 pairs on the launch stream around every launch), against the 8 TB/s HBM peak;
                 the GPU's measured streaming rate (float4 triad) is reported beside it.
                 Algorithmic bytes per sample are those of the REFERENCE's traversal (SURVEY.md
                 §8d: 40 B per node fetched + 36 B per triangle test), counted by the
                 instrumented CPU oracle on this same scene and frame.
  cpu_baseline — the CPU oracle (restatement of the reference's algorithm) timed on this
                 host's cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
W, H = 1920, 1080
SPP_PER_PASS = 256
LAT_LON = 187


def cpu_baseline(nodes, tris, ncores):
    """Oracle timed on the host: the same scene and full 1920x1080 frame, one pass, with the spp
    chosen (after a short calibration) so that the run is ~10-30 s of CPU work.
    Returns (dict, counters)."""
    import oracle_lib as O
    O.set_libm(1)
    sc = O.Scene(nodes.tobytes(), tris)
    cam = O.make_camera(W, H)
    t0 = time.time()
    sc.render(cam, O.make_params(W, H, 1, 1, window=(0, H // 2 - 32, W, H // 2 + 32)), ncores)   # 64 rows, 1 spp
    per_frame_spp = (time.time() - t0) * H / 64.0
    spp = int(max(1, min(64, round(15.0 / max(per_frame_spp, 1e-3)))))
    t0 = time.time()
    _, cnt = sc.render(cam, O.make_params(W, H, 1, spp), ncores)
    dt = time.time() - t0
    n = int(cnt[5])
    return ({"value": n / dt / 1e6, "unit": "Msamples/s", "cores": ncores, "kind": "port",
             "sample": f"same scene, full 1920x1080 frame, 1 pass x {spp} spp ({n} paths, {dt:.1f} s, "
                       f"oracle/pt_oracle.cpp on {ncores} threads)"}, cnt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--spp", type=int, default=SPP_PER_PASS, help="spp per pass (non-default values are for profiling only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch
    import ptamd
    from ptamd.dist import TileRenderer, gather_tiles

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback for the render path)")
    # one rank per GPU; BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks
    # (ranks then share devices and the gather goes through host memory)
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if backend == "nccl" and local_rank >= ndev:
        raise SystemExit(f"rank {rank}: LOCAL_RANK {local_rank} but only {ndev} GPU(s) visible")
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # ---- scene (host): every rank builds and uploads its own replica (no collective needed) ----
    t0 = time.time()
    prims = ptamd.gen_scene(1, LAT_LON)
    nodes, tris, depth = ptamd.build_bvh(prims)
    t_build = time.time() - t0
    scene = ptamd.Scene(nodes, tris, device=dev_index)
    cam = ptamd.make_camera(W, H)

    def params(first_pass, passes):
        return ptamd.default_params(passes=passes, spp_per_pass=args.spp, first_pass=first_pass, rank=rank, world=world)

    # The K timed steps are K consecutive passes (SampleIDX W..W+K-1) submitted as ONE render call,
    # exactly as PathTracer::Render runs its NUM_MULTI_SAMPLE passes: the pipeline keeps all of
    # their streams in flight, and `image += mean(pass)` happens in pass order inside (sum_passes).
    tr = TileRenderer(scene, cam, params(args.warmup, args.steps), dev)
    warm = TileRenderer(scene, cam, params(0, args.warmup), dev) if args.warmup > 0 else None

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if warm is not None:
        warm.render()
        if world > 1:   # warm the collective once as well
            gather_tiles(warm.tiles if backend == "nccl" else warm.tiles.cpu(), rank, world)
        del warm
    barrier()
    scene.render_timings(reset=True)
    scene.enable_trace_timing(16384)         # HIP event pair around every wf_trace launch, on the launch stream
    t0 = time.perf_counter()
    tr.render()
    if world > 1 and backend != "nccl":                      # rehearsal: gloo gathers host tensors
        g = gather_tiles(tr.tiles.cpu(), rank, world)
        gathered = g.to(dev) if rank == 0 else None
    else:
        gathered = gather_tiles(tr.tiles, rank, world)       # the single exchange step (RCCL over xGMI)
    if rank == 0:
        frame = tr.assemble(gathered, world)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = scene.render_timings(reset=True)

    if rank == 0:
        samples = float(W) * H * args.spp * args.steps
        value = samples / dt / 1e6
        out = {
            "metric": "Msamples/sec (pixels x spp) at 1080p on bunny scene", "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[2]: Cornell room + 69,564-tri bunny stand-in (specular reflection BRDF), "
                                   "1920x1080, NEE on; step = one full-frame pass of %d spp" % args.spp,
                       "triangles": int(tris.shape[0]), "bvh_nodes": int(nodes.shape[0]), "bvh_depth": int(depth),
                       "spp_per_pass": args.spp, "parallelism": f"tile-split x{world}, one gather",
                       "host_bvh_build_s": round(t_build, 3)},
        }
        cpu, cnt = (None, None)
        if world == 1 and not args.no_cpu_baseline:
            ncores = min(len(os.sched_getaffinity(0)), 16)      # the box's CPU share for one GPU
            cpu, cnt = cpu_baseline(nodes, tris, ncores)
        # reference-algorithm counters of this exact workload (full 1080p frame), committed by oracle/gen_counters.py
        tc = json.load(open(os.path.join(ROOT, "tests", "golden", "traversal_counters.json")))["config3_standin"]
        # the dominant kernel is the traversal kernel wf_trace: price it with the traversal terms of
        # SURVEY.md 8(d) (40 B per node fetched + 36 B per triangle test of the REFERENCE algorithm)
        bps_trav = (40.0 * tc["nodes_fetched"] + 36.0 * tc["tri_tests"]) / tc["paths"]
        bps_all = tc["bytes_per_sample_traversal"] + 24.0 / args.spp
        t_sum_ms, t_launches, t_max_ms = scene.trace_timing()
        call_samples = float(W) * H * args.spp * args.steps / world          # samples this rank's render call covered
        if t_launches > 0:
            k_ms = t_sum_ms / t_launches
            units_per_launch = call_samples / t_launches
            achieved = units_per_launch * bps_trav / (k_ms * 1e-3) / 1e9
        else:   # mode 0 (one-kernel state machine): one launch per call
            k_ms = float(np.sum(kern_ms)); units_per_launch = call_samples; t_launches = int(len(kern_ms))
            achieved = units_per_launch * bps_all / (k_ms * 1e-3) / 1e9
        traffic = None
        traffic_spp = None
        tp = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if os.path.exists(tp):
            try:
                tj = json.load(open(tp))
                # per-launch traffic of the same K-pass call (same streams in flight per launch); the PMC passes may have been
                # collected at a smaller spp_per_pass (fewer launches of the same kind) — the file says which
                if tj.get("n_gpus", 1) == world and tj.get("steps") == args.steps:
                    traffic = tj.get("hbm_bytes_per_launch")
                    traffic_spp = tj.get("spp_per_pass")
            except Exception:
                traffic = None
        try:
            triad = ptamd.triad_gbps(1 << 30, 10, dev.index or 0)       # this GPU's measured streaming rate
        except Exception:
            triad = None
        out["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                           "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_pmc_spp_per_pass": traffic_spp, "peak_measured_triad": triad,
                           "kernel": "wf_trace", "kernel_ms_avg": k_ms, "kernel_ms_max": t_max_ms, "launches_timed": int(t_launches),
                           "kernel_ms_sum": t_sum_ms, "samples_per_launch": units_per_launch,
                           "algorithmic_bytes_per_sample": bps_trav,
                           "algorithmic_bytes_per_sample_incl_shading_and_accum": bps_all,
                           "bounce_iterations": int(scene.last_iterations()),
                           "pipeline_ms_per_step": (float(np.sum(kern_ms)) / args.steps) if len(kern_ms) else None,
                           "note": "algorithmic bytes are those of the REFERENCE traversal (brute force on degenerate rays); this kernel's "
                                   "own fetches are far fewer (a 4-wide quantised tree, camera rays traced once per pass) and mostly cache "
                                   "hits, so frac > 1 means faster than a bandwidth-perfect execution of the reference's traversal, not a "
                                   "saturated HBM; the kernel is VALU-issue bound (DESIGN.md section 5)"}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        # a cheap sanity guard on the timed output (not a parity test): finite, plausible brightness
        m = float(frame.mean().item()) / args.steps
        if not (0.05 < m < 5.0) or not bool(torch.isfinite(frame).all().item()):
            raise SystemExit(f"bench output implausible (mean {m})")
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
