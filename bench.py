#!/usr/bin/env python3
"""bench.py — Msamples/s (pixels x spp) of the radiance integrator on MI355X.

  python bench.py --gpus N --steps K --warmup W [--config C]
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`)

Workload (default --config 2) = BASELINE.json configs[2], the configuration the metric is quoted on: Cornell room
+ 69,564-triangle "bunny" stand-in with the specular-reflection BRDF, 1920x1080, NEE on, the reference's camera.
--config 1 / 3 / 4 select configs[1] (Cornell 1080p), configs[3] (+ glass sphere, depth 12) and configs[4]
(4 instanced stand-ins, 3840x2160).  Scene data is synthetic (the reference ships no assets).

A STEP is one full-frame PASS of the hot path: every pixel of the frame runs `spp_per_pass` camera paths (config 2:
256; 2048 spp = 8 passes x 256, as SURVEY.md 8d maps the config).  The K timed steps are K consecutive passes
submitted as ONE render call, exactly as PathTracer::Render runs its NUM_MULTI_SAMPLE passes.  With N GPUs the
frame's 8x8 tiles are dealt round-robin to the ranks (total work fixed -> "strong" scaling); after the K timed steps
the finished tiles are gathered to rank 0 with ONE collective (RCCL over xGMI) and de-interleaved — that exchange is
inside the timed region.  value = W*H*spp_per_pass*K / seconds / 1e6, whole job, inputs resident in HBM.

Also on the JSON line:
  roofline     — for the dominant kernel (wf_trace, the traversal kernel).  The kernel is bound by vector-ALU issue,
                 not by HBM (no dense contraction -> no MFMA; its working set is cache resident), so the roofline is a
                 VALU one: achieved = lane-operations per second = (active-lane VALU instructions per sample, from the
                 PMC passes committed under profiles/, SQ_THREAD_CYCLES_VALU) x samples per launch / the kernel's
                 average launch duration, MEASURED LIVE here with HIP event pairs on the launch stream; peak = 256 CUs
                 x 4 SIMDs x 32 lanes/clk x 2.4 GHz (the fp32 vector rate; pt_dbg_valu_rate measures what the chip really
                 issues and is reported beside it).  traffic = HBM-side bytes per launch from the L2's fabric
                 request counters (same PMC passes, per sample x samples per launch); hbm_frac prices them against
                 8 TB/s.  The figure SURVEY.md 8(d) defines — algorithmic bytes of the REFERENCE's traversal divided by
                 this kernel's time — is reported separately as vs_reference_algorithm (it is a speed-up over a
                 bandwidth-perfect execution of the reference's traversal, not a bandwidth).
  cpu_baseline — the CPU oracle (restatement of the reference's algorithm) timed on this host's cores on a FIXED
                 bounded sample of the same workload (rank 0, N=1 only).

--emulate-world W --rank R renders only rank R's share of a W-way tile split on this one GPU (n_gpus stays 1; the
line says so): the per-rank times of a split can be measured without the multi-GPU node.
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
# VALU peak in lane-operations per second: 256 CUs x 4 SIMDs x 32 lanes per clock x 2.4 GHz = the 157 TFLOP/s fp32 vector
# rate / 2.  Only plain 2-operand / fma / integer-add instructions issue at that rate (one wave64 instruction per SIMD per 2 clocks,
# and only with >= 2 waves per SIMD); v_max3 / v_cvt_f32_ubyte / packed and fp64 instructions take 4 clocks (all measured by
# pt_dbg_valu_rate: gpurun_out -> profiles/r02_valu_probe.json).  The traversal kernel's mix is about half of each, so its
# practical ceiling is ~0.7 of this peak; the fraction is quoted against the hard one.
VALU_PEAK_GLOPS = 256 * 4 * 32 * 2.4
LAT_LON = 187

# BASELINE.json configs[1..4] -> passes x spp_per_pass as SURVEY.md 8(d) maps them
CONFIGS = {
    1: dict(name="configs[1]: Cornell-box diffuse room, 1920x1080, 512 spp = 8 passes x 64, NEE on", kind=0, W=1920, H=1080, passes=8, spp=64,
            depth=8, glass=False, counters="config2_cornell", metric="Msamples/sec (pixels x spp) at 1080p on Cornell scene",
            cpu_sample=dict(window=None, spp=16)),
    2: dict(name="configs[2]: Cornell room + 69,564-tri bunny stand-in (specular reflection BRDF), 1920x1080, 2048 spp = 8 passes x 256, NEE on",
            kind=1, W=1920, H=1080, passes=8, spp=256, depth=8, glass=False, counters="config3_standin",
            metric="Msamples/sec (pixels x spp) at 1080p on bunny scene", cpu_sample=dict(window=None, spp=4)),
    3: dict(name="configs[3]: bunny stand-in + glass sphere (refraction path, max depth 12), 1920x1080, 2048 spp = 8 passes x 256",
            kind=1, W=1920, H=1080, passes=8, spp=256, depth=12, glass=True, counters="config4_glass",
            metric="Msamples/sec (pixels x spp) at 1080p on bunny + glass sphere scene", cpu_sample=dict(window=None, spp=4)),
    4: dict(name="configs[4]: 4x instanced bunny stand-ins (278,268 tris), 3840x2160, 4096 spp = 8 passes x 512",
            kind=2, W=3840, H=2160, passes=8, spp=512, depth=8, glass=False, counters="config5_4x",
            metric="Msamples/sec (pixels x spp) at 2160p on 4x instanced bunny scene", cpu_sample=dict(window=(960, 540, 2880, 1620), spp=2)),
}


def make_scene(cfg, dev_index):
    """Host BVH build + upload of one config's scene.  Returns (scene, nodes, tris, build seconds)."""
    import ptamd
    t0 = time.time()
    prims = ptamd.gen_scene(cfg["kind"], LAT_LON)
    nodes, tris, depth = ptamd.build_bvh(prims)
    t_build = time.time() - t0
    spheres = None
    if cfg["glass"]:    # analytic Sphere r=6 at (10,6,8), opacity 0, roughness 0 (pure_refractive), specular .04
        spheres = ptamd.make_sphere((10, 6, 8), 6.0, albedo=(1, 1, 1), opacity=0.0, roughness=0.0, metallic=0.0)
    scene = ptamd.Scene(nodes, tris, spheres, device=dev_index)
    scene.bvh_depth = depth
    return scene, nodes, tris, t_build


def cpu_baseline(cfg, nodes, tris, ncores):
    """Oracle timed on the host: the same scene and frame, one pass, a FIXED window and spp per config (so the sample —
    and with it the number — does not change from run to run).  Returns the cpu_baseline dict."""
    import numpy as np
    import oracle_lib as O
    O.set_libm(1)
    sph = None
    if cfg["glass"]:
        import ptamd
        sph = ptamd.make_sphere((10, 6, 8), 6.0, albedo=(1, 1, 1), opacity=0.0, roughness=0.0, metallic=0.0).reshape(1, 16)
    sc = O.Scene(nodes.tobytes(), tris, sph)
    W, H = cfg["W"], cfg["H"]
    cam = O.make_camera(W, H)
    win, spp = cfg["cpu_sample"]["window"], cfg["cpu_sample"]["spp"]
    t0 = time.time()
    _, cnt = sc.render(cam, O.make_params(W, H, 1, spp, window=win, max_bounce=cfg["depth"]), ncores)
    dt = time.time() - t0
    n = int(cnt[5])
    where = "full %dx%d frame" % (W, H) if win is None else "window %s of the %dx%d frame" % (str(tuple(win)), W, H)
    return {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": ncores, "kind": "port",
            "sample": f"same scene, {where}, 1 pass x {spp} spp ({n} paths, {dt:.1f} s, oracle/pt_oracle.cpp on {ncores} threads)"}


def load_pmc(config):
    """Per-sample PMC sums of this config's kernels (tools/pmc_summary.py -> profiles/r02_pmc_config<C>.json), or None."""
    p = os.path.join(ROOT, "profiles", "r02_pmc_config%d.json" % config)
    if not os.path.exists(p):
        return None
    try:
        return json.load(open(p))
    except Exception:
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS))
    ap.add_argument("--spp", type=int, default=None, help="spp per pass (non-default values are for profiling only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-probes", action="store_true", help="skip the triad / VALU-rate machine probes")
    ap.add_argument("--emulate-world", type=int, default=0, help="render only --rank's share of a W-way tile split on this one GPU")
    ap.add_argument("--rank", type=int, default=0)
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    W, H = cfg["W"], cfg["H"]
    spp = args.spp or cfg["spp"]

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
    if args.emulate_world and world != 1:
        raise SystemExit("--emulate-world is a single-process, single-GPU mode")
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
    scene, nodes, tris, t_build = make_scene(cfg, dev_index)
    cam = ptamd.make_camera(W, H)
    # the tile split this process renders: its real rank, or the emulated one
    split_rank, split_world = (args.rank, args.emulate_world) if args.emulate_world else (rank, world)

    def params(first_pass, passes):
        return ptamd.default_params(passes=passes, spp_per_pass=spp, first_pass=first_pass, rank=split_rank, world=split_world,
                                    max_bounce=cfg["depth"])

    tr = TileRenderer(scene, cam, params(args.warmup, args.steps), dev)
    tr.work.zero_(); tr.tiles.zero_()        # scratch and output buffers are mapped before the timed region (torch.empty leaves first touch to the render)
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
    if rank == 0 and not args.emulate_world:
        frame = tr.assemble(gathered, world)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = scene.render_timings(reset=True)

    if rank == 0:
        samples = float(W) * H * spp * args.steps           # whole job
        call_samples = samples / split_world                 # what this process's render call covered
        value = (call_samples if args.emulate_world else samples) / dt / 1e6
        out = {
            "metric": cfg["metric"], "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"] + "; step = one full-frame pass of %d spp" % spp,
                       "triangles": int(tris.shape[0]), "bvh_nodes": int(nodes.shape[0]), "bvh_depth": int(scene.bvh_depth),
                       "spp_per_pass": spp, "max_bounce": cfg["depth"], "parallelism": f"tile-split x{world}, one gather",
                       "host_bvh_build_s": round(t_build, 3)},
        }
        if args.emulate_world:
            out["emulated"] = {"world": args.emulate_world, "rank": args.rank, "seconds": dt,
                               "note": "only this rank's share of the tile split was rendered, on one GPU; value = that share's samples / its time"}
        t_sum_ms, t_launches, t_max_ms = scene.trace_timing()
        roof = {"kernel": "wf_trace", "bound": "valu", "unit": "Glane-op/s", "peak": VALU_PEAK_GLOPS,
                "achieved": None, "frac": None, "traffic": None}
        if t_launches > 0:
            k_ms = t_sum_ms / t_launches
            spl = call_samples / t_launches                      # samples' worth of rays one launch advances
            roof.update({"kernel_ms_avg": k_ms, "kernel_ms_max": t_max_ms, "launches_timed": int(t_launches), "kernel_ms_sum": t_sum_ms,
                         "samples_per_launch": spl, "bounce_iterations": int(scene.last_iterations()),
                         "pipeline_ms_per_step": (float(np.sum(kern_ms)) / args.steps) if len(kern_ms) else None})
            pmc = load_pmc(args.config)
            k = (pmc or {}).get("kernels", {}).get("wf_trace")
            if k:
                lane_ops = k["valu_lane_ops_per_sample"] * spl           # per launch
                roof["achieved"] = lane_ops / (k_ms * 1e-3) / 1e9
                roof["frac"] = roof["achieved"] / VALU_PEAK_GLOPS
                roof["valu_lane_ops_per_sample"] = k["valu_lane_ops_per_sample"]
                roof["valu_lane_utilisation"] = k.get("valu_lane_utilisation")
                roof["valu_issue_frac"] = (k["valu_insts_per_sample"] * spl * 64.0 / (k_ms * 1e-3) / 1e9) / VALU_PEAK_GLOPS
                roof["traffic"] = k["hbm_bytes_per_sample"] * spl         # HBM-side bytes per launch (fabric requests; Infinity-Cache hits included)
                roof["hbm_frac"] = roof["traffic"] / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS
                roof["l2_hit_rate"] = k.get("l2_hit_rate")
                roof["pmc_source"] = "profiles/r02_pmc_config%d.json (%s)" % (args.config, pmc.get("collected_with", ""))
            # algorithmic bytes of the REFERENCE traversal (40 B per node fetched + 36 B per triangle test, counted
            # by the instrumented oracle on this scene and frame) over this kernel's time.  Not a bandwidth: see the docstring.
            try:
                tc = json.load(open(os.path.join(ROOT, "tests", "golden", "traversal_counters.json")))[cfg["counters"]]
                bps = (40.0 * tc["nodes_fetched"] + 36.0 * tc["tri_tests"]) / tc["paths"]
                eff = spl * bps / (k_ms * 1e-3) / 1e9
                out["vs_reference_algorithm"] = {"reference_traversal_bytes_per_sample": bps, "effective_GBps": eff,
                                                 "ratio_to_hbm_peak": eff / HBM_PEAK_GBS,
                                                 "note": "speed-up over a bandwidth-perfect execution of the reference's traversal "
                                                         "(brute force on degenerate rays, camera ray re-traced per sample); not a bandwidth"}
            except (KeyError, OSError):
                pass
        if not args.no_probes:
            try:
                roof["hbm_peak_measured_triad_GBps"] = ptamd.triad_gbps(1 << 30, 10, dev.index or 0)
                r, g = ptamd.valu_rate(0, 4, 20000, dev.index or 0)
                roof["valu_peak_measured"] = {"op": "v_fma_f32, 4 waves/SIMD", "Glane_op_per_s": r * 64 / 1e9, "clock_ghz": g,
                                              "clocks_per_wave_inst_per_simd": 1024.0 * g * 1e9 / r}
            except Exception as e:      # probes never fail the bench
                roof["probe_error"] = str(e)
        out["roofline"] = roof
        if world == 1 and not args.no_cpu_baseline and not args.emulate_world:
            ncores = min(len(os.sched_getaffinity(0)), 16)      # the box's CPU share for one GPU
            out["cpu_baseline"] = cpu_baseline(cfg, nodes, tris, ncores)
        if not args.emulate_world:
            # a cheap sanity guard on the timed output (not a parity test): finite, plausible brightness
            m = float(torch.nan_to_num(frame, nan=0.0).mean().item()) / args.steps
            nan_px = int(torch.isnan(frame).any(-1).sum().item())
            if not (0.05 < m < 5.0) or bool(torch.isinf(frame).any().item()):
                raise SystemExit(f"bench output implausible (mean {m})")
            if nan_px:      # the reference's NaN pixels (0/0 in its own BxDF arithmetic, reproduced bit for bit; tests/test_gpu_parity.py)
                out["config"]["nan_pixels"] = nan_px
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
