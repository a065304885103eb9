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
submitted in render calls of AT MOST the config's own pass count (configs[2]: 8), back to back — the shape PathTracer::Render
has: its NUM_MULTI_SAMPLE passes are all a render call can keep in flight.  --steps 20 is therefore 8 + 8 + 4 (the line says
so: config.render_calls); the figure for all K passes submitted as ONE call (more streams in flight than the config can have)
is measured after the timed region and reported separately as all_in_flight, never as value.  With N GPUs the
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
  roofline_shade — the other half of every step (wf_shade): HBM-side bytes per launch (same PMC passes) over its average launch
                 duration measured live (HIP events after each wf_trace and after the wf_shade that follows it) against 8 TB/s and
                 against the triad rate measured on this device, plus its VALU lane utilisation and lane-op fraction.
  cpu_baseline — the CPU oracle (restatement of the reference's algorithm) timed on this host's cores on a FIXED
                 bounded sample of the same workload (rank 0, N=1 only), on all the box's threads (<= 16) and on ONE thread.

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
    out = {"value": n / dt / 1e6, "unit": "Msamples/s", "cores": ncores, "kind": "port",
           "sample": f"same scene, {where}, 1 pass x {spp} spp ({n} paths, {dt:.1f} s, oracle/pt_oracle.cpp on {ncores} threads)"}
    # the 1-thread figure beside the all-core one, on a sixteenth of that sample (a fixed window in the
    # middle of the frame: mesh and walls both in view) so that it takes about as long
    x0, y0, x1, y1 = win if win is not None else (0, 0, W, H)
    w1 = (x0 + (x1 - x0) * 3 // 8, y0 + (y1 - y0) * 3 // 8, x0 + (x1 - x0) * 5 // 8, y0 + (y1 - y0) * 5 // 8)
    t0 = time.time()
    _, cnt1 = sc.render(cam, O.make_params(W, H, 1, spp, window=w1, max_bounce=cfg["depth"]), 1)
    dt1 = time.time() - t0
    n1 = int(cnt1[5])
    out["one_thread"] = {"value": n1 / dt1 / 1e6, "unit": "Msamples/s", "cores": 1,
                         "sample": f"window {w1} of the {W}x{H} frame, 1 pass x {spp} spp ({n1} paths, {dt1:.1f} s)"}
    return out


def render_calls(steps, passes_per_call, first_pass=0):
    """The K timed steps as render calls of at most the config's own pass count (PathTracer::Render keeps NUM_MULTI_SAMPLE passes in
    flight, no more): [(first pass index, passes)], e.g. 20 steps of an 8-pass config -> 8 + 8 + 4."""
    calls, done = [], 0
    while done < steps:
        n = min(passes_per_call, steps - done)
        calls.append((first_pass + done, n))
        done += n
    return calls


def load_pmc(config):
    """Per-sample PMC sums of this config's kernels (tools/pmc_summary.py -> profiles/r02_pmc_config<C>.json), or None."""
    for rnd in ("r03", "r02"):
        p = os.path.join(ROOT, "profiles", "%s_pmc_config%d.json" % (rnd, config))
        if os.path.exists(p):
            try:
                d = json.load(open(p))
                d["_path"] = "profiles/" + os.path.basename(p)
                return d
            except Exception:
                pass
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
    ap.add_argument("--all-in-flight", action="store_true", help="submit all --steps passes as ONE render call (more passes in flight than the config has; not the config's shape)")
    ap.add_argument("--no-all-in-flight-extra", action="store_true", help="skip the extra all-in-flight measurement after the timed region")
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

    # The K timed steps go out in render calls of at most the config's own pass count (PathTracer::Render keeps NUM_MULTI_SAMPLE
    # passes in flight, no more): --steps 20 on configs[2] = 8 + 8 + 4.  One work buffer serves all calls (they run one after the
    # other); every call has its own tile buffer, summed on the device into one before the single gather.
    calls = render_calls(args.steps, args.steps if args.all_in_flight else cfg["passes"], args.warmup)
    big = max(n for _, n in calls)
    n_warm = min(args.warmup, cfg["passes"])
    shared_work = torch.zeros(ptamd.work_bytes(cam, params(0, max(big, n_warm))) // 4, dtype=torch.float32, device=dev)
    trs = [TileRenderer(scene, cam, params(fp, n), dev, work=shared_work) for fp, n in calls]
    for t in trs:
        t.tiles.zero_()                      # scratch and output buffers are mapped before the timed region (torch.empty leaves first touch to the render)
    warm = TileRenderer(scene, cam, params(0, n_warm), dev, work=shared_work) if args.warmup > 0 else None

    def barrier():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    if warm is not None:
        # W untimed warm-up steps, in calls of at most the config's pass count like the timed ones
        left = args.warmup
        while left > 0:
            warm.render()
            left -= n_warm
        if world > 1:   # warm the collective once as well
            gather_tiles(warm.tiles if backend == "nccl" else warm.tiles.cpu(), rank, world)
        del warm
    barrier()
    scene.render_timings(reset=True)
    scene.enable_trace_timing(16384)         # HIP events around every wf_trace and wf_shade launch, on the launch stream
    t_sum_ms = t_max_ms = s_sum_ms = s_max_ms = 0.0
    t_launches = s_launches = iters = 0
    t0 = time.perf_counter()
    for i, t in enumerate(trs):
        t.render()
        if i > 0:
            trs[0].tiles.add_(t.tiles)
        # per-launch kernel times of this call (reading the events blocks only on work that has already drained)
        a, n, m = scene.trace_timing(); t_sum_ms += a; t_launches += n; t_max_ms = max(t_max_ms, m)
        a, n, m = scene.shade_timing(); s_sum_ms += a; s_launches += n; s_max_ms = max(s_max_ms, m)
        iters += int(scene.last_iterations())
    tiles = trs[0].tiles
    if world > 1 and backend != "nccl":                      # rehearsal: gloo gathers host tensors
        g = gather_tiles(tiles.cpu(), rank, world)
        gathered = g.to(dev) if rank == 0 else None
    else:
        gathered = gather_tiles(tiles, rank, world)          # the single exchange step (RCCL over xGMI)
    if rank == 0 and not args.emulate_world:
        frame = trs[0].assemble(gathered, world)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    kern_ms = scene.render_timings(reset=True)

    if rank == 0:
        samples = float(W) * H * spp * args.steps           # whole job
        call_samples = samples / split_world                 # what this process's render calls covered
        value = (call_samples if args.emulate_world else samples) / dt / 1e6
        out = {
            "metric": cfg["metric"], "value": value, "unit": "Msamples/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": cfg["name"] + "; step = one full-frame pass of %d spp" % spp,
                       "render_calls": [n for _, n in calls], "passes_in_flight_max": big,
                       "triangles": int(tris.shape[0]), "bvh_nodes": int(nodes.shape[0]), "bvh_depth": int(scene.bvh_depth),
                       "spp_per_pass": spp, "max_bounce": cfg["depth"], "parallelism": f"tile-split x{world}, one gather",
                       "host_bvh_build_s": round(t_build, 3)},
        }
        if args.emulate_world:
            out["emulated"] = {"world": args.emulate_world, "rank": args.rank, "seconds": dt,
                               "note": "only this rank's share of the tile split was rendered, on one GPU; value = that share's samples / its time"}
        roof = {"kernel": "wf_trace", "bound": "valu", "unit": "Glane-op/s", "peak": VALU_PEAK_GLOPS,
                "achieved": None, "frac": None, "traffic": None}
        roof_s = {"kernel": "wf_shade", "bound": "hbm", "unit": "GB/s", "peak": HBM_PEAK_GBS, "achieved": None, "frac": None, "traffic": None}
        pmc = load_pmc(args.config)
        pmc_src = None
        if pmc:
            pmc_src = "%s (%s)" % (pmc.get("_path"), pmc.get("collected_with", ""))
        if t_launches > 0:
            k_ms = t_sum_ms / t_launches
            spl = call_samples / t_launches                      # samples' worth of rays one launch advances
            roof.update({"kernel_ms_avg": k_ms, "kernel_ms_max": t_max_ms, "launches_timed": int(t_launches), "kernel_ms_sum": t_sum_ms,
                         "samples_per_launch": spl, "bounce_iterations": iters,
                         "pipeline_ms_per_step": (float(np.sum(kern_ms)) / args.steps) if len(kern_ms) else None})
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
                roof["pmc_source"] = pmc_src
                # the counters are per-sample figures from a separate profiled run; say what that run was next to what this one is
                roof["pmc_samples_profiled"] = pmc.get("samples_profiled"); roof["pmc_note"] = pmc.get("note")
                roof["live"] = {"spp_per_pass": spp, "steps": args.steps}
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
        if s_launches > 0:
            ks_ms = s_sum_ms / s_launches
            spl_s = call_samples / s_launches
            roof_s.update({"kernel_ms_avg": ks_ms, "kernel_ms_max": s_max_ms, "launches_timed": int(s_launches), "kernel_ms_sum": s_sum_ms,
                           "samples_per_launch": spl_s})
            # every wf_shade* kernel of the profiled run (the schedule variants are separate kernels), per sample
            ks = [v for n, v in (pmc or {}).get("kernels", {}).items() if n.startswith("wf_shade") and "hbm_bytes_per_sample" in v]
            if ks:
                bps = sum(v["hbm_bytes_per_sample"] for v in ks)
                lops = sum(v.get("valu_lane_ops_per_sample", 0.0) for v in ks)
                insts = sum(v.get("valu_insts_per_sample", 0.0) for v in ks)
                roof_s["traffic"] = bps * spl_s                              # HBM-side bytes per launch
                roof_s["achieved"] = roof_s["traffic"] / (ks_ms * 1e-3) / 1e9
                roof_s["frac"] = roof_s["achieved"] / HBM_PEAK_GBS
                roof_s["hbm_bytes_per_sample"] = bps
                roof_s["valu_lane_ops_per_sample"] = lops
                roof_s["valu_lane_utilisation"] = (lops / (64.0 * insts)) if insts else None
                roof_s["valu_frac"] = lops * spl_s / (ks_ms * 1e-3) / 1e9 / VALU_PEAK_GLOPS
                roof_s["pmc_source"] = pmc_src
        if not args.no_probes:
            try:
                roof["hbm_peak_measured_triad_GBps"] = ptamd.triad_gbps(1 << 30, 10, dev.index or 0)
                r, g = ptamd.valu_rate(0, 4, 20000, dev.index or 0)
                roof["valu_peak_measured"] = {"op": "v_fma_f32, 4 waves/SIMD", "Glane_op_per_s": r * 64 / 1e9, "clock_ghz": g,
                                              "clocks_per_wave_inst_per_simd": 1024.0 * g * 1e9 / r}
                if roof.get("achieved"):
                    roof["frac_of_measured_peak"] = roof["achieved"] / (r * 64 / 1e9)
                if roof_s.get("achieved"):
                    roof_s["hbm_peak_measured_triad_GBps"] = roof["hbm_peak_measured_triad_GBps"]
                    roof_s["frac_of_measured_triad"] = roof_s["achieved"] / roof["hbm_peak_measured_triad_GBps"]
            except Exception as e:      # probes never fail the bench
                roof["probe_error"] = str(e)
        out["roofline"] = roof
        out["roofline_shade"] = roof_s
        if world == 1 and len(calls) > 1 and not args.no_all_in_flight_extra and not args.emulate_world:
            # what the same K passes take when they are all in flight together (the previous rounds' bench line); NOT the config's shape
            one = TileRenderer(scene, cam, params(args.warmup, args.steps), dev)
            one.work.zero_(); one.tiles.zero_()
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            one.render()
            torch.cuda.synchronize(dev)
            d1 = time.perf_counter() - t1
            out["all_in_flight"] = {"value": samples / d1 / 1e6, "unit": "Msamples/s", "passes_in_flight": args.steps, "ms_per_step": d1 * 1e3 / args.steps,
                                    "note": "all %d passes submitted as one render call: more streams in flight than the config's %d passes allow" % (args.steps, cfg["passes"])}
            del one
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
