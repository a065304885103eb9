#!/usr/bin/env python3
"""Per-rank render time of a W-way tile split, measured on ONE GPU: rank r's share (tiles t with t % world == r) is
rendered alone, one rank after the other, each with the same 8 passes x 256 spp call bench.py times.  The implied
multi-GPU time is the slowest rank's (ranks run concurrently on their own GPUs; the gather of 3.1 MB/rank is not
included and is ~0.1 ms).  This is an emulation: no xGMI, no RCCL, no second device is involved.
  python3 tools/emulate_world.py [--config 2] [--worlds 1,8] [--passes 8] [--spp 256]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
sys.path.insert(0, ROOT)
import torch, ptamd
from ptamd.dist import TileRenderer


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=2)
    ap.add_argument("--worlds", default="1,8")
    ap.add_argument("--passes", type=int, default=None)
    ap.add_argument("--spp", type=int, default=None)
    ap.add_argument("--ranks", default=None, help="comma list of ranks to time (default: all)")
    a = ap.parse_args()
    import bench
    cfg = bench.CONFIGS[a.config]
    passes = a.passes or cfg["passes"]; spp = a.spp or cfg["spp"]
    scene, nodes, tris, _ = bench.make_scene(cfg, 0)
    cam = ptamd.make_camera(cfg["W"], cfg["H"])
    dev = torch.device("cuda", 0)
    out = {"config": a.config, "workload": cfg["name"], "passes": passes, "spp_per_pass": spp, "worlds": {}}
    for world in [int(x) for x in a.worlds.split(",")]:
        ranks = range(world) if a.ranks is None else [int(x) for x in a.ranks.split(",") if int(x) < world]
        per = []
        for r in ranks:
            prm = ptamd.default_params(passes=passes, spp_per_pass=spp, first_pass=1, rank=r, world=world, max_bounce=cfg["depth"])
            warm = TileRenderer(scene, cam, ptamd.default_params(passes=1, spp_per_pass=spp, first_pass=0, rank=r, world=world, max_bounce=cfg["depth"]), dev)
            warm.render(); torch.cuda.synchronize(); del warm
            tr = TileRenderer(scene, cam, prm, dev)
            torch.cuda.synchronize(); t0 = time.perf_counter(); tr.render(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
            per.append({"rank": r, "seconds": dt, "iterations": scene.last_iterations()})
            print("world %d rank %d: %.4f s, %d iterations" % (world, r, dt, scene.last_iterations()), flush=True)
            del tr
        samples = float(cfg["W"]) * cfg["H"] * passes * spp
        slow = max(p["seconds"] for p in per)
        out["worlds"][str(world)] = {"per_rank": per, "slowest_s": slow, "implied_Msamples_per_s": samples / slow / 1e6}
    w = out["worlds"]
    if "1" in w:
        for k, v in w.items():
            v["implied_speedup_vs_1"] = w["1"]["slowest_s"] / v["slowest_s"]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
