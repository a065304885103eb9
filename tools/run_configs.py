#!/usr/bin/env python3
"""Kernel-time Msamples/s of BASELINE.json's configs 2-5 on one GPU (reduced pass counts where noted)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import numpy as np, torch, ptamd
from ptamd.dist import TileRenderer
glass = ptamd.make_sphere((10, 6, 8), 6.0, albedo=(1, 1, 1), opacity=0.0, roughness=0.0, metallic=0.0)
cfgs = [("config2 cornell 1080p 8x64", 0, None, 1920, 1080, 8, 64, 8),
        ("config3 standin 1080p 8x256", 1, None, 1920, 1080, 8, 256, 8),
        ("config4 standin+glass depth12 1080p 8x256", 1, glass, 1920, 1080, 8, 256, 12),
        ("config5 4x standin 4K 2x512 (of 8 passes)", 2, None, 3840, 2160, 2, 512, 8)]
dev = torch.device("cuda:0")
for name, kind, sph, W, H, passes, spp, depth in cfgs:
    t0 = time.time(); nodes, tris, d = ptamd.build_bvh(ptamd.gen_scene(kind, 187)); tb = time.time() - t0
    sc = ptamd.Scene(nodes, tris, sph)
    cam = ptamd.make_camera(W, H); prm = ptamd.default_params(passes=passes, spp_per_pass=spp, max_bounce=depth)
    tr = TileRenderer(sc, cam, prm, dev)
    torch.cuda.synchronize(); t0 = time.time(); tr.render(); torch.cuda.synchronize(); dt = time.time() - t0
    ms = sc.render_timings()
    print(json.dumps({"config": name, "tris": int(tris.shape[0]), "host_bvh_s": round(tb, 3), "wall_s": round(dt, 3), "pipeline_ms": float(ms.sum()),
                      "Msamples_per_s": W * H * passes * spp / dt / 1e6, "iterations": sc.last_iterations(), "mean": float(tr.tiles.mean().item()) / passes}), flush=True)
    del tr, sc
