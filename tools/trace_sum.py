#!/usr/bin/env python3
"""Sum and mean of the wf_trace launch durations of one render (HIP events), early shade off — run it with and without PTAMD_TSTAT=2 to
see what the timestamped build costs.  usage: trace_sum.py [kind W H passes spp [world rank]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import torch, ptamd
from ptamd.dist import TileRenderer
a = [int(x) for x in sys.argv[1:]]
kind, W, H, passes, spp = (a + [1, 1920, 1080, 8, 64][len(a):])[:5]
world, rank = (a[5], a[6]) if len(a) > 6 else (1, 0)
nodes, tris, depth = ptamd.build_bvh(ptamd.gen_scene(kind, 187))
sc = ptamd.Scene(nodes, tris)
sc.set_early_shade(0)
sc.enable_trace_timing(8192)
tr = TileRenderer(sc, ptamd.make_camera(W, H), ptamd.default_params(passes=passes, spp_per_pass=spp, rank=rank, world=world), torch.device("cuda:0"))
for rep in range(2):
    tr.render(); torch.cuda.synchronize()
    s, n, m = sc.trace_timing()
    print("TSTAT=%s kind %d world %d: wf_trace %d launches, sum %.1f ms, mean %.1f us, max %.1f us" % (os.environ.get("PTAMD_TSTAT", "0"), kind, world, n, s, 1000 * s / max(n, 1), 1000 * m))
