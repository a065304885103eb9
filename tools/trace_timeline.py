#!/usr/bin/env python3
"""Launch timeline of wf_trace with the production code path (PTAMD_TSTAT=2): per launch, when the ray queue ran dry and
when the last wave left.  usage: trace_timeline.py [kind W H passes spp [world rank]]"""
import os, sys, time
os.environ["PTAMD_TSTAT"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import numpy as np, torch, ptamd
from ptamd.dist import TileRenderer
a = [int(x) for x in sys.argv[1:] if not x.startswith('--')]
kind, W, H, passes, spp = (a + [1, 1920, 1080, 8, 32][len(a):])[:5]
world, rank = (a[5], a[6]) if len(a) > 6 else (1, 0)
nodes, tris, depth = ptamd.build_bvh(ptamd.gen_scene(kind, 187))
sc = ptamd.Scene(nodes, tris)
cam = ptamd.make_camera(W, H)
dev = torch.device("cuda:0")
tr = TileRenderer(sc, cam, ptamd.default_params(passes=passes, spp_per_pass=spp, rank=rank, world=world), dev)
tr.render(); torch.cuda.synchronize()
t0 = time.perf_counter(); tr.render(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
it = sc.last_iterations()
tl = sc.trace_timeline(min(it, 2700)).astype(np.float64)
ok = (tl[:, 0] != 0) & (tl[:, 2] != 0)
tl = tl[ok]
dur = (tl[:, 2] - tl[:, 0]) / 100.0                       # microseconds
exh = np.where(tl[:, 1] != 0, (tl[:, 1] - tl[:, 0]) / 100.0, dur)
print("render %.3f s, %d iterations, %.1f Msamples/s (timeline build); wf_trace: sum %.1f ms, mean %.1f us; queue dry after %.1f us on average; drain %.1f us = %.1f%% of trace time"
      % (dt, it, W * H * passes * spp / world / dt / 1e6, dur.sum() / 1e3, dur.mean(), exh.mean(), (dur - exh).mean(), 100 * (dur - exh).sum() / dur.sum()))
nr = sc.trace_launch_rays(min(it, 2700)).astype(np.float64)[ok]
if "--curve" in sys.argv or True:
    step = max(1, len(tl) // 24)
    print("  launch: rays, duration us, queue dry us")
    for i in range(0, len(tl), step):
        print("  %5d: %10d %8.1f %8.1f" % (i, nr[i], dur[i], exh[i]))
n = len(tl)
for lo, hi in ((0, n // 8), (n // 8, n // 4), (n // 4, n // 2), (n // 2, 3 * n // 4), (3 * n // 4, n)):
    print("  launches %4d-%4d: duration %8.1f us, queue dry at %8.1f us, drain %6.1f us" % (lo, hi, dur[lo:hi].mean(), exh[lo:hi].mean(), (dur - exh)[lo:hi].mean()))
