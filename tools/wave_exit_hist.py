#!/usr/bin/env python3
"""When do the waves of wf_trace leave?  Histogram of wave lifetimes (32-us bins, all launches of one render pooled; production code path with
timestamps, PTAMD_TSTAT=2) next to the launch timeline.  usage: wave_exit_hist.py [kind W H passes spp [world rank]]"""
import ctypes as C, os, sys
os.environ["PTAMD_TSTAT"] = "2"
os.environ["PTAMD_TPOOL"] = "1"      # the pooled histograms: their atomics lengthen the tail (see r03_b27.log) — shapes only
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import numpy as np, torch, ptamd
from ptamd.dist import TileRenderer
a = [int(x) for x in sys.argv[1:]]
kind, W, H, passes, spp = (a + [0, 1920, 1080, 8, 64][len(a):])[:5]
world, rank = (a[5], a[6]) if len(a) > 6 else (1, 0)
nodes, tris, depth = ptamd.build_bvh(ptamd.gen_scene(kind, 187))
sc = ptamd.Scene(nodes, tris)
tr = TileRenderer(sc, ptamd.make_camera(W, H), ptamd.default_params(passes=passes, spp_per_pass=spp, rank=rank, world=world), torch.device("cuda:0"))
tr.render(); torch.cuda.synchronize()
it = sc.last_iterations()
tl = sc.trace_timeline(min(it, 2700)).astype(np.float64)
ok = (tl[:, 0] != 0) & (tl[:, 2] != 0)
dur = (tl[ok, 2] - tl[ok, 0]) / 100.0
exh = np.where(tl[ok, 1] != 0, (tl[ok, 1] - tl[ok, 0]) / 100.0, dur)
print("%d launches; first half: duration %.1f us, queue dry at %.1f us" % (len(dur), dur[:len(dur) // 2].mean(), exh[:len(dur) // 2].mean()))
h = np.zeros(32, np.int64)
ptamd._check(ptamd.lib().pt_dbg_trace_timeline(sc._h, ptamd._ptr(h), 0), "pt_dbg_trace_timeline")
tot = h.sum()
print("wave lifetimes (first wave start of its launch -> this wave's exit), 32-us bins, %% of %d waves:" % tot)
print("  " + " ".join("%d-%d:%.1f" % (32 * i, 32 * i + 32, 100.0 * h[i] / tot) for i in range(32) if h[i]))
h64 = np.zeros(64, np.int64); ptamd._check(ptamd.lib().pt_dbg_trace_timeline(sc._h, ptamd._ptr(h64), -3000), "pt_dbg_trace_timeline")
h32 = np.zeros(32, np.int64); ptamd._check(ptamd.lib().pt_dbg_trace_timeline(sc._h, ptamd._ptr(h32), -3001), "pt_dbg_trace_timeline")
c = sc.counters().astype(np.float64)
print("per wave (all launches pooled): %.1f rays, %.1f trips, of them %.1f after the wave found the queue dry" % (c[7] / tot, c[0] / tot, c[2] / tot))
print("trips per wave, bins of 4, %% of waves: " + " ".join("%d:%.1f" % (4 * i, 100.0 * h64[i] / tot) for i in range(64) if h64[i] * 200 > tot))
print("trips after queue-dry per wave, bins of 2, %% of waves: " + " ".join("%d:%.1f" % (2 * i, 100.0 * h32[i] / tot) for i in range(32) if h32[i] * 200 > tot))
