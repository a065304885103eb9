#!/usr/bin/env python3
"""Trip statistics of wf_trace (diagnostic build, PTAMD_TSTAT=1): how full the node / triangle trips are."""
import os, sys
os.environ.setdefault("PTAMD_TSTAT", "1")      # 3: trip counters + section clocks without the per-step histogram atomics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import numpy as np, torch, ptamd
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H, passes, spp = (int(x) for x in (sys.argv[2:6] if len(sys.argv) > 5 else (1920, 1080, 2, 32)))
prims = ptamd.gen_scene(kind, 187)
nodes, tris, depth = ptamd.build_bvh(prims)
sc = ptamd.Scene(nodes, tris)
cam = ptamd.make_camera(W, H); prm = ptamd.default_params(passes=passes, spp_per_pass=spp)
sc.render(cam, prm); torch.cuda.synchronize()
nT, nL, tT, tL, rf, rfL, noRay, rays = (float(x) for x in sc.counters())
trips = nT + tT
print("rays %.4g  trips/ray: node %.2f (lanes served/trip %.1f)  tri %.2f (lanes/trip %.1f)" % (rays, nL / rays, nL / nT, tL / rays, tL / tT))
print("wave trips %.4g: node %.1f%% tri %.1f%%;  lanes without a ray per trip %.1f;  useful lane-trips / (64 x trips) = %.3f" %
      (trips, 100 * nT / trips, 100 * tT / trips, noRay / trips, (nL + tL) / (64 * trips)))
print("refills %.4g, %.1f rays each; trips per refill %.1f" % (rf, rfL / rf, trips / rf))
it = sc.last_iterations()
tl = sc.trace_timeline(min(it, 2700)).astype(np.float64)
ok = (tl[:, 0] != 0) & (tl[:, 2] != 0)
tl = tl[ok]
dur = (tl[:, 2] - tl[:, 0]) / 100.0                       # microseconds
exh = np.where(tl[:, 1] != 0, (tl[:, 1] - tl[:, 0]) / 100.0, dur)
print("launches %d: mean duration %.1f us (first wave start -> last wave exit); queue first seen empty after %.1f us; drain after that %.1f us" %
      (len(tl), dur.mean(), exh.mean(), (dur - exh).mean()))
for lo, hi in ((0, 50), (len(tl) // 4, len(tl) // 4 + 50), (len(tl) // 2, len(tl) // 2 + 50), (3 * len(tl) // 4, 3 * len(tl) // 4 + 50)):
    print("  launches %4d-%4d: duration %.1f us, queue empty at %.1f us" % (lo, hi, dur[lo:hi].mean(), exh[lo:hi].mean()))
import ctypes as C
h = np.zeros(32, np.int64)
ptamd.lib().pt_dbg_trace_timeline(sc._h, h.ctypes.data_as(C.c_void_p), 0)
tot = max(1, int(h.sum()))
print("wave lifetimes, 32-us bins (all launches pooled), % of waves:", " ".join("%.1f" % (100.0 * x / tot) for x in h[:24]))
hs = np.zeros(64, np.int64)
ptamd.lib().pt_dbg_trace_timeline(sc._h, hs.ctypes.data_as(C.c_void_p), -3000)
tot = max(1, int(hs.sum())); cum = np.cumsum(hs[::-1])[::-1] / tot
print("node steps per ray, fraction of rays with >= N steps:", " ".join("%d:%.2e" % (4 * k, cum[k]) for k in (0, 2, 4, 6, 8, 12, 16, 24, 32, 40, 48, 63)))
hd = np.zeros(32, np.int64)
ptamd.lib().pt_dbg_trace_timeline(sc._h, hd.ctypes.data_as(C.c_void_p), -3001)
tot = max(1, int(hd.sum())); cum = np.cumsum(hd[::-1])[::-1] / tot
print("stack depth after a node step: mean %.2f; fraction of node steps with depth >= N:" % (float((hd * np.arange(32)).sum()) / tot), " ".join("%d:%.2e" % (k, cum[k]) for k in (1, 2, 4, 6, 8, 10, 12, 14, 16, 20)))
hc = np.zeros(5, np.int64)
ptamd.lib().pt_dbg_trace_timeline(sc._h, hc.ctypes.data_as(C.c_void_p), -3002)
tot = max(1, int(hc.sum()))
print("wave clocks by section of the loop (counting build): refill %.1f%%  vote+budget %.1f%%  node step %.1f%%  triangle step %.1f%%  ray epilogue %.1f%%;  per trip: node %.0f clk, triangle %.0f clk, per refill %.0f clk" %
      tuple([100.0 * x / tot for x in hc] + [hc[2] / max(nT, 1), hc[3] / max(tT, 1), hc[0] / max(rf, 1)]))
