#!/usr/bin/env python3
"""How much of a wf_trace launch is waiting for a few late waves?  The timestamped build (PTAMD_TSTAT=2) keeps the latest wave exit of
each of 64 stripes of workgroups per launch: the launch ends at the maximum, the typical stripe at the median — the difference is what
the launch waits for its stragglers.  usage: straggler_cost.py [kind W H passes spp [world rank]]"""
import os, sys
os.environ["PTAMD_TSTAT"] = "2"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import numpy as np, torch, ptamd
from ptamd.dist import TileRenderer
a = [int(x) for x in sys.argv[1:]]
kind, W, H, passes, spp = (a + [1, 1920, 1080, 8, 256][len(a):])[:5]
world, rank = (a[5], a[6]) if len(a) > 6 else (1, 0)
nodes, tris, depth = ptamd.build_bvh(ptamd.gen_scene(kind, 187))
sc = ptamd.Scene(nodes, tris)
sc.set_early_shade(0); sc.set_drain_threshold(0)
tr = TileRenderer(sc, ptamd.make_camera(W, H), ptamd.default_params(passes=passes, spp_per_pass=spp, rank=rank, world=world), torch.device("cuda:0"))
tr.render(); torch.cuda.synchronize()
it = min(sc.last_iterations(), 2700)
raw = np.zeros(2700 * 64 * 3, np.int64); ptamd._check(ptamd.lib().pt_dbg_trace_timeline(sc._h, ptamd._ptr(raw), -3005), "pt_dbg_trace_timeline")
raw = raw.reshape(2700, 64, 3)[:it]
nr = sc.trace_launch_rays(it).astype(np.float64)
rows = []
for l in range(it):
    st, dr, en = raw[l, :, 0], raw[l, :, 1], raw[l, :, 2]
    ok = en != 0
    if ok.sum() < 8: continue
    t0 = (~st[ok]).min()
    ends = np.sort((en[ok] - t0) / 100.0)
    dry = ((~dr[ok & (dr != 0)]).min() - t0) / 100.0 if (ok & (dr != 0)).any() else np.nan
    rows.append((l, nr[l], ends[-1], np.median(ends), ends[-2], ends[int(0.9 * (len(ends) - 1))], dry))
R = np.array(rows)
print("%d launches with >= 8 stripes; columns: rays | launch end | median / 90 %% / second-latest stripe end | queue dry (us)" % len(R))
step = max(1, len(R) // 30)
for r in R[::step]: print("  %5d: %9d | %7.1f | %7.1f %7.1f %7.1f | %7.1f" % (r[0], r[1], r[2], r[3], r[5], r[4], r[6]))
big = R[:, 1] > 0.5 * R[:, 1].max()
for name, m in (("launches above half the largest", big), ("the others", ~big)):
    if m.sum() == 0: continue
    x = R[m]
    print("%s (%d): end %.1f us, median stripe %.1f, 90 %% stripe %.1f, second-latest %.1f  ->  waiting for the latest stripe %.1f us = %.1f %% of the launch; for the latest tenth %.1f us = %.1f %%"
          % (name, m.sum(), x[:, 2].mean(), x[:, 3].mean(), x[:, 5].mean(), x[:, 4].mean(), (x[:, 2] - x[:, 4]).mean(), 100 * (x[:, 2] - x[:, 4]).sum() / x[:, 2].sum(),
             (x[:, 2] - x[:, 5]).mean(), 100 * (x[:, 2] - x[:, 5]).sum() / x[:, 2].sum()))
d = R[:, 2] - R[:, 4]
print("launches waiting more than 25 / 50 / 100 us for their latest stripe: %.1f %% / %.1f %% / %.1f %%" % tuple(100.0 * (d > t).mean() for t in (25, 50, 100)))
