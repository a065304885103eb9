#!/usr/bin/env python3
"""One launch of wf_trace, wave by wave (production code path with timestamps: PTAMD_TSTAT=2, PTAMD_TDUMP=<launch>): when each wave starts,
finds the queue dry and leaves, how many trips it makes, where it runs (XCC / SE / CU / SIMD / slot), and for every 112th wave the time and
lane count of each trip.  usage: wave_dump.py launch [kind W H passes spp [world rank]]"""
import os, sys
a = [int(x) for x in sys.argv[1:]]
launch = a[0] if a else 4
os.environ["PTAMD_TSTAT"] = "2"
os.environ["PTAMD_TDUMP"] = str(launch)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import numpy as np, torch, ptamd
from ptamd.dist import TileRenderer
kind, W, H, passes, spp = (a[1:] + [1, 1920, 1080, 8, 64][len(a[1:]):])[:5]
world, rank = (a[6], a[7]) if len(a) > 7 else (1, 0)
nodes, tris, depth = ptamd.build_bvh(ptamd.gen_scene(kind, 187))
sc = ptamd.Scene(nodes, tris)
tr = TileRenderer(sc, ptamd.make_camera(W, H), ptamd.default_params(passes=passes, spp_per_pass=spp, rank=rank, world=world), torch.device("cuda:0"))
tr.render(); torch.cuda.synchronize()
NW, LW, LT = 8192, 64, 1024
rec = np.zeros(NW * 8, np.int64); ptamd._check(ptamd.lib().pt_dbg_trace_timeline(sc._h, ptamd._ptr(rec), -3003), "pt_dbg_trace_timeline")
log = np.zeros(LW * LT // 2, np.int64); ptamd._check(ptamd.lib().pt_dbg_trace_timeline(sc._h, ptamd._ptr(log), -3004), "pt_dbg_trace_timeline")
rec = rec.reshape(NW, 8); log = log.view(np.uint32).reshape(LW, LT)
ran = rec[:, 2] != 0
r = rec[ran]
t00 = r[:, 0].min()
st = (r[:, 0] - t00) / 100.0; ex = (r[:, 2] - t00) / 100.0; dry = np.where(r[:, 1] != 0, (r[:, 1] - t00) / 100.0, np.nan)
trips = r[:, 3].astype(np.float64); tdry = r[:, 4].astype(np.float64); rays = r[:, 5].astype(np.float64)
hw = r[:, 6]; xcc = r[:, 7] & 15
slot = hw & 15; simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
pc = lambda v: " ".join("%.0f" % x for x in np.nanpercentile(v, [0, 10, 25, 50, 75, 90, 99, 100]))
print("launch %d of kind %d %dx%d %dx%d world %d rank %d: %d waves ran, %d rays, span %.1f us" % (launch, kind, W, H, passes, spp, world, rank, ran.sum(), rays.sum(), ex.max()))
print("percentiles 0 10 25 50 75 90 99 100")
print("  wave start  us:", pc(st))
print("  queue dry   us:", pc(dry), "(waves that saw it dry: %d)" % np.isfinite(dry).sum())
print("  wave exit   us:", pc(ex))
print("  life        us:", pc(ex - st))
print("  trips         :", pc(trips), " after dry:", pc(tdry))
print("  rays          :", pc(rays))
print("  us per trip   :", " ".join("%.2f" % x for x in np.percentile((ex - st) / np.maximum(trips, 1), [0, 10, 25, 50, 75, 90, 99, 100])))
cc = lambda x, y: float(np.corrcoef(x, y)[0, 1]) if x.std() > 0 and y.std() > 0 else float("nan")
print("correlation of exit time with: trips %.2f  rays %.2f  start %.2f  slot %.2f  trips-after-dry %.2f" % (cc(ex, trips), cc(ex, rays), cc(ex, st), cc(ex, slot.astype(float)), cc(ex, tdry)))
for name, key in (("slot", slot), ("simd", simd), ("xcc", xcc), ("se", se), ("cu", cu)):
    ks = np.unique(key)
    print("  by %-4s:" % name, " ".join("%d:n%d,st%.0f,ex%.0f,tr%.0f" % (k, (key == k).sum(), st[key == k].mean(), ex[key == k].mean(), trips[key == k].mean()) for k in ks))
# waves per SIMD alive over time
simd_id = (xcc.astype(np.int64) << 12) | (se << 8) | (sh << 7) | (cu << 2) | simd
print("distinct SIMDs seen: %d (waves per SIMD: %s)" % (len(np.unique(simd_id)), pc(np.unique(simd_id, return_counts=True)[1])))
edges = np.arange(0, ex.max() + 25, 25.0)
alive = [(int(((st <= t) & (ex > t)).sum())) for t in edges]
print("waves alive at t (25-us steps):", " ".join(str(v) for v in alive))
# per-trip logs: four words per trip (csrc/pt_wavefront.hip, MODE 2)
print("per-trip logs (every 112-th wave).  A trip = [shader clock over the trip, GHz | refill | vote + node fetch (node trips) | rest]; times in us")
rows = []      # t_top, wait, refill, fetch, rest, lanes, dry, atomics, took, is_node
for w in range(LW):
    g = w * 112
    if g >= NW or not ran[g]: continue
    n = int(min(rec[g, 3], LT // 4))
    if n < 2: continue
    v = log[w, :4 * n].reshape(n, 4)
    base = (rec[g, 0] - t00) / 100.0
    tk = lambda c: (v[:, c] & 0xfffff) / 100.0 + base
    A, Cc, D = tk(0), tk(2), tk(3)
    clk = v[:, 1].astype(np.int64)
    node = (v[:, 3] != 0)
    lanes = (v[:, 0] >> 20) & 127; dryf = (v[:, 0] >> 27) & 1; atom = (v[:, 0] >> 28) & 3
    took = (v[:, 2] >> 20) & 127
    nxt = np.concatenate([A[1:], [(rec[g, 2] - t00) / 100.0]])
    dclk = np.diff(np.concatenate([clk, clk[-1:]])) & 0xffffffff
    ghz = np.where(nxt > A, dclk / np.maximum((nxt - A) * 1000.0, 1e-9), 0.0); ghz[-1] = 0.0      # shader clocks per ns over this trip
    wait = ghz; refill = Cc - A; fetch = np.where(node, D - Cc, 0.0); rest = np.where(node, nxt - D, nxt - Cc)
    for k in range(n): rows.append((A[k], wait[k], refill[k], fetch[k], rest[k], lanes[k], dryf[k], atom[k], took[k], node[k]))
    if w % 8 == 0:
        print("  wave %d (slot %d xcc %d): %d trips, exit %.0f" % (g, rec[g, 6] & 15, rec[g, 7] & 15, n, (rec[g, 2] - t00) / 100.0))
        print("    " + " ".join("%.0f:%d%s%s%s/%.1f+%.1f+%.1f+%.1f" % (A[k], lanes[k], "a%d" % atom[k] if atom[k] else "", "+%d" % took[k] if took[k] else "", "" if node[k] else "T", wait[k], refill[k], fetch[k], rest[k]) for k in range(n)))
R = np.array(rows, dtype=np.float64)
if len(R):
    print("all logged waves, by time in the launch (25-us bins): trips | mean shader clock (GHz), refill, node fetch (node trips), rest (us) | lanes | share with a queue atomic | with rays taken")
    for lo in edges:
        m = (R[:, 0] >= lo) & (R[:, 0] < lo + 25)
        if m.sum() == 0: continue
        mn = m & (R[:, 9] > 0)
        print("  %4.0f: %5d | %.2f %.2f %.2f %.2f | %.1f | %.2f | %.2f" % (lo, m.sum(), R[m, 1].mean(), R[m, 2].mean(), R[mn, 3].mean() if mn.sum() else 0, R[m, 4].mean(), R[m, 5].mean(), (R[m, 7] > 0).mean(), (R[m, 8] > 0).mean()))
    tot = R[:, 2:5].sum()
    print("  share of the logged time: refill %.2f  node fetch %.2f  rest %.2f" % tuple(R[:, c].sum() / tot for c in (2, 3, 4)))
    for name, m in (("before the queue is dry", R[:, 6] == 0), ("after", R[:, 6] == 1)):
        if m.sum(): print("  trips %-24s: %6d  GHz %.2f  refill %.2f  fetch %.2f  rest %.2f us  lanes %.1f" % (name, m.sum(), R[m, 1].mean(), R[m, 2].mean(), R[m & (R[:, 9] > 0), 3].mean(), R[m, 4].mean(), R[m, 5].mean()))
