#!/usr/bin/env python3
"""Print the work counters of the HIP integrator (counting build) next to the oracle's."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, ptamd
kind = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W, H, spp = (int(x) for x in (sys.argv[2:5] if len(sys.argv) > 4 else (1920, 1080, 2)))
prims = ptamd.gen_scene(kind, 187)
nodes, tris, depth = ptamd.build_bvh(prims)
sc = ptamd.Scene(nodes, tris)
cam = ptamd.make_camera(W, H); prm = ptamd.default_params(passes=1, spp_per_pass=spp)
sc.render(cam, prm)
t0 = time.time(); sc.render(cam, prm); torch.cuda.synchronize(); dt = time.time() - t0
print("plain build: %.3f s -> %.2f Msamples/s, kernel ms %s" % (dt, W * H * spp / dt / 1e6, sc.render_timings()))
sc.enable_counters(True)
sc.render(cam, prm)
c = sc.counters()
print("counters", c.tolist())
rays, nd, tr, sp, hits, paths, trips, act = (float(x) for x in c)
print("rays/path %.3f  wide-nodes/ray %.2f  tris/ray %.2f  hit frac %.3f" % (rays / paths, nd / rays, tr / rays, hits / rays))
print("scheduler trips (wave) %.0f, active lane-trips %.0f -> lane utilisation %.3f; rays per active trip %.3f" % (trips, act, act / (trips * 64), rays / act))
