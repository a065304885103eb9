#!/usr/bin/env python3
"""G6: traversal counters of the REFERENCE algorithm (instrumented oracle) per config scene,
full frame (1920x1080; config 5: 3840x2160), 1 pass x 1 spp -> tests/golden/traversal_counters.json.
They fix the algorithmic bytes per sample of SURVEY.md §8(d) that bench.py prices the
roofline with.  TEST INFRASTRUCTURE; run in the build container."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import oracle_lib as O, ptamd
out = {}
O.set_libm(1)
glass = ptamd.make_sphere((10, 6, 8), 6.0, albedo=(1, 1, 1), opacity=0.0, roughness=0.0, metallic=0.0).reshape(1, 16)
path = os.path.join(ROOT, "tests", "golden", "traversal_counters.json")
if os.path.exists(path) and "--all" not in sys.argv:
    out = json.load(open(path))          # keep what is there; only missing configs are counted (pass --all to redo everything)
for name, kind, W, H, sph, depth_max in (("config2_cornell", 0, 1920, 1080, None, 8), ("config3_standin", 1, 1920, 1080, None, 8),
                                         ("config4_glass", 1, 1920, 1080, glass, 12), ("config5_4x", 2, 3840, 2160, None, 8)):
    if name in out:
        continue
    nodes, tris, depth = ptamd.build_bvh(ptamd.gen_scene(kind, 187))
    t = time.time()
    _, c = O.Scene(nodes.tobytes(), tris, sph).render(O.make_camera(W, H), O.make_params(W, H, 1, 1, max_bounce=depth_max), 8)
    rays, nd, tr, sp, hits, paths = (int(x) for x in c[:6])
    out[name] = {"W": W, "H": H, "passes": 1, "spp": 1, "rays": rays, "nodes_fetched": nd, "tri_tests": tr, "rays_with_hit": hits, "paths": paths,
                 "rays_per_sample": rays / paths, "nodes_per_ray": nd / rays, "tris_per_ray": tr / rays,
                 "bytes_per_sample_traversal": (40.0 * nd + 36.0 * tr + 156.0 * hits) / paths,
                 "formula": "sum_rays(40*nodes + 36*tris + 156*[hit]) / paths  (+ 24/spp_per_pass added by bench.py), SURVEY.md 8(d)",
                 "oracle_seconds_8_threads": round(time.time() - t, 1)}
    print(name, out[name])
json.dump(out, open(path, "w"), indent=1)
