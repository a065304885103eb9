#!/usr/bin/env python3
"""G6: traversal counters of the REFERENCE algorithm (instrumented oracle) per config scene,
full 1920x1080 frame, 1 pass x 1 spp -> tests/golden/traversal_counters.json.
They fix the algorithmic bytes per sample of SURVEY.md §8(d) that bench.py prices the
roofline with.  TEST INFRASTRUCTURE; run in the build container."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import oracle_lib as O, ptamd
out = {}
O.set_libm(1)
for name, kind, W, H in (("config2_cornell", 0, 1920, 1080), ("config3_standin", 1, 1920, 1080)):
    nodes, tris, depth = ptamd.build_bvh(ptamd.gen_scene(kind, 187))
    t = time.time()
    _, c = O.Scene(nodes.tobytes(), tris).render(O.make_camera(W, H), O.make_params(W, H, 1, 1), 8)
    rays, nd, tr, sp, hits, paths = (int(x) for x in c[:6])
    out[name] = {"W": W, "H": H, "passes": 1, "spp": 1, "rays": rays, "nodes_fetched": nd, "tri_tests": tr, "rays_with_hit": hits, "paths": paths,
                 "rays_per_sample": rays / paths, "nodes_per_ray": nd / rays, "tris_per_ray": tr / rays,
                 "bytes_per_sample_traversal": (40.0 * nd + 36.0 * tr + 156.0 * hits) / paths,
                 "formula": "sum_rays(40*nodes + 36*tris + 156*[hit]) / paths  (+ 24/spp_per_pass added by bench.py), SURVEY.md 8(d)",
                 "oracle_seconds_8_threads": round(time.time() - t, 1)}
    print(name, out[name])
json.dump(out, open(os.path.join(ROOT, "tests", "golden", "traversal_counters.json"), "w"), indent=1)
