import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, ptamd, oracle_lib as O
glass = ptamd.make_sphere((10, 6, 8), 6.0, albedo=(1, 1, 1), opacity=0.0, roughness=0.0, metallic=0.0)
nodes, tris, d = ptamd.build_bvh(ptamd.gen_scene(1, 187))
sc = ptamd.Scene(nodes, tris, glass)
W, H = 1920, 1080
cam = ptamd.make_camera(W, H)
O.set_libm(1)
so = O.Scene(nodes.tobytes(), tris, glass)
found = []
for p in range(8):
    img = sc.render(cam, ptamd.default_params(passes=1, spp_per_pass=256, max_bounce=12, first_pass=p))
    bad = np.argwhere(~np.isfinite(img).all(-1))
    print("pass", p, "non-finite pixels", len(bad), bad[:5].tolist(), flush=True)
    for (y, x) in bad[:3]:
        ref, _ = so.render(O.make_camera(W, H), O.make_params(W, H, 1, 256, max_bounce=12, first_pass=p, window=(int(x), int(y), int(x) + 1, int(y) + 1)), 1)
        print("   pixel", (int(x), int(y)), "gpu", img[y, x].tolist(), "oracle", ref[y, x].tolist(), flush=True)
