"""The oracle's integrator half has no reference fixture to pin it (the reference ships no
tests and its integrator headers need cuRAND).  The outside anchors that exist are the
image means SURVEY.md Appendix A recorded from the reference's own source; the oracle must
reproduce them to all six printed digits under the same conditions (glibc float libm)."""
import json
import os

import numpy as np

import oracle_lib as O
import ptamd


def _mean(kind, W, H, spp, lat_lon=187):
    prims = ptamd.gen_scene(kind, lat_lon)
    nodes, tris, _ = O.bvh_build(prims)
    sc = O.Scene(nodes, tris)
    old = O.set_libm(0)
    try:
        img, cnt = sc.render(O.make_camera(W, H), O.make_params(W, H, 1, spp), 8)
    finally:
        O.set_libm(old)
    return float(img.mean(dtype=np.float64)), cnt


def test_cornell_anchor(golden_dir):
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    m, cnt = _mean(0, 256, 256, 16)
    assert f"{m:.6f}" == f"{a['cornell_256x256_1x16']:.6f}"
    # traversal counters of the reference algorithm: 7 nodes + 12 tris per ray
    assert cnt[1] == 7 * cnt[0] and cnt[2] == 12 * cnt[0]
    assert abs(cnt[0] / cnt[5] - 6.72) < 0.01


def test_standin_anchor(golden_dir):
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    m, cnt = _mean(1, 320, 180, 4)
    assert f"{m:.6f}" == f"{a['standin1_320x180_1x4']:.6f}"
    assert abs(cnt[1] / cnt[0] - 166.4) < 0.1 and abs(cnt[2] / cnt[0] - 174.8) < 0.1


def test_standin4x_anchor(golden_dir):
    """The third anchor of SURVEY.md Appendix A: four instanced stand-ins (278,268 triangles), 320x180, 1 x 4 spp."""
    a = json.load(open(os.path.join(golden_dir, "anchors.json")))
    m, cnt = _mean(2, 320, 180, 4)
    assert f"{m:.6f}" == f"{a['standin4_320x180_1x4']:.6f}"


def test_golden_images_are_stable(golden_dir):
    """The committed oracle images (pinned contract, correctly-rounded libm) regenerate bit for bit."""
    for name, kind in (("cornell", 0), ("standin24", 1), ("standin24_spheres", 1)):
        g = np.load(os.path.join(golden_dir, f"oracle_{name}.npz"))
        prims = ptamd.gen_scene(kind, 24)
        nodes, tris, _ = O.bvh_build(prims)
        sph = g["spheres"] if g["spheres"].shape[0] else None
        sc = O.Scene(nodes, tris, sph)
        O.set_libm(1)
        img, _ = sc.render(O.make_camera(64, 64), O.make_params(64, 64, int(g["passes"]), int(g["spp"]), int(g["max_bounce"])), 8)
        assert np.array_equal(img.view(np.uint32), g["image"].view(np.uint32))
        hits, prim, _ = sc.raycast(g["rays8"])
        assert np.array_equal(prim, g["prim"]) and np.array_equal(hits.view(np.uint32), g["hits"].view(np.uint32))
