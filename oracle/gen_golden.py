#!/usr/bin/env python3
"""Generate tests/golden/* — run in the build container only (needs /root/reference).

TEST INFRASTRUCTURE.  Two kinds of fixture are written:

 * ref_*.npz   inputs + outputs of oracle/_ref/ptref, the partial build of the REAL
               reference (srcs/bvh.cpp, srcs/CudaPrimitive.cu, srcs/camera.cpp, include/CudaPrimitive.cuh,
               include/CudaVector.cuh, include/image.h compiled unmodified — see oracle/Makefile).  These are
               data (vectors), not reference source.
 * oracle_*.npz images / ray tables produced by the CPU restatement under the pinned
               contract (o_set_libm(1)); they pin the oracle against regressions and let
               the GPU tests run where the oracle build is unavailable.
 * anchors.json the three image means recorded in SURVEY.md Appendix A (measured by the
               survey on the reference's own source) — reproduced here with o_set_libm(0).
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "pathtrace-on-cuda_amd"))
import oracle_lib as O  # noqa: E402
import ptamd  # noqa: E402
from scenes_util import jittered_grid, random_rays10, random_tris48, random_spheres16, scene_rays8, test_spheres  # noqa: E402

G = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def gen_camera_u8():
    """ref_camera.npz / ref_u8.npz: the REAL srcs/camera.cpp (Camera ctor + SetRotation + GetForward/GetUp/GetRight) and
    include/image.h (ConverToUint8) through oracle/_ref/ptref.  Own random stream, so the other fixtures are untouched."""
    rs = np.random.RandomState(4321)
    rot = np.concatenate([
        np.array([[0, 90, 0], [0, 0, 0], [0, 180, 0], [0, 90, 360], [0, 90, -90], [0, 200, 725], [0, -30, -725],
                  [359.5, 45, 45], [-10, 135, 180], [0, 1e-3, 270], [0, 179.999, 90]], np.float32),
        np.stack([rs.uniform(-400, 400, 500), rs.uniform(-20, 200, 500), rs.uniform(-800, 800, 500)], 1).astype(np.float32)])
    np.savez_compressed(os.path.join(G, "ref_camera.npz"), rot=rot, basis=O.ref_camera(rot))
    k = np.arange(0, 257, dtype=np.float64)
    edges = (k / 255.99).astype(np.float32)                     # the values where the result steps, and their float neighbours
    vals = np.concatenate([np.linspace(0.0, 1.0, 4097, dtype=np.float32), edges, np.nextafter(edges, np.float32(-1)), np.nextafter(edges, np.float32(2)),
                           rs.uniform(0.0, 1.0, 4096).astype(np.float32)])
    vals = vals[(vals >= 0.0) & (vals <= 1.0)]                   # ACESFilm saturates to [0,1]; outside it the cast is undefined behaviour
    np.savez_compressed(os.path.join(G, "ref_u8.npz"), values=vals, u8=O.ref_u8(vals))
    print("ref_camera.npz", rot.shape, "ref_u8.npz", vals.shape)


def gen_png():
    """ref_png.npz: small images and the PNG files the REAL srcs/image.cpp (Image(W,H,C) + Image::WriteTo, the reference's own vendored
    stb_image_write.h) writes for them, through oracle/_ref/ptref pngwrite.  Data: pixels in, file bytes out."""
    import tempfile
    rs = np.random.RandomState(777)
    out = {}
    for key, (H, W, C) in {"rgb": (21, 37, 3), "gray": (5, 9, 1), "rgba": (8, 8, 4)}.items():
        px = rs.randint(0, 256, (H, W, C)).astype(np.uint8)
        px[: H // 2, :, :] = (np.arange(W)[None, :, None] * 7 % 256).astype(np.uint8)      # smooth rows too, so filters / matches get used
        with tempfile.TemporaryDirectory() as d:
            q = os.path.join(d, "o.png")
            O.ref_png_write(px, q)
            assert np.array_equal(O.ref_png_read(q), px)
            out[key] = px
            out[key + "_png"] = np.frombuffer(open(q, "rb").read(), np.uint8)
    np.savez_compressed(os.path.join(G, "ref_png.npz"), **out)
    print("ref_png.npz", {k: v.shape for k, v in out.items()})


def gen_rocrand():
    """ref_rocrand_xorwow.npz: rocRAND's own XORWOW engine (oracle/rocrand_ref.cpp, host build of /opt/rocm/include/rocrand/rocrand_xorwow.h)
    for the seeds the renderer uses: pixel offset + pass * W * H, small and beyond 2^32."""
    seeds = np.array([0, 1, 2, 255, 256 * 256, 1920 * 1080 - 1, 1920 * 1080 * 7 + 12345, 2 ** 31 - 1, 2 ** 32 - 1, 2 ** 32 + 5, 2 ** 40 + 123456789,
                      2 ** 63 + 17], dtype=np.uint64)
    raw, uni = [], []
    for s in seeds:
        r, u = O.rocrand_ref(int(s), 96)
        raw.append(r); uni.append(u)
    np.savez_compressed(os.path.join(G, "ref_rocrand_xorwow.npz"), seeds=seeds, raw=np.stack(raw), uniform=np.stack(uni))
    print("ref_rocrand_xorwow.npz", np.stack(raw).shape)


def main():
    os.makedirs(G, exist_ok=True)
    if not O.have_ref():
        raise SystemExit("oracle/_ref/ptref missing: run `make -C oracle ref` first")
    if len(sys.argv) > 1 and sys.argv[1] == "rocrand":
        if not O.have_rocrand_ref():
            raise SystemExit("oracle/_build/rocrand_ref missing: run `make -C oracle rocrand` first")
        gen_rocrand()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "png":
        gen_png()
        return
    if len(sys.argv) > 1 and sys.argv[1] == "camera_u8":
        gen_camera_u8()
        return
    gen_camera_u8()
    rs = np.random.RandomState(1234)

    # ---- G1: BVH build + flatten, real reference ----
    cornell = ptamd.gen_scene(0)
    n, t = O.ref_bvh(cornell)
    np.savez_compressed(os.path.join(G, "ref_bvh_cornell.npz"), prims=cornell, nodes=n, tris=t)
    grid = jittered_grid(50, 50, rs)            # 5000 triangles, many centroid ties
    n, t = O.ref_bvh(grid)
    np.savez_compressed(os.path.join(G, "ref_bvh_grid5000.npz"), prims=grid, nodes=n, tris=t)
    hashes = {}
    for name, kind, ll in (("standin187", 1, 187), ("standin4x187", 2, 187), ("standin24", 1, 24)):
        prims = ptamd.gen_scene(kind, ll)
        n, t = O.ref_bvh(prims)
        hashes[name] = {"kind": kind, "lat_lon": ll, "n_prims": int(prims.shape[0]), "n_nodes": int(n.size // 40),
                        "prims_sha256": sha(prims), "nodes_sha256": sha(n), "tris_sha256": sha(t)}
    json.dump(hashes, open(os.path.join(G, "ref_bvh_hashes.json"), "w"), indent=1)

    # ---- G2a: Triangle::hit / Sphere::hit / vec3, real reference ----
    tris48 = random_tris48(64, rs)
    rays = random_rays10(4096, 64, tris48, rs)
    np.savez_compressed(os.path.join(G, "ref_trihit.npz"), tris48=tris48, rays10=rays, hits=O.ref_tri_hit(tris48, rays))
    sph = random_spheres16(8, rs)
    srays = random_rays10(2048, 8, None, rs, spheres=sph)
    np.savez_compressed(os.path.join(G, "ref_sphit.npz"), sph16=sph, rays10=srays, hits=O.ref_sphere_hit(sph, srays))
    v = rs.standard_normal((2048, 7)).astype(np.float32)
    v[:, 6] = rs.uniform(0.3, 3.0, 2048).astype(np.float32)
    v[::7, 3:6] /= np.linalg.norm(v[::7, 3:6], axis=1, keepdims=True)
    np.savez_compressed(os.path.join(G, "ref_vecmath.npz"), in7=v, out21=O.ref_vecmath(v))

    # ---- anchors from SURVEY.md Appendix A ----
    json.dump({"cornell_256x256_1x16": 0.478260, "standin1_320x180_1x4": 0.294783, "standin4_320x180_1x4": 0.296273,
               "source": "SURVEY.md Appendix A (survey probe of the reference's own source, glibc libm, no FMA)"},
              open(os.path.join(G, "anchors.json"), "w"), indent=1)

    # ---- G2b/G5: oracle ray tables and images under the pinned contract ----
    O.set_libm(1)
    scenes = {
        "cornell": (ptamd.gen_scene(0), None),
        "standin24": (ptamd.gen_scene(1, 24), None),
        "standin24_spheres": (ptamd.gen_scene(1, 24), test_spheres()),
    }
    for name, (prims, sph) in scenes.items():
        nodes, tris, _ = O.bvh_build(prims)
        sc = O.Scene(nodes, tris, sph)
        rays8 = scene_rays8(4096, rs)
        hits, prim, cnt = sc.raycast(rays8)
        W = H = 64
        cam = O.make_camera(W, H)
        prm = O.make_params(W, H, passes=2, spp=8, max_bounce=12 if sph is not None else 8)
        img, icnt = sc.render(cam, prm, 8)
        np.savez_compressed(os.path.join(G, f"oracle_{name}.npz"), rays8=rays8, hits=hits, prim=prim, ray_counters=cnt,
                            image=img, image_counters=icnt, passes=2, spp=8, max_bounce=prm.max_bounce,
                            spheres=np.zeros((0, 16), np.float32) if sph is None else sph)
        print(name, "image mean", img.mean(dtype=np.float64), "hits", int((prim >= 0).sum()))
    print("golden fixtures written to", G)


if __name__ == "__main__":
    main()
