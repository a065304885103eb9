"""OBJ ingest and the headless CLI on the reference-shaped host surface."""
import os
import struct
import subprocess
import zlib

import numpy as np
import pytest

import oracle_lib as O
import ptamd
from scenes_util import V_ALB, V_BIT, V_EMIT, V_MET, V_NRM, V_OPA, V_POS, V_ROU, V_SPEC, V_TAN

PTRENDER = os.path.join(ptamd.PKG_ROOT, "ptrender")

CUBE_OBJ = """# unit cube, quads, no normals
mtllib cube.mtl
v 0 0 0
v 1 0 0
v 1 1 0
v 0 1 0
v 0 0 1
v 1 0 1
v 1 1 1
v 0 1 1
usemtl red
f 1 4 3 2
f 5 6 7 8
usemtl glow
f 1 2 6 5
f 2 3 7 6
f 3 4 8 7
f -8 -4 -1 -5
"""
CUBE_MTL = """newmtl red
Kd 0.8 0.1 0.1
Ks 0.5 0.5 0.5
Pr 0.7
Pm 0.25
d 0.5
newmtl glow
Kd 0 0 0
Ke 3 2 1
"""


def load_obj(path, scale=1.0, translate=(0, 0, 0)):
    import ctypes as C
    t = (C.c_float * 3)(*translate)
    n = ptamd.lib().pt_load_obj(path.encode(), scale, C.byref(t), None, 0)
    if n < 0:
        raise ptamd.PtError(ptamd.lib().pt_last_error().decode())
    prims = np.zeros((n, 84), np.float32)
    assert ptamd.lib().pt_load_obj(path.encode(), scale, C.byref(t), prims.ctypes.data_as(C.c_void_p), n) == n
    return prims


def test_obj_cube_geometry_materials_and_bake(tmp_path):
    (tmp_path / "cube.obj").write_text(CUBE_OBJ)
    (tmp_path / "cube.mtl").write_text(CUBE_MTL)
    prims = load_obj(str(tmp_path / "cube.obj"), scale=2.0, translate=(10, 0, -1))
    assert prims.shape == (12, 84)                                   # 6 quads -> 12 triangles (fan triangulation)
    v = prims.reshape(12, 3, 28)
    pos = v[:, :, V_POS:V_POS + 3]
    assert pos[..., 0].min() == 10 and pos[..., 0].max() == 12 and pos[..., 2].min() == -1 and pos[..., 2].max() == 1
    # first face: "f 1 4 3 2" -> (1,4,3) and (1,3,2), baked by M = T*S
    assert np.array_equal(pos[0], np.array([[10, 0, -1], [10, 2, -1], [12, 2, -1]], np.float32))
    assert np.array_equal(pos[1], np.array([[10, 0, -1], [12, 2, -1], [12, 0, -1]], np.float32))
    # materials: Kd/Ks/Pr/Pm/d from the MTL, the reference's defaults where a key is absent
    assert np.allclose(v[0, 0, V_ALB:V_ALB + 3], (0.8, 0.1, 0.1)) and np.allclose(v[0, 0, V_SPEC:V_SPEC + 3], 0.5)
    assert np.isclose(v[0, 0, V_ROU], 0.7) and np.isclose(v[0, 0, V_MET], 0.25) and np.isclose(v[0, 0, V_OPA], 0.5)
    assert np.allclose(v[4, 0, V_EMIT:V_EMIT + 3], (3, 2, 1)) and np.allclose(v[4, 0, V_SPEC:V_SPEC + 3], 0.04)
    assert v[4, 0, V_ROU] == 0 and v[4, 0, V_OPA] == 1
    # smooth normals: a cube corner's normal is the (area-weighted) diagonal; AddModel leaves it scaled by S
    n0 = v[0, 0, V_NRM:V_NRM + 3]
    assert np.allclose(n0 / np.linalg.norm(n0), -np.ones(3) / np.sqrt(3), atol=1e-6) and np.isclose(np.linalg.norm(n0), 2.0, atol=1e-5)
    # tangent fallback of include/model.h:159-171 and bitangent = cross(n, t)
    n, t, b = n0 / 2.0, v[0, 0, V_TAN:V_TAN + 3] / 2.0, v[0, 0, V_BIT:V_BIT + 3] / 2.0
    exp_t = np.array([0, n[2], -n[1]]) / np.hypot(n[2], n[1])
    assert np.allclose(t, exp_t, atol=1e-6) and np.allclose(b, np.cross(n, t), atol=1e-6)
    # it feeds the BVH build like any other primitive list, identically in product and oracle
    n_p, t_p, _ = ptamd.build_bvh(prims)
    n_o, t_o, _ = O.bvh_build(prims)
    assert n_p.tobytes() == n_o.tobytes() and np.array_equal(t_p.view(np.uint32), t_o.view(np.uint32))


def test_obj_with_normals_and_errors(tmp_path):
    (tmp_path / "t.obj").write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nvn 0 0 1\nf 1//1 2//1 3//1\n")
    prims = load_obj(str(tmp_path / "t.obj"))
    v = prims.reshape(1, 3, 28)
    assert np.array_equal(v[0, :, V_NRM:V_NRM + 3], np.tile(np.array([0, 0, 1], np.float32), (3, 1)))
    assert np.allclose(v[0, 0, V_ALB:V_ALB + 3], 0) and v[0, 0, V_OPA] == 1      # default material
    with pytest.raises(ptamd.PtError):
        load_obj(str(tmp_path / "missing.obj"))
    (tmp_path / "bad.obj").write_text("v 0 0 0\nf 1 2 3\n")
    with pytest.raises(ptamd.PtError):
        load_obj(str(tmp_path / "bad.obj"))
    (tmp_path / "empty.obj").write_text("v 0 0 0\n")
    with pytest.raises(ptamd.PtError):
        load_obj(str(tmp_path / "empty.obj"))


def test_cli_usage():
    assert os.path.exists(PTRENDER)
    r = subprocess.run([PTRENDER, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage: ptrender" in r.stdout
    assert subprocess.run([PTRENDER, "--bogus"], capture_output=True).returncode == 2
    assert subprocess.run([PTRENDER, "--scene", "nope"], capture_output=True).returncode == 2


def _read_png(path):
    b = open(path, "rb").read()
    pos, idat, ihdr = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", b[pos + 8:pos + 8 + n])
        if typ == b"IDAT":
            idat += b[pos + 8:pos + 8 + n]
        pos += 12 + n
    W, H = ihdr[0], ihdr[1]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(H, 1 + W * 3)
    return raw[:, 1:].reshape(H, W, 3)


@pytest.mark.gpu
def test_cli_render_matches_library_and_oracle(tmp_path):
    """ptrender (C++ PathTracer::Render on the reference-shaped surface) writes the same pixels as the
    Python path, progressively, with the glass sphere of config 4 and a dropped-in OBJ."""
    (tmp_path / "cube.obj").write_text(CUBE_OBJ)
    (tmp_path / "cube.mtl").write_text(CUBE_MTL)
    W, H, passes, spp = 96, 56, 3, 4
    r = subprocess.run([PTRENDER, "--scene", "standin", "--lat-lon", "16", "--glass-sphere", "--obj", str(tmp_path / "cube.obj"),
                        "--obj-scale", "6", "--obj-translate", "-14,0,4", "--width", str(W), "--height", str(H),
                        "--passes", str(passes), "--spp", str(spp), "--depth", "12", "--raw", str(tmp_path / "accum.f32")],
                       cwd=tmp_path, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Export Success") == passes + 1 and "Maximum depth of tree" in r.stdout and "ADD light" in r.stdout
    prims = np.concatenate([ptamd.gen_scene(1, 16), load_obj(str(tmp_path / "cube.obj"), 6.0, (-14, 0, 4))])
    nodes, tris, _ = ptamd.build_bvh(prims)
    glass = ptamd.make_sphere((10, 6, 8), 6.0, albedo=(1, 1, 1), opacity=0.0, roughness=0.0, metallic=0.0)
    img = ptamd.Scene(nodes, tris, glass).render(ptamd.make_camera(W, H), ptamd.default_params(passes=passes, spp_per_pass=spp, max_bounce=12))
    assert np.array_equal(_read_png(str(tmp_path / "result.png")), ptamd.tonemap_u8(img, passes))
    assert os.path.exists(tmp_path / "temp.png")
    raw = np.fromfile(str(tmp_path / "accum.f32"), np.float32).reshape(H, W, 3)       # --raw: the float accumulation buffer (viewer hook)
    assert np.array_equal(raw.view(np.uint32), img.view(np.uint32))
    O.set_libm(1)
    ref, _ = O.Scene(nodes.tobytes(), tris, glass).render(O.make_camera(W, H), O.make_params(W, H, passes, spp, max_bounce=12), 8)
    assert np.array_equal(img.view(np.uint32), ref.view(np.uint32))
    assert np.array_equal(_read_png(str(tmp_path / "result.png")), O.tonemap(ref, passes))


RENDER_MGPU = os.path.join(ptamd.PKG_ROOT, "render_mgpu")


def test_render_mgpu_builds_and_fails_loudly_without_a_gpu(tmp_path):
    """integration/render_mgpu.cpp — INTEGRATION.md's one-process-per-GPU C++ program — is a compiled artefact of the package Makefile."""
    assert os.path.exists(RENDER_MGPU)
    r = subprocess.run([RENDER_MGPU, "--help"], capture_output=True, text=True)
    assert r.returncode == 0 and "usage: render_mgpu" in r.stdout
    assert subprocess.run([RENDER_MGPU, "--bogus"], capture_output=True).returncode == 2
    assert subprocess.run([RENDER_MGPU, "--rank", "3", "--world", "2"], capture_output=True).returncode == 2
    import torch
    if not torch.cuda.is_available():
        r = subprocess.run([RENDER_MGPU, "--size", "64x48", "--passes", "1", "--spp", "1", "--scene", "0"], capture_output=True, text=True, cwd=tmp_path)
        assert r.returncode == 99 and "pt_scene_create" in r.stderr          # the reference's convention: message, exit(99)


@pytest.mark.gpu
def test_render_mgpu_world1_matches_library(tmp_path):
    """The C++ tile-split program with one rank (no RCCL involved): tiles -> pt_gather_frame -> frame, bit-identical to pt_render, and the PNG it
    writes is pt_tonemap_u8 of that frame."""
    W, H, passes, spp = 160, 96, 2, 4
    raw = tmp_path / "frame.f32"
    r = subprocess.run([RENDER_MGPU, "--size", f"{W}x{H}", "--passes", str(passes), "--spp", str(spp), "--scene", "1", "--lat-lon", "16",
                        "--out", str(tmp_path / "o.png"), "--raw", str(raw), "--id-file", str(tmp_path / "job.id"), "--job-tag", "77"],
                       capture_output=True, text=True, cwd=tmp_path, timeout=600)
    assert r.returncode == 0, r.stderr[-500:]
    assert "Export Success" in r.stdout and "rank 0 of 1" in r.stdout
    assert not os.path.exists(tmp_path / "job.id")                              # world 1 never touches the rendezvous file
    frame = np.fromfile(raw, np.float32).reshape(H, W, 3)
    img = ptamd.Scene.from_prims(ptamd.gen_scene(1, 16)).render(ptamd.make_camera(W, H), ptamd.default_params(passes=passes, spp_per_pass=spp))
    assert np.array_equal(frame.view(np.uint32), img.view(np.uint32))
    from PIL import Image
    assert np.array_equal(np.array(Image.open(tmp_path / "o.png")), ptamd.tonemap_u8(img, passes).reshape(H, W, 3))
