"""The reference-side binding (integration/pathtracer_mi355x.cpp) as a compiled artefact, and Image::WriteTo pinned by the real
srcs/image.cpp.

 * not gpu: the binding compiles against the reference's REAL headers (oracle/Makefile `ref`, where /root/reference exists), the
   object defines PathTracer::Render(Camera&, BVH*), LoadFromBVH(BVH*) and the three globals exactly once, the headless viewer host
   links against libptamd.so with no undefined symbol, and without a GPU PathTracer::Render fails the reference's way (exit 99).
 * gpu: oracle/_ref/ptviewer — the reference's own SAHBVH::GenBVHTree, Camera, Image + the binding + libptamd.so — renders the
   Cornell room and the frame is bit-identical to the same render through ctypes and to the oracle; result.png decodes to
   pt_tonemap_u8's bytes.
 * PNG: the golden tests/golden/ref_png.npz holds a PNG written by the reference's Image::WriteTo (srcs/image.cpp:22-25, its own
   stb_image_write.h); pt_write_png's file must decode to the same pixels (PIL here, and the reference's stbi_load where ptref is).
"""
import io
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

import oracle_lib as O
import ptamd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HAVE_REFERENCE = os.path.isdir("/root/reference/srcs")


def _png_decode(data):
    from PIL import Image
    return np.array(Image.open(io.BytesIO(data)))


@pytest.mark.skipif(not HAVE_REFERENCE, reason="needs the reference tree (build container only)")
def test_binding_compiles_against_reference_headers():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "pathtrace-on-cuda_amd")], check=True)
    # touch nothing: make rebuilds oracle/_ref only when a source is newer; force the binding's compile by asking make for it
    subprocess.run(["make", "-s", "-B", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    assert os.path.exists(O.BINDING_OBJ) and os.path.exists(O.PTVIEWER)
    sym = subprocess.run(["nm", "-C", O.BINDING_OBJ], check=True, capture_output=True, text=True).stdout.splitlines()
    defined = {" ".join(l.split()[2:]) for l in sym if len(l.split()) >= 3 and l.split()[1] in "TBD"}
    for want in ("PathTracer::Render(Camera&, BVH*)", "LoadFromBVH(BVH*)", "CudaSpheres", "CudaBVH", "CudaPrims"):
        assert want in defined, (want, sorted(defined))
    # the C-ABI symbols it needs are exactly imports (U), resolved by libptamd.so at link time (-Wl,--no-undefined in the recipe)
    undefined = {" ".join(l.split()[1:]) for l in sym if l.split()[0] == "U"}
    assert {"pt_bvh_build_sah", "pt_scene_create", "pt_render", "pt_tonemap_u8", "pt_scene_destroy"} <= undefined
    ldd = subprocess.run(["ldd", O.PTVIEWER], check=True, capture_output=True, text=True).stdout
    assert "libptamd.so" in ldd and "not found" not in ldd


def _run_viewer(d, W, H, env_extra, spheres=None):
    prims = ptamd.gen_scene(0, 16)
    pp = os.path.join(d, "prims.bin")
    np.ascontiguousarray(prims, np.float32).tofile(pp)
    args = [O.PTVIEWER, pp, str(W), str(H)]
    if spheres is not None:
        sp = os.path.join(d, "spheres.bin")
        np.ascontiguousarray(spheres, np.float32).tofile(sp)
        args.append(sp)
    env = dict(os.environ, **env_extra)
    return prims, subprocess.run(args, cwd=d, env=env, capture_output=True, text=True, timeout=600)


@pytest.mark.skipif(not os.path.exists(O.PTVIEWER), reason="oracle/_ref/ptviewer not built")
def test_binding_fails_the_reference_way_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with tempfile.TemporaryDirectory() as d:
        _, r = _run_viewer(d, 32, 24, {"PT_NUM_MULTI_SAMPLE": "1", "PT_NUM_SAMPLE": "1"})
    assert r.returncode == 99, (r.returncode, r.stderr[-400:])          # include/CudaUtil.cuh:28-36
    assert "GPU error in" in r.stderr
    assert "Tree on GPU Size : 7" in r.stdout and "Prim on GPU Size : 12" in r.stdout      # LoadFromBVH ran on the host


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(O.PTVIEWER), reason="oracle/_ref/ptviewer not built")
@pytest.mark.parametrize("with_sphere", [False, True])
def test_reference_host_code_renders_through_the_binding(with_sphere):
    W, H, passes, spp = 64, 48, 2, 4
    sph = ptamd.make_sphere((10, 6, 8), 6.0, albedo=(1, 1, 1), opacity=0.0, roughness=0.0, metallic=0.0).reshape(1, 16) if with_sphere else None
    with tempfile.TemporaryDirectory() as d:
        prims, r = _run_viewer(d, W, H, {"PT_NUM_MULTI_SAMPLE": str(passes), "PT_NUM_SAMPLE": str(spp), "PT_RAW_OUT": os.path.join(d, "raw.bin")}, sph)
        assert r.returncode == 0, r.stderr[-800:]
        out = r.stdout
        for line in ("Camera : 64 x 48", "Tree on GPU Size : 7", "Prim on GPU Size : 12", "ADD light", "Sample 1 : Delta time", "Export Success"):
            assert line in out, (line, out)
        raw = np.fromfile(os.path.join(d, "raw.bin"), np.float32).reshape(H, W, 3)
        png = open(os.path.join(d, "result.png"), "rb").read()
        assert os.path.exists(os.path.join(d, "temp.png"))
    # the same render through ctypes (the reference's tree comes from its own GenBVHTree there, from pt_bvh_build_sah here)
    nodes, tris, _ = ptamd.build_bvh(prims)
    img = ptamd.Scene(nodes, tris, sph, device=0).render(ptamd.make_camera(W, H), ptamd.default_params(passes=passes, spp_per_pass=spp))
    assert np.array_equal(raw.view(np.uint32), img.view(np.uint32))
    # and the oracle
    O.set_libm(1)
    ref, _ = O.Scene(nodes.tobytes(), tris, sph).render(O.make_camera(W, H), O.make_params(W, H, passes, spp), 8)
    assert np.array_equal(raw.view(np.uint32), ref.view(np.uint32))
    # result.png = exportImage of the accumulation buffer, written by the reference's own Image::WriteTo
    assert np.array_equal(_png_decode(png), ptamd.tonemap_u8(img, passes).reshape(H, W, 3))


def test_png_writer_against_reference_image_writeto(golden_dir):
    g = np.load(os.path.join(golden_dir, "ref_png.npz"))
    for key in ("rgb", "gray", "rgba"):
        px = g[key]
        ref_file = g[key + "_png"].tobytes()
        assert np.array_equal(_png_decode(ref_file).reshape(px.shape), px)          # the reference's file holds these pixels
        with tempfile.TemporaryDirectory() as d:
            p = os.path.join(d, "o.png")
            ptamd.write_png(p, px)
            mine = open(p, "rb").read()
            assert np.array_equal(_png_decode(mine).reshape(px.shape), px)          # pt_write_png: same pixels, same shape
            if O.have_ref():                                                        # live: the reference's own decoder reads our file
                assert np.array_equal(O.ref_png_read(p), px)
                q = os.path.join(d, "r.png")
                O.ref_png_write(px, q)
                assert open(q, "rb").read() == ref_file                             # the golden is what the reference writes today
