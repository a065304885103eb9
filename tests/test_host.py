"""Host half of the product (no GPU): BVH build/flatten, tone map, PNG, camera, and that
libptamd.so loads and exports every symbol include/pt_api.h declares."""
import ctypes as C
import os
import re
import struct
import zlib

import numpy as np
import pytest

import oracle_lib as O
import ptamd
from scenes_util import jittered_grid

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "pt_api.h")).read()
    declared = set(re.findall(r"PT_API\s+[\w\s\*]+?\b(pt_\w+)\s*\(", hdr))
    assert len(declared) >= 30
    l = C.CDLL(ptamd.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(l, name), f"libptamd.so does not export {name}"
    assert declared == {n for n, _, _ in ptamd.API}       # the binding covers the header, nothing more
    assert b"gfx950" in ptamd.lib().pt_version()


def test_struct_sizes_match_reference_layout():
    # sizes printed by the real reference build (`oracle/_ref/ptref sizes`): Vertex 112, Primitive 336,
    # CudaBVHNode 40, Sphere 64, Material 48; Triangle 360 = 352 + vptr
    assert ptamd.PRIM_FLOATS * 4 == 336 and ptamd.NODE_BYTES == 40 and ptamd.TRI_FLOATS * 4 == 352
    assert C.sizeof(ptamd.PtCamera) == 64 and C.sizeof(ptamd.PtParams) == 36


def test_default_params_are_the_reference_defines():
    p = ptamd.default_params()
    assert (p.passes, p.spp_per_pass, p.max_bounce, p.rr_bounce, p.max_refract) == (8, 1024, 8, 3, 8)
    assert p.rr_floor == 0.5 and (p.rank, p.world, p.first_pass) == (0, 1, 0)


@pytest.mark.parametrize("name", ["ref_bvh_cornell", "ref_bvh_grid5000"])
def test_product_bvh_matches_reference_golden(golden_dir, name):
    g = np.load(os.path.join(golden_dir, name + ".npz"))
    nodes, tris, _ = ptamd.build_bvh(g["prims"])
    assert nodes.tobytes() == g["nodes"].tobytes()
    assert np.array_equal(bits(tris), bits(g["tris"]))


@pytest.mark.parametrize("kind,ll", [(0, 3), (1, 24), (1, 187), (2, 40)])
def test_product_bvh_matches_oracle(kind, ll):
    prims = ptamd.gen_scene(kind, ll)
    n_p, t_p, d_p = ptamd.build_bvh(prims)
    n_o, t_o, d_o = O.bvh_build(prims)
    assert n_p.tobytes() == n_o.tobytes() and np.array_equal(bits(t_p), bits(t_o)) and d_p == d_o
    # structure facts SURVEY.md §3.4 records: pre-order with childL == idx+1, contiguous increasing leaf ranges
    inter = n_p["primStart"] == -1
    assert np.all(n_p["childL"][inter] == np.nonzero(inter)[0] + 1)
    leaves = n_p[~inter]
    assert np.all(leaves["primStart"][1:] == leaves["primEnd"][:-1] + 1) and leaves["primStart"][0] == 0
    assert np.all(leaves["primEnd"] - leaves["primStart"] < 4)


def test_bvh_edge_cases():
    rs = np.random.RandomState(5)
    grid = jittered_grid(3, 3, rs)
    for n in (1, 2, 4, 5, 18):                            # single leaf, exactly stopNumber, first split
        n_p, t_p, _ = ptamd.build_bvh(grid[:n])
        n_o, t_o, _ = O.bvh_build(grid[:n])
        assert n_p.tobytes() == n_o.tobytes() and np.array_equal(bits(t_p), bits(t_o))
    dup = np.repeat(grid[:1], 9, 0)                        # all centroids equal: pure tie order
    n_p, t_p, _ = ptamd.build_bvh(dup)
    n_o, t_o, _ = O.bvh_build(dup)
    assert n_p.tobytes() == n_o.tobytes() and np.array_equal(bits(t_p), bits(t_o))
    with pytest.raises(ptamd.PtError):
        ptamd.build_bvh(np.zeros((0, 84), np.float32))


def test_tonemap_matches_oracle():
    rs = np.random.RandomState(3)
    raw = np.concatenate([rs.uniform(0, 40, (4000, 3)), rs.uniform(0, 0.01, (96, 3)), [[0, 1e9, -1.0]]]).astype(np.float32)
    for cnt in (1, 3, 8):
        assert np.array_equal(ptamd.tonemap_u8(raw, cnt), O.tonemap(raw, cnt))


def test_camera_basis_matches_oracle_and_reference_defaults():
    for rot in ((0, 90, 0), (0, 60, 30), (10, 200, 400), (0, 0.5, 359)):
        for a, b in zip(ptamd.camera_basis(rot), O.camera_basis(rot)):
            assert np.array_equal(bits(a), bits(b))
    f, u, r = ptamd.camera_basis((0, 90, 0))              # forward ~ (0,-4.4e-8,-1), up ~ (0,1,-4.4e-8)
    assert abs(f[2] + 1) < 1e-6 and abs(u[1] - 1) < 1e-6 and abs(r[0] - 1) < 1e-6 and abs(f[1]) < 1e-6


def test_png_roundtrip(tmp_path):
    rs = np.random.RandomState(1)
    img = rs.randint(0, 256, (37, 53, 3)).astype(np.uint8)
    p = str(tmp_path / "x.png")
    ptamd.write_png(p, img)
    b = open(p, "rb").read()
    assert b[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, ihdr = 8, b"", None
    while pos < len(b):
        n, typ = struct.unpack(">I4s", b[pos:pos + 8])
        data = b[pos + 8:pos + 8 + n]
        assert struct.unpack(">I", b[pos + 8 + n:pos + 12 + n])[0] == zlib.crc32(typ + data)
        if typ == b"IHDR":
            ihdr = struct.unpack(">IIBBBBB", data)
        if typ == b"IDAT":
            idat += data
        pos += 12 + n
    assert ihdr == (53, 37, 8, 2, 0, 0, 0)
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(37, 1 + 53 * 3)
    assert np.all(raw[:, 0] == 0) and np.array_equal(raw[:, 1:].reshape(37, 53, 3), img)
    with pytest.raises(ptamd.PtError):
        ptamd.write_png("/nonexistent_dir/x.png", img)


def test_scene_gen_counts():
    assert ptamd.gen_scene(0).shape == (12, 84)
    assert ptamd.gen_scene(1, 187).shape[0] == 12 + 69564
    assert ptamd.gen_scene(2, 187).shape[0] == 12 + 4 * 69564


def test_comm_single_rank_and_argument_checks(tmp_path):
    """pt_comm_* on the host: a world of one needs neither RCCL nor a GPU; bad geometry is rejected; a rank that never
    gets the id file times out with PT_ERR_IO instead of hanging."""
    c = ptamd.Comm(rank=0, world=1)
    assert (c.rank, c.world) == (0, 1)
    c.close()
    for rank, world in ((1, 1), (-1, 2), (2, 2), (0, 0)):
        with pytest.raises(ptamd.PtError):
            ptamd.Comm(rank=rank, world=world, unique_id=bytes(128))
    with pytest.raises(ptamd.PtError, match="timed out"):
        ptamd.Comm(rank=1, world=2, id_file=str(tmp_path / "never_written.id"), timeout_s=0)


def test_sampler_sincos_matches_glibc_on_every_float_of_its_domain(tmp_path):
    """csrc/pt_sincos.h (the device's sin / cos for the BxDF samplers) compiled for the host and compared with the oracle's definition,
    (float)sin((double)x) / (float)cos((double)x) of glibc, for EVERY float in [0, 6.283186]: 1,086,918,621 values, zero mismatches
    (tools/sincos_check.c; a few seconds on 8 threads)."""
    import json
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "sincos_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-I", os.path.join(root, "pathtrace-on-cuda_amd", "csrc"),
                    os.path.join(root, "tools", "sincos_check.c"), "-o", exe, "-lm", "-lpthread"], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    d = json.loads(r.stdout)
    assert r.returncode == 0 and d["floats_checked"] > 1_000_000_000
    assert d["sin_mismatches"] == 0 and d["cos_mismatches"] == 0 and d["nan_in_nan_out"]


def test_bench_times_render_calls_of_the_configs_own_shape():
    """bench.py submits the K timed steps in render calls of at most the config's pass count (configs[2]: 8): --steps 20 = 8 + 8 + 4."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec); spec.loader.exec_module(bench)
    assert [n for _, n in bench.render_calls(20, 8, 5)] == [8, 8, 4]
    assert bench.render_calls(20, 8, 5) == [(5, 8), (13, 8), (21, 4)]
    assert bench.render_calls(8, 8, 1) == [(1, 8)] and bench.render_calls(3, 8) == [(0, 3)] and bench.render_calls(20, 20) == [(0, 20)]
    for c in bench.CONFIGS.values():
        assert c["passes"] == 8
