"""Host-only check of the 4-wide quantised traversal tree (host/accel_build.cpp) that wf_trace walks: every triangle is
reachable exactly once and every quantised child box contains the triangles below it (tools/check_quad.cpp)."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "pathtrace-on-cuda_amd")
OBJS = ["accel_build.o", "bvh_build.o", "scenes.o", "pt_host.o", "obj_loader.o"]


@pytest.fixture(scope="module")
def check_quad(tmp_path_factory):
    objs = [os.path.join(PKG, "build", o) for o in OBJS]
    if not all(os.path.exists(o) for o in objs):
        subprocess.run(["make", "-s", "-C", PKG], check=True)
    exe = str(tmp_path_factory.mktemp("quad") / "check_quad")
    subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tools", "check_quad.cpp")] + objs +
                   ["-pthread", "-o", exe], check=True)
    return exe


@pytest.mark.parametrize("kind,lat_lon", [(0, 4), (1, 8), (1, 40), (2, 16)])
def test_quad_tree_is_complete_and_conservative(check_quad, kind, lat_lon):
    r = subprocess.run([check_quad, str(kind), str(lat_lon)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "tris not seen exactly once 0" in r.stdout and "bad boxes 0" in r.stdout, r.stdout
