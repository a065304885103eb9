"""ctypes binding of the CPU oracle (oracle/pt_oracle.cpp) — test infrastructure only.

Also wraps oracle/_ref/ptref, the partial build of the real reference, when it exists.
Nothing under pathtrace-on-cuda_amd/ imports this module.
"""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
ORACLE_SO = os.path.join(ORACLE_DIR, "_build", "libpt_oracle.so")
PTREF = os.path.join(ORACLE_DIR, "_ref", "ptref")

TRI_FLOATS, HIT_FLOATS, NUM_COUNTERS = 88, 29, 8


class OCamera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("forward", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("fovy_deg", C.c_float), ("aspect", C.c_float), ("W", C.c_int), ("H", C.c_int)]


class OParams(C.Structure):
    _fields_ = [("passes", C.c_int), ("spp_per_pass", C.c_int), ("max_bounce", C.c_int), ("rr_bounce", C.c_int),
                ("rr_floor", C.c_float), ("max_refract", C.c_int), ("first_pass", C.c_int),
                ("x0", C.c_int), ("y0", C.c_int), ("x1", C.c_int), ("y1", C.c_int)]


_lib = None


def build():
    subprocess.run(["make", "-s", "-C", ORACLE_DIR, "oracle"], check=True)


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(ORACLE_DIR, "pt_oracle.cpp")
        if not os.path.exists(ORACLE_SO) or os.path.getmtime(ORACLE_SO) < os.path.getmtime(src):
            build()
        l = C.CDLL(ORACLE_SO)
        P = C.c_void_p
        l.o_set_libm.restype = C.c_int
        l.o_set_libm.argtypes = [C.c_int]
        l.o_rng.argtypes = [C.c_uint64, C.c_int, P, P]
        l.o_bvh_build.restype = C.c_int
        l.o_bvh_build.argtypes = [P, C.c_int, C.POINTER(P), C.POINTER(C.c_int), C.POINTER(P), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        l.o_free.argtypes = [P]
        l.o_tri_hit.argtypes = [P, C.c_int, P, C.c_int, P]
        l.o_sphere_hit.argtypes = [P, C.c_int, P, C.c_int, P]
        l.o_vecmath.argtypes = [P, C.c_int, P]
        l.o_scene_create.restype = P
        l.o_scene_create.argtypes = [P, C.c_int, P, C.c_int, P, C.c_int]
        l.o_scene_destroy.argtypes = [P]
        l.o_scene_num_lights.restype = C.c_int
        l.o_scene_num_lights.argtypes = [P]
        l.o_raycast.argtypes = [P, P, C.c_int, P, P, P]
        l.o_camera_basis.argtypes = [P, P, P, P]
        l.o_render.restype = C.c_int
        l.o_render.argtypes = [P, C.POINTER(OCamera), C.POINTER(OParams), P, P, C.c_int]
        l.o_tonemap.argtypes = [P, C.c_int, C.c_int, P]
        l.o_u8.argtypes = [P, C.c_int, P]
        l.o_pixel_dir.argtypes = [C.POINTER(OCamera), P, C.c_int, P]
        l.o_nee.argtypes = [P, P, C.c_int, P]
        l.o_bxdf.argtypes = [C.c_int, P, C.c_int, P]
        _lib = l
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def set_libm(mode):
    return lib().o_set_libm(mode)


def rng(seed, n):
    raw = np.zeros(n, np.uint32)
    uni = np.zeros(n, np.float32)
    lib().o_rng(seed, n, _p(raw), _p(uni))
    return raw, uni


def bvh_build(prims):
    prims = np.ascontiguousarray(prims, np.float32)
    nodes_p, tris_p = C.c_void_p(), C.c_void_p()
    nn, nt, md = C.c_int(), C.c_int(), C.c_int()
    lib().o_bvh_build(_p(prims), prims.shape[0], C.byref(nodes_p), C.byref(nn), C.byref(tris_p), C.byref(nt), C.byref(md))
    nodes = np.frombuffer(C.string_at(nodes_p, nn.value * 40), np.uint8).copy()
    tris = np.frombuffer(C.string_at(tris_p, nt.value * TRI_FLOATS * 4), np.float32).reshape(nt.value, TRI_FLOATS).copy()
    lib().o_free(nodes_p)
    lib().o_free(tris_p)
    return nodes, tris, md.value


def tri_hit(tris48, rays10):
    tris48 = np.ascontiguousarray(tris48, np.float32).reshape(-1, 48)
    rays10 = np.ascontiguousarray(rays10, np.float32).reshape(-1, 10)
    out = np.zeros((rays10.shape[0], HIT_FLOATS), np.float32)
    lib().o_tri_hit(_p(tris48), tris48.shape[0], _p(rays10), rays10.shape[0], _p(out))
    return out


def sphere_hit(sph16, rays10):
    sph16 = np.ascontiguousarray(sph16, np.float32).reshape(-1, 16)
    rays10 = np.ascontiguousarray(rays10, np.float32).reshape(-1, 10)
    out = np.zeros((rays10.shape[0], HIT_FLOATS), np.float32)
    lib().o_sphere_hit(_p(sph16), sph16.shape[0], _p(rays10), rays10.shape[0], _p(out))
    return out


def vecmath(in7):
    in7 = np.ascontiguousarray(in7, np.float32).reshape(-1, 7)
    out = np.zeros((in7.shape[0], 21), np.float32)
    lib().o_vecmath(_p(in7), in7.shape[0], _p(out))
    return out


def camera_basis(rot=(0.0, 90.0, 0.0)):
    r = np.array(rot, np.float32)
    f, u, rt = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
    lib().o_camera_basis(_p(r), _p(f), _p(u), _p(rt))
    return f, u, rt


def make_camera(W, H, pos=(0.0, 20.0, 60.0), rot=(0.0, 90.0, 0.0), fovy_deg=45.0):
    f, u, r = camera_basis(rot)
    c = OCamera()
    c.pos[:] = pos
    c.forward[:] = f.tolist()
    c.up[:] = u.tolist()
    c.right[:] = r.tolist()
    c.fovy_deg = fovy_deg
    c.aspect = np.float32(W) / np.float32(H)
    c.W, c.H = W, H
    return c


def make_params(W, H, passes=1, spp=16, max_bounce=8, rr_bounce=3, rr_floor=0.5, max_refract=8, first_pass=0, window=None):
    p = OParams()
    p.passes, p.spp_per_pass, p.max_bounce, p.rr_bounce = passes, spp, max_bounce, rr_bounce
    p.rr_floor, p.max_refract, p.first_pass = rr_floor, max_refract, first_pass
    p.x0, p.y0, p.x1, p.y1 = window if window else (0, 0, W, H)
    return p


class Scene:
    def __init__(self, nodes_bytes, tris88, spheres16=None):
        nodes = np.ascontiguousarray(np.frombuffer(np.ascontiguousarray(nodes_bytes).tobytes(), np.uint8))
        tris = np.ascontiguousarray(tris88, np.float32).reshape(-1, TRI_FLOATS)
        sph = np.zeros((0, 16), np.float32) if spheres16 is None else np.ascontiguousarray(spheres16, np.float32).reshape(-1, 16)
        self.n_tris = tris.shape[0]
        self._h = lib().o_scene_create(_p(nodes), nodes.size // 40, _p(tris), tris.shape[0], _p(sph), sph.shape[0])

    def __del__(self):
        try:
            if self._h:
                lib().o_scene_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def num_lights(self):
        return lib().o_scene_num_lights(self._h)

    def raycast(self, rays8):
        rays8 = np.ascontiguousarray(rays8, np.float32).reshape(-1, 8)
        n = rays8.shape[0]
        hits = np.zeros((n, HIT_FLOATS), np.float32)
        prim = np.zeros(n, np.int32)
        cnt = np.zeros(NUM_COUNTERS, np.int64)
        lib().o_raycast(self._h, _p(rays8), n, _p(hits), _p(prim), _p(cnt))
        return hits, prim, cnt

    def nee(self, in5):
        a = np.ascontiguousarray(in5, np.float32).reshape(-1, 5)
        out = np.zeros((a.shape[0], 12), np.float32)
        lib().o_nee(self._h, _p(a), a.shape[0], _p(out))
        return out

    def render(self, cam, prm, nthreads=8, accum=None):
        if accum is None:
            accum = np.zeros((cam.H, cam.W, 3), np.float32)
        cnt = np.zeros(NUM_COUNTERS, np.int64)
        rc = lib().o_render(self._h, C.byref(cam), C.byref(prm), _p(accum), _p(cnt), nthreads)
        if rc != 0:
            raise RuntimeError(f"o_render rc={rc}")
        return accum, cnt


def tonemap(raw, sample_cnt):
    raw = np.ascontiguousarray(raw, np.float32)
    out = np.zeros(raw.shape, np.uint8)
    lib().o_tonemap(_p(raw), raw.size // 3, sample_cnt, _p(out))
    return out


def pixel_dir(cam, pxpypass):
    a = np.ascontiguousarray(pxpypass, np.int32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 8), np.float32)
    lib().o_pixel_dir(C.byref(cam), _p(a), a.shape[0], _p(out))
    return out


def u8(values):
    v = np.ascontiguousarray(values, np.float32)
    out = np.zeros(v.shape, np.uint8)
    lib().o_u8(_p(v), v.size, _p(out))
    return out


def bxdf(lobe, in28):
    in28 = np.ascontiguousarray(in28, np.float32).reshape(-1, 28)
    out = np.zeros((in28.shape[0], 12), np.float32)
    lib().o_bxdf(lobe, _p(in28), in28.shape[0], _p(out))
    return out


# ---------------------------------------------------------------------------------------
# oracle/_ref/ptref — partial build of the real reference (present only where it was built)
# ---------------------------------------------------------------------------------------
def have_ref():
    return os.path.exists(PTREF) and os.access(PTREF, os.X_OK)


def _run_ref(cmd, inputs, out_specs):
    with tempfile.TemporaryDirectory() as d:
        args = [PTREF, cmd]
        for i, a in enumerate(inputs):
            p = os.path.join(d, f"in{i}.bin")
            np.ascontiguousarray(a).tofile(p)
            args.append(p)
        outs = [os.path.join(d, f"out{i}.bin") for i in range(len(out_specs))]
        args += outs
        subprocess.run(args, check=True, stderr=subprocess.DEVNULL)
        return [np.fromfile(p, dt) for p, dt in zip(outs, out_specs)]


def ref_bvh(prims):
    nodes, tris = _run_ref("bvh", [np.ascontiguousarray(prims, np.float32)], [np.uint8, np.float32])
    return nodes, tris.reshape(-1, TRI_FLOATS)


def ref_tri_hit(tris48, rays10):
    (o,) = _run_ref("trihit", [np.ascontiguousarray(tris48, np.float32), np.ascontiguousarray(rays10, np.float32)], [np.float32])
    return o.reshape(-1, HIT_FLOATS)


def ref_sphere_hit(sph16, rays10):
    (o,) = _run_ref("sphit", [np.ascontiguousarray(sph16, np.float32), np.ascontiguousarray(rays10, np.float32)], [np.float32])
    return o.reshape(-1, HIT_FLOATS)


def ref_vecmath(in7):
    (o,) = _run_ref("vecmath", [np.ascontiguousarray(in7, np.float32)], [np.float32])
    return o.reshape(-1, 21)


def ref_camera(rot3):
    """forward | up | right (9 floats) of the reference's Camera after SetRotation(rot) — real srcs/camera.cpp."""
    (o,) = _run_ref("camera", [np.ascontiguousarray(rot3, np.float32)], [np.float32])
    return o.reshape(-1, 9)


def ref_u8(values):
    """ConverToUint8 of the real include/image.h."""
    (o,) = _run_ref("u8", [np.ascontiguousarray(values, np.float32)], [np.uint8])
    return o


PTVIEWER = os.path.join(ORACLE_DIR, "_ref", "ptviewer")
BINDING_OBJ = os.path.join(ORACLE_DIR, "_ref", "pathtracer_mi355x.obj")


def ref_png_write(pixels_hwc, path):
    """PNG through the reference's Image(W,H,C) + Image::WriteTo — real srcs/image.cpp with its vendored stb_image_write.h."""
    px = np.ascontiguousarray(pixels_hwc, np.uint8)
    H, W, C = px.shape
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "in.bin")
        px.tofile(p)
        subprocess.run([PTREF, "pngwrite", p, str(W), str(H), str(C), path], check=True, stderr=subprocess.DEVNULL)


def ref_png_read(path):
    """Decode through the reference's Image(path) (stbi_load, srcs/image.cpp:12-15).  Returns (H, W, C) uint8."""
    with tempfile.TemporaryDirectory() as d:
        o = os.path.join(d, "out.bin")
        subprocess.run([PTREF, "pngread", path, o], check=True, stderr=subprocess.DEVNULL)
        raw = np.fromfile(o, np.uint8)
    W, H, C = (int(v) for v in raw[:12].view(np.int32))
    return raw[12:].reshape(H, W, C)


ROCRAND_REF = os.path.join(ORACLE_DIR, "_build", "rocrand_ref")


def have_rocrand_ref():
    return os.path.exists(ROCRAND_REF) and os.access(ROCRAND_REF, os.X_OK)


def rocrand_ref(seed, n):
    """(raw uint32, uniform float32) of rocRAND's own XORWOW engine on the host: rocrand_init(seed, 0, 0) then n draws."""
    out = subprocess.run([ROCRAND_REF, str(int(seed)), str(int(n))], check=True, capture_output=True, text=True).stdout.split()
    a = np.array(out, dtype=np.uint64).astype(np.uint32).reshape(-1, 2)
    return a[:, 0].copy(), a[:, 1].copy().view(np.float32)
