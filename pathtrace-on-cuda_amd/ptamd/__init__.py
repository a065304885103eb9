"""ptamd — Python binding of the C-ABI in include/pt_api.h (libptamd.so).

This is plumbing only: ctypes declarations, numpy views of the C structs and thin
wrappers.  The renderer itself is the HIP library; nothing here computes radiance, and
there is no CPU fallback — if libptamd.so is missing or a HIP call fails, these functions
raise.  torch is used (by ptamd.dist and bench.py) for device buffers, streams and
torch.distributed only.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)
LIB_PATH = os.environ.get("PTAMD_LIB") or os.path.join(PKG_ROOT, "libptamd.so")      # PTAMD_LIB: A/B builds of the same library

PRIM_FLOATS = 84      # sizeof(PtPrimitive) / 4   (3 x 112-byte Vertex, include/mesh.h:21-37)
TRI_FLOATS = 88       # sizeof(PtTriangle) / 4
NODE_BYTES = 40       # sizeof(PtBVHNode)        (CudaBVHNode, include/CudaPrimitive.cuh:237-247)
SPHERE_FLOATS = 16
TILE = 8

NODE_DTYPE = np.dtype([("bMin", "<f4", 3), ("bMax", "<f4", 3), ("childL", "<i4"), ("childR", "<i4"),
                       ("primStart", "<i4"), ("primEnd", "<i4")])
assert NODE_DTYPE.itemsize == NODE_BYTES


class PtError(RuntimeError):
    pass


class PtCamera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("forward", C.c_float * 3), ("up", C.c_float * 3), ("right", C.c_float * 3),
                ("fovy_deg", C.c_float), ("aspect", C.c_float), ("W", C.c_int32), ("H", C.c_int32)]


class PtParams(C.Structure):
    _fields_ = [("passes", C.c_int32), ("spp_per_pass", C.c_int32), ("max_bounce", C.c_int32), ("rr_bounce", C.c_int32),
                ("rr_floor", C.c_float), ("max_refract", C.c_int32), ("first_pass", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32)]


_lib = None

# every symbol include/pt_api.h declares: (name, restype, argtypes)
_P = C.c_void_p
API = [
    ("pt_params_default", None, [C.POINTER(PtParams)]),
    ("pt_last_error", C.c_char_p, []),
    ("pt_version", C.c_char_p, []),
    ("pt_bvh_build_sah", C.c_int, [_P, C.c_int32, C.POINTER(_P)]),
    ("pt_bvh_free", None, [_P]),
    ("pt_bvh_num_nodes", C.c_int32, [_P]),
    ("pt_bvh_num_tris", C.c_int32, [_P]),
    ("pt_bvh_max_depth", C.c_int32, [_P]),
    ("pt_bvh_nodes", _P, [_P]),
    ("pt_bvh_tris", _P, [_P]),
    ("pt_scene_create", C.c_int, [_P, C.c_int32, _P, C.c_int32, _P, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("pt_scene_destroy", None, [_P]),
    ("pt_scene_num_lights", C.c_int32, [_P]),
    ("pt_scene_device_bytes", C.c_int64, [_P]),
    ("pt_tiles_floats", C.c_int64, [C.POINTER(PtCamera), C.POINTER(PtParams)]),
    ("pt_work_bytes", C.c_int64, [C.POINTER(PtCamera), C.POINTER(PtParams)]),
    ("pt_render_tiles", C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), _P, _P, _P]),
    ("pt_untile", C.c_int, [_P, C.POINTER(PtCamera), C.c_int32, _P, _P]),
    ("pt_render", C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), _P]),
    ("pt_last_render_ms", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("pt_render_timings", C.c_int, [_P, _P, C.c_int32, C.c_int32]),
    ("pt_comm_unique_id", C.c_int, [_P]),
    ("pt_comm_create", C.c_int, [_P, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("pt_comm_create_from_file", C.c_int, [C.c_char_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("pt_comm_create_from_file_tagged", C.c_int, [C.c_char_p, C.c_uint64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(_P)]),
    ("pt_comm_destroy", None, [_P]),
    ("pt_comm_rank", C.c_int32, [_P]),
    ("pt_comm_world", C.c_int32, [_P]),
    ("pt_gather_tiles", C.c_int, [_P, _P, C.c_int64, _P, _P]),
    ("pt_gather_frame", C.c_int, [_P, _P, C.POINTER(PtCamera), C.POINTER(PtParams), _P, _P, _P]),
    ("pt_render_split", C.c_int, [_P, C.POINTER(PtCamera), C.POINTER(PtParams), _P, _P]),
    ("pt_tonemap_u8", C.c_int, [_P, C.c_int64, C.c_int32, _P]),
    ("pt_convert_u8", C.c_int, [_P, C.c_int64, _P]),
    ("pt_write_png", C.c_int, [C.c_char_p, _P, C.c_int32, C.c_int32, C.c_int32]),
    ("pt_camera_basis", None, [C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3), C.POINTER(C.c_float * 3)]),
    ("pt_scene_gen", C.c_int32, [C.c_int32, C.c_int32, _P, C.c_int32]),
    ("pt_load_obj", C.c_int32, [C.c_char_p, C.c_float, C.POINTER(C.c_float * 3), _P, C.c_int32]),
    ("pt_dbg_raycast", C.c_int, [_P, _P, C.c_int32, _P, _P]),
    ("pt_dbg_bxdf", C.c_int, [C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    ("pt_dbg_rng", C.c_int, [C.c_int32, C.c_uint64, C.c_int32, _P, _P]),
    ("pt_dbg_math", C.c_int, [C.c_int32, _P, C.c_int32, _P]),
    ("pt_dbg_sincos", C.c_int, [C.c_int32, _P, C.c_int32, _P]),
    ("pt_dbg_ray_setup", C.c_int, [C.c_int32, _P, C.c_int32, _P]),
    ("pt_dbg_pixel_dir", C.c_int, [C.c_int32, C.POINTER(PtCamera), _P, C.c_int32, _P]),
    ("pt_dbg_nee", C.c_int, [_P, _P, C.c_int32, _P]),
    ("pt_dbg_triad", C.c_int, [C.c_int32, C.c_int64, C.c_int32, _P]),
    ("pt_dbg_valu_rate", C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    ("pt_last_counters", C.c_int, [_P, _P]),
    ("pt_dbg_trace_timeline", C.c_int, [_P, _P, C.c_int32]),
    ("pt_enable_counters", C.c_int, [_P, C.c_int32]),
    ("pt_set_mode", C.c_int, [_P, C.c_int32]),
    ("pt_enable_trace_timing", C.c_int, [_P, C.c_int32]),
    ("pt_trace_timing", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    ("pt_shade_timing", C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_double)]),
    ("pt_last_iterations", C.c_int, [_P]),
    ("pt_set_drain_threshold", C.c_int, [_P, C.c_int32]),
    ("pt_set_shade_rounds", C.c_int, [_P, C.c_int32]),
    ("pt_set_early_shade", C.c_int, [_P, C.c_int32]),
]


def lib():
    """Load libptamd.so (built by __graft_entry__.build() / `make -C pathtrace-on-cuda_amd`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PtError(f"{LIB_PATH} is missing: build it with `make -C {PKG_ROOT}` "
                          "(there is no CPU fallback for the render path)")
        l = C.CDLL(LIB_PATH)
        for name, res, args in API:
            fn = getattr(l, name)       # AttributeError if the library does not export it
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def _check(rc, what):
    if rc != 0:
        raise PtError(f"{what} failed ({rc}): {lib().pt_last_error().decode()}")


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def default_params(**kw):
    p = PtParams()
    lib().pt_params_default(C.byref(p))
    for k, v in kw.items():
        setattr(p, k, v)
    return p


def camera_basis(rot_deg=(0.0, 90.0, 0.0)):
    r = (C.c_float * 3)(*rot_deg)
    f, u, rt = (C.c_float * 3)(), (C.c_float * 3)(), (C.c_float * 3)()
    lib().pt_camera_basis(C.byref(r), C.byref(f), C.byref(u), C.byref(rt))
    return np.array(f[:], np.float32), np.array(u[:], np.float32), np.array(rt[:], np.float32)


def make_camera(W, H, pos=(0.0, 20.0, 60.0), rot_deg=(0.0, 90.0, 0.0), fovy_deg=45.0):
    """The reference app's camera: Renderer ctor puts it at (0,20,60) with rotation (0,90,0)
    (srcs/renderer.cpp:28-30); SetScreenSize sets aspect = W/H (srcs/renderer.cpp:47-53)."""
    f, u, r = camera_basis(rot_deg)
    c = PtCamera()
    c.pos[:] = pos
    c.forward[:] = f.tolist()
    c.up[:] = u.tolist()
    c.right[:] = r.tolist()
    c.fovy_deg = fovy_deg
    c.aspect = np.float32(W) / np.float32(H)
    c.W, c.H = W, H
    return c


def gen_scene(kind, lat_lon=187):
    """Procedural scene as a (n, 84) float32 array of reference `Primitive` records."""
    n = lib().pt_scene_gen(kind, lat_lon, None, 0)
    if n < 0:
        _check(n, "pt_scene_gen")
    prims = np.zeros((n, PRIM_FLOATS), np.float32)
    n2 = lib().pt_scene_gen(kind, lat_lon, _ptr(prims), n)
    assert n2 == n
    return prims


def build_bvh(prims):
    """SAH build + flatten.  Returns (nodes[NODE_DTYPE], tris float32 (n,88), max_depth)."""
    prims = np.ascontiguousarray(prims, np.float32)
    assert prims.ndim == 2 and prims.shape[1] == PRIM_FLOATS
    h = C.c_void_p()
    _check(lib().pt_bvh_build_sah(_ptr(prims), prims.shape[0], C.byref(h)), "pt_bvh_build_sah")
    try:
        nn, nt = lib().pt_bvh_num_nodes(h), lib().pt_bvh_num_tris(h)
        nodes = np.frombuffer(C.string_at(lib().pt_bvh_nodes(h), nn * NODE_BYTES), NODE_DTYPE).copy()
        tris = np.frombuffer(C.string_at(lib().pt_bvh_tris(h), nt * TRI_FLOATS * 4), np.float32).reshape(nt, TRI_FLOATS).copy()
        depth = lib().pt_bvh_max_depth(h)
    finally:
        lib().pt_bvh_free(h)
    return nodes, tris, depth


def make_sphere(center, rad, emittance=(0, 0, 0), albedo=(1, 1, 1), specular=(0.04, 0.04, 0.04),
                opacity=1.0, roughness=0.2, metallic=1.0):
    """One PtSphere record (16 float32): center rad | emittance albedo specular opacity roughness metallic."""
    return np.array([*center, rad, *emittance, *albedo, *specular, opacity, roughness, metallic], np.float32)


def tonemap_u8(raw_rgb, sample_cnt):
    raw = np.ascontiguousarray(raw_rgb, np.float32)
    out = np.zeros(raw.shape, np.uint8)
    _check(lib().pt_tonemap_u8(_ptr(raw), raw.size // 3, sample_cnt, _ptr(out)), "pt_tonemap_u8")
    return out


def convert_u8(values):
    """ConverToUint8 (include/image.h:5-8), element-wise."""
    v = np.ascontiguousarray(values, np.float32)
    out = np.zeros(v.shape, np.uint8)
    _check(lib().pt_convert_u8(_ptr(v), v.size, _ptr(out)), "pt_convert_u8")
    return out


def write_png(path, rgb8):
    a = np.ascontiguousarray(rgb8, np.uint8)
    H, W, ch = a.shape
    _check(lib().pt_write_png(path.encode(), _ptr(a), W, H, ch), "pt_write_png")


class Scene:
    """An uploaded scene (PtScene): HBM-resident wide-node BVH, triangle records, lights."""

    def __init__(self, nodes, tris, spheres=None, device=0):
        nodes = np.ascontiguousarray(nodes)
        tris = np.ascontiguousarray(tris, np.float32)
        assert nodes.dtype == NODE_DTYPE and tris.ndim == 2 and tris.shape[1] == TRI_FLOATS
        if spheres is None:
            spheres = np.zeros((0, SPHERE_FLOATS), np.float32)
        spheres = np.ascontiguousarray(spheres, np.float32).reshape(-1, SPHERE_FLOATS)
        self.device = device
        self._h = C.c_void_p()
        _check(lib().pt_scene_create(_ptr(nodes), nodes.shape[0], _ptr(tris), tris.shape[0],
                                     _ptr(spheres) if spheres.shape[0] else None, spheres.shape[0], device, C.byref(self._h)),
               "pt_scene_create")

    @classmethod
    def from_prims(cls, prims, spheres=None, device=0):
        nodes, tris, _ = build_bvh(prims)
        return cls(nodes, tris, spheres, device)

    def close(self):
        if self._h:
            lib().pt_scene_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    @property
    def num_lights(self):
        return lib().pt_scene_num_lights(self._h)

    @property
    def device_bytes(self):
        return lib().pt_scene_device_bytes(self._h)

    def render(self, cam, prm):
        """Whole frame, synchronous; returns (H, W, 3) float32 accumulated radiance."""
        out = np.zeros((cam.H, cam.W, 3), np.float32)
        _check(lib().pt_render(self._h, C.byref(cam), C.byref(prm), _ptr(out)), "pt_render")
        return out

    def render_tiles(self, cam, prm, d_tiles_ptr, d_work_ptr, stream_ptr=0):
        """Device-resident render of this rank's tiles (raw device pointers), enqueued on the given stream; blocks until the
        render has drained in the default mode 1 (include/pt_api.h: pt_render_tiles)."""
        _check(lib().pt_render_tiles(self._h, C.byref(cam), C.byref(prm), C.c_void_p(d_tiles_ptr), C.c_void_p(d_work_ptr),
                                     C.c_void_p(stream_ptr)), "pt_render_tiles")

    def last_render_ms(self):
        ms = C.c_float()
        _check(lib().pt_last_render_ms(self._h, C.byref(ms)), "pt_last_render_ms")
        return ms.value

    def render_timings(self, reset=True):
        """ms of each recent render_units launch (HIP events on the launch stream)."""
        out = np.zeros(64, np.float32)
        n = lib().pt_render_timings(self._h, _ptr(out), 64, 1 if reset else 0)
        if n < 0:
            _check(n, "pt_render_timings")
        return out[:n].copy()

    def set_mode(self, mode):
        """1 = wavefront pipeline (default), 0 = one-kernel state machine."""
        _check(lib().pt_set_mode(self._h, mode), "pt_set_mode")

    def enable_trace_timing(self, max_launches=8192):
        _check(lib().pt_enable_trace_timing(self._h, max_launches), "pt_enable_trace_timing")

    def trace_timing(self):
        """(sum_ms, launches, max_ms) of the wf_trace launches of the last render (HIP events)."""
        s, n, m = C.c_double(), C.c_int32(), C.c_double()
        _check(lib().pt_trace_timing(self._h, C.byref(s), C.byref(n), C.byref(m)), "pt_trace_timing")
        return s.value, n.value, m.value

    def shade_timing(self):
        """(sum_ms, launches, max_ms) of the wf_shade launches of the last render (HIP events on the launch stream)."""
        s, n, m = C.c_double(), C.c_int32(), C.c_double()
        _check(lib().pt_shade_timing(self._h, C.byref(s), C.byref(n), C.byref(m)), "pt_shade_timing")
        return s.value, n.value, m.value

    def set_drain_threshold(self, live_streams):
        _check(lib().pt_set_drain_threshold(self._h, live_streams), "pt_set_drain_threshold")

    def set_early_shade(self, live_streams):
        """wf_shade starts beside the draining wf_trace while at most this many streams are alive (0 = never).  Result-neutral."""
        _check(lib().pt_set_early_shade(self._h, live_streams), "pt_set_early_shade")

    def set_shade_rounds(self, mode):
        """1: next sample starts in the step a path ends; 0: one bounce per step; -1: by live-stream count (result-neutral)."""
        _check(lib().pt_set_shade_rounds(self._h, mode), "pt_set_shade_rounds")

    def last_iterations(self):
        return lib().pt_last_iterations(self._h)

    def enable_counters(self, on=True):
        _check(lib().pt_enable_counters(self._h, 1 if on else 0), "pt_enable_counters")

    def counters(self):
        out = np.zeros(8, np.int64)
        _check(lib().pt_last_counters(self._h, _ptr(out)), "pt_last_counters")
        return out

    def trace_timeline(self, n_launches):
        """Diagnostic (PTAMD_TSTAT=1): per wf_trace launch (start, queue-empty, end) in 100 MHz ticks; 0 = not recorded."""
        raw = np.zeros((int(n_launches), 3), np.int64)
        _check(lib().pt_dbg_trace_timeline(self._h, _ptr(raw), int(n_launches)), "pt_dbg_trace_timeline")
        out = raw.copy()
        out[:, 0] = np.where(raw[:, 0] != 0, ~raw[:, 0], 0)
        out[:, 1] = np.where(raw[:, 1] != 0, ~raw[:, 1], 0)
        return out

    def nee(self, in5):
        """pt_dbg_nee: rows of (p.xyz, seed lo, seed hi as uint32 bits) -> (n, 12) float32 (see include/pt_api.h)."""
        a = np.ascontiguousarray(in5, np.float32).reshape(-1, 5)
        out = np.zeros((a.shape[0], 12), np.float32)
        _check(lib().pt_dbg_nee(self._h, _ptr(a), a.shape[0], _ptr(out)), "pt_dbg_nee")
        return out

    def trace_launch_rays(self, n_launches):
        """Diagnostic (PTAMD_TSTAT=1 or 2): rays traced by each of the first n wf_trace launches of the last render."""
        raw = np.zeros(int(n_launches), np.int64)
        _check(lib().pt_dbg_trace_timeline(self._h, _ptr(raw), -int(n_launches)), "pt_dbg_trace_timeline")
        return raw

    def raycast(self, rays8):
        rays8 = np.ascontiguousarray(rays8, np.float32).reshape(-1, 8)
        n = rays8.shape[0]
        hits = np.zeros((n, 29), np.float32)
        prim = np.zeros(n, np.int32)
        _check(lib().pt_dbg_raycast(self._h, _ptr(rays8), n, _ptr(hits), _ptr(prim)), "pt_dbg_raycast")
        return hits, prim


def tiles_floats(cam, prm):
    n = lib().pt_tiles_floats(C.byref(cam), C.byref(prm))
    if n < 0:
        raise PtError(lib().pt_last_error().decode())
    return n


def work_bytes(cam, prm):
    n = lib().pt_work_bytes(C.byref(cam), C.byref(prm))
    if n < 0:
        raise PtError(lib().pt_last_error().decode())
    return n


def untile(d_gathered_ptr, cam, world, d_frame_ptr, stream_ptr=0):
    _check(lib().pt_untile(C.c_void_p(d_gathered_ptr), C.byref(cam), world, C.c_void_p(d_frame_ptr), C.c_void_p(stream_ptr)), "pt_untile")


class Comm:
    """PtComm: the C-ABI's communicator for the single gather (RCCL under it when world > 1)."""

    def __init__(self, rank=0, world=1, device=0, unique_id=None, id_file=None, timeout_s=60, job_tag=0):
        self._h = C.c_void_p()
        if id_file is not None:
            _check(lib().pt_comm_create_from_file_tagged(id_file.encode(), job_tag, rank, world, device, timeout_s, C.byref(self._h)), "pt_comm_create_from_file_tagged")
        else:
            buf = (C.c_uint8 * 128)(*(unique_id or bytes(128)))
            _check(lib().pt_comm_create(buf, rank, world, device, C.byref(self._h)), "pt_comm_create")

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        _check(lib().pt_comm_unique_id(buf), "pt_comm_unique_id")
        return bytes(buf)

    @property
    def rank(self):
        return lib().pt_comm_rank(self._h)

    @property
    def world(self):
        return lib().pt_comm_world(self._h)

    def gather_tiles(self, d_tiles_ptr, n_floats, d_gathered_ptr, stream_ptr=0):
        _check(lib().pt_gather_tiles(self._h, C.c_void_p(d_tiles_ptr), n_floats, C.c_void_p(d_gathered_ptr), C.c_void_p(stream_ptr)), "pt_gather_tiles")

    def gather_frame(self, d_tiles_ptr, cam, prm, d_gathered_ptr, d_frame_ptr, stream_ptr=0):
        _check(lib().pt_gather_frame(self._h, C.c_void_p(d_tiles_ptr), C.byref(cam), C.byref(prm), C.c_void_p(d_gathered_ptr),
                                     C.c_void_p(d_frame_ptr), C.c_void_p(stream_ptr)), "pt_gather_frame")

    def render_split(self, scene, cam, prm):
        """pt_render_split: this rank's share + the gather; returns the (H, W, 3) frame on rank 0, None elsewhere."""
        out = np.zeros((cam.H, cam.W, 3), np.float32) if self.rank == 0 else None
        _check(lib().pt_render_split(scene.handle, C.byref(cam), C.byref(prm), self._h, _ptr(out) if out is not None else None), "pt_render_split")
        return out

    def close(self):
        if self._h:
            lib().pt_comm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def dbg_bxdf(lobe, in28, device=0):
    in28 = np.ascontiguousarray(in28, np.float32).reshape(-1, 28)
    out = np.zeros((in28.shape[0], 12), np.float32)
    _check(lib().pt_dbg_bxdf(device, lobe, _ptr(in28), in28.shape[0], _ptr(out)), "pt_dbg_bxdf")
    return out


def dbg_rng(seed, n, device=0):
    raw = np.zeros(n, np.uint32)
    uni = np.zeros(n, np.float32)
    _check(lib().pt_dbg_rng(device, seed, n, _ptr(raw), _ptr(uni)), "pt_dbg_rng")
    return raw, uni


def triad_gbps(bytes_per_array=1 << 30, iters=10, device=0):
    """Measured streaming bandwidth of this GPU (float4 triad, 2 reads + 1 write), GB/s."""
    out = C.c_double(0.0)
    _check(lib().pt_dbg_triad(int(device), int(bytes_per_array), int(iters), C.byref(out)), "pt_dbg_triad")
    return float(out.value)


def valu_rate(op=0, waves_per_simd=4, iters=20000, device=0):
    """Measured VALU issue rate: (wave-instructions per second chip-wide, shader clock in GHz) for one instruction kind
    or short instruction group (pt_dbg_valu_rate: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_max3_f32, 3 v_cvt_f32_ubyte1, 4 v_add_u32,
    5 v_fma_f64, 6 v_cndmask_b32, 7 v_pk_mul_f32, 8..67: the table in tools/valu_probe.py; 16 = op 0 with half the lanes masked
    off, op + 256 = any op with half the lanes masked off)."""
    r, g = C.c_double(0.0), C.c_double(0.0)
    _check(lib().pt_dbg_valu_rate(int(device), int(op), int(waves_per_simd), int(iters), C.byref(r), C.byref(g)), "pt_dbg_valu_rate")
    return float(r.value), float(g.value)


def dbg_pixel_dir(cam, pxpypass, device=0):
    a = np.ascontiguousarray(pxpypass, np.int32).reshape(-1, 3)
    out = np.zeros((a.shape[0], 8), np.float32)
    _check(lib().pt_dbg_pixel_dir(device, C.byref(cam), _ptr(a), a.shape[0], _ptr(out)), "pt_dbg_pixel_dir")
    return out


def dbg_ray_setup(dirs, device=0):
    """wf_trace's per-ray set-up on (n, 3) directions -> (n, 5): Normalize(inv(dir)) xyz, cull scale, degenerate flag."""
    d = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
    out = np.zeros((d.shape[0], 5), np.float32)
    _check(lib().pt_dbg_ray_setup(device, _ptr(d), d.shape[0], _ptr(out)), "pt_dbg_ray_setup")
    return out


def dbg_sincos(x, device=0):
    """(n, 2) float32: sin, cos of the samplers' device function (angles in [0, 2 pi])."""
    x = np.ascontiguousarray(x, np.float32)
    out = np.zeros((x.shape[0], 2), np.float32)
    _check(lib().pt_dbg_sincos(device, _ptr(x), x.shape[0], _ptr(out)), "pt_dbg_sincos")
    return out


def dbg_math(x, device=0):
    x = np.ascontiguousarray(x, np.float32).ravel()
    out = np.zeros((x.size, 8), np.float32)
    _check(lib().pt_dbg_math(device, _ptr(x), x.size, _ptr(out)), "pt_dbg_math")
    return out
