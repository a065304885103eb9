"""Seeded input generators shared by the tests and oracle/gen_golden.py (pure numpy)."""
import numpy as np

# float offsets inside one 28-float Vertex record (include/mesh.h:21-37)
V_POS, V_NRM, V_UV, V_TAN, V_BIT, V_EMIT, V_ALB, V_SPEC, V_OPA, V_MET, V_ROU, V_U, V_V = 0, 3, 6, 8, 11, 14, 17, 20, 23, 24, 25, 26, 27


def _norm(v):
    return (v / np.sqrt((v * v).sum(-1, keepdims=True))).astype(np.float32)


def make_prims(a, b, c, albedo=(0.7, 0.7, 0.7), emit=(0, 0, 0), rough=1.0, metal=0.0, opacity=1.0, smooth_normals=None):
    """(n,3) float32 vertex arrays -> (n,84) Primitive records with flat normals and the
    tangent fallback of include/model.h:159-171."""
    a, b, c = (np.asarray(x, np.float32) for x in (a, b, c))
    n = a.shape[0]
    nrm = _norm(np.cross(b - a, c - a).astype(np.float32))
    t1 = np.stack([-nrm[:, 2], np.zeros(n, np.float32), nrm[:, 0]], 1)
    t2 = np.stack([np.zeros(n, np.float32), nrm[:, 2], -nrm[:, 1]], 1)
    use1 = (np.abs(nrm[:, 0]) > np.abs(nrm[:, 1]))[:, None]
    tan = _norm(np.where(use1, t1, t2))
    bit = np.cross(nrm, tan).astype(np.float32)
    out = np.zeros((n, 3, 28), np.float32)
    for k, p in enumerate((a, b, c)):
        out[:, k, V_POS:V_POS + 3] = p
        out[:, k, V_NRM:V_NRM + 3] = nrm if smooth_normals is None else smooth_normals[k]
        out[:, k, V_TAN:V_TAN + 3] = tan
        out[:, k, V_BIT:V_BIT + 3] = bit
        out[:, k, V_EMIT:V_EMIT + 3] = emit
        out[:, k, V_ALB:V_ALB + 3] = albedo
        out[:, k, V_SPEC:V_SPEC + 3] = 0.04
        out[:, k, V_OPA] = opacity
        out[:, k, V_MET] = metal
        out[:, k, V_ROU] = rough
    return out.reshape(n, 84)


def jittered_grid(nx, nz, rs):
    """2*nx*nz triangles on a height field; x/z on a regular lattice so centroids tie a lot
    (exercises the unstable-sort tie order of the BVH build)."""
    xs = np.linspace(-10, 10, nx + 1, dtype=np.float32)
    zs = np.linspace(-10, 10, nz + 1, dtype=np.float32)
    h = rs.uniform(0.0, 1.0, (nx + 1, nz + 1)).astype(np.float32)
    h[::3] = 0.5       # flat rows: exact ties on the y axis as well
    A, B, Cc = [], [], []
    for i in range(nx):
        for j in range(nz):
            p00 = (xs[i], h[i, j], zs[j]); p10 = (xs[i + 1], h[i + 1, j], zs[j])
            p11 = (xs[i + 1], h[i + 1, j + 1], zs[j + 1]); p01 = (xs[i], h[i, j + 1], zs[j + 1])
            A += [p00, p00]; B += [p11, p01]; Cc += [p10, p11]
    prims = make_prims(np.array(A), np.array(B), np.array(Cc))
    prims[0].reshape(3, 28)[:, V_EMIT:V_EMIT + 3] = 5.0     # one emissive triangle so the scene is renderable
    return prims


def random_tris48(m, rs):
    """TRI48 records: V0 V1 V2 N0 N1 N2 T0 T1 T2 B0 B1 B2 | MAT(12).  Smooth, distinct vertex frames."""
    v = rs.uniform(-5, 5, (m, 3, 3)).astype(np.float32)
    v[: m // 8, :, 1] = 0.0                                  # some axis-aligned (flat boxes)
    fr = _norm(rs.standard_normal((m, 9, 3)).astype(np.float32))
    mat = rs.uniform(0, 1, (m, 12)).astype(np.float32)
    return np.concatenate([v.reshape(m, 9), fr.reshape(m, 27), mat], 1).astype(np.float32)


def random_spheres16(s, rs):
    c = rs.uniform(-5, 5, (s, 3)).astype(np.float32)
    r = rs.uniform(0.5, 3, (s, 1)).astype(np.float32)
    mat = rs.uniform(0, 1, (s, 12)).astype(np.float32)
    return np.concatenate([c, r, mat], 1).astype(np.float32)


def random_rays10(r, m, tris48, rs, spheres=None):
    """RAY10 records aimed at their primitive; includes edge/vertex hits, back faces, tiny
    t ranges, un-normalised directions, and rays with zero direction components."""
    idx = rs.randint(0, m, r)
    org = rs.uniform(-12, 12, (r, 3)).astype(np.float32)
    if tris48 is not None:
        v = tris48[idx, :9].reshape(r, 3, 3)
        w = rs.dirichlet((1, 1, 1), r).astype(np.float32)
        w[::16] = (1, 0, 0); w[1::16] = (0.5, 0.5, 0); w[2::16] = (0, 0, 1)      # vertices / edge midpoints
        tgt = (v * w[:, :, None]).sum(1).astype(np.float32)
    else:
        tgt = (spheres[idx, :3] + rs.uniform(-1, 1, (r, 3)) * spheres[idx, 3:4]).astype(np.float32)
    d = (tgt - org).astype(np.float32)
    d[3::16] *= -1                                           # pointing away
    d[4::32, 0] = 0.0                                        # zero components
    d[5::64, 1:] = 0.0
    tmin = np.zeros(r, np.float32)
    tmax = np.full(r, 999999.0, np.float32)
    tmax[6::16] = rs.uniform(0, 10, tmax[6::16].shape)       # ranges that cut the hit off
    tmin[7::16] = rs.uniform(0, 10, tmin[7::16].shape)
    normalise = (rs.uniform(0, 1, r) < 0.7).astype(np.float32)
    scale = rs.uniform(0.2, 3.0, (r, 1)).astype(np.float32)
    d = (d * np.where(normalise[:, None] > 0, 1.0, scale / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-6))).astype(np.float32)
    return np.concatenate([idx[:, None].astype(np.float32), org, d, tmin[:, None], tmax[:, None], normalise[:, None]], 1).astype(np.float32)


def scene_rays8(r, rs):
    """RAY8 records for the Cornell-room scenes (room is x,z in [-20,20], y in [0,40])."""
    org = np.stack([rs.uniform(-19, 19, r), rs.uniform(1, 39, r), rs.uniform(-19, 30, r)], 1).astype(np.float32)
    d = _norm(rs.standard_normal((r, 3)).astype(np.float32))
    d[::32] = (0, -1, 0); d[1::32] = (1, 0, 0); d[2::32] = (0, 0, -1)            # axis-aligned (degenerate box test)
    d[3::32, 1] = 0.0                                                             # one zero component, not normalised
    # a share of rays aimed at the mesh in the middle of the room
    aim = np.array([0, 11, 0], np.float32) + rs.uniform(-8, 8, (r, 3)).astype(np.float32)
    k = np.arange(r) % 4 == 0
    d[k] = _norm((aim - org)[k])
    d[3::32, 1] = 0.0
    tmax = np.full(r, 999999.0, np.float32)
    tmax[5::16] = rs.uniform(1, 40, tmax[5::16].shape)
    return np.concatenate([org, d, np.zeros((r, 1), np.float32), tmax[:, None]], 1).astype(np.float32)


def test_spheres():
    """Analytic spheres covering the remaining lobes: rough metal (gltfpbr), delta glass
    (pure_refractive), rough glass (refractive) — cf. srcs/renderer.cpp:125-144."""
    def sph(c, r, albedo, opacity, rough, metal):
        return [*c, r, 0, 0, 0, *albedo, 0.04, 0.04, 0.04, opacity, rough, metal]
    return np.array([
        sph((10, 6, 8), 6.0, (1, 1, 1), 0.0, 0.0, 0.0),          # config-4 glass sphere: pure_refractive
        sph((-11, 5, 9), 5.0, (0.9, 0.9, 1.0), 0.0, 0.05, 1.0),  # rough glass: refractive (renderer.cpp:137-144)
        sph((-9, 30, -8), 6.0, (1, 1, 1), 1.0, 0.2, 1.0),        # rough metal: gltfpbr (renderer.cpp:125-135)
    ], np.float32)


def rel_rms(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.sqrt(((a - b) ** 2).sum() / (b ** 2).sum()))
