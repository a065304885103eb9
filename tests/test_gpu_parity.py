"""Parity of the HIP path (through the C-ABI) against the CPU oracle, on a real MI355X.

Bar (BASELINE.json north_star): float radiance within 1e-4 relative RMS of the reference
at a fixed RNG seed.  Integer work (RNG words, primitive indices) must be bit-exact.  The
float pipeline is built to be bit-exact too (same operation order, no FMA contraction,
IEEE div/sqrt, correctly rounded transcendentals), so these tests additionally report /
bound the fraction of pixels that are not bit-identical.
"""
import os

import numpy as np
import pytest

import oracle_lib as O
import ptamd
from scenes_util import rel_rms, scene_rays8
from scenes_util import test_spheres as make_test_spheres

pytestmark = pytest.mark.gpu

REL_RMS_TOL = 1e-4          # north_star tolerance


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def same_bits_or_nan(a, b):
    a, b = np.asarray(a, np.float32), np.asarray(b, np.float32)
    return (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))


@pytest.fixture(autouse=True, params=["tail_in_wf_drain", "pipeline_to_the_end"])
def _render_tail(request, monkeypatch):
    """Every test of this file runs twice: with the library's default hand-over of a render's last live streams to wf_drain (80,000: the
    small renders used here are then finished by that kernel after their first 16 bounce iterations) and with the pipeline running to
    the last stream (PTAMD_DRAIN=0, read when a scene is created) — both paths must give the oracle's bits."""
    if request.param == "pipeline_to_the_end":
        monkeypatch.setenv("PTAMD_DRAIN", "0")
    else:
        monkeypatch.delenv("PTAMD_DRAIN", raising=False)
    yield


@pytest.fixture(scope="module", autouse=True)
def _contract():
    import torch
    assert torch.cuda.is_available(), "-m gpu tests need a GPU"
    O.set_libm(1)            # the pinned contract: correctly rounded float transcendentals
    yield


def test_rng_words_and_uniforms_bit_exact():
    for seed in (0, 1, 256 * 256, 2 ** 32 + 5, 1920 * 1080 * 7 + 12345):
        r_g, u_g = ptamd.dbg_rng(seed, 64)
        r_o, u_o = O.rng(seed, 64)
        assert np.array_equal(r_g, r_o) and np.array_equal(bits(u_g), bits(u_o))
        assert u_g.min() > 0.0 and u_g.max() <= 1.0


def test_rng_matches_rocrand_engine_golden(golden_dir):
    """The device RNG against rocRAND's own XORWOW engine (tests/golden/ref_rocrand_xorwow.npz, oracle/rocrand_ref.cpp): the published
    implementation the RNG contract names."""
    g = np.load(os.path.join(golden_dir, "ref_rocrand_xorwow.npz"))
    for seed, raw, uni in zip(g["seeds"], g["raw"], g["uniform"]):
        r_g, u_g = ptamd.dbg_rng(int(seed), raw.shape[0])
        assert np.array_equal(r_g, raw) and np.array_equal(bits(u_g), bits(uni)), int(seed)


def test_device_arithmetic_is_ieee_and_correctly_rounded():
    rs = np.random.RandomState(0)
    x = np.concatenate([rs.uniform(-7, 7, 100000), rs.uniform(-1e-3, 1e-3, 1000), 10.0 ** rs.uniform(-6, 6, 5000)]).astype(np.float32)
    x = x[x != 0]
    out = ptamd.dbg_math(x)
    x64 = x.astype(np.float64)
    # exact: sqrt, reciprocal, division, length (no FMA contraction: (x*x + y*y) + z*z rounded at every step)
    assert np.array_equal(bits(out[:, 4]), bits(np.sqrt(np.abs(x))))
    assert np.array_equal(bits(out[:, 5]), bits(np.float32(1) / x))
    assert np.array_equal(bits(out[:, 6]), bits(x / np.float32(3)))
    y, z = x + np.float32(1), x + np.float32(2)
    assert np.array_equal(bits(out[:, 7]), bits(np.sqrt((x * x + y * y) + z * z)))
    # correctly rounded through fp64 (a double-rounding miss has probability ~1e-8 per call)
    c = np.clip(np.abs(x), np.float32(1e-4), np.float32(0.999)).astype(np.float64)
    for col, ref in ((0, np.sin(x64)), (1, np.cos(x64)), (2, np.arctan(x64)), (3, c ** 5)):
        bad = int((bits(out[:, col]) != bits(ref.astype(np.float32))).sum())
        assert bad <= 1, f"column {col}: {bad} results not correctly rounded"


def test_sampler_sincos_is_the_oracles_on_its_whole_domain():
    """The samplers' sin / cos (csrc/pt_sincos.h, angles in [0, 2 pi]) against the oracle's definition — glibc double sin / cos rounded once
    to float — on 8 M floats: dense random angles, every float around the quadrant boundaries, tiny angles, the largest phi.  The host
    build of the same function is checked on EVERY float of the domain (tools/sincos_check.c, tests/test_host.py): zero mismatches
    there, so zero are allowed here."""
    rs = np.random.RandomState(5)
    parts = [rs.uniform(0, 6.283186, 6000000), 10.0 ** rs.uniform(-30, 0.8, 500000)]
    for k in range(5):      # every float within 2^17 ulps of k * pi / 2
        c = np.float32(k * np.pi / 2)
        b = np.array([c], np.float32).view(np.uint32)[0]
        lo = max(int(b) - (1 << 17), 0)
        parts.append(np.arange(lo, int(b) + (1 << 17), dtype=np.uint32).view(np.float32).astype(np.float64))
    x = np.concatenate(parts).astype(np.float32)
    x = x[(x >= 0) & (x <= np.float32(6.283186))]
    x = np.concatenate([x, np.array([0.0, 6.283184, 6.283186, 1.5707964, 3.1415927], np.float32)])
    out = ptamd.dbg_sincos(x)
    x64 = x.astype(np.float64)
    assert np.array_equal(bits(out[:, 0]), bits(np.sin(x64).astype(np.float32)))
    assert np.array_equal(bits(out[:, 1]), bits(np.cos(x64).astype(np.float32)))
    assert np.isnan(ptamd.dbg_sincos(np.array([np.nan], np.float32))).all()


def test_ray_setup_matches_ieee_arithmetic():
    """wf_trace's per-ray set-up (pt_trace.h: ray_setup — the reference's Normalize(inv(dir)), its degenerate case and the clamp)
    bit for bit against IEEE arithmetic (numpy float32) on a million directions: unit vectors, components down to 1e-30, exact
    powers of two, zeros (degenerate), very large and very small magnitudes."""
    rs = np.random.RandomState(5)
    n = 1 << 20
    d = rs.standard_normal((n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d[: n // 8] *= 10.0 ** rs.uniform(-30, 0, (n // 8, 3))                 # tiny components: huge inverses
    d[n // 8: n // 4] *= 2.0 ** rs.randint(-60, 60, (n // 8, 3))          # beyond +-2^40
    d[n // 4: n // 4 + 4096, rs.randint(0, 3)] = 0.0                        # degenerate
    d[n // 4 + 4096: n // 4 + 8192] = 2.0 ** rs.randint(-3, 3, (4096, 3))  # exact powers of two
    d = d.astype(np.float32)
    out = ptamd.dbg_ray_setup(d)
    with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
        inv = np.float32(1) / d
        L = np.sqrt((inv[:, 0] * inv[:, 0] + inv[:, 1] * inv[:, 1]) + inv[:, 2] * inv[:, 2])
        deg = ~(L < np.float32(np.inf))
        ref = inv / L[:, None]
    assert np.array_equal(out[:, 4] != 0, deg)
    ok = ~deg
    assert ok.sum() > n // 2
    assert np.array_equal(bits(out[ok, :3]), bits(ref[ok])), "normalised inverse direction differs from IEEE division"
    clamp = np.where(np.abs(inv[deg]) <= np.float32(1e30), inv[deg], np.copysign(np.float32(1e30), d[deg]))
    assert np.array_equal(bits(out[deg, :3]), bits(clamp.astype(np.float32)))


def _bxdf_inputs(n, rs, lobe):
    nrm = rs.standard_normal((n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    t = np.cross(nrm, rs.standard_normal((n, 3))); t /= np.linalg.norm(t, axis=1, keepdims=True)
    b = np.cross(nrm, t)
    front = (rs.uniform(0, 1, (n, 1)) < 0.5).astype(np.float64)
    albedo = rs.uniform(0, 1, (n, 3)); spec = rs.uniform(0, 0.2, (n, 3))
    spec[::5] = 0.04; spec[1::11] = 0.0
    rough = rs.uniform(0.02, 1.0, (n, 1)); rough[::7] = 1.0
    if lobe in (1, 3):
        rough[:] = 0.0
    metal = rs.uniform(0, 1, (n, 1)); metal[::3] = 0.0; metal[1::3] = 1.0
    wo = rs.standard_normal((n, 3)); wo /= np.linalg.norm(wo, axis=1, keepdims=True)
    wi = rs.standard_normal((n, 3)); wi /= np.linalg.norm(wi, axis=1, keepdims=True)
    if lobe < 2:   # opaque lobes are evaluated with wo on the normal's side, as the integrator does
        s = np.sign((wo * nrm).sum(1, keepdims=True)); wo *= np.where(s == 0, 1, s)
    seeds = rs.randint(0, 2 ** 31, (n, 2)).astype(np.uint32).view(np.float32)
    return np.concatenate([nrm, t, b, front, albedo, spec, rough, metal, wo, wi], 1).astype(np.float32), seeds


@pytest.mark.parametrize("lobe", [0, 1, 2, 3])
def test_bxdf_tables_match_oracle(lobe):
    rs = np.random.RandomState(10 + lobe)
    a, seeds = _bxdf_inputs(20000, rs, lobe)
    in28 = np.concatenate([a, seeds, np.zeros((a.shape[0], 2), np.float32)], 1)
    g = ptamd.dbg_bxdf(lobe, in28)
    o = O.bxdf(lobe, in28)
    ok = same_bits_or_nan(g, o).all(1)
    # a transcendental double-rounding miss may flip a handful of rows; anything systematic fails
    assert (~ok).sum() <= 2, f"lobe {lobe}: {(~ok).sum()} of {len(ok)} rows differ, first {np.nonzero(~ok)[0][:5]}"
    assert np.isfinite(o[:, :3]).all(1).mean() > 0.9 and (np.abs(o[:, :3]).sum(1) > 0).mean() > 0.2


@pytest.mark.parametrize("name,kind", [("cornell", 0), ("standin24", 1), ("standin24_spheres", 1)])
def test_closest_hit_matches_oracle_golden(golden_dir, name, kind):
    g = np.load(os.path.join(golden_dir, f"oracle_{name}.npz"))
    sph = g["spheres"] if g["spheres"].shape[0] else None
    sc = ptamd.Scene.from_prims(ptamd.gen_scene(kind, 24), sph)
    hits, prim = sc.raycast(g["rays8"])
    assert np.array_equal(prim, g["prim"])                    # same primitive, including the tie rule
    assert same_bits_or_nan(hits, g["hits"]).all()            # t, p, normal, tangent, bitangent, material


def test_closest_hit_big_scene_live():
    rs = np.random.RandomState(77)
    prims = ptamd.gen_scene(1, 187)
    nodes, tris, _ = ptamd.build_bvh(prims)
    rays = scene_rays8(20000, rs)
    h_o, p_o, cnt = O.Scene(nodes.tobytes(), tris, make_test_spheres()).raycast(rays)
    h_g, p_g = ptamd.Scene(nodes, tris, make_test_spheres()).raycast(rays)
    assert np.array_equal(p_g, p_o) and same_bits_or_nan(h_g, h_o).all()
    assert (p_o >= 0).mean() > 0.7


def _check_image(img_g, img_o, what):
    rr = rel_rms(img_g, img_o)
    same = (bits(img_g) == bits(img_o)).all(-1)
    print(f"{what}: relRMS {rr:.3e}, bit-identical pixels {same.mean():.6f}")
    assert np.isfinite(img_g).all()
    assert rr <= REL_RMS_TOL, f"{what}: relative RMS {rr:.3e} > {REL_RMS_TOL}"
    assert same.mean() >= 0.999, f"{what}: only {same.mean():.5f} of pixels bit-identical"


@pytest.mark.parametrize("name,kind", [("cornell", 0), ("standin24", 1), ("standin24_spheres", 1)])
def test_image_matches_oracle_golden(golden_dir, name, kind):
    g = np.load(os.path.join(golden_dir, f"oracle_{name}.npz"))
    sph = g["spheres"] if g["spheres"].shape[0] else None
    sc = ptamd.Scene.from_prims(ptamd.gen_scene(kind, 24), sph)
    cam = ptamd.make_camera(64, 64)
    prm = ptamd.default_params(passes=int(g["passes"]), spp_per_pass=int(g["spp"]), max_bounce=int(g["max_bounce"]))
    _check_image(sc.render(cam, prm), g["image"], name)
    for rounds in (0, 1):      # both shading schedules of the pipeline (pt_set_shade_rounds)
        sc.set_shade_rounds(rounds)
        _check_image(sc.render(cam, prm), g["image"], f"{name}, shade rounds {rounds}")


def test_config1_cornell_256_16spp_live():
    """BASELINE.json configs[0]: Cornell 256x256, 16 spp — the reference's CPU-runnable case."""
    prims = ptamd.gen_scene(0)
    nodes, tris, _ = ptamd.build_bvh(prims)
    img_o, cnt = O.Scene(nodes.tobytes(), tris).render(O.make_camera(256, 256), O.make_params(256, 256, 1, 16), 16)
    img_g = ptamd.Scene(nodes, tris).render(ptamd.make_camera(256, 256), ptamd.default_params(passes=1, spp_per_pass=16))
    _check_image(img_g, img_o, "config1")


def test_ragged_frame_and_odd_sizes():
    """Frame sizes that are not multiples of the 8x8 tile; 2x2 is the smallest legal frame."""
    prims = ptamd.gen_scene(1, 12)
    nodes, tris, _ = ptamd.build_bvh(prims)
    so, sg = O.Scene(nodes.tobytes(), tris), ptamd.Scene(nodes, tris)
    for W, H in ((2, 2), (13, 9), (65, 31)):
        img_o, _ = so.render(O.make_camera(W, H), O.make_params(W, H, 2, 4), 8)
        img_g = sg.render(ptamd.make_camera(W, H), ptamd.default_params(passes=2, spp_per_pass=4))
        assert img_g.shape == (H, W, 3)
        _check_image(img_g, img_o, f"{W}x{H}")


def test_pass_accumulation_order():
    """image += mean(pass) in pass order: a 3-pass call equals three 1-pass calls summed in order."""
    sc = ptamd.Scene.from_prims(ptamd.gen_scene(1, 12))
    cam = ptamd.make_camera(48, 40)
    whole = sc.render(cam, ptamd.default_params(passes=3, spp_per_pass=4))
    acc = np.zeros_like(whole)
    for p in range(3):
        acc = acc + sc.render(cam, ptamd.default_params(passes=1, spp_per_pass=4, first_pass=p))
    assert np.array_equal(bits(whole), bits(acc))


def test_tile_split_is_bitwise_invariant():
    """1-GPU frame == N-rank tiled frame, bit for bit (virtual ranks on one device)."""
    import torch
    from ptamd.dist import TileRenderer, untile_index
    sc = ptamd.Scene.from_prims(ptamd.gen_scene(1, 16), make_test_spheres())
    W, H = 100, 52
    cam = ptamd.make_camera(W, H)
    base = sc.render(cam, ptamd.default_params(passes=2, spp_per_pass=4, max_bounce=12))
    for world in (2, 3, 8):
        parts = []
        for rank in range(world):
            prm = ptamd.default_params(passes=2, spp_per_pass=4, max_bounce=12, rank=rank, world=world)
            tr = TileRenderer(sc, cam, prm, torch.device("cuda:0"))
            parts.append(tr.render().clone())
        gathered = torch.cat(parts)
        frame = tr.assemble(gathered, world).cpu().numpy()
        assert np.array_equal(bits(frame), bits(base)), f"world={world}"
        # the HIP untile kernel implements exactly ptamd.dist.untile_index
        idx = untile_index(W, H, world)
        assert np.array_equal(bits(gathered.cpu().numpy().reshape(-1, 3)[idx].reshape(H, W, 3)), bits(base))


def test_scaled_scene_shadow_early_out():
    """Cornell room + small stand-in scaled x100 (coordinates in the thousands, where ulp(coordinate) exceeds the reference's EPS = 1e-4):
    the shadow rays' any-hit early-out keeps its margin above the rounding of org + t*dir (pt_stream.h: shadow_stop_t), so the image
    still equals the oracle's — which always takes the closest hit — bit for bit."""
    prims = ptamd.gen_scene(1, 12).copy()
    for v in range(3):
        prims[:, 28 * v:28 * v + 3] *= 100.0                      # Vertex.Position of the three vertices
    nodes, tris, _ = ptamd.build_bvh(prims)
    W, H = 96, 64
    pos = (0.0, 2000.0, 6000.0)
    img = ptamd.Scene(nodes, tris).render(ptamd.make_camera(W, H, pos=pos), ptamd.default_params(passes=2, spp_per_pass=8))
    O.set_libm(1)
    ref, _ = O.Scene(nodes.tobytes(), tris).render(O.make_camera(W, H, pos=pos), O.make_params(W, H, 2, 8), 8)
    _check_image(img, ref, "scaled x100")
    assert np.array_equal(bits(img), bits(ref))


def test_pixel_direction_table():
    """StartRender's prologue + GetPixelDirection (srcs/pathtracer.cu:33-40,70-74), row by row: jitter draws, direction bits and
    the RNG position afterwards, for corner / edge / random pixels of three frame shapes and several passes."""
    rs = np.random.RandomState(5)
    for W, H in ((1920, 1080), (3840, 2160), (100, 52)):
        px = np.concatenate([[0, W - 1, 0, W - 1, W // 2], rs.randint(0, W, 3000)])
        py = np.concatenate([[0, 0, H - 1, H - 1, H // 2], rs.randint(0, H, 3000)])
        ps = np.concatenate([[0, 1, 7, 7, 3], rs.randint(0, 8, 3000)])
        rows = np.stack([px, py, ps], 1).astype(np.int32)
        got = ptamd.dbg_pixel_dir(ptamd.make_camera(W, H), rows)
        want = O.pixel_dir(O.make_camera(W, H), rows)
        assert np.array_equal(bits(got), bits(want)), (W, H)
        assert np.all(np.abs(np.linalg.norm(got[:, 2:5], axis=1) - 1) < 1e-6)


def test_nee_table():
    """One NEE sample per row (SamplePrimitive, pdf, cosA: include/CudaUtil.cuh:38-48,235-241) and its visibility (GetLightColor,
    :150-166: shadow ray's closest hit, EPS test, emittance), on the stand-in scene with the test spheres, from surface points
    found by casting rays into the scene plus points on and above the light."""
    prims = ptamd.gen_scene(1, 24)
    nodes, tris, _ = ptamd.build_bvh(prims)
    sph = make_test_spheres()
    sg, so = ptamd.Scene(nodes, tris, sph), O.Scene(nodes.tobytes(), tris, sph)
    rs = np.random.RandomState(11)
    hits, prim, _ = so.raycast(scene_rays8(6000, rs))
    pts = hits[prim >= 0][:, 5:8]                                  # HitResult.p of rays that hit something
    pts = np.concatenate([pts, np.stack([rs.uniform(-5, 5, 200), np.full(200, 39.98), rs.uniform(-5, 5, 200)], 1),
                          np.stack([rs.uniform(-20, 20, 200), np.full(200, 39.999), rs.uniform(-20, 20, 200)], 1)]).astype(np.float32)
    seeds = rs.randint(0, 2**32, (pts.shape[0], 2), dtype=np.uint64).astype(np.uint32)
    in5 = np.concatenate([pts, seeds.view(np.float32)], 1)
    got, want = sg.nee(in5), so.nee(in5)
    assert (want[:, 8:11].sum(1) > 0).sum() > 1000 and (want[:, 8:11].sum(1) == 0).sum() > 1000      # lit and shadowed rows both present
    assert np.array_equal(bits(got), bits(want))


def test_c_abi_gather_frame_world1():
    """The C-ABI's exchange step (pt_comm_create / pt_gather_frame) with a world of one: a device copy + pt_untile, bit for bit
    the frame pt_render returns.  (world > 1 runs ncclGather and needs one GPU per rank: unmeasured on this one-GPU box.)"""
    import torch
    from ptamd.dist import TileRenderer
    sc = ptamd.Scene.from_prims(ptamd.gen_scene(1, 16))
    W, H = 100, 52
    cam = ptamd.make_camera(W, H)
    prm = ptamd.default_params(passes=2, spp_per_pass=4)
    base = sc.render(cam, prm)
    dev = torch.device("cuda:0")
    tr = TileRenderer(sc, cam, prm, dev)
    tiles = tr.render()
    gathered = torch.empty_like(tiles)
    frame = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    comm = ptamd.Comm(rank=0, world=1, device=0)
    stream = torch.cuda.current_stream(dev).cuda_stream
    comm.gather_frame(tiles.data_ptr(), cam, prm, gathered.data_ptr(), frame.data_ptr(), stream)
    torch.cuda.synchronize()
    assert np.array_equal(bits(frame.cpu().numpy()), bits(base))
    assert np.array_equal(bits(comm.render_split(sc, cam, prm)), bits(base))        # pt_render_split: the whole thing in one call
    with pytest.raises(ptamd.PtError):          # params of another split than the communicator's
        comm.gather_frame(tiles.data_ptr(), cam, ptamd.default_params(passes=2, spp_per_pass=4, rank=0, world=2), gathered.data_ptr(), frame.data_ptr(), stream)
    comm.close()


def test_full_frame_1080p_window_parity_and_split():
    """At BASELINE's frame size (1920x1080, stand-in scene, 69,576 triangles): a pixel window
    of the GPU frame equals the oracle's render of that window (same full-frame seeds), and
    the 8-way tile split reproduces the 1-GPU frame bit for bit."""
    import torch
    from ptamd.dist import TileRenderer
    prims = ptamd.gen_scene(1, 187)
    nodes, tris, _ = ptamd.build_bvh(prims)
    sg = ptamd.Scene(nodes, tris)
    W, H = 1920, 1080
    cam = ptamd.make_camera(W, H)
    prm = ptamd.default_params(passes=1, spp_per_pass=2)
    full = sg.render(cam, prm)
    assert np.isfinite(full).all() and 0.2 < full.mean() < 0.6
    win = (900, 500, 964, 532)                                  # 64x32 pixels over the mesh
    img_o, _ = O.Scene(nodes.tobytes(), tris).render(O.make_camera(W, H), O.make_params(W, H, 1, 2, window=win), 16)
    x0, y0, x1, y1 = win
    _check_image(full[y0:y1, x0:x1], img_o[y0:y1, x0:x1], "1080p window")
    world = 8
    parts = []
    for rank in range(world):
        tr = TileRenderer(sg, cam, ptamd.default_params(passes=1, spp_per_pass=2, rank=rank, world=world), torch.device("cuda:0"))
        parts.append(tr.render().clone())
    frame = tr.assemble(torch.cat(parts), world).cpu().numpy()
    assert np.array_equal(bits(frame), bits(full))


def test_shading_schedule_switches_inside_a_render():
    """With the schedule left to the live-stream count (pt_set_shade_rounds(-1)) a 1080p x 3-pass render starts above the switch
    point (6.2 M streams: one bounce per step) and crosses it as streams retire (two rounds per step): the mix must give the
    same bits as either pure schedule."""
    sc = ptamd.Scene.from_prims(ptamd.gen_scene(1, 24))
    cam = ptamd.make_camera(1920, 1080)
    prm = ptamd.default_params(passes=3, spp_per_pass=3)
    frames = []
    for rounds in (-1, 1, 0):
        sc.set_shade_rounds(rounds)
        frames.append(sc.render(cam, prm))
    assert np.isfinite(frames[0]).all() and frames[0].mean() > 0.05
    assert np.array_equal(bits(frames[0]), bits(frames[1])) and np.array_equal(bits(frames[0]), bits(frames[2]))


def test_early_shade_is_result_neutral():
    """pt_set_early_shade: the shade step starts on a second stream beside the draining traversal kernel (streams whose rays are all back),
    the rest follows.  Same bits as the one-launch step — on and off by the render's stream count, with both shading schedules, with
    spheres (all four lobes) —, for a small frame and for one rank of an 8-way split of the 1080p frame (the size the feature is for)."""
    prims = ptamd.gen_scene(1, 24)
    nodes, tris, _ = ptamd.build_bvh(prims)
    sph = make_test_spheres()
    sc = ptamd.Scene(nodes, tris, sph)
    cam = ptamd.make_camera(160, 90)
    prm = ptamd.default_params(passes=2, spp_per_pass=6)
    sc.set_early_shade(0)
    base = sc.render(cam, prm)
    assert np.isfinite(base).all() and base.mean() > 0.05
    for rounds in (1, 0):
        sc.set_shade_rounds(rounds)
        for below in (28800 * 8, 28800, 28799, 28800 * 8 + 8):      # on / on (the render has exactly 28,800 streams) / off (too many) / off (too few)
            sc.set_early_shade(below)
            assert np.array_equal(bits(sc.render(cam, prm)), bits(base)), (rounds, below)
    # a larger frame: a window of 1080p rendered as one rank of an 8-way split (the size the feature is for) against the plain step
    big = ptamd.Scene.from_prims(ptamd.gen_scene(1, 187))
    cam = ptamd.make_camera(1920, 1080)
    import torch
    from ptamd.dist import TileRenderer
    out = []
    for below in (0, 4000000):
        big.set_early_shade(below)
        tr = TileRenderer(big, cam, ptamd.default_params(passes=2, spp_per_pass=4, rank=3, world=8), torch.device("cuda:0"))
        out.append(tr.render().cpu().numpy().copy())
    assert np.isfinite(out[0]).all() and out[0].mean() > 0.05
    assert np.array_equal(bits(out[0]), bits(out[1]))


def test_very_bright_light_keeps_its_shadow_rays():
    """Dead-NEE pruning assumes (weight * brdfcos) * Le stays finite (pt_stream.h: bounce); the upload switches it off for a scene with
    an emittance above 1e8, where the product can overflow and the reference then adds inf * 0 = NaN (include/CudaUtil.cuh:271-272).
    With a light of 3e38 the frame is mostly inf / NaN — and must be so exactly where the oracle's is."""
    from scenes_util import V_EMIT
    prims = ptamd.gen_scene(1, 16).copy().reshape(-1, 3, 28)
    lit = prims[:, 0, V_EMIT] > 0
    assert lit.sum() == 2
    prims[lit, :, V_EMIT:V_EMIT + 3] = np.float32(3e38)
    prims = prims.reshape(-1, 84)
    nodes, tris, _ = ptamd.build_bvh(prims)
    W, H = 64, 48
    img = ptamd.Scene(nodes, tris).render(ptamd.make_camera(W, H), ptamd.default_params(passes=2, spp_per_pass=16))
    ref, _ = O.Scene(nodes.tobytes(), tris).render(O.make_camera(W, H), O.make_params(W, H, 2, 16), 8)
    assert same_bits_or_nan(img, ref).all()
    assert np.isnan(ref).any() or np.isinf(ref).any()
    # the same scene with an ordinary light still prunes and still matches (the guard is per scene)
    prims = prims.reshape(-1, 3, 28); prims[lit, :, V_EMIT:V_EMIT + 3] = np.float32(9e7); prims = prims.reshape(-1, 84)
    nodes, tris, _ = ptamd.build_bvh(prims)
    img = ptamd.Scene(nodes, tris).render(ptamd.make_camera(W, H), ptamd.default_params(passes=1, spp_per_pass=8))
    ref, _ = O.Scene(nodes.tobytes(), tris).render(O.make_camera(W, H), O.make_params(W, H, 1, 8), 8)
    assert same_bits_or_nan(img, ref).all() and np.isfinite(ref).all()


def test_error_paths():
    prims = ptamd.gen_scene(0)
    nodes, tris, _ = ptamd.build_bvh(prims[:10])               # drop the light quad -> no emissive triangle
    sc = ptamd.Scene(nodes, tris)
    assert sc.num_lights == 0
    with pytest.raises(ptamd.PtError, match="emissive"):
        sc.render(ptamd.make_camera(16, 16), ptamd.default_params(passes=1, spp_per_pass=1))
    ok = ptamd.Scene.from_prims(prims)
    assert ok.num_lights == 2
    with pytest.raises(ptamd.PtError):
        ok.render(ptamd.make_camera(1, 16), ptamd.default_params(passes=1, spp_per_pass=1))
    with pytest.raises(ptamd.PtError):
        ok.render(ptamd.make_camera(16, 16), ptamd.default_params(passes=0))
    bad = nodes.copy(); bad["childL"][0] = 99                   # child index out of range is caught on the host
    with pytest.raises(ptamd.PtError):
        ptamd.Scene(bad, tris)


def test_drain_kernel_on_the_config_scene_tree_equals_the_pipeline():
    """wf_drain (the run-to-completion launch that takes over a render's last live streams, on by default) walks the 4-wide tree with a 40-entry
    per-lane stack: on the bunny stand-in of configs[2] — a tree of the depth that stack is sized for — a frame finished entirely by it, one
    handed over half-way and one rendered by the pipeline alone are the same bits, and all equal the CPU oracle."""
    prims = ptamd.gen_scene(1, 187)
    nodes, tris, _ = ptamd.build_bvh(prims)
    sc = ptamd.Scene(nodes, tris)
    cam = ptamd.make_camera(96, 54)
    prm = ptamd.default_params(passes=2, spp_per_pass=6)
    sc.set_drain_threshold(0)
    a = sc.render(cam, prm)
    it0 = sc.last_iterations()
    sc.set_drain_threshold(1 << 30)
    b = sc.render(cam, prm)
    sc.set_drain_threshold(3000)
    c = sc.render(cam, prm)
    assert sc.last_iterations() < it0
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and np.array_equal(a.view(np.uint32), c.view(np.uint32))
    O.set_libm(1)
    img, _ = O.Scene(nodes.tobytes(), tris).render(O.make_camera(96, 54), O.make_params(96, 54, 2, 6), 16)
    _check_image(a, img, "drain kernel, bunny stand-in")


def test_wavefront_pipeline_equals_state_machine_kernel():
    """The two render paths (queue-driven pipeline, one-kernel state machine) are bit-identical,
    including refraction chains, on a frame that is not tile-aligned."""
    sc = ptamd.Scene.from_prims(ptamd.gen_scene(1, 20), make_test_spheres())
    cam = ptamd.make_camera(150, 77)
    prm = ptamd.default_params(passes=3, spp_per_pass=5, max_bounce=12)
    sc.set_mode(1)
    sc.set_drain_threshold(0)                 # pure pipeline: every bounce is a trace + shade launch pair
    a = sc.render(cam, prm)
    iters = sc.last_iterations()
    sc.set_drain_threshold(2000)              # pipeline, then the last <= 2000 streams run to completion in wf_drain
    a2 = sc.render(cam, prm)
    assert sc.last_iterations() < iters
    sc.set_drain_threshold(1 << 30)           # everything drained after the first poll
    a3 = sc.render(cam, prm)
    sc.set_drain_threshold(0)
    sc.set_shade_rounds(0)                    # one bounce evaluation per step: one more iteration per sample
    a4 = sc.render(cam, prm)
    sc.set_shade_rounds(1)
    a5 = sc.render(cam, prm)
    sc.set_shade_rounds(-1)
    sc.set_mode(0)
    b = sc.render(cam, prm)
    assert np.array_equal(bits(a), bits(b)) and np.array_equal(bits(a2), bits(b)) and np.array_equal(bits(a3), bits(b))
    assert np.array_equal(bits(a4), bits(b)) and np.array_equal(bits(a5), bits(b))
    assert 5 <= iters <= 5 * (12 + 8 + 3) + 8


def test_config4_glass_sphere_depth12_live():
    """BASELINE.json configs[3] analogue at test size: stand-in + analytic glass sphere, max depth 12."""
    prims = ptamd.gen_scene(1, 48)
    nodes, tris, _ = ptamd.build_bvh(prims)
    glass = make_test_spheres()[:1]                     # r=6 at (10,6,8), opacity 0, roughness 0 -> pure_refractive
    W, H = 128, 72
    img_o, _ = O.Scene(nodes.tobytes(), tris, glass).render(O.make_camera(W, H), O.make_params(W, H, 2, 8, max_bounce=12), 16)
    img_g = ptamd.Scene(nodes, tris, glass).render(ptamd.make_camera(W, H), ptamd.default_params(passes=2, spp_per_pass=8, max_bounce=12))
    _check_image(img_g, img_o, "config4")


def test_config5_four_instances_deep_tree():
    """BASELINE.json configs[4] analogue: 4 instanced stand-ins (278,268 triangles, reference tree depth 21):
    closest-hit table on 20k rays and a pixel window of the 3840x2160 frame, against the oracle."""
    rs = np.random.RandomState(5)
    prims = ptamd.gen_scene(2, 187)
    nodes, tris, depth = ptamd.build_bvh(prims)
    assert tris.shape[0] == 278268 and depth == 21
    so, sg = O.Scene(nodes.tobytes(), tris), ptamd.Scene(nodes, tris)
    rays = scene_rays8(20000, rs)
    h_o, p_o, _ = so.raycast(rays)
    h_g, p_g = sg.raycast(rays)
    assert np.array_equal(p_g, p_o) and same_bits_or_nan(h_g, h_o).all()
    W, H = 3840, 2160
    win = (1500, 1400, 1564, 1416)                       # 64x16 pixels over an instance
    img_o, _ = so.render(O.make_camera(W, H), O.make_params(W, H, 1, 2, window=win), 16)
    # render only the tiles of that window on the GPU: a 2-rank split would still render half the frame, so use the
    # whole-frame call at 1 spp x 2 ... 8.3M pixels x 2 spp is ~20 ms on the GPU
    full = sg.render(ptamd.make_camera(W, H), ptamd.default_params(passes=1, spp_per_pass=2))
    x0, y0, x1, y1 = win
    _check_image(full[y0:y1, x0:x1], img_o[y0:y1, x0:x1], "4K window")


def test_config4_full_frame_nan_pixels_are_the_references():
    """At BASELINE's config-4 size the reference's own arithmetic produces a few NaN pixels (the glass lobe
    divides by a vanishing pdf product on rare paths); they tone-map to black (saturate(NaN) = 0).  The HIP
    frame must have NaNs exactly where the oracle has them — here pass 1 of 1920x1080 x 256 spp, depth 12 —
    and identical bits around them."""
    prims = ptamd.gen_scene(1, 187)
    nodes, tris, _ = ptamd.build_bvh(prims)
    glass = make_test_spheres()[:1]
    W, H = 1920, 1080
    img = ptamd.Scene(nodes, tris, glass).render(ptamd.make_camera(W, H), ptamd.default_params(passes=1, spp_per_pass=256, max_bounce=12, first_pass=1))
    bad = np.argwhere(~np.isfinite(img).all(-1))
    assert [tuple(b) for b in bad.tolist()] == [(213, 645), (337, 1267)]
    so = O.Scene(nodes.tobytes(), tris, glass)
    for (y, x) in ((213, 645), (337, 1267)):
        win = (x - 1, y, x + 2, y + 1)                           # the NaN pixel and its two neighbours
        ref, _ = so.render(O.make_camera(W, H), O.make_params(W, H, 1, 256, max_bounce=12, first_pass=1, window=win), 3)
        assert same_bits_or_nan(img[y, x - 1:x + 2], ref[y, x - 1:x + 2]).all()
        assert np.isnan(ref[y, x]).all() and np.isfinite(ref[y, x - 1]).all()
    assert np.array_equal(ptamd.tonemap_u8(img[213:214, 645:646], 1), np.zeros((1, 1, 3), np.uint8))   # NaN -> black


def test_tiny_scenes_single_leaf_tree():
    """Trees whose root is a leaf (<= 4 triangles) and whose only geometry is the light."""
    cornell = ptamd.gen_scene(0)
    for prims in (cornell[10:12], cornell[8:12], cornell[[0, 1, 10, 11, 4]]):
        nodes, tris, _ = ptamd.build_bvh(prims)
        W, H = 40, 24
        img_o, _ = O.Scene(nodes.tobytes(), tris).render(O.make_camera(W, H), O.make_params(W, H, 2, 4), 4)
        sc = ptamd.Scene(nodes, tris)
        for mode in (1, 0):
            sc.set_mode(mode)
            _check_image(sc.render(ptamd.make_camera(W, H), ptamd.default_params(passes=2, spp_per_pass=4)), img_o, f"{len(prims)} tris mode {mode}")


@pytest.mark.parametrize("max_bounce,rr_bounce,spp,passes", [(1, 3, 1, 1), (1, 3, 7, 2), (2, 0, 5, 2), (3, 1, 6, 1), (8, 3, 1, 3)])
def test_short_paths_and_single_samples(max_bounce, rr_bounce, spp, passes):
    """The stream step overlaps consecutive samples (the next sample's first bounce is shaded in the step that ends
    a path).  Its corner cases: paths that end at their very first bounce (max_bounce 1: every step closes one
    sample and opens another), roulette from bounce 0, a single sample per pass, pixels that see only background
    — with analytic spheres (refraction chains) in the frame, in both render modes."""
    prims = ptamd.gen_scene(1, 16)
    nodes, tris, _ = ptamd.build_bvh(prims)
    sph = make_test_spheres()
    W, H = 96, 40
    img_o, _ = O.Scene(nodes.tobytes(), tris, sph).render(
        O.make_camera(W, H), O.make_params(W, H, passes, spp, max_bounce=max_bounce, rr_bounce=rr_bounce), 4)
    sc = ptamd.Scene(nodes, tris, sph)
    for mode, rounds in ((1, 1), (1, 0), (0, -1)):
        sc.set_mode(mode)
        sc.set_shade_rounds(rounds)
        img = sc.render(ptamd.make_camera(W, H), ptamd.default_params(passes=passes, spp_per_pass=spp, max_bounce=max_bounce, rr_bounce=rr_bounce))
        _check_image(img, img_o, f"max_bounce {max_bounce} rr {rr_bounce} spp {spp} mode {mode} shade rounds {rounds}")


def test_stream_triad_measures_a_plausible_bandwidth():
    gbps = ptamd.triad_gbps(1 << 28, 5)
    assert 500.0 < gbps < 9000.0, gbps
