"""GPU: the HIP path, through the C ABI, against the golden outputs of the real reference
(tests/golden/reference_outputs.npz) and size-independent properties at the benchmark's full size."""
import importlib
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "reference_outputs.npz"))


def _cam(rta, v):
    c = rta.rt_camera()
    for k in range(3):
        c.C[k], c.right[k], c.up[k], c.view[k] = v[k], v[3 + k], v[6 + k], v[9 + k]
    c.zdist, c.focus, c.aa_sigma, c.dof_sigma = v[12], v[13], v[14], v[15]
    return c


@pytest.mark.parametrize("name", ["room", "meshroom", "battle_small"])
def test_reference_hits_bit_exact(gpu, rta, scenes, ob, name):
    mg = importlib.import_module("tests.golden.make_golden")
    builder, kw = mg.scene_specs()[name]
    s = builder(**kw)[0]
    W = kw["W"]
    cam = _cam(rta, GOLD[name + "_cam"])
    gpu.upload(s)
    rays, _ = ob.primary_rays(cam, W, W)
    h = gpu.intersect_rays(rays)
    assert np.array_equal(h["obj"], GOLD[name + "_primary_obj"])
    assert np.array_equal(h["tri"], GOLD[name + "_primary_tri"])
    assert np.array_equal(h["t"], GOLD[name + "_primary_t"])
    h = gpu.intersect_rays(mg.random_rays(3000, 5))
    assert np.array_equal(h["obj"], GOLD[name + "_random_obj"])
    assert np.array_equal(h["tri"], GOLD[name + "_random_tri"])
    assert np.array_equal(h["t"], GOLD[name + "_random_t"])
    assert np.array_equal(h["N"], GOLD[name + "_random_N"], equal_nan=True)
    assert np.allclose(h["albedo"], GOLD[name + "_random_albedo"], rtol=1e-14, atol=0)


@pytest.mark.parametrize("name", ["room", "meshroom", "battle_small"])
def test_reference_sample_radiance(gpu, rta, scenes, ob, name):
    """Per-sample radiance of the REAL reference (fixture; it samples with the host's libm) against the GPU (which samples
    with the portable cos / sin / log of rt_math.h). Every sample beyond 1e-9 has to be accounted for: the oracle in libm
    mode reproduces the fixture bit for bit, so its path trace is the reference's path; the GPU's trace must leave it at
    a silhouette or a flipped branch behind a cos / sin / log-dependent direction (ob.explain_divergence), and samples
    with equal traces must agree."""
    mg = importlib.import_module("tests.golden.make_golden")
    builder, kw = mg.scene_specs()[name]
    s = builder(**kw)[0]
    W = kw["W"]
    cam = _cam(rta, GOLD[name + "_cam"])
    gpu.upload(s)
    ijk = mg.sample_list(W, 4 if name != "battle_small" else 2, 2)
    p = rta.make_params(W, W, 2)
    got, tg = gpu.sample_trace(cam, p, ijk)
    exp = GOLD[name + "_samples"]
    orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L)
    ref, tr = orc.sample_trace(cam, p, ijk, math="libm")
    assert np.array_equal(ref, exp, equal_nan=True)  # the oracle's libm mode IS the reference, here too
    both_nan = np.isnan(got) & np.isnan(exp)
    rel = np.abs(got - exp) / np.maximum(np.abs(exp), 1e-3)
    rel[both_nan] = 0
    rel[np.isnan(rel)] = np.inf
    rel = rel.max(axis=1)
    same = (tg == tr).all(axis=(1, 2))
    assert not (same & (rel > 1e-9)).any()
    for q in np.nonzero(rel > 1e-9)[0]:
        assert ob.explain_divergence(tr[q], tg[q]) is not None, (name, ijk[q], rel[q], tr[q], tg[q])
    assert (rel > 1e-9).sum() <= max(3, len(ijk) // 200)  # sanity only: every one of them is explained above


def test_hit_without_triangle(gpu, rta, scenes):
    """a triangle met at or beyond t = 1E9 makes the mesh report a hit at 1E9 with zeroed P, N and colour (raytracer.cpp:689-690)"""
    mg = importlib.import_module("tests.golden.make_golden")
    objs, rays = mg.far_scene()
    s = rta.Scene(I=1.0, L=(0.0, 0.0, 0.0))
    s.add(rta.Sphere(objs[0]["O"], objs[0]["R"], objs[0]["albedo"]))
    m = rta.TriangleMesh(objs[1]["albedo"])
    m.set_raw(objs[1]["verts"], objs[1]["normals"], objs[1]["uvs"], objs[1]["idx"])
    m.buildBVH()
    s.add(m)
    gpu.upload(s)
    for prune in (1, 0):
        rta.lib.rt_debug_probe_prune(gpu._h, prune)
        h = gpu.intersect_rays(rays)
        assert np.array_equal(h["obj"], GOLD["far_obj"])
        assert np.array_equal(h["t"], GOLD["far_t"])
        assert np.array_equal(h["tri"] < 0, GOLD["far_no_tri"])
        assert np.array_equal(np.concatenate([h["P"], h["N"], h["albedo"]], 1), GOLD["far_PNalbedo"], equal_nan=True)
    rta.lib.rt_debug_probe_prune(gpu._h, 1)


def test_flat_boxes(gpu, rta, scenes):
    """meshes in coordinate planes against the real reference's answers: a box without thickness is never entered"""
    mg = importlib.import_module("tests.golden.make_golden")
    s, cam, _ = scenes.axis_aligned()
    gpu.upload(s)
    rays = mg.axis_rays()
    for prune in (1, 0):
        rta.lib.rt_debug_probe_prune(gpu._h, prune)
        h = gpu.intersect_rays(rays)
        assert np.array_equal(h["obj"], GOLD["axis_obj"]) and np.array_equal(h["tri"], GOLD["axis_tri"]) and np.array_equal(h["t"], GOLD["axis_t"])
    rta.lib.rt_debug_probe_prune(gpu._h, 1)


def test_resolve_matches_reference(gpu, rta, scenes):
    acc = (scenes._hash01(np.arange(64 * 64 * 3), 3).reshape(64, 64, 3) ** 4) * 4e5
    acc[0, 0] = (np.nan, -1.0, 0.0)
    acc[0, 1] = (1e300, 255.0 ** (1 / 0.45), 2.0)
    gpu.accum_clear(64, 64)
    gpu.accum_upload(acc)
    got = gpu.resolve(64, 64, 3)
    exp = GOLD["resolve_expected_nbrays3"]
    assert got[63, 0, 0] == 255 and got[63, 0, 1] == 255 and got[63, 0, 2] == 0
    assert np.array_equal(got, exp)  # the tone map goes through thresholds made with the host's pow: the reference's bytes


class TestFullSize:
    """properties that need no CPU render, on the benchmark's own scene and frame size (1024 x 1024)"""

    @pytest.fixture(scope="class")
    def full(self, gpu, rta, scenes):
        s, cam, flags, meta = scenes.battle(W=1024)
        gpu.upload(s)
        return s, cam, flags

    def test_deterministic_and_rect_consistent(self, gpu, rta, full):
        s, cam, flags = full
        W = 1024
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 2, flags=flags))
        a = gpu.accum_download(W, W)
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 2, flags=flags))
        assert np.array_equal(a, gpu.accum_download(W, W), equal_nan=True)  # run-to-run identical
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 2, flags=flags, rect=(384, 512, 640, 704)))
        b = gpu.accum_download(W, W)
        assert np.array_equal(b[512:704, 384:640], a[512:704, 384:640], equal_nan=True)  # a tile alone == the tile in the frame
        assert not b[:512].any() and not b[704:].any()
        assert a.min() >= 0 and (a > 0).mean() > 0.9

    def test_sample_sharding_is_additive(self, gpu, rta, full):
        s, cam, flags = full
        W = 1024
        rect = (256, 384, 768, 640)
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 6, flags=flags, rect=rect))
        a = gpu.accum_download(W, W)
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 2, spp_offset=4, flags=flags, rect=rect))  # shards in any order
        gpu.render(cam, rta.make_params(W, W, 4, spp_offset=0, flags=flags, rect=rect))
        b = gpu.accum_download(W, W)
        assert np.allclose(a, b, rtol=1e-13, atol=1e-9)

    def test_sample_units_add_up(self, gpu, rta, full):
        """On a frame this large a launch hands out a pixel's samples in units of a few samples (added to the
        accumulator in sample order by a second kernel): 20 samples in one call == the same samples rendered in calls too short
        to be cut."""
        s, cam, flags = full
        W = 1024
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 20, flags=flags))
        a = gpu.accum_download(W, W)
        gpu.accum_clear(W, W)
        for k0, n in ((0, 8), (8, 8), (16, 4)):  # the units themselves, one call each: the same additions in the same order
            gpu.render(cam, rta.make_params(W, W, n, spp_offset=k0, flags=flags))
        b = gpu.accum_download(W, W)
        assert np.array_equal(a, b, equal_nan=True)
        gpu.accum_clear(W, W)
        for k0, n in ((0, 5), (5, 5), (10, 5), (15, 5)):
            gpu.render(cam, rta.make_params(W, W, n, spp_offset=k0, flags=flags))
        assert np.allclose(a, gpu.accum_download(W, W), rtol=1e-13, atol=1e-9, equal_nan=True)
        assert (a > 0).mean() > 0.9
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 20, flags=flags))
        assert np.array_equal(a, gpu.accum_download(W, W), equal_nan=True)  # run-to-run identical

    def test_pruned_equals_unpruned_and_counted(self, gpu, rta, full):
        s, cam, flags = full
        W = 1024
        rect = (448, 448, 576, 576)
        out = []
        for fl in (0, rta.RT_F_NO_PRUNE, rta.RT_F_COUNTERS):
            gpu.accum_clear(W, W)
            gpu.render(cam, rta.make_params(W, W, 3, flags=flags | fl, rect=rect))
            out.append(gpu.accum_download(W, W))
        assert np.array_equal(out[0], out[1], equal_nan=True)
        assert np.array_equal(out[0], out[2], equal_nan=True)

    def test_tile_against_oracle(self, gpu, rta, scenes, ob, full):
        """a 48 x 48 tile of the full-size frame against the CPU oracle, per sample"""
        s, cam, flags = full
        W = 1024
        orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L, math="portable")
        ijk = np.array([(i, j, k) for i in range(600, 648, 2) for j in range(200, 248, 2) for k in range(2)], np.int32)
        p = rta.make_params(W, W, 2, flags=flags)
        (g, tg), (o, to) = gpu.sample_trace(cam, p, ijk), orc.sample_trace(cam, p, ijk)
        rel = np.abs(g - o) / np.maximum(np.abs(o), 1e-3)
        assert (rel > 1e-9).sum() == 0, np.sort(rel.ravel())[-5:]
        assert np.array_equal(tg, to)

    def test_converged_tile_256spp(self, gpu, rta, scenes, ob, full, converged_crop_check):
        """a 128 x 128 tile of the full cfg4 frame (soldiers, tank, floor, textures) at 256 samples with independent seeds:
        measured RMSE 4.52 against a floor of 4.49 levels, bias 0.02 (bars: 1.15 x floor, 0.25 level)"""
        s, cam, flags = full
        orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L, math="portable")
        converged_crop_check(gpu, rta, ob, orc, cam, flags, 1024, (448, 512, 576, 640), 256, "cfg4 tile")

    def test_all_primary_rays_against_oracle(self, gpu, rta, scenes, ob, full):
        """every camera ray of the 1024 x 1024 frame through the full 698 316-triangle scene: object id, triangle id and t
        equal to the oracle's bit for bit (the oracle's Scene::intersect is pinned to the real reference's)"""
        s, cam, flags = full
        W = 1024
        orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L)
        rays, _ = ob.primary_rays(cam, W, W)
        g, o = gpu.intersect_rays(rays), orc.intersect_rays(rays)
        assert (o["tri"] >= 0).sum() > 100000
        assert np.array_equal(g["obj"], o["obj"]) and np.array_equal(g["tri"], o["tri"]) and np.array_equal(g["t"], o["t"])
        hit = o["obj"] >= 0
        assert np.array_equal(g["N"][hit], o["N"][hit], equal_nan=True) and np.array_equal(g["albedo"][hit], o["albedo"][hit])


def test_sample_units_with_the_sphere_tree(gpu, rta, scenes):
    """the same property for the kernel that walks the tree over a long sphere list (1024 x 1024: large enough for units)"""
    W = 1024
    s, cam, flags = scenes.stress(W=W, n_tris=20000, n_spheres=150)[:3]
    gpu.upload(s)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 12, flags=flags))
    a = gpu.accum_download(W, W)
    gpu.accum_clear(W, W)
    for k0, n in ((0, 8), (8, 4)):
        gpu.render(cam, rta.make_params(W, W, n, spp_offset=k0, flags=flags))
    b = gpu.accum_download(W, W)
    assert np.array_equal(a, b, equal_nan=True)
    assert (a > 0).mean() > 0.9


def test_sample_units_on_small_frames(gpu, rta, scenes):
    """512 x 512 (the reference's default frame) is two pixels per resident lane on an MI355X: units of 8 samples, the
    sums equal to those of calls of one unit each and identical from run to run. 128 x 128: units of one sample, so any
    split of the samples over calls gives the same bits."""
    W = 512
    s, cam, flags = scenes.sphere_room(W)
    gpu.upload(s)
    out = []
    for _ in range(2):
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 24, flags=flags))
        out.append(gpu.accum_download(W, W))
    assert np.array_equal(out[0], out[1], equal_nan=True)
    gpu.accum_clear(W, W)
    for k0 in (0, 8, 16):
        gpu.render(cam, rta.make_params(W, W, 8, spp_offset=k0, flags=flags))
    assert np.array_equal(out[0], gpu.accum_download(W, W), equal_nan=True)
    W = 128
    s, cam, flags = scenes.sphere_room(W)
    gpu.upload(s)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 16, flags=flags))
    a = gpu.accum_download(W, W)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 5, spp_offset=0, flags=flags))
    gpu.render(cam, rta.make_params(W, W, 11, spp_offset=5, flags=flags))
    assert np.array_equal(a, gpu.accum_download(W, W), equal_nan=True)
