"""CPU: the oracle (oracle/ref_cpu.cpp) against the golden outputs of the real reference
(tests/golden/reference_outputs.npz, made by tests/golden/make_golden.py). Hit identity and hit
parameter must be bit-exact; radiance may differ in the last bits where the libm of this machine
differs from the one the fixture was made with."""
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = np.load(os.path.join(HERE, "golden", "reference_outputs.npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def cam_from(rta, v):
    c = rta.rt_camera()
    for k in range(3):
        c.C[k], c.right[k], c.up[k], c.view[k] = v[k], v[3 + k], v[6 + k], v[9 + k]
    c.zdist, c.focus, c.aa_sigma, c.dof_sigma = v[12], v[13], v[14], v[15]
    return c


def build(scenes, name):
    import importlib
    mg = importlib.import_module("tests.golden.make_golden")
    builder, kw = mg.scene_specs()[name]
    res = builder(**kw)
    return res[0], kw["W"], mg


@pytest.mark.parametrize("name", ["room", "meshroom", "battle_small"])
def test_hits_bit_exact(rta, scenes, ob, name):
    s, W, mg = build(scenes, name)
    cam = cam_from(rta, GOLD[name + "_cam"])
    o = ob.OracleScene(scenes.export_objects(s), s.I, s.L)
    rays, _ = ob.primary_rays(cam, W, W)
    h = o.intersect_rays(rays)
    assert np.array_equal(h["obj"], GOLD[name + "_primary_obj"])
    assert np.array_equal(h["tri"], GOLD[name + "_primary_tri"])
    assert np.array_equal(h["t"], GOLD[name + "_primary_t"])
    h = o.intersect_rays(mg.random_rays(3000, 5))
    assert np.array_equal(h["obj"], GOLD[name + "_random_obj"])
    assert np.array_equal(h["tri"], GOLD[name + "_random_tri"])
    assert np.array_equal(h["t"], GOLD[name + "_random_t"])
    assert np.array_equal(h["N"], GOLD[name + "_random_N"], equal_nan=True)
    assert np.allclose(h["albedo"], GOLD[name + "_random_albedo"], rtol=1e-14, atol=0)  # pow() of the texture table


@pytest.mark.parametrize("name", ["room", "meshroom", "battle_small"])
def test_sample_radiance(rta, scenes, ob, name):
    s, W, mg = build(scenes, name)
    cam = cam_from(rta, GOLD[name + "_cam"])
    o = ob.OracleScene(scenes.export_objects(s), s.I, s.L)
    ijk = mg.sample_list(W, 4 if name != "battle_small" else 2, 2)
    got = o.sample_radiance(cam, rta.make_params(W, W, 2), ijk)
    exp = GOLD[name + "_samples"]
    both_nan = np.isnan(got) & np.isnan(exp)
    rel = np.abs(got - exp) / np.maximum(np.abs(exp), 1e-3)
    rel[both_nan] = 0
    rel[np.isnan(rel)] = np.inf
    assert (rel > 1e-9).sum() <= 3, np.sort(rel.ravel())[-5:]


@pytest.mark.parametrize("name", ["meshroom", "battle_small"])
def test_bvh_matches_reference(rta, scenes, ob, name):
    s, W, mg = build(scenes, name)
    objs = scenes.export_objects(s)
    o = ob.OracleScene(objs, s.I, s.L)
    for oid in o.mesh_ids:
        n_nodes, n_leaves = GOLD["%s_bvh_%d" % (name, oid)]
        info = o.bvh_info(oid, len(objs[oid]["idx"]))
        assert info["n_nodes"] == n_nodes and info["n_leaves"] == n_leaves
        boxes, dfl = o.bvh_dump(oid, int(n_nodes))
        hb, hd, hp = GOLD["%s_bvhhash_%d" % (name, oid)]
        assert sha(boxes) == hb and sha(dfl) == hd and sha(info["perm"]) == hp


def _check_far(h):
    assert np.array_equal(h["obj"], GOLD["far_obj"])
    assert np.array_equal(h["t"], GOLD["far_t"])
    assert np.array_equal(h["tri"] < 0, GOLD["far_no_tri"])
    assert np.array_equal(np.concatenate([h["P"], h["N"], h["albedo"]], 1), GOLD["far_PNalbedo"], equal_nan=True)
    assert ((h["obj"] == 1) & (h["tri"] < 0) & (h["t"] == 1e9)).sum() > 300


def test_hit_without_triangle(scenes, ob):
    """a triangle met at or beyond t = 1E9 makes the mesh report a hit at 1E9 with zeroed P, N and colour (raytracer.cpp:689-690)"""
    import importlib
    objs, rays = importlib.import_module("tests.golden.make_golden").far_scene()
    _check_far(ob.OracleScene(objs, 1.0, (0, 0, 0)).intersect_rays(rays))


def test_flat_boxes(scenes, ob):
    """meshes in coordinate planes: a box without thickness is never entered (raytracer.cpp:184)"""
    import importlib
    mg = importlib.import_module("tests.golden.make_golden")
    s, cam, _ = scenes.axis_aligned()
    h = ob.OracleScene(scenes.export_objects(s), s.I, s.L).intersect_rays(mg.axis_rays())
    assert np.array_equal(h["obj"], GOLD["axis_obj"]) and np.array_equal(h["tri"], GOLD["axis_tri"]) and np.array_equal(h["t"], GOLD["axis_t"])
    assert not np.any(h["obj"] == 2) and (h["obj"] == 3).sum() > 200


def test_resolve(scenes, ob):
    acc = (scenes._hash01(np.arange(64 * 64 * 3), 3).reshape(64, 64, 3) ** 4) * 4e5
    acc[0, 0] = (np.nan, -1.0, 0.0)
    acc[0, 1] = (1e300, 255.0 ** (1 / 0.45), 2.0)
    got = ob.resolve(acc, 3)
    exp = GOLD["resolve_expected_nbrays3"]
    assert got[63, 0, 0] == 255 and got[63, 0, 1] == 255 and got[63, 0, 2] == 0  # NaN -> 255 (std::min keeps its first argument)
    assert np.abs(got.astype(int) - exp.astype(int)).max() <= 1 and (got != exp).mean() < 1e-3


def test_rng_known_answers(ob):
    """the counter-based stream: fixed values, range, and the two Monte-Carlo integrals the reference
    carries as exercises (raytracer.cpp:1096-1151: 63 pi / 256 and pi^4 / 2) with Box-Muller sampling"""
    key = ob.orc.orc_rng_key(10, 12345, 7)
    assert key == ob.orc.orc_rng_key(10, 12345, 7) != ob.orc.orc_rng_key(10, 12345, 8)
    u = np.array([ob.orc.orc_rng_uniform(key, n) for n in range(20000)])
    assert u.min() >= 0 and u.max() < 1 and abs(u.mean() - 0.5) < 0.01 and abs(u.var() - 1 / 12) < 0.005
    assert ob.orc.orc_rng_draw(0, 0) == 0xE220A8397B1DCDAF  # splitmix64(0): first output of the published generator
    sigma, n = 0.25, 10000
    u1, u2 = u[0:2 * n:2], u[1:2 * n:2]
    xi = sigma * np.cos(2 * np.pi * u1) * np.sqrt(-2 * np.log(u2))
    ok = np.abs(xi) <= np.pi / 2
    p = np.exp(-xi * xi / (2 * sigma * sigma)) / (sigma * np.sqrt(2 * np.pi))
    est = (np.cos(xi[ok]) ** 10 / p[ok]).sum() / n
    assert abs(est - 63 * np.pi / 256) < 0.03
