"""GPU: the synthetic stress configuration at FULL size - one 5-million-triangle mesh + 1000 random spheres + light + walls,
2048 x 2048 - against the CPU oracle and through size-independent properties. The scene is built once for the class."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
W = 2048


@pytest.fixture(scope="module")
def stress(gpu, rta, scenes, ob):
    s, cam, flags = scenes.stress()  # defaults: 5 000 000 triangles, 1000 spheres, 2048 x 2048
    info = s.flat_info()
    assert info["n_tris"] >= 4_900_000 and info["n_spheres"] == 1006
    gpu.upload(s)
    orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L, math="portable")
    return s, cam, flags, orc


def _same_hits(g, o):
    assert np.array_equal(g["obj"], o["obj"]) and np.array_equal(g["tri"], o["tri"]) and np.array_equal(g["t"], o["t"])  # bit-exact
    hit = o["obj"] >= 0
    assert np.array_equal(g["N"][hit], o["N"][hit], equal_nan=True) and np.array_equal(g["P"][hit], o["P"][hit])


def test_primary_and_incoherent_rays(gpu, rta, scenes, ob, stress):
    s, cam, flags, orc = stress
    rays, _ = ob.primary_rays(cam, W, W)
    sel = np.ascontiguousarray(rays[::97])  # 43 000 camera rays spread over the frame
    o = orc.intersect_rays(sel)
    assert (o["tri"] >= 0).sum() > 3000 and len(np.unique(o["obj"])) > 300
    for prune in (1, 0):
        rta.lib.rt_debug_probe_prune(gpu._h, prune)
        _same_hits(gpu.intersect_rays(sel), o)
    rta.lib.rt_debug_probe_prune(gpu._h, 1)
    inc = scenes.rays_into_box(30000, 41, (-40, -5, -55), (40, 40, 50), (-30, -16, -40), (30, 10, 25))
    o = orc.intersect_rays(inc)
    assert (o["tri"] >= 0).sum() > 3000
    _same_hits(gpu.intersect_rays(inc), o)


def test_per_sample_radiance_and_paths(gpu, rta, stress):
    s, cam, flags, orc = stress
    ijk = np.array([(i, j, k) for i in range(37, W, 53) for j in range(11, W, 53) for k in range(2)], np.int32)  # 2 x 38 x 39 samples
    p = rta.make_params(W, W, 2, flags=flags)
    (g, tg), (o, to) = gpu.sample_trace(cam, p, ijk), orc.sample_trace(cam, p, ijk)
    both_nan = np.isnan(g) & np.isnan(o)
    rel = np.abs(g - o) / np.maximum(np.abs(o), 1e-3)
    rel[both_nan] = 0
    rel[np.isnan(rel)] = np.inf
    assert (rel > 1e-9).sum() == 0, np.sort(rel.ravel())[-5:]
    assert np.array_equal(tg, to)
    assert (to[:, 1, 3] != 0).mean() > 0.5  # most paths go on beyond the first hit


def test_full_frame_properties(gpu, rta, stress):
    """2048 x 2048: run-to-run identical; a rect alone == the rect within the frame; pruned == unpruned == counted"""
    s, cam, flags, orc = stress
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 2, flags=flags))
    a = gpu.accum_download(W, W)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 2, flags=flags))
    assert np.array_equal(a, gpu.accum_download(W, W), equal_nan=True)
    assert (a > 0).mean() > 0.9
    rect = (900, 1100, 1156, 1292)
    for fl in (0, rta.RT_F_NO_PRUNE, rta.RT_F_COUNTERS):
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 2, flags=flags | fl, rect=rect))
        b = gpu.accum_download(W, W)
        assert np.array_equal(b[1100:1292, 900:1156], a[1100:1292, 900:1156], equal_nan=True), fl
        assert not b[:1100].any()


def test_tile_of_the_frame_against_the_oracle(gpu, rta, ob, stress):
    """a 24 x 24 pixel tile, 2 samples, summed by rt_render: equal to the oracle's pixel loop on the same tile"""
    s, cam, flags, orc = stress
    rect = (1000, 1200, 1024, 1224)
    p = rta.make_params(W, W, 2, flags=flags, rect=rect)
    gpu.accum_clear(W, W)
    gpu.render(cam, p)
    g = gpu.accum_download(W, W)[1200:1224, 1000:1024]
    o = orc.render(cam, p)[1200:1224, 1000:1024]
    rel = np.abs(g - o) / np.maximum(np.abs(o), 1e-3)
    assert np.nanmax(rel) < 1e-9 and np.array_equal(np.isnan(g), np.isnan(o))
