"""GPU: the single-precision render (rt_params.flags = RT_F_FAST). It promises no bits, so it is held to the statistical bar
of the double-precision render - against the CPU oracle with independent seeds at 256 samples: |mean signed difference| <
0.25 level of 255, RMSE <= 1.15 x the oracle-vs-oracle noise floor - and, with the same seed, to staying on the
double-precision render's paths almost everywhere (same random stream, same estimator: the two images differ only where a
hit or a branch flips in the last digits)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _pair(gpu, rta, cam, W, spp, flags=0, rect=None):
    out = []
    for fl in (0, rta.RT_F_FAST):
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, spp, flags=flags | fl, rect=rect))
        out.append((gpu.accum_download(W, W), gpu.resolve(W, W, spp).astype(np.float64)))
    return out


@pytest.mark.parametrize("name,kw,rmse_bar", [("sphere_room", dict(W=256), 2.0), ("mesh_room", dict(W=256, n_tris=20000, tex_size=128), 3.0),
                                              ("battle", dict(W=256, tri_scale=0.1, tex_size=64), 2.0)])
def test_same_seed_stays_on_the_double_precision_paths(gpu, rta, scenes, name, kw, rmse_bar):
    W = kw["W"]
    out = getattr(scenes, name)(**kw)
    s, cam, flags = out[0], out[1], out[2]
    gpu.upload(s)
    (a64, i64), (af, ifast) = _pair(gpu, rta, cam, W, 32, flags)
    d = ifast - i64
    print("%s: fast - f64 with one seed: mean %.4f, RMSE %.3f levels, radiance ratio %.5f" % (name, d.mean(), np.sqrt((d * d).mean()), np.nansum(af) / np.nansum(a64)))
    assert np.array_equal(np.isnan(af).any(2), np.isnan(a64).any(2)) or np.isnan(af).mean() < 1e-3  # NaN pixels (zero tangent frames) are the reference's too
    assert abs(d.mean()) < 0.1
    assert np.sqrt((d * d).mean()) < rmse_bar
    assert abs(np.nansum(af) / np.nansum(a64) - 1) < 2e-3
    # run-to-run identical
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 32, flags=flags | rta.RT_F_FAST))
    assert np.array_equal(gpu.accum_download(W, W), af, equal_nan=True)
    # samples-per-pixel sharding works as for the double-precision render: two calls on unit boundaries == one call
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 16, flags=flags | rta.RT_F_FAST))
    gpu.render(cam, rta.make_params(W, W, 16, spp_offset=16, flags=flags | rta.RT_F_FAST))
    assert np.allclose(gpu.accum_download(W, W), af, rtol=1e-12, atol=0, equal_nan=True)


def test_converged_crop_cfg2_fast(gpu, rta, scenes, ob, converged_crop_check):
    W = 512
    s, cam, flags = scenes.sphere_room(W=W)[:3]
    orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L, math="portable")
    gpu.upload(s)
    converged_crop_check(gpu, rta, ob, orc, cam, flags, W, (192, 192, 320, 320), 256, "cfg2, single precision", gpu_flags=rta.RT_F_FAST)


def test_converged_tile_cfg4_fast(gpu, rta, scenes, ob, converged_crop_check):
    """the full battle scene at its frame size, the tile of test_gpu_golden.py's double-precision check"""
    W = 1024
    s, cam, flags, meta = scenes.battle(W=W)
    orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L, math="portable")
    gpu.upload(s)
    converged_crop_check(gpu, rta, ob, orc, cam, flags, W, (448, 512, 576, 640), 256, "cfg4 tile, single precision", gpu_flags=rta.RT_F_FAST)


def test_fast_mode_refusals(gpu, rta, scenes):
    W = 64
    s, cam, flags = scenes.sphere_room(W=W)[:3]
    gpu.upload(s)
    gpu.accum_clear(W, W)
    with pytest.raises(rta.RTError, match="combined"):
        gpu.render(cam, rta.make_params(W, W, 2, flags=rta.RT_F_FAST | rta.RT_F_NO_INDIRECT))
    with pytest.raises(rta.RTError, match="rt_render_static"):
        gpu.render_static(cam, rta.make_params(W, W, 2, flags=rta.RT_F_FAST))
    s2, cam2, fl2 = scenes.stress(W=W, n_tris=2000, n_spheres=100)[:3]
    gpu.upload(s2)
    gpu.accum_clear(W, W)
    with pytest.raises(rta.RTError, match="sphere lists"):
        gpu.render(cam2, rta.make_params(W, W, 2, flags=rta.RT_F_FAST))
    gpu.render(cam2, rta.make_params(W, W, 2))  # the double-precision render takes the same scene
    gpu.sync()


def test_fast_mode_on_random_scene_mixes(gpu, rta):
    """the random scene mixes of tests/fuzz_parity.py (nested and repeated spheres, doubled / flat / repeated meshes, mirror and
    glass everywhere) through the single-precision kernel: no stack or index fault (rt_sync reports them), and the frame's
    radiance within a few percent of the double-precision render with the same seed - whole paths differ where a hit
    flips, so this is a sanity bound, not a parity bar; both leaf sizes of the filter tree"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    W, spp, done = 96, 8, 0
    cam = rta.make_camera((0, 18, 60), 60.0, -10.0, 0.0, W)
    for seed in range(3, 40):
        s, verts, n_sph, n_mesh, notes = fuzz.make_scene(seed)
        if s.flat_info()["n_spheres"] > 64:
            continue
        for own_leaf in (0, 4):
            assert rta.lib.rt_debug_own_leaf(gpu._h, own_leaf) == 0
            try:
                gpu.upload(s)
                (a64, i64), (af, ifast) = _pair(gpu, rta, cam, W, spp)
                gpu.sync()
            finally:
                rta.lib.rt_debug_own_leaf(gpu._h, 0)
            ok64, okf = np.isfinite(a64).all(2), np.isfinite(af).all(2)
            assert (ok64 != okf).mean() < 0.02, (seed, own_leaf)
            both = ok64 & okf
            if a64[both].sum() > 0:
                ratio = af[both].sum() / a64[both].sum()
                assert abs(ratio - 1) < 0.05, (seed, own_leaf, ratio, notes)
        done += 1
    assert done >= 15
