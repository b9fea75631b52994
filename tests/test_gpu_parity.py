"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same seeded inputs. Bars: hit identity (object id, triangle id) and hit parameter t bit-exact;
noise-free radiance within 1e-4 relative per pixel (in practice ~1e-15); per-sample stochastic
radiance within 1e-9 relative for all but a handful of samples (libm differences in sin/cos/log can
move a ray across a silhouette or flip a Fresnel coin)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scene_pair(ob, scenes, builder, **kw):
    out = builder(**kw)
    s, cam, flags = out[0], out[1], out[2]
    objs = scenes.export_objects(s)
    return s, cam, flags, ob.OracleScene(objs, s.I, s.L)


@pytest.fixture(params=["prune", "noprune"])
def probe_mode(request, gpu, rta):
    """rt_intersect_rays with distance pruning (the default) and visiting every node as the reference does"""
    rta.lib.rt_debug_probe_prune(gpu._h, 1 if request.param == "prune" else 0)
    yield request.param
    rta.lib.rt_debug_probe_prune(gpu._h, 1)


def _check_hits(g, o):
    assert np.array_equal(g["obj"], o["obj"])
    assert np.array_equal(g["tri"], o["tri"])
    assert np.array_equal(g["t"], o["t"])  # bit-exact
    hit = o["obj"] >= 0
    assert np.array_equal(g["P"][hit], o["P"][hit])
    assert np.array_equal(g["N"][hit], o["N"][hit], equal_nan=True)
    assert np.array_equal(g["albedo"][hit], o["albedo"][hit])


def _rel_err(a, b):
    """per-sample relative error, NaN == NaN counts as equal (axis-aligned normals, random_cos)"""
    both_nan = np.isnan(a) & np.isnan(b)
    d = np.abs(a - b) / np.maximum(np.abs(b), 1e-3)
    d[both_nan] = 0
    d[np.isnan(d)] = np.inf
    return d.max(axis=-1)


def test_wave_scans(gpu, rta):
    """the DPP prefix sums the triangle step relies on"""
    import ctypes as C
    rng = np.random.default_rng(1)
    for trial in range(8):
        v = rng.integers(-1, 300, 64).astype(np.int32)
        if trial == 0:
            v[:] = 1
        a = np.empty(64, np.int32); m = np.empty(64, np.int32)
        rc = rta.lib.rt_debug_wave_scan(gpu._h, v.ctypes.data_as(C.c_void_p), a.ctypes.data_as(C.c_void_p), m.ctypes.data_as(C.c_void_p))
        assert rc == 0
        assert np.array_equal(a, np.cumsum(v)), (v, a)
        assert np.array_equal(m, np.maximum.accumulate(v)), (v, m)


@pytest.mark.parametrize("W", [64, 257])
def test_primary_hits_sphere_room(gpu, rta, scenes, ob, W, probe_mode):
    s, cam, flags, orc = _scene_pair(ob, scenes, scenes.sphere_room, W=W)
    gpu.upload(s)
    rays, _ = ob.primary_rays(cam, W, W)
    _check_hits(gpu.intersect_rays(rays), orc.intersect_rays(rays))


def test_primary_hits_mesh_room(gpu, rta, scenes, ob, probe_mode):
    W = 160
    s, cam, flags, orc = _scene_pair(ob, scenes, scenes.mesh_room, W=W, n_tris=30000, tex_size=128)
    gpu.upload(s)
    rays, _ = ob.primary_rays(cam, W, W)
    g, o = gpu.intersect_rays(rays), orc.intersect_rays(rays)
    assert (o["tri"] >= 0).sum() > 500
    _check_hits(g, o)


def test_random_rays_battle(gpu, rta, scenes, ob, probe_mode):
    """incoherent rays from inside the room against the 12-mesh scene (reduced triangle count)"""
    s, cam, flags, meta = scenes.battle(W=64, tri_scale=0.1, tex_size=64)
    orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L)
    gpu.upload(s)
    rng = np.random.default_rng(5)
    n = 20000
    org = np.stack([rng.uniform(-30, 30, n), rng.uniform(-5, 40, n), rng.uniform(-40, 50, n)], 1)
    d = rng.normal(size=(n, 3))
    d /= np.sqrt((d * d).sum(1))[:, None]
    rays = np.concatenate([org, d], 1)
    g, o = gpu.intersect_rays(rays), orc.intersect_rays(rays)
    assert (o["tri"] >= 0).sum() > 300
    _check_hits(g, o)


def test_degenerate_rays(gpu, rta, scenes, ob, probe_mode):
    """axis-parallel directions (division by zero in the slab test), zero direction, NaN"""
    s, cam, flags, orc = _scene_pair(ob, scenes, scenes.mesh_room, W=64, n_tris=3000, tex_size=32)
    gpu.upload(s)
    rays = np.array([[0, 30, 55, 0, 0, -1], [0, 30, 55, 0, -1, 0], [0, 30, 55, 1, 0, 0], [0, -2, 0, 0, 0, -1], [0, 5, 0, 0, -1, 0],
                     [0, 30, 55, 0, 0, 0], [0, 30, 55, np.nan, 0, -1], [0, 0, 0, 0, 1, 0], [0, 30, 55, 0, -0.5, -1]], np.float64)
    _check_hits(gpu.intersect_rays(rays), orc.intersect_rays(rays))
    assert gpu.intersect_rays(np.empty((0, 6)))["obj"].shape == (0,)


def test_deterministic_room_radiance(gpu, rta, scenes, ob):
    """cfg1 at reduced size: noise-off path (point light, hard shadows, mirror, refraction), 1 spp"""
    W = 128
    s, cam, flags, orc = _scene_pair(ob, scenes, scenes.sphere_room_deterministic, W=W)
    gpu.upload(s)
    p = rta.make_params(W, W, 1, flags=flags)
    gpu.accum_clear(W, W)
    gpu.render(cam, p)
    g = gpu.accum_download(W, W)
    o = orc.render(cam, p)
    assert o.max() > 1e3 and (o > 0).mean() > 0.5
    rel = np.abs(g - o) / np.maximum(np.abs(o), 1e-6)
    assert rel.max() < 1e-4, rel.max()  # the bar; measured ~1e-15
    assert rel.max() < 1e-12
    assert np.array_equal(gpu.resolve(W, W, 1), ob.resolve(o, 1))


@pytest.mark.parametrize("name,kw", [("sphere_room", dict(W=96)), ("mesh_room", dict(W=96, n_tris=20000, tex_size=128))])
def test_stochastic_per_sample(gpu, rta, scenes, ob, name, kw):
    s, cam, flags, orc = _scene_pair(ob, scenes, getattr(scenes, name), **kw)
    gpu.upload(s)
    W = kw["W"]
    p = rta.make_params(W, W, 4, flags=flags)
    ijk = np.array([(i, j, k) for i in range(0, W, 2) for j in range(0, W, 2) for k in range(4)], np.int32)
    g = gpu.sample_radiance(cam, p, ijk)
    o = orc.sample_radiance(cam, p, ijk)
    rel = _rel_err(g, o)
    bad = (rel > 1e-9).sum()
    assert bad <= max(2, len(ijk) // 2000), (bad, len(ijk), np.sort(rel)[-5:])


def test_render_equals_sum_of_samples(gpu, rta, scenes, ob):
    """rt_render adds samples k = spp_offset.. in order; rect and spp_offset sharding agree with the oracle"""
    W = 64
    s, cam, flags, orc = _scene_pair(ob, scenes, scenes.sphere_room, W=W)
    gpu.upload(s)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 3, spp_offset=0, rect=(8, 16, 56, 48)))
    gpu.render(cam, rta.make_params(W, W, 2, spp_offset=3, rect=(8, 16, 56, 48)))
    g = gpu.accum_download(W, W)
    o = orc.render(cam, rta.make_params(W, W, 5, rect=(8, 16, 56, 48)))
    assert np.all(g[:16] == 0) and np.all(g[:, :8] == 0) and np.all(g[48:] == 0) and np.all(g[:, 56:] == 0)
    rel = np.abs(g - o) / np.maximum(np.abs(o), 1e-3)
    assert (rel > 1e-9).sum() <= 6, np.sort(rel.ravel())[-8:]


def test_counters_match_oracle(gpu, rta, scenes, ob):
    """rays / box tests / nodes / triangle tests counted under the reference's traversal semantics"""
    W = 64
    s, cam, flags, orc = _scene_pair(ob, scenes, scenes.mesh_room, W=W, n_tris=5000, tex_size=32)
    gpu.upload(s)
    p = rta.make_params(W, W, 2, flags=rta.RT_F_COUNTERS | rta.RT_F_NO_PRUNE)
    gpu.reset_counters()
    gpu.accum_clear(W, W)
    gpu.render(cam, p)
    gc = gpu.counters()
    oc = rta.rt_counters()
    orc.render(cam, p, counters=oc)
    oc = oc.as_dict()
    assert gc["paths"] == oc["paths"] == W * W * 2
    for k in ("rays", "box_tests", "nodes", "tris", "mesh_calls", "diffuse", "specular"):
        assert abs(gc[k] - oc[k]) <= 1e-4 * oc[k] + 50, (k, gc[k], oc[k])


def test_pruning_changes_work_not_results(gpu, rta, scenes, ob):
    """default (pruned, any-hit shadow rays) render == RT_F_NO_PRUNE render bit for bit, with fewer node visits"""
    W = 96
    s, cam, flags, orc = _scene_pair(ob, scenes, scenes.mesh_room, W=W, n_tris=20000, tex_size=64)
    gpu.upload(s)
    out = {}
    for name, fl in (("prune", 0), ("noprune", rta.RT_F_NO_PRUNE)):
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 3, flags=fl))
        out[name] = gpu.accum_download(W, W)
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 3, flags=fl | rta.RT_F_WAVEFRONT))
        assert np.array_equal(gpu.accum_download(W, W), out[name], equal_nan=True)  # two-kernel form == single kernel
        gpu.reset_counters()
        gpu.accum_clear(W, W)
        gpu.render(cam, rta.make_params(W, W, 3, flags=fl | rta.RT_F_COUNTERS))
        out[name + "_cnt"] = gpu.counters()
        assert np.array_equal(gpu.accum_download(W, W), out[name], equal_nan=True)  # counted variant == production variant
    assert np.array_equal(out["prune"], out["noprune"], equal_nan=True)
    assert out["prune_cnt"]["rays"] == out["noprune_cnt"]["rays"]
    assert out["prune_cnt"]["nodes"] < out["noprune_cnt"]["nodes"]
    assert out["prune_cnt"]["tris"] < out["noprune_cnt"]["tris"]


def test_errors(gpu, rta, scenes):
    s = rta.Scene()
    m = rta.TriangleMesh((1, 1, 1))
    s.add(m)  # objects[0] is not a sphere and the mesh has no BVH
    with pytest.raises(rta.RTError):
        gpu.upload(s)
    s2, cam, flags = scenes.sphere_room_deterministic(32)
    gpu.upload(s2)
    with pytest.raises(rta.RTError):  # area-light estimator without a light sphere at objects[0]... objects[0] IS a sphere here
        gpu.render(cam, rta.make_params(32, 32, 1, rect=(0, 0, 64, 64)))
    with pytest.raises(rta.RTError):
        gpu.render(cam, rta.make_params(32, 32, 1))  # accumulator not allocated for 32x32


def test_cpp_host_program_matches_python_path(gpu, rta, scenes, tmp_path):
    """examples/render_room.cpp = the reference's main() against rt_host.hpp (readOBJ, PPM loadTexture, transforms,
    buildBVH, rt::render). Same bytes as the same scene built through the Python binding."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "examples", "render_room")
    assert os.path.exists(exe), "build() makes it"
    v, n, uv, ix = scenes.blob_mesh(12, size=(0.3, 0.6, 0.72), offset=(0, 0, 0.72), seed=5)
    obj, tex, out = str(tmp_path / "m.obj"), str(tmp_path / "t.ppm"), str(tmp_path / "o.ppm")
    scenes.write_obj(obj, v, n, uv, ix)
    t = scenes.procedural_texture(64, 3)
    with open(tex, "wb") as f:
        f.write(b"P6\n# made by the test\n64 64\n255\n" + t.tobytes())
    W, spp = 96, 3
    r = subprocess.run([exe, out, str(W), str(spp), obj, tex], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    data = open(out, "rb").read()
    img = np.frombuffer(data[-W * W * 3:], np.uint8).reshape(W, W, 3)
    s, cam, flags = scenes.sphere_room(W)
    m = rta.TriangleMesh((0, 0, 1)).readOBJ(obj).loadTexture(tex)
    m.zoom(4).rotate(1, 2).rotate(0, 2).translate(1, -10).rotateAroundAxe(1, -30).translate(0, 5).translate(2, 25).buildBVH()
    s.add(m)
    gpu.upload(s)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, spp))
    assert np.array_equal(gpu.resolve(W, W, spp), img)
    r = subprocess.run([exe, out, "32", "1", "/nonexistent.obj"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "cannot open" in r.stderr
