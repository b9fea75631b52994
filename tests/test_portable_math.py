"""CPU: the portable cos / sin / log of ray-tracing_amd/csrc/rt_math.h (what the GPU samples with), the tone-map
threshold table, and the account of every sample on which they make the oracle's answer differ from its answer with
the host's libm (the mode that is bit-identical to the real reference).

The chain the GPU tests rely on:  real reference == oracle(libm) bit for bit (test_oracle_vs_reference, golden fixtures);
oracle(libm) vs oracle(portable): equal to 1e-9 except samples whose path traces show a silhouette or a branch flipped
behind a direction that came from cos / sin / log (here);  oracle(portable) vs GPU: equal to 1e-9 with NO exception
(tests/test_gpu_*.py)."""
import ctypes as C
import importlib

import numpy as np
import pytest


def _ulp_err(got, ref_ld):
    r = ref_ld.astype(np.float64)
    return np.abs((got.astype(np.longdouble) - ref_ld) / np.spacing(np.abs(r)).astype(np.longdouble)).astype(np.float64)


def _uniforms(n, seed):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 2 ** 53, n, dtype=np.uint64).astype(np.float64) * 2.0 ** -53
    u[:10] = [0, 0.25, 0.5, 0.75, 2.0 ** -53, 1 - 2.0 ** -53, 0.125, 0.375, 0.625, 0.875]
    return u


def test_sincos_accuracy(ob):
    u = _uniforms(1_000_000, 1)
    x = 2 * 3.14159265358979323846 * u  # the argument as raytracer.cpp:116-117, :1468-1469 form it
    s, c = ob.portable_sincos(x)
    xl = x.astype(np.longdouble)
    assert _ulp_err(s, np.sin(xl)).max() < 1.0
    assert _ulp_err(c, np.cos(xl)).max() < 1.0
    assert (s != np.sin(x)).mean() < 0.05 and (c != np.cos(x)).mean() < 0.05  # mostly the very same double as libm
    s, c = ob.portable_sincos(np.array([0.0, 7.0, -1e-300, np.nan, np.inf]))
    assert s[0] == 0.0 and c[0] == 1.0 and np.isnan(s[1:]).all() and np.isnan(c[1:]).all()


def test_log_accuracy(ob):
    u = _uniforms(1_000_000, 2)
    u[0] = 2.0 ** -53
    lg = ob.portable_log(u)
    assert _ulp_err(lg, np.log(u.astype(np.longdouble))).max() < 1.0
    assert (lg != np.log(u)).mean() < 0.1
    z = ob.portable_log(np.array([0.0, -0.0, -1.0, np.nan, np.inf, 1.0, 2.0, 1e300, 3.0e-300]))
    assert z[0] == -np.inf and z[1] == -np.inf and np.isnan(z[2]) and np.isnan(z[3]) and z[4] == np.inf and z[5] == 0.0
    assert np.allclose(z[6:], np.log(np.array([2.0, 1e300, 3.0e-300])), rtol=2e-16, atol=0)


def test_box_muller_known_answers(ob):
    """the Monte-Carlo exercises of raytracer.cpp:1096-1151 with the counter-based stream and the portable functions:
    integral of cos^10 over [-pi/2, pi/2] = 63 pi / 256 by Gaussian importance sampling (sigma = 0.25 as the reference)"""
    n = 400_000
    u1 = np.array([ob.orc.orc_rng_uniform(1234, 2 * k) for k in range(n // 40)])  # the stream itself, a sample of it
    assert 0.45 < u1.mean() < 0.55
    rng = np.random.default_rng(3)
    u1, u2 = rng.random(n), rng.random(n)
    s, c = ob.portable_sincos(2 * np.pi * u1)
    sigma = 0.25
    x = sigma * c * np.sqrt(-2 * ob.portable_log(u2))
    p = np.exp(-x * x / (2 * sigma * sigma)) / (sigma * np.sqrt(2 * np.pi))
    inside = np.abs(x) < np.pi / 2
    est = (np.where(inside, np.cos(x) ** 10, 0) / p).mean()
    assert abs(est - 63 * np.pi / 256) < 5e-3


def test_tone_table_equals_pow(rta, ob):
    """the byte of raytracer.cpp:1499-1501 from the threshold table == (unsigned char)min(255, pow(v, 0.45)) with the host's
    pow: on random values, and on the 129 doubles around every threshold (where pow must not decrease)"""
    T = np.empty(256)
    assert rta.lib.rt_debug_tone_table(T.ctypes.data_as(C.c_void_p)) == 0
    assert T[0] == 0 and np.all(np.diff(T) > 0)

    def by_table(v):
        return np.where(np.isnan(v) | (v < 0), 255, np.searchsorted(T, v, side="right") - 1).astype(np.uint8)

    def by_pow(v):  # the oracle's tone map: C pow of the host's libm, as the reference (numpy's own power differs from it in the last bit)
        n = len(v) // 3 * 3
        return ob.resolve(np.ascontiguousarray(v[:n]).reshape(1, n // 3, 3), 1.0).ravel()

    rng = np.random.default_rng(5)
    v = np.concatenate([rng.random(200000) ** 4 * 3e5, rng.random(50000) * 10, [0.0, -0.0, -1.0, np.nan, np.inf, 1e300, 5e-324]])
    assert np.array_equal(by_table(v)[:len(v) // 3 * 3], by_pow(v))
    bits = T[1:].view(np.uint64)[:, None] + np.arange(-64, 65, dtype=np.int64)[None, :].astype(np.uint64)
    near = bits.view(np.float64).ravel()
    assert np.array_equal(by_table(near)[:len(near) // 3 * 3], by_pow(near))


SCENES = [("sphere_room", dict(W=96), 4), ("mesh_room", dict(W=96, n_tris=20000, tex_size=128), 4),
          ("battle", dict(W=64, tri_scale=0.1, tex_size=64), 2), ("stress", dict(W=96, n_tris=8000, n_spheres=59), 2),
          ("axis_aligned", dict(), 2)]


@pytest.mark.parametrize("name,kw,spp", SCENES, ids=[s[0] for s in SCENES])
def test_libm_vs_portable_paths_are_explained(rta, scenes, ob, name, kw, spp):
    """Every sample whose radiance differs by more than 1e-9 between the two math modes must show, in its path traces,
    a different hit or a different branch behind a cos / sin / log-dependent direction; all other samples must have
    identical traces. No allowance for unexplained samples."""
    out = getattr(scenes, name)(**kw)
    s, cam, flags = out[0], out[1], out[2]
    W = kw.get("W", 64)
    orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L)
    p = rta.make_params(W, W, spp, flags=flags)
    ijk = np.array([(i, j, k) for i in range(0, W, 2) for j in range(0, W, 2) for k in range(spp)], np.int32)
    a, ta = orc.sample_trace(cam, p, ijk, math="libm")
    b, tb = orc.sample_trace(cam, p, ijk, math="portable")
    assert np.array_equal(a, orc.sample_radiance(cam, p, ijk, math="libm"), equal_nan=True)
    both_nan = np.isnan(a) & np.isnan(b)
    rel = np.abs(a - b) / np.maximum(np.abs(a), 1e-3)
    rel[both_nan] = 0
    rel[np.isnan(rel)] = np.inf
    rel = rel.max(axis=1)
    same_trace = (ta == tb).all(axis=(1, 2))
    bad = np.nonzero(rel > 1e-9)[0]
    for q in bad:
        why = ob.explain_divergence(ta[q], tb[q], jittered=not (flags & rta.RT_F_NO_JITTER))
        assert why is not None, ("unexplained difference", name, ijk[q], rel[q], ta[q], tb[q])
    # and the other way round: where the traces agree, so does the radiance
    assert not (same_trace & (rel > 1e-9)).any()
    assert len(bad) <= len(ijk) // 200, (len(bad), len(ijk))  # sanity: a 1-ulp change rarely crosses a silhouette
