"""GPU (one device is enough): samples-per-pixel sharding with two HIP contexts - what a multi-GPU caller does with one
context per GPU. Two contexts on device 0 render disjoint sample ranges of the same frame; their accumulators, added
with rt_accum_add (the peer-copy + add step of the C++ rt::render(..., devices)), equal a single context's render of
all the samples within 1e-13 and give the identical 8-bit image."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_two_contexts_sum_to_one_render(gpu, rta, scenes):
    W, spp = 256, 24
    s, cam, flags = scenes.mesh_room(W=W, n_tris=20000, tex_size=64)[:3]
    gpu.upload(s)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, spp, flags=flags))
    whole = gpu.accum_download(W, W)
    whole_img = gpu.resolve(W, W, spp)
    a, b = rta.Renderer(0), rta.Renderer(0)
    try:
        for r in (a, b):
            r.upload(s)
            r.accum_clear(W, W)
        # two streams, two copies of the scene, disjoint global sample indices
        a.render(cam, rta.make_params(W, W, 10, spp_offset=0, flags=flags))
        b.render(cam, rta.make_params(W, W, 14, spp_offset=10, flags=flags))
        a.accum_add(b)
        got = a.accum_download(W, W)
        assert np.allclose(got, whole, rtol=1e-13, atol=1e-9, equal_nan=True)
        assert np.array_equal(a.resolve(W, W, spp), whole_img)
        # the same split again gives the same bits (fixed order of the additions)
        for r in (a, b):
            r.accum_clear(W, W)
        a.render(cam, rta.make_params(W, W, 10, spp_offset=0, flags=flags))
        b.render(cam, rta.make_params(W, W, 14, spp_offset=10, flags=flags))
        a.accum_add(b)
        assert np.array_equal(got, a.accum_download(W, W), equal_nan=True)
        # the way round a refused peer copy (pinned host buffer): the same bits
        rta.lib.rt_debug_force_host_bounce(a._h, 1)
        for r in (a, b):
            r.accum_clear(W, W)
        a.render(cam, rta.make_params(W, W, 10, spp_offset=0, flags=flags))
        b.render(cam, rta.make_params(W, W, 14, spp_offset=10, flags=flags))
        a.accum_add(b)
        assert np.array_equal(got, a.accum_download(W, W), equal_nan=True)
        rta.lib.rt_debug_force_host_bounce(a._h, 0)
        with pytest.raises(rta.RTError):
            a.accum_add(a)
    finally:
        a.close()
        b.close()


def test_render_devices_entry_point(gpu, rta, scenes):
    """rt::render(scene, cam, params, image, devices) through its C wrapper, with device 0 listed twice: the image of one
    context rendering all samples"""
    W, spp = 192, 16
    s, cam, flags = scenes.sphere_room(W)
    gpu.upload(s)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, spp, flags=flags))
    want = gpu.resolve(W, W, spp)
    got = rta.render_devices(s, cam, W, W, spp, [0, 0], flags=flags)
    assert np.array_equal(got, want)
    assert np.array_equal(rta.render_devices(s, cam, W, W, spp, [0], flags=flags), want)
    with pytest.raises(rta.RTError):
        rta.render_devices(s, cam, W, W, spp, [99], flags=flags)
