"""rt_render_static - the entry point that allocates nothing - against rt_render, through the C ABI.

Both run the same kernels; the static variant keeps the tile counter and the unit sums in the library's static device
array instead of buffers allocated on demand. Its output must equal rt_render's bit for bit (the unit structure, and
so the order of every addition, is the same), on two calls in a row, the second with a smaller input: the array keeps
the first call's contents, which must not leak into the second."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(gpu, rta, cam, W, H, spp, flags=0, rect=None, spp_offset=0):
    out = []
    for fn in (gpu.render, gpu.render_static):
        gpu.accum_clear(W, H)
        fn(cam, rta.make_params(W, H, spp, spp_offset=spp_offset, flags=flags, rect=rect))
        gpu.sync()
        out.append((gpu.accum_download(W, H), gpu.resolve(W, H, spp)))
    return out


def test_static_scratch_render_equals_render(gpu, rta, scenes):
    W = 512
    s, cam, flags = scenes.mesh_room(W=W, n_tris=20000, tex_size=64)[:3]
    gpu.upload(s)
    # first call: 512 x 512 is two pixels per resident lane -> units of 8 samples, 24 samples = 3 units of 6.3 MB each
    (a, ia), (b, ib) = _both(gpu, rta, cam, W, W, 24, flags)
    assert (a > 0).mean() > 0.9
    assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(ia, ib)
    # second call, smaller input: a 192 x 128 rectangle, 5 samples from sample 3 on -> units of one sample
    rect = (64, 64, 256, 192)
    (a2, ia2), (b2, ib2) = _both(gpu, rta, cam, W, W, 5, flags, rect=rect, spp_offset=3)
    assert a2[64:192, 64:256].any() and not a2[:64].any() and not a2[192:].any()
    assert np.array_equal(a2, b2, equal_nan=True)
    assert np.array_equal(ia2, ib2)
    # and the first input again: what the second call left in the array does not matter either
    gpu.accum_clear(W, W)
    gpu.render_static(cam, rta.make_params(W, W, 24, flags=flags))
    assert np.array_equal(a, gpu.accum_download(W, W), equal_nan=True)


def _free_device_bytes(gpu, rta):
    import ctypes
    f, t = ctypes.c_uint64(), ctypes.c_uint64()
    assert rta.lib.rt_debug_mem_info(gpu._h, ctypes.byref(f), ctypes.byref(t)) == 0  # hipMemGetInfo on the context's device
    return f.value


def test_static_render_allocates_no_device_memory(gpu, rta, scenes):
    """the property the entry point exists for: between two calls nothing is taken from (or given back to) the device's
    memory - hipMemGetInfo's free bytes are the same before and after static renders that make rt_render grow its
    unit-sum buffer"""
    W = 768
    s, cam, flags = scenes.sphere_room(W)[:3]
    gpu.upload(s)
    gpu.accum_clear(W, W)
    gpu.render_static(cam, rta.make_params(W, W, 8, flags=flags))  # code objects, events: whatever a first launch sets up
    gpu.sync()
    before = _free_device_bytes(gpu, rta)
    gpu.render_static(cam, rta.make_params(W, W, 32, flags=flags))  # four units of 14 MB: rt_render would allocate 57 MB here
    gpu.render_static(cam, rta.make_params(W, W, 16, flags=flags, rect=(0, 0, 512, 512)))
    gpu.sync()
    assert _free_device_bytes(gpu, rta) == before
    gpu.accum_clear(W, W)
    fresh = rta.Renderer(0)  # a context that has never rendered: its rt_render has no unit-sum buffer yet and must allocate
    try:
        fresh.upload(s)
        fresh.accum_clear(W, W)
        fresh.sync()
        b0 = _free_device_bytes(gpu, rta)
        fresh.render(cam, rta.make_params(W, W, 32, flags=flags))
        fresh.sync()
        assert _free_device_bytes(gpu, rta) < b0  # the check does see an allocation when there is one
    finally:
        fresh.close()


def test_static_scratch_limit(gpu, rta, scenes):
    """a frame whose launch does not fit the array is rendered in more launches of fewer samples, with rt_render's bits; only a
    rectangle of which not even two units fit is refused, with a message; a call of one unit needs no unit sums and fits any frame"""
    W = 1024
    s, cam, flags = scenes.sphere_room(W)[:3]
    gpu.upload(s)
    gpu.accum_clear(W, W)
    gpu.render_static(cam, rta.make_params(W, W, 64, flags=flags))  # 8 units x 25 MB: four launches of two units
    assert gpu.last_render_ms()[1] == 4
    b = gpu.accum_download(W, W)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 64, flags=flags))  # one launch of eight units
    assert gpu.last_render_ms()[1] == 1
    assert np.array_equal(gpu.accum_download(W, W), b, equal_nan=True)
    gpu.accum_clear(W, W)
    gpu.render_static(cam, rta.make_params(W, W, 8, flags=flags))  # one unit: straight into the accumulator
    c = gpu.accum_download(W, W)
    gpu.accum_clear(W, W)
    gpu.render(cam, rta.make_params(W, W, 8, flags=flags))
    assert np.array_equal(gpu.accum_download(W, W), c, equal_nan=True)
    W = 1280  # 1 638 400 pixels: one unit is 39 MB, two do not fit 64 MiB
    s, cam, flags = scenes.sphere_room(W)[:3]
    gpu.upload(s)
    gpu.accum_clear(W, W)
    with pytest.raises(rta.RTError, match="static"):
        gpu.render_static(cam, rta.make_params(W, W, 16, flags=flags))
    gpu.render_static(cam, rta.make_params(W, W, 8, flags=flags))
    gpu.sync()


def test_static_renders_of_two_contexts_do_not_overlap(gpu, rta, scenes):
    """the array is shared by every context of a GPU: a static render while another context's is still running is refused
    (RT_E_STATE) instead of racing on the tile counter and the unit sums"""
    W = 1024
    s, cam, flags = scenes.sphere_room(W)[:3]
    other = rta.Renderer(0)
    try:
        for r in (gpu, other):
            r.upload(s)
            r.accum_clear(W, W)
        gpu.render_static(cam, rta.make_params(W, W, 64, flags=flags))  # tens of milliseconds of launches, not waited for
        with pytest.raises(rta.RTError, match="still running"):
            other.render_static(cam, rta.make_params(W, W, 16, flags=flags))
        gpu.sync()
        other.render_static(cam, rta.make_params(W, W, 16, flags=flags))  # the first context has finished: the array is free
        other.sync()
        a = other.accum_download(W, W)
        other.accum_clear(W, W)
        other.render(cam, rta.make_params(W, W, 16, flags=flags))
        assert np.array_equal(other.accum_download(W, W), a, equal_nan=True)
    finally:
        other.close()
