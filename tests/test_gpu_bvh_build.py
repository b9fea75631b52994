"""TriangleMesh::buildBVH (raytracer.cpp:220-291) on the GPU against the host build, which the golden fixtures pin
to the real reference: the boxes, the tree and the order of the triangles must be the same, bit for bit."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(rta, gpu, make):
    a, b = make(), make()
    a.buildBVH()
    b.buildBVH(gpu)
    ba, da = a.bvh_dump()
    bb, db = b.bvh_dump()
    assert ba.shape == bb.shape
    assert np.array_equal(da, db)
    assert ba.tobytes() == bb.tobytes()  # also the sign of zeros
    pa, pb = a.arrays()[4], b.arrays()[4]
    assert np.array_equal(pa, pb)
    assert np.array_equal(a.arrays()[3], b.arrays()[3])
    assert a.sizes()["max_depth"] == b.sizes()["max_depth"]
    return a.sizes()


def _mesh(rta, v, n, uv, ix, ops=()):
    m = rta.TriangleMesh((1, 1, 1))
    m.set_raw(v, n, uv, ix)
    for op in ops:
        getattr(m, op[0])(*op[1:])
    return m


@pytest.mark.parametrize("grid,ops", [(3, ()), (20, ()), (60, (("zoom", 4), ("rotate", 1, 2), ("rotateAroundAxe", 1, -30), ("translate", 0, 25))),
                                      (90, (("zoom", 0), ("rotate", 0, 2), ("rotateAroundAxe", 1, 30), ("rotateAroundAxe", 0, 15)))])
def test_blob_meshes(gpu, rta, scenes, grid, ops):
    """smooth and blocky (integer-truncated: many equal coordinates and centres) meshes"""
    def make():
        return _mesh(rta, *scenes.blob_mesh(grid, size=(1.2, 2.1, 1.0), offset=(0.0, 0.0, 1.0), seed=grid), ops=ops)
    s = _both(rta, gpu, make)
    assert s["n_nodes"] >= 1


def test_coordinate_planes_and_tiny_meshes(gpu, rta, scenes):
    """zeros of both signs among the coordinates, ranges that never split, fewer than five triangles"""
    def plane():
        return _mesh(rta, *scenes.grid_mesh(24, lambda a, b: np.stack([a - 12, np.where((a + b) % 2 == 0, 0.0, -0.0), b - 12], 1)))
    _both(rta, gpu, plane)

    def stairs():
        return _mesh(rta, *scenes.grid_mesh(40, lambda a, b: np.stack([np.trunc(a * 0.3) - 6, np.trunc((a + b) * 0.2) - 4, b * 0.5 - 10], 1)))
    _both(rta, gpu, stairs)
    for n in (1, 2):
        _both(rta, gpu, lambda: _mesh(rta, *scenes.grid_mesh(n, lambda a, b: np.stack([a * 1.5, a * b * 0.25, b * 2.0], 1))))

    def lopsided():  # nearly everything on one side of the first cuts: long runs of S after the first U, and the reverse
        v, nrm, uv, ix = scenes.grid_mesh(30, lambda a, b: np.stack([np.where(a < 29, a * 0.001, 50.0 + b), b * 0.01, (a * b) % 7 * 0.1], 1))
        return _mesh(rta, v, nrm, uv, ix[::-1].copy())
    _both(rta, gpu, lopsided)


def test_scene_built_on_the_gpu_renders_the_same(gpu, rta, scenes, ob):
    s, cam, flags = scenes.mesh_room(W=64, n_tris=8000, tex_size=32)
    s2, _, _ = scenes.mesh_room(W=64, n_tris=8000, tex_size=32)
    for o in s2.objects:
        if isinstance(o, rta.TriangleMesh):
            o.buildBVH(gpu)
    rays, _ = ob.primary_rays(cam, 64, 64)
    gpu.upload(s)
    h1 = gpu.intersect_rays(rays)
    gpu.upload(s2)
    h2 = gpu.intersect_rays(rays)
    # the scene builder has built the trees once already; a second build numbers the triangles by their positions after
    # the first, so the triangle ids are labels of another numbering: everything else must agree
    for k in ("obj", "t", "P", "N", "albedo"):
        assert np.array_equal(h1[k], h2[k], equal_nan=True), k
    assert (h1["tri"] >= 0).sum() > 100
