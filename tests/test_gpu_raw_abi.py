"""The C ABI taken at its word: a scene description built by hand from the PODs of include/rt_amd.h (no host
classes), uploaded and probed; then one field at a time is made wrong - every such description must come
back as an error from rt_upload_scene, never reach a kernel."""
import ctypes as C

import numpy as np
import pytest

SPHERE = np.dtype([("O", "f8", 3), ("R", "f8"), ("albedo", "f8", 3), ("flags", "i4"), ("object_id", "i4")])
NODE = np.dtype([("boxL", "f8", 6), ("boxR", "f8", 6), ("L", "i4"), ("R", "i4"), ("nL", "i4"), ("nR", "i4"), ("pad", "i4", 4)])
TRI = np.dtype([("A", "f8", 3), ("e1", "f8", 3), ("e2", "f8", 3), ("orig_id", "i4"), ("mesh", "i4")])
ATTR = np.dtype([("n", "f8", (3, 3)), ("uv", "f8", (3, 2)), ("group", "i4"), ("pad", "i4")])
MESH = np.dtype([("root_box", "f8", 6), ("albedo", "f8", 3), ("flags", "i4"), ("object_id", "i4"), ("root", "i4"), ("root_count", "i4"),
                 ("first_tex", "i4"), ("n_tex", "i4"), ("first_tri", "i4"), ("n_tris", "i4"), ("max_depth", "i4"), ("pad", "i4")])
TEX = np.dtype([("offset", "i8"), ("W", "i4"), ("H", "i4")])


class SceneDesc(C.Structure):
    _fields_ = [("n_spheres", C.c_int32), ("n_meshes", C.c_int32), ("spheres", C.c_void_p), ("meshes", C.c_void_p),
                ("n_nodes", C.c_int64), ("nodes", C.c_void_p), ("n_tris", C.c_int64), ("tris", C.c_void_p), ("attrs", C.c_void_p),
                ("n_tex", C.c_int32), ("pad", C.c_int32), ("tex", C.c_void_p), ("texels", C.c_void_p), ("texel_bytes", C.c_int64),
                ("srgb_lut", C.c_void_p), ("I", C.c_double), ("L", C.c_double * 3)]


def test_layouts():
    assert (SPHERE.itemsize, NODE.itemsize, TRI.itemsize, ATTR.itemsize, MESH.itemsize, TEX.itemsize) == (64, 128, 80, 128, 112, 16)
    assert C.sizeof(SceneDesc) == 136


def _scene():
    """light sphere + one mesh of four triangles (two quads, z = -1 and z = -3, tilted a little) under one BVH record + one 2x2 texture"""
    a = {}
    sp = np.zeros(1, SPHERE)
    sp["O"][0] = (-10, 50, 40); sp["R"] = 5; sp["albedo"][0] = (1, 1, 1); sp["object_id"] = 0
    tr = np.zeros(4, TRI)
    quads = [((-1, -1, -1.0), (1, -1, -1.1), (1, 1, -1.2), (-1, 1, -1.1)), ((-1, -1, -3.0), (1, -1, -3.1), (1, 1, -3.2), (-1, 1, -3.1))]
    k = 0
    for q in quads:
        p = np.array(q, float)
        for (i0, i1, i2) in ((0, 1, 2), (0, 2, 3)):
            tr["A"][k] = p[i0]; tr["e1"][k] = p[i1] - p[i0]; tr["e2"][k] = p[i2] - p[i0]; tr["orig_id"][k] = k; tr["mesh"][k] = 0
            k += 1
    at = np.zeros(4, ATTR)
    at["n"][:] = (0, 0, 1)
    at["uv"][:, 1] = (1, 0); at["uv"][:, 2] = (0, 1)
    nd = np.zeros(1, NODE)
    nd["boxL"][0] = (-1, -1, -1.2, 1, 1, -1.0); nd["boxR"][0] = (-1, -1, -3.2, 1, 1, -3.0)
    nd["L"] = 0; nd["nL"] = 2; nd["R"] = 2; nd["nR"] = 2
    me = np.zeros(1, MESH)
    me["root_box"][0] = (-1, -1, -3.2, 1, 1, -1.0); me["albedo"][0] = (1, 0, 0); me["object_id"] = 1
    me["root"] = 0; me["root_count"] = 0; me["first_tex"] = 0; me["n_tex"] = 1; me["first_tri"] = 0; me["n_tris"] = 4; me["max_depth"] = 1
    tx = np.zeros(1, TEX)
    tx["offset"] = 0; tx["W"] = 2; tx["H"] = 2
    texels = np.arange(12, dtype=np.uint8) * 20
    a.update(spheres=sp, tris=tr, attrs=at, nodes=nd, meshes=me, tex=tx, texels=texels)
    return a


def _desc(a):
    d = SceneDesc()
    d.n_spheres, d.n_meshes = len(a["spheres"]), len(a["meshes"])
    d.n_nodes, d.n_tris, d.n_tex = len(a["nodes"]), len(a["tris"]), len(a["tex"])
    for k in ("spheres", "meshes", "nodes", "tris", "attrs", "tex", "texels"):
        setattr(d, k, a[k].ctypes.data)
    d.texel_bytes = len(a["texels"])
    d.srgb_lut = None
    d.I = 6e9
    d.L = (C.c_double * 3)(-10, 50, 40)
    return d


@pytest.mark.gpu
def test_hand_built_scene_and_its_corruptions(gpu, rta):
    lib = rta.lib
    lib.rt_upload_scene.argtypes = [C.c_void_p, C.c_void_p]
    lib.rt_upload_scene.restype = C.c_int
    a = _scene()
    assert lib.rt_upload_scene(gpu._h, C.byref(_desc(a))) == 0, lib.rt_last_error(gpu._h)
    rays = np.array([[0.2, 0.3, 5, 0, 0, -1], [0.2, 0.3, -2, 0, 0, -1], [5, 5, 5, 0, 0, -1], [0.2, 0.3, -5, 0, 0, 1]], float)
    h = gpu.intersect_rays(rays)
    assert list(h["obj"]) == [1, 1, -1, 1]
    assert list(h["tri"]) == [1, 3, -1, 3]          # (0.2, 0.3) lies in the second triangle of a quad; ray 4 comes from below: the lower quad again
    assert np.all(h["t"][[0, 1, 3]] > 0)
    # texture: the colour is one of the four texels, through the 2.2 table
    lut = (np.arange(256) / 255.0) ** 2.2
    texel_colours = [lut[np.array([0, 20, 40]) + 60 * q] for q in range(4)]
    assert any(np.array_equal(h["albedo"][0], c) for c in texel_colours), h["albedo"][0]

    def broken(change):
        b = {k: v.copy() for k, v in a.items()}
        d = _desc(b)
        change(b, d)
        rc = lib.rt_upload_scene(gpu._h, C.byref(d))
        assert rc != 0, "accepted a broken scene"
        assert len(lib.rt_last_error(gpu._h)) > 0

    def set_field(arr, field, value, i=0):
        def f(b, d):
            b[arr][field][i] = value
        return f

    broken(set_field("nodes", "L", 3))               # leaf range beyond the triangles
    broken(set_field("nodes", "nL", -1))
    broken(set_field("nodes", "nR", 0))              # inner child that does not point forward
    broken(set_field("nodes", "R", 1))               # the two leaves share triangle 1
    broken(set_field("meshes", "n_tris", 3))         # does not cover the triangle array
    broken(set_field("meshes", "root", 7))
    broken(set_field("meshes", "max_depth", 61))
    broken(set_field("meshes", "object_id", 0))      # repeats the sphere's
    broken(set_field("meshes", "first_tex", 1))
    broken(set_field("attrs", "group", 1, 2))        # texture group the mesh does not have
    broken(set_field("tris", "mesh", 1, 3))
    broken(set_field("tex", "W", 3))                 # texture beyond the texel pool
    broken(set_field("spheres", "object_id", 5))
    broken(lambda b, d: setattr(d, "n_tris", 5))
    broken(lambda b, d: setattr(d, "nodes", None))
    broken(lambda b, d: setattr(d, "n_meshes", -1))
    # and the good description still uploads and answers
    assert lib.rt_upload_scene(gpu._h, C.byref(_desc(a))) == 0
    assert list(gpu.intersect_rays(rays)["obj"]) == [1, 1, -1, 1]
