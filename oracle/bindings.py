"""ctypes bindings of the CPU oracle (oracle/libref_cpu.so) and, where it was built, of the real
reference classes (oracle/_ref/libref_true.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg. The product (ray-tracing_amd/) never imports this module.

Scenes are given as the plain-data object list of `scenes.export_objects` (dicts with final arrays),
so the checkers see exactly the geometry the GPU path sees, without sharing any of its code.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_LIB = os.path.join(_HERE, "libref_cpu.so")
TRUEREF_LIB = os.path.join(_HERE, "_ref", "libref_true.so")
PLAINREF_LIB = os.path.join(_HERE, "_ref", "libref_plain.so")  # the reference with its own shared RNG: for timing only

_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_pd = C.POINTER(C.c_double)


def build(quiet=True):
    """(re)build the oracle and, where /root/reference exists, oracle/_ref"""
    subprocess.check_call(["make", "-C", _HERE] + (["-s"] if quiet else []))


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _load_oracle():
    if not os.path.exists(ORACLE_LIB):
        build()
    lib = C.CDLL(ORACLE_LIB)
    lib.orc_scene_create.restype = _vp
    lib.orc_scene_create.argtypes = [_dbl, _pd]
    lib.orc_scene_destroy.argtypes = [_vp]
    lib.orc_add_sphere.argtypes = [_vp, _pd, _dbl, _pd, C.c_int, C.c_int]
    lib.orc_add_mesh.argtypes = [_vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _pd, C.c_int, C.c_int, C.c_int, _vp, _vp, _vp]
    lib.orc_mesh_bvh_info.argtypes = [_vp, C.c_int, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32), C.POINTER(_i32)]
    lib.orc_mesh_bvh_dump.restype = _i64
    lib.orc_mesh_bvh_dump.argtypes = [_vp, C.c_int, _vp, _vp, _i64]
    lib.orc_intersect_rays.argtypes = [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp]
    lib.orc_sample_radiance.argtypes = [_vp, _vp, _vp, _vp, _i64, _vp]
    lib.orc_sample_trace.argtypes = [_vp, _vp, _vp, _vp, _i64, _vp, _vp]
    lib.orc_set_math.argtypes = [C.c_int]
    lib.orc_math_sincos.argtypes = [_vp, _i64, _vp, _vp]
    lib.orc_math_log.argtypes = [_vp, _i64, _vp]
    lib.orc_render.argtypes = [_vp, _vp, _vp, _vp, C.c_int, _vp]
    lib.orc_resolve.argtypes = [_vp, _i32, _i32, _dbl, _vp]
    lib.orc_rng_key.restype = C.c_uint64
    lib.orc_rng_key.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32]
    lib.orc_rng_draw.restype = C.c_uint64
    lib.orc_rng_draw.argtypes = [C.c_uint64, C.c_uint32]
    lib.orc_rng_uniform.restype = _dbl
    lib.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint32]
    lib.orc_max_threads.restype = C.c_int
    return lib


def _load_trueref(path=TRUEREF_LIB):
    if not os.path.exists(path):
        return None
    lib = C.CDLL(path)
    lib.trf_scene_create.restype = _vp
    lib.trf_scene_create.argtypes = [_dbl, _pd]
    lib.trf_scene_destroy.argtypes = [_vp]
    lib.trf_add_sphere.argtypes = [_vp, _pd, _dbl, _pd, C.c_int, C.c_int]
    lib.trf_add_mesh.argtypes = [_vp, _pd, C.c_int, C.c_int]
    lib.trf_mesh_read_obj.argtypes = [_vp, C.c_int, C.c_char_p]
    lib.trf_mesh_load_texture.argtypes = [_vp, C.c_int, C.c_char_p]
    lib.trf_mesh_add_texture_raw.argtypes = [_vp, C.c_int, _vp, C.c_int, C.c_int]
    lib.trf_mesh_set_raw.argtypes = [_vp, C.c_int, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp]
    lib.trf_mesh_op.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.c_int, _dbl]
    lib.trf_mesh_build_bvh.argtypes = [_vp, C.c_int]
    lib.trf_mesh_sizes.argtypes = [_vp, C.c_int, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32)]
    lib.trf_mesh_get.argtypes = [_vp, C.c_int, _vp, _vp, _vp, _vp, _vp]
    lib.trf_mesh_texture.argtypes = [_vp, C.c_int, C.c_int, C.POINTER(_i32), C.POINTER(_i32), _vp]
    lib.trf_bvh_dump.restype = _i64
    lib.trf_bvh_dump.argtypes = [_vp, C.c_int, _vp, _vp, _i64]
    lib.trf_intersect_rays.argtypes = [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp]
    lib.trf_sample_radiance.argtypes = [_vp, _vp, _vp, _vp, _i64, _vp]
    lib.trf_render.argtypes = [_vp, _vp, _vp, _vp]
    lib.trf_render_omp.argtypes = [_vp, _vp, _vp, _vp, C.c_int]
    lib.trf_resolve.argtypes = [_vp, C.c_int, C.c_int, C.c_int, _vp]
    lib.trf_camera_basis.argtypes = [_dbl, _dbl, _dbl, C.c_int, _pd, _pd, _pd, _pd]
    lib.trf_draws.restype = C.c_uint64
    return lib


orc = _load_oracle()
trf = _load_trueref()


def have_trueref():
    return trf is not None


def _hit_arrays(n):
    return (np.empty(n, np.int32), np.empty(n, np.int32), np.empty(n), np.empty((n, 6)), np.empty((n, 3)))


MATH_MODES = {"libm": 0, "portable": 1}


class OracleScene:
    """The CPU restatement (oracle/ref_cpu.cpp) on an exported object list.

    math: which cos / sin / log the sampling code uses - "libm" (the host's, as the reference: the mode that is
    bit-identical to the real reference) or "portable" (ray-tracing_amd/csrc/rt_math.h, the functions the GPU computes
    with: the mode the GPU is compared against sample by sample). Closest-hit queries do not depend on it."""

    def __init__(self, objects, I, L, math="libm"):
        self.math = math
        self._keep = []
        self._h = orc.orc_scene_create(float(I), _d3(L))
        self.mesh_ids = []
        for o in objects:
            if o["kind"] == "sphere":
                orc.orc_add_sphere(self._h, _d3(o["O"]), float(o["R"]), _d3(o["albedo"]), int(o["mirror"]), int(o["transp"]))
            else:
                v = np.ascontiguousarray(o["verts"], np.float64)
                n = np.ascontiguousarray(o["normals"], np.float64)
                uv = np.ascontiguousarray(o["uvs"], np.float64)
                ix = np.ascontiguousarray(o["idx"], np.int32)
                texs = [np.ascontiguousarray(t, np.uint8) for t in o.get("textures", [])]
                self._keep += texs
                nt = len(texs)
                tp = (_vp * max(nt, 1))(*[t.ctypes.data for t in texs])
                tw = (_i32 * max(nt, 1))(*[t.shape[1] for t in texs])
                th = (_i32 * max(nt, 1))(*[t.shape[0] for t in texs])
                oid = orc.orc_add_mesh(self._h, len(v), _ptr(v), len(n), _ptr(n), len(uv), _ptr(uv), len(ix), _ptr(ix), _d3(o["albedo"]),
                                       int(o["mirror"]), int(o["transp"]), nt, tp, tw, th)
                self.mesh_ids.append(oid)

    def __del__(self):
        if getattr(self, "_h", None):
            orc.orc_scene_destroy(self._h)
            self._h = None

    def intersect_rays(self, rays, counters=None):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        obj, tri, t, PN, alb = _hit_arrays(len(rays))
        orc.orc_intersect_rays(self._h, _ptr(rays), len(rays), _ptr(obj), _ptr(tri), _ptr(t), _ptr(PN), _ptr(alb),
                               C.byref(counters) if counters is not None else None)
        return dict(obj=obj, tri=tri, t=t, P=PN[:, :3].copy(), N=PN[:, 3:].copy(), albedo=alb)

    def sample_radiance(self, cam, params, ijk, math=None):
        return self.sample_trace(cam, params, ijk, math, want_trace=False)[0]

    def sample_trace(self, cam, params, ijk, math=None, want_trace=True):
        """radiance (n, 3) and the path of every sample (n, 12, 4), coded as rt_sample_trace of rt_amd.h"""
        ijk = np.ascontiguousarray(ijk, np.int32).reshape(-1, 3)
        out = np.empty((len(ijk), 3))
        trace = np.empty((len(ijk), 12, 4), np.int32) if want_trace else None
        orc.orc_set_math(MATH_MODES[math or self.math])
        orc.orc_sample_trace(self._h, C.byref(cam), C.byref(params), _ptr(ijk), len(ijk), _ptr(out), _ptr(trace))
        return out, trace

    def render(self, cam, params, accum=None, n_threads=0, counters=None, math=None):
        if accum is None:
            accum = np.zeros((params.H, params.W, 3))
        orc.orc_set_math(MATH_MODES[math or self.math])
        orc.orc_render(self._h, C.byref(cam), C.byref(params), _ptr(accum), int(n_threads),
                       C.byref(counters) if counters is not None else None)
        return accum

    def bvh_info(self, object_id, n_tris):
        perm = np.empty(n_tris, np.int32)
        nn, nl, md, ml = _i64(), _i64(), _i32(), _i32()
        rc = orc.orc_mesh_bvh_info(self._h, object_id, _ptr(perm), C.byref(nn), C.byref(nl), C.byref(md), C.byref(ml))
        assert rc == 0
        return dict(perm=perm, n_nodes=nn.value, n_leaves=nl.value, max_depth=md.value, max_leaf=ml.value)

    def bvh_dump(self, object_id, n_nodes):
        boxes = np.empty((n_nodes, 6)); dfl = np.empty((n_nodes, 3), np.int32)
        k = orc.orc_mesh_bvh_dump(self._h, object_id, _ptr(boxes), _ptr(dfl), n_nodes)
        assert k == n_nodes, (k, n_nodes)
        return boxes, dfl


def portable_sincos(x):
    x = np.ascontiguousarray(x, np.float64)
    s, c = np.empty_like(x), np.empty_like(x)
    orc.orc_math_sincos(_ptr(x), x.size, _ptr(s), _ptr(c))
    return s, c


def portable_log(x):
    x = np.ascontiguousarray(x, np.float64)
    out = np.empty_like(x)
    orc.orc_math_log(_ptr(x), x.size, _ptr(out))
    return out


def explain_divergence(trace_a, trace_b, jittered=True):
    """Why two implementations' radiance of ONE sample may differ by more than rounding: compares their path traces
    (12 x 4: object id, triangle id, draws used, RT_TRACE_* code per recursion depth) and returns a reason string, or
    None if the difference is NOT explained. Explained means: the paths agree entry by entry up to some depth d, at
    which they hit a different object / triangle (a ray that grazes a silhouette), or take a different branch at the
    same hit (the Fresnel coin's threshold 100 R, or a shadow ray grazing an occluder), AND the ray that reached depth d
    had a direction that depends on cos / sin / log - the camera jitter (every path when `jittered`) or a diffuse
    vertex before d. Identical traces explain nothing."""
    for d in range(12):
        a, b = trace_a[d], trace_b[d]
        if np.array_equal(a, b):
            continue
        transcendental = jittered or any(trace_a[e][3] in (7, 8) for e in range(d))
        if not transcendental:
            return None
        if a[0] != b[0] or a[1] != b[1]:
            return "depth %d: hit object/triangle %d/%d vs %d/%d (silhouette)" % (d, a[0], a[1], b[0], b[1])
        if a[3] != b[3]:
            return "depth %d: same hit %d/%d, branch %d vs %d (Fresnel coin or shadow ray at a limit)" % (d, a[0], a[1], a[3], b[3])
        return None  # same hit, same branch, another draw count: the streams themselves disagree - never legitimate
    return None


def resolve(accum, spp_total):
    accum = np.ascontiguousarray(accum, np.float64)
    H, W, _ = accum.shape
    out = np.empty((H, W, 3), np.uint8)
    orc.orc_resolve(_ptr(accum), W, H, float(spp_total), _ptr(out))
    return out


def time_plain_reference(objects, I, L, cam, params, n_threads):
    """Wall seconds of the REAL reference's pixel loop (OpenMP rows, its own shared random engine and rand(): the
    program as committed) on an exported object list, or None where oracle/_ref/libref_plain.so is absent. The values it
    renders are not deterministic with several threads and are not used."""
    import time
    lib = _load_trueref(PLAINREF_LIB)
    if lib is None:
        return None
    global trf
    keep, trf = trf, lib  # TrueRefScene drives whichever library `trf` names
    try:
        t = TrueRefScene(objects, I, L)
        acc = np.zeros((params.H, params.W, 3))
        t0 = time.time()
        rc = lib.trf_render_omp(t._h, C.byref(cam), C.byref(params), _ptr(acc), int(n_threads))
        dt = time.time() - t0
        assert rc == 0 and np.isfinite(acc).any()
        del t
        return dt
    finally:
        trf = keep


class TrueRefScene:
    """The reference's own Scene / Sphere / TriangleMesh objects (oracle/_ref)."""

    def __init__(self, objects, I, L, build_bvh=True):
        assert trf is not None, "oracle/_ref not built (needs /root/reference)"
        self._h = trf.trf_scene_create(float(I), _d3(L))
        self.mesh_ids = []
        for o in objects:
            if o["kind"] == "sphere":
                trf.trf_add_sphere(self._h, _d3(o["O"]), float(o["R"]), _d3(o["albedo"]), int(o["mirror"]), int(o["transp"]))
            else:
                oid = trf.trf_add_mesh(self._h, _d3(o["albedo"]), int(o["mirror"]), int(o["transp"]))
                self.mesh_ids.append(oid)
                if "obj_path" in o:
                    assert trf.trf_mesh_read_obj(self._h, oid, os.fsencode(o["obj_path"])) == 0
                else:
                    v = np.ascontiguousarray(o["verts"], np.float64)
                    n = np.ascontiguousarray(o["normals"], np.float64)
                    uv = np.ascontiguousarray(o["uvs"], np.float64)
                    ix = np.ascontiguousarray(o["idx"], np.int32)
                    trf.trf_mesh_set_raw(self._h, oid, len(v), _ptr(v), len(n), _ptr(n), len(uv), _ptr(uv), len(ix), _ptr(ix))
                for t in o.get("textures", []):
                    t = np.ascontiguousarray(t, np.uint8)
                    trf.trf_mesh_add_texture_raw(self._h, oid, _ptr(t), t.shape[1], t.shape[0])
                for p in o.get("texture_paths", []):
                    assert trf.trf_mesh_load_texture(self._h, oid, os.fsencode(p)) == 0
                for op in o.get("ops", []):
                    self.mesh_op(oid, *op)
                if build_bvh:
                    trf.trf_mesh_build_bvh(self._h, oid)

    def __del__(self):
        if getattr(self, "_h", None) and trf is not None:
            trf.trf_scene_destroy(self._h)
            self._h = None

    def mesh_op(self, oid, name, a=0, b=0, angle=0.0):
        ops = {"invertNormals": 0, "translate": 1, "rotateNormals": 2, "rotate": 3, "rotateAroundAxe": 4, "zoom": 5}
        if name == "rotateAroundAxe":
            a, b, angle = a, 0, b
        if name == "zoom":
            a, b = 0, a
        assert trf.trf_mesh_op(self._h, oid, ops[name], int(a), int(b), float(angle)) == 0

    def build_bvh(self, oid):
        trf.trf_mesh_build_bvh(self._h, oid)

    def mesh_arrays(self, oid):
        nv, nn, nuv, nt, ntex = _i64(), _i64(), _i64(), _i64(), _i32()
        trf.trf_mesh_sizes(self._h, oid, C.byref(nv), C.byref(nn), C.byref(nuv), C.byref(nt), C.byref(ntex))
        v = np.empty((nv.value, 3)); n = np.empty((nn.value, 3)); uv = np.empty((nuv.value, 2))
        ix = np.empty((nt.value, 10), np.int32); perm = np.full(nt.value, -1, np.int32)
        trf.trf_mesh_get(self._h, oid, _ptr(v), _ptr(n), _ptr(uv), _ptr(ix), _ptr(perm))
        return v, n, uv, ix, perm

    def mesh_texture(self, oid, k):
        W, H = _i32(), _i32()
        assert trf.trf_mesh_texture(self._h, oid, k, C.byref(W), C.byref(H), None) == 0
        out = np.empty((H.value, W.value, 3), np.uint8)
        trf.trf_mesh_texture(self._h, oid, k, C.byref(W), C.byref(H), _ptr(out))
        return out

    def bvh_dump(self, oid):
        n = trf.trf_bvh_dump(self._h, oid, None, None, 0)
        boxes = np.empty((n, 6)); dfl = np.empty((n, 3), np.int32)
        trf.trf_bvh_dump(self._h, oid, _ptr(boxes), _ptr(dfl), n)
        return boxes, dfl

    def intersect_rays(self, rays):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        obj, tri, t, PN, alb = _hit_arrays(len(rays))
        trf.trf_intersect_rays(self._h, _ptr(rays), len(rays), _ptr(obj), _ptr(tri), _ptr(t), _ptr(PN), _ptr(alb))
        return dict(obj=obj, tri=tri, t=t, P=PN[:, :3].copy(), N=PN[:, 3:].copy(), albedo=alb)

    def sample_radiance(self, cam, params, ijk):
        ijk = np.ascontiguousarray(ijk, np.int32).reshape(-1, 3)
        out = np.empty((len(ijk), 3))
        assert trf.trf_sample_radiance(self._h, C.byref(cam), C.byref(params), _ptr(ijk), len(ijk), _ptr(out)) == 0
        return out

    def render(self, cam, params, accum=None):
        if accum is None:
            accum = np.zeros((params.H, params.W, 3))
        assert trf.trf_render(self._h, C.byref(cam), C.byref(params), _ptr(accum)) == 0
        return accum


def trueref_resolve(accum, nbrays):
    accum = np.ascontiguousarray(accum, np.float64)
    H, W, _ = accum.shape
    out = np.empty((H, W, 3), np.uint8)
    trf.trf_resolve(_ptr(accum), W, H, int(nbrays), _ptr(out))
    return out


def primary_rays(cam, W, H, rect=None):
    """Un-jittered camera rays Ray(C, u) through pixel centres (raytracer.cpp:1475-1477, :1482),
    computed with exactly-rounded operations only; returns (rays (n,6), ij (n,2))."""
    j0, i0, j1, i1 = rect if rect else (0, 0, W, H)
    ii, jj = np.meshgrid(np.arange(i0, i1), np.arange(j0, j1), indexing="ij")
    x = (jj - W // 2 + 0.5).astype(np.float64).ravel()
    y = (ii - H // 2 + 0.5).astype(np.float64).ravel()
    z = np.full_like(x, cam.zdist)
    n = np.sqrt(x * x + y * y + z * z)
    x, y, z = x / n, y / n, z / n
    r, u, v = np.array(cam.right[:]), np.array(cam.up[:]), np.array(cam.view[:])
    d = (x[:, None] * r[None, :] + y[:, None] * u[None, :]) + z[:, None] * v[None, :]
    rays = np.concatenate([np.broadcast_to(np.array(cam.C[:]), d.shape), d], 1)
    return np.ascontiguousarray(rays), np.stack([ii.ravel(), jj.ravel()], 1)
