"""MI355X (gfx950) path tracer: Python binding of the C ABI in include/rt_amd.h.

NOTE for processes that also use PyTorch: import torch BEFORE this package. torch ships its own copy of
the HIP runtime under the same soname; whichever is loaded first is shared by both, and loading this
library first makes torch load a second runtime whose device pointers the first one does not know.

The compute path is the HIP library ``csrc/librt_amd.so`` (hand-written
kernels + C ABI + C++ host classes). This module only loads it with ctypes and
mirrors the reference's scene surface (Sphere / TriangleMesh / Scene and the
render loop of /root/reference/raytracer.cpp:1153-1503) for Python callers.
There is no CPU fallback: if the library is missing, importing fails loudly.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "librt_amd.so")

if not os.path.exists(LIB_PATH):
    raise ImportError(
        "HIP extension not built: %s is missing (run `python -c 'import __graft_entry__ as g; g.build()'` "
        "or `make -C ray-tracing_amd/csrc`)" % LIB_PATH)
lib = C.CDLL(LIB_PATH)

# --- PODs of include/rt_amd.h --------------------------------------------------------------------
RT_MAT_MIRROR, RT_MAT_TRANSPARENT = 1, 2
RT_F_POINT_LIGHT, RT_F_NO_INDIRECT, RT_F_NO_FRESNEL, RT_F_NO_JITTER, RT_F_COUNTERS, RT_F_NO_PRUNE, RT_F_FAST = 1, 2, 4, 8, 16, 32, 64
RT_TRACE_LEN = 12
(RT_TRACE_UNUSED, RT_TRACE_NONE, RT_TRACE_LIGHT, RT_TRACE_MIRROR, RT_TRACE_TOTAL_REFLECTION, RT_TRACE_REFLECT, RT_TRACE_REFRACT,
 RT_TRACE_DIFFUSE_LIT, RT_TRACE_DIFFUSE_SHADOWED) = range(9)


class rt_camera(C.Structure):
    _fields_ = [("C", C.c_double * 3), ("right", C.c_double * 3), ("up", C.c_double * 3), ("view", C.c_double * 3),
                ("zdist", C.c_double), ("focus", C.c_double), ("aa_sigma", C.c_double), ("dof_sigma", C.c_double)]


class rt_params(C.Structure):
    _fields_ = [("W", C.c_int32), ("H", C.c_int32), ("spp", C.c_int32), ("spp_offset", C.c_int32), ("seed", C.c_uint64),
                ("flags", C.c_uint32), ("max_rebond", C.c_int32), ("ior", C.c_double), ("epsilon", C.c_double),
                ("shadow_bias", C.c_double), ("rect", C.c_int32 * 4)]


class rt_counters(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("box_tests", C.c_uint64), ("nodes", C.c_uint64),
                ("tris", C.c_uint64), ("mesh_calls", C.c_uint64), ("diffuse", C.c_uint64), ("specular", C.c_uint64)]

    def as_dict(self):
        return {k: int(getattr(self, k)) for k, _ in self._fields_}


# every symbol include/rt_amd.h and include/rt_amd_host.h declare
RT_SYMBOLS = ["rt_create", "rt_destroy", "rt_last_error", "rt_upload_scene", "rt_render", "rt_render_static", "rt_sync", "rt_accum_clear",
              "rt_accum_download", "rt_accum_upload", "rt_accum_device_ptr", "rt_accum_add", "rt_resolve", "rt_intersect_rays", "rt_sample_radiance", "rt_sample_trace",
              "rt_get_counters", "rt_reset_counters", "rt_last_render_ms", "rt_render_time_ms", "rt_stream", "rt_version", "rt_build_bvh"]
RTH_SYMBOLS = ["rth_last_error", "rth_mesh_create", "rth_mesh_destroy", "rth_mesh_read_obj", "rth_mesh_load_texture",
               "rth_mesh_add_texture", "rth_mesh_texture", "rth_decode_jpeg", "rth_mesh_set_raw", "rth_mesh_op", "rth_mesh_build_bvh", "rth_mesh_build_bvh_device", "rth_mesh_sizes",
               "rth_mesh_get", "rth_mesh_bvh_dump", "rth_scene_create", "rth_scene_destroy", "rth_scene_add_sphere",
               "rth_scene_add_mesh", "rth_scene_upload", "rth_scene_render_devices", "rth_scene_flat_info", "rth_scene_flat_arrays", "rth_camera_make", "rth_params_default",
               "rth_write_png", "rth_write_execution_time"]

_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
# the little-endian PODs of include/rt_amd.h as numpy record types
SPHERE_DTYPE = np.dtype([("O", "f8", 3), ("R", "f8"), ("albedo", "f8", 3), ("flags", "i4"), ("object_id", "i4")])
NODE_DTYPE = np.dtype([("boxL", "f8", 6), ("boxR", "f8", 6), ("L", "i4"), ("R", "i4"), ("nL", "i4"), ("nR", "i4"), ("pad", "i4", 4)])
TRI_DTYPE = np.dtype([("A", "f8", 3), ("e1", "f8", 3), ("e2", "f8", 3), ("orig_id", "i4"), ("mesh", "i4")])
ATTR_DTYPE = np.dtype([("n", "f8", (3, 3)), ("uv", "f8", (3, 2)), ("group", "i4"), ("pad", "i4")])
MESH_DTYPE = np.dtype([("root_box", "f8", 6), ("albedo", "f8", 3), ("flags", "i4"), ("object_id", "i4"), ("root", "i4"), ("root_count", "i4"),
                       ("first_tex", "i4"), ("n_tex", "i4"), ("first_tri", "i4"), ("n_tris", "i4"), ("max_depth", "i4"), ("pad", "i4")])
TEX_DTYPE = np.dtype([("offset", "i8"), ("W", "i4"), ("H", "i4")])
_pd = C.POINTER(C.c_double)


def _sig(name, restype, *argtypes):
    f = getattr(lib, name)
    f.restype = restype
    f.argtypes = list(argtypes)
    return f


_sig("rt_create", C.c_int, C.c_int, C.POINTER(_vp))
_sig("rt_destroy", None, _vp)
_sig("rt_last_error", C.c_char_p, _vp)
_sig("rt_render", C.c_int, _vp, C.POINTER(rt_camera), C.POINTER(rt_params), _vp)
_sig("rt_render_static", C.c_int, _vp, C.POINTER(rt_camera), C.POINTER(rt_params), _vp)
_sig("rt_sync", C.c_int, _vp)
_sig("rt_accum_clear", C.c_int, _vp, _i32, _i32)
_sig("rt_accum_download", C.c_int, _vp, _vp, _i64)
_sig("rt_accum_upload", C.c_int, _vp, _vp, _i64)
_sig("rt_accum_device_ptr", _vp, _vp)
_sig("rt_accum_add", C.c_int, _vp, _vp)
_sig("rt_resolve", C.c_int, _vp, _vp, _i32, _i32, _dbl, _vp)
_sig("rt_intersect_rays", C.c_int, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp)
_sig("rt_sample_radiance", C.c_int, _vp, C.POINTER(rt_camera), C.POINTER(rt_params), _vp, _i64, _vp)
_sig("rt_sample_trace", C.c_int, _vp, C.POINTER(rt_camera), C.POINTER(rt_params), _vp, _i64, _vp, _vp)
_sig("rt_get_counters", C.c_int, _vp, C.POINTER(rt_counters))
_sig("rt_reset_counters", C.c_int, _vp)
_sig("rt_last_render_ms", C.c_int, _vp, _pd, C.POINTER(_i32))
_sig("rt_render_time_ms", C.c_int, _vp, _pd, C.POINTER(_i32), _i32)
_sig("rt_stream", _vp, _vp)
_sig("rt_version", C.c_char_p)
_sig("rth_last_error", C.c_char_p)
_sig("rth_mesh_create", _vp, _pd, C.c_int, C.c_int)
_sig("rth_mesh_destroy", None, _vp)
_sig("rth_mesh_read_obj", C.c_int, _vp, C.c_char_p)
_sig("rth_mesh_load_texture", C.c_int, _vp, C.c_char_p)
_sig("rth_mesh_add_texture", C.c_int, _vp, _vp, _i32, _i32)
_sig("rth_mesh_texture", C.c_int, _vp, _i32, C.POINTER(_i32), C.POINTER(_i32), _vp)
_sig("rth_decode_jpeg", C.c_int, _vp, _i64, C.POINTER(_i32), C.POINTER(_i32), _vp, _i64)
_sig("rth_mesh_set_raw", C.c_int, _vp, _i64, _vp, _i64, _vp, _i64, _vp, _i64, _vp)
_sig("rth_mesh_op", C.c_int, _vp, C.c_int, C.c_int, C.c_int, _dbl)
_sig("rth_mesh_build_bvh", C.c_int, _vp)
_sig("rth_mesh_build_bvh_device", C.c_int, _vp, _vp)
_sig("rth_mesh_sizes", C.c_int, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i32),
     C.POINTER(_i64), C.POINTER(_i32))
_sig("rth_mesh_get", C.c_int, _vp, _vp, _vp, _vp, _vp, _vp)
_sig("rth_mesh_bvh_dump", _i64, _vp, _vp, _vp, _i64)
_sig("rth_scene_create", _vp, _dbl, _pd)
_sig("rth_scene_destroy", None, _vp)
_sig("rth_scene_add_sphere", C.c_int, _vp, _pd, _dbl, _pd, C.c_int, C.c_int)
_sig("rth_scene_add_mesh", C.c_int, _vp, _vp)
_sig("rth_scene_upload", C.c_int, _vp, _vp, C.POINTER(_i64))
_sig("rth_scene_render_devices", C.c_int, _vp, C.POINTER(rt_camera), C.POINTER(rt_params), _vp, _i32, _vp)
_sig("rth_scene_flat_info", C.c_int, _vp, C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64), C.POINTER(_i64))
_sig("rth_scene_flat_arrays", C.c_int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64))
_sig("rth_camera_make", C.c_int, _pd, _dbl, _dbl, _dbl, _i32, _dbl, _dbl, _dbl, C.POINTER(rt_camera))
_sig("rth_params_default", C.c_int, _i32, _i32, _i32, C.POINTER(rt_params))
_sig("rth_write_png", C.c_int, C.c_char_p, _i32, _i32, _vp)
_sig("rth_write_execution_time", C.c_int, C.c_char_p, _dbl)


class RTError(RuntimeError):
    pass


def _d3(v):
    return (C.c_double * 3)(*[float(x) for x in v])


def _ptr(a):
    return a.ctypes.data_as(_vp) if a is not None else None


def _host_check(rc):
    if rc != 0:
        raise RTError(lib.rth_last_error().decode())


def make_camera(Cpos=(0, 30, 55), fov_deg=60.0, angle_v_deg=-25.0, angle_h_deg=0.0, W=1024, focus=55.0, aa_sigma=0.25,
                dof_sigma=0.01):
    """Camera of main(): raytracer.cpp:1162, :1441-1453, :1468, :1472, :1479."""
    cam = rt_camera()
    lib.rth_camera_make(_d3(Cpos), fov_deg, angle_v_deg, angle_h_deg, W, focus, aa_sigma, dof_sigma, C.byref(cam))
    return cam


def make_params(W, H, spp, spp_offset=0, seed=10, flags=0, rect=None):
    p = rt_params()
    lib.rth_params_default(W, H, spp, C.byref(p))
    p.spp_offset = spp_offset
    p.seed = seed
    p.flags = flags
    if rect is not None:
        for k in range(4):
            p.rect[k] = int(rect[k])
    return p


OPS = {"invertNormals": 0, "translate": 1, "rotateNormals": 2, "rotate": 3, "rotateAroundAxe": 4, "zoom": 5}


class TriangleMesh:
    """TriangleMesh of the reference (raytracer.cpp:208-872) backed by the C++ host class."""

    def __init__(self, albedo=(1, 1, 1), mirror=False, transp=False):
        self.albedo, self.mirror, self.transp = tuple(float(x) for x in albedo), bool(mirror), bool(transp)
        self._h = lib.rth_mesh_create(_d3(albedo), int(mirror), int(transp))
        self.textures = []

    def __del__(self):
        if getattr(self, "_h", None) and lib is not None:  # at interpreter shutdown the module's globals may be gone already
            lib.rth_mesh_destroy(self._h)
            self._h = None

    def readOBJ(self, path):
        _host_check(lib.rth_mesh_read_obj(self._h, os.fsencode(path)))
        return self

    def set_raw(self, verts, normals, uvs, idx10):
        v = np.ascontiguousarray(verts, np.float64).reshape(-1, 3)
        n = np.ascontiguousarray(normals, np.float64).reshape(-1, 3)
        uv = np.ascontiguousarray(uvs, np.float64).reshape(-1, 2)
        ix = np.ascontiguousarray(idx10, np.int32).reshape(-1, 10)
        _host_check(lib.rth_mesh_set_raw(self._h, len(v), _ptr(v), len(n), _ptr(n), len(uv), _ptr(uv), len(ix), _ptr(ix)))
        return self

    def addTexture(self, rgb):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        assert rgb.ndim == 3 and rgb.shape[2] == 3
        _host_check(lib.rth_mesh_add_texture(self._h, _ptr(rgb), rgb.shape[1], rgb.shape[0]))
        self.textures.append(rgb)
        return self

    def loadTexture(self, path):
        """TriangleMesh::loadTexture (raytracer.cpp:759-766) through the C++ loader: JPEG files with the texels of the
        reference's stbi_load call (:762), binary PPM files as they are."""
        _host_check(lib.rth_mesh_load_texture(self._h, os.fsencode(path)))
        k = len(self.textures)
        W, H = _i32(), _i32()
        _host_check(lib.rth_mesh_texture(self._h, k, C.byref(W), C.byref(H), None))
        rgb = np.empty((H.value, W.value, 3), np.uint8)
        _host_check(lib.rth_mesh_texture(self._h, k, C.byref(W), C.byref(H), _ptr(rgb)))
        self.textures.append(rgb)
        return self

    def _op(self, name, a=0, b=0, angle=0.0):
        _host_check(lib.rth_mesh_op(self._h, OPS[name], int(a), int(b), float(angle)))
        return self

    def invertNormals(self):
        return self._op("invertNormals")

    def translate(self, d, dist):
        return self._op("translate", d, int(dist))

    def rotateNormals(self, a, b):
        return self._op("rotateNormals", a, b)

    def rotate(self, a, b):
        return self._op("rotate", a, b)

    def rotateAroundAxe(self, axe, angle):
        return self._op("rotateAroundAxe", axe, 0, angle)

    def zoom(self, k):
        return self._op("zoom", 0, int(k))  # int argument: zoom(0.8) == zoom(0), raytracer.cpp:840, :1285

    def buildBVH(self, renderer=None):
        """TriangleMesh::buildBVH (raytracer.cpp:244-291) on the host, or on the GPU of `renderer`: same result."""
        if renderer is None:
            _host_check(lib.rth_mesh_build_bvh(self._h))
        else:
            _host_check(lib.rth_mesh_build_bvh_device(self._h, renderer._h))
        return self

    def sizes(self):
        nv, nn, nuv, nt, nnod = _i64(), _i64(), _i64(), _i64(), _i64()
        ntex, md = _i32(), _i32()
        lib.rth_mesh_sizes(self._h, C.byref(nv), C.byref(nn), C.byref(nuv), C.byref(nt), C.byref(ntex), C.byref(nnod), C.byref(md))
        return dict(nv=nv.value, nn=nn.value, nuv=nuv.value, nt=nt.value, ntex=ntex.value, n_nodes=nnod.value, max_depth=md.value)

    def arrays(self):
        """Current vertices, normals, uvs, TriangleIndices (n,10) and permutation (after buildBVH)."""
        s = self.sizes()
        v = np.empty((s["nv"], 3)); n = np.empty((s["nn"], 3)); uv = np.empty((s["nuv"], 2))
        ix = np.empty((s["nt"], 10), np.int32)
        perm = np.empty(s["nt"], np.int32) if s["n_nodes"] else None
        lib.rth_mesh_get(self._h, _ptr(v), _ptr(n), _ptr(uv), _ptr(ix), _ptr(perm))
        return v, n, uv, ix, perm

    def bvh_dump(self):
        s = self.sizes()
        boxes = np.empty((s["n_nodes"], 6)); dfl = np.empty((s["n_nodes"], 3), np.int32)
        k = lib.rth_mesh_bvh_dump(self._h, _ptr(boxes), _ptr(dfl), s["n_nodes"])
        assert k == s["n_nodes"]
        return boxes, dfl


class Sphere:
    """Sphere(O, R, albedo, isMirror, isTransparent), raytracer.cpp:877."""

    def __init__(self, O, R, albedo, mirror=False, transp=False):
        self.O, self.R = tuple(float(x) for x in O), float(R)
        self.albedo, self.mirror, self.transp = tuple(float(x) for x in albedo), bool(mirror), bool(transp)


class Scene:
    """Scene (raytracer.cpp:923-929): objects in order, light position L, intensity I."""

    def __init__(self, I=6e9, L=(-10, 50, 40)):
        self.I, self.L = float(I), tuple(float(x) for x in L)
        self.objects = []

    def add(self, obj):
        self.objects.append(obj)
        return len(self.objects) - 1

    def _build_handle(self):
        h = lib.rth_scene_create(self.I, _d3(self.L))
        for o in self.objects:
            if isinstance(o, Sphere):
                lib.rth_scene_add_sphere(h, _d3(o.O), o.R, _d3(o.albedo), int(o.mirror), int(o.transp))
            else:
                lib.rth_scene_add_mesh(h, o._h)
        return h

    def flat_arrays(self):
        """the flattened scene as numpy record arrays with the layouts of rt_amd.h (what upload() hands to rt_upload_scene)"""
        info = self.flat_info()
        h = self._build_handle()
        try:
            nt, nb = _i64(), _i64()
            _host_check(lib.rth_scene_flat_arrays(h, None, None, None, None, None, None, None, C.byref(nt), C.byref(nb)))
            out = dict(spheres=np.zeros(info["n_spheres"], SPHERE_DTYPE), meshes=np.zeros(info["n_meshes"], MESH_DTYPE),
                       nodes=np.zeros(info["n_nodes"], NODE_DTYPE), tris=np.zeros(info["n_tris"], TRI_DTYPE),
                       attrs=np.zeros(info["n_tris"], ATTR_DTYPE), tex=np.zeros(nt.value, TEX_DTYPE), texels=np.zeros(nb.value, np.uint8))
            _host_check(lib.rth_scene_flat_arrays(h, *[C.c_void_p(out[k].ctypes.data) for k in ("spheres", "meshes", "nodes", "tris", "attrs", "tex", "texels")],
                                                  None, None))
            return out
        finally:
            lib.rth_scene_destroy(h)

    def flat_info(self):
        h = self._build_handle()
        try:
            a = [_i64() for _ in range(5)]
            _host_check(lib.rth_scene_flat_info(h, *[C.byref(x) for x in a]))
            return dict(zip(["n_spheres", "n_meshes", "n_nodes", "n_tris", "bytes"], [x.value for x in a]))
        finally:
            lib.rth_scene_destroy(h)


class Renderer:
    """One context = one HIP device + stream (rt_create .. rt_destroy)."""

    def __init__(self, device=0):
        h = _vp()
        rc = lib.rt_create(device, C.byref(h))
        if rc != 0:
            raise RTError("rt_create: " + lib.rt_last_error(None).decode())
        self._h = h
        self.flat_bytes = 0

    def close(self):
        if getattr(self, "_h", None) and lib is not None:  # at interpreter shutdown the module's globals may be gone already
            lib.rt_destroy(self._h)
            self._h = None

    __del__ = close

    def _check(self, rc):
        if rc != 0:
            raise RTError(lib.rt_last_error(self._h).decode())

    def upload(self, scene):
        h = scene._build_handle()
        try:
            nb = _i64()
            rc = lib.rth_scene_upload(h, self._h, C.byref(nb))
            if rc != 0:
                raise RTError(lib.rth_last_error().decode())
            self.flat_bytes = nb.value
        finally:
            lib.rth_scene_destroy(h)

    def accum_clear(self, W, H):
        self._check(lib.rt_accum_clear(self._h, W, H))

    def render(self, cam, params, accum_ptr=None):
        """Adds params.spp samples per pixel into the accumulator (own one if accum_ptr is None)."""
        self._check(lib.rt_render(self._h, C.byref(cam), C.byref(params), accum_ptr))

    def render_static(self, cam, params, accum_ptr=None):
        """rt_render without any device allocation (static scratch of the library; see rt_render_static in rt_amd.h)."""
        self._check(lib.rt_render_static(self._h, C.byref(cam), C.byref(params), accum_ptr))

    def sync(self):
        self._check(lib.rt_sync(self._h))

    def accum_download(self, W, H):
        out = np.empty((H, W, 3), np.float64)
        self._check(lib.rt_accum_download(self._h, _ptr(out), out.size))
        return out

    def accum_upload(self, acc):
        acc = np.ascontiguousarray(acc, np.float64)
        self._check(lib.rt_accum_upload(self._h, _ptr(acc), acc.size))

    def accum_add(self, other):
        """own accumulator += other's own accumulator (another context, on this GPU or another one)"""
        self._check(lib.rt_accum_add(self._h, other._h))

    def accum_device_ptr(self):
        return lib.rt_accum_device_ptr(self._h)

    def resolve(self, W, H, spp_total, accum_ptr=None):
        out = np.empty((H, W, 3), np.uint8)
        self._check(lib.rt_resolve(self._h, accum_ptr, W, H, float(spp_total), _ptr(out)))
        return out

    def intersect_rays(self, rays):
        rays = np.ascontiguousarray(rays, np.float64).reshape(-1, 6)
        n = len(rays)
        obj = np.empty(n, np.int32); tri = np.empty(n, np.int32); t = np.empty(n)
        PN = np.empty((n, 6)); alb = np.empty((n, 3))
        self._check(lib.rt_intersect_rays(self._h, _ptr(rays), n, _ptr(obj), _ptr(tri), _ptr(t), _ptr(PN), _ptr(alb)))
        return dict(obj=obj, tri=tri, t=t, P=PN[:, :3].copy(), N=PN[:, 3:].copy(), albedo=alb)

    def sample_radiance(self, cam, params, ijk):
        ijk = np.ascontiguousarray(ijk, np.int32).reshape(-1, 3)
        out = np.empty((len(ijk), 3))
        self._check(lib.rt_sample_radiance(self._h, C.byref(cam), C.byref(params), _ptr(ijk), len(ijk), _ptr(out)))
        return out

    def sample_trace(self, cam, params, ijk):
        """radiance (n, 3) and the path of every sample (n, RT_TRACE_LEN, 4): object id, triangle id, draws used, RT_TRACE_* code"""
        ijk = np.ascontiguousarray(ijk, np.int32).reshape(-1, 3)
        out = np.empty((len(ijk), 3))
        trace = np.empty((len(ijk), RT_TRACE_LEN, 4), np.int32)
        self._check(lib.rt_sample_trace(self._h, C.byref(cam), C.byref(params), _ptr(ijk), len(ijk), _ptr(out), _ptr(trace)))
        return out, trace

    def counters(self):
        c = rt_counters()
        self._check(lib.rt_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def reset_counters(self):
        self._check(lib.rt_reset_counters(self._h))

    def last_render_ms(self):
        ms, n = C.c_double(), _i32()
        self._check(lib.rt_last_render_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def render_time_ms(self, reset=True):
        """(total milliseconds, kernel launches) of the render calls since the last reset; waits for them"""
        ms, n = C.c_double(), _i32()
        self._check(lib.rt_render_time_ms(self._h, C.byref(ms), C.byref(n), 1 if reset else 0))
        return ms.value, n.value

    def stream(self):
        return lib.rt_stream(self._h)


def write_png(path, image):
    """The image (H, W, 3) uint8, rows top to bottom, as a PNG file (raytracer.cpp:1506)."""
    image = np.ascontiguousarray(image, np.uint8)
    H, W = image.shape[:2]
    _host_check(lib.rth_write_png(os.fsencode(path), W, H, _ptr(image)))


def write_execution_time(path, seconds):
    """Whole seconds of a duration, as the reference's execution_time.txt has them (raytracer.cpp:1509-1512)."""
    _host_check(lib.rth_write_execution_time(os.fsencode(path), float(seconds)))


def render(scene, cam, W, H, spp, seed=10, flags=0, device=0):
    """Whole-frame convenience: what the reference's main() does up to the PNG write."""
    r = Renderer(device)
    try:
        r.upload(scene)
        r.accum_clear(W, H)
        r.render(cam, make_params(W, H, spp, seed=seed, flags=flags))
        return r.resolve(W, H, spp)
    finally:
        r.close()


def render_devices(scene, cam, W, H, spp, devices, seed=10, flags=0):
    """rt::render over several GPUs of one process: the samples split over `devices`, summed on the first, tone-mapped"""
    h = scene._build_handle()
    try:
        out = np.empty((H, W, 3), np.uint8)
        dv = np.ascontiguousarray(devices, np.int32)
        p = make_params(W, H, spp, seed=seed, flags=flags)
        _host_check(lib.rth_scene_render_devices(h, C.byref(cam), C.byref(p), _ptr(dv), len(dv), _ptr(out)))
        return out
    finally:
        lib.rth_scene_destroy(h)


def decode_jpeg(data):
    """the loader's JPEG decoder on the bytes of a file: (H, W, 3) uint8"""
    buf = np.frombuffer(bytes(data), np.uint8)
    W, H = _i32(), _i32()
    _host_check(lib.rth_decode_jpeg(_ptr(buf), buf.size, C.byref(W), C.byref(H), None, 0))
    out = np.empty((H.value, W.value, 3), np.uint8)
    _host_check(lib.rth_decode_jpeg(_ptr(buf), buf.size, C.byref(W), C.byref(H), _ptr(out), out.size))
    return out


def version():
    return lib.rt_version().decode()
