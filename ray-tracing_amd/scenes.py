"""Scene builders for the five benchmark configurations and procedural stand-in geometry.

Geometry is generated with +, -, *, / and sqrt only (no sin/cos/exp), so every machine produces
the same vertex bits and the golden fixtures under tests/golden/ stay valid anywhere.

Scenes follow main() of the reference (/root/reference/raytracer.cpp):
  sphere room  : light :1164-1167, S1..S3 :1171-1173, walls/floor :1174-1178
  battle scene : object order and transform chains :1222-1355, :1401-1435
The reference's OBJ/JPG assets are third-party files that are not shipped with this repository;
procedural meshes of matching triangle count, object-space size and transform chain stand in for
them. `assets_dir` lets a caller use the real files where they exist.
"""
import os

import numpy as np

from . import Scene, Sphere, TriangleMesh, make_camera

WALL = (138 / 255., 168 / 255., 167 / 255.)  # raytracer.cpp:1174-1177


# ---------------------------------------------------------------------------------------------
# procedural geometry
# ---------------------------------------------------------------------------------------------
def _hash01(ix, seed):
    """integer hash -> float64 in [0,1), exact integer arithmetic"""
    with np.errstate(over="ignore"):
        x = np.asarray(ix).astype(np.uint64) + np.uint64((int(seed) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF)
        x = x ^ (x >> np.uint64(30))
        x = x * np.uint64(0xBF58476D1CE4E5B9)
        x = x ^ (x >> np.uint64(27))
        x = x * np.uint64(0x94D049BB133111EB)
        x = x ^ (x >> np.uint64(31))
    return (x >> np.uint64(11)).astype(np.float64) / 9007199254740992.0


def blob_mesh(n, size=(1.0, 1.0, 1.0), offset=(0.0, 0.0, 0.0), seed=1, bump=0.15, n_groups=1):
    """Closed lumpy surface: a cube-sphere (6 faces of n x n quads = 12 n^2 triangles) displaced by a
    low-order polynomial, scaled by `size` and moved by `offset`. Returns verts, normals, uvs,
    TriangleIndices rows (n,10) in face-fan order (each quad -> (0,1,2), (0,2,3), as the reference's
    OBJ reader fans polygons, raytracer.cpp:487-636)."""
    with np.errstate(over="ignore"):
        c = _hash01(np.arange(12), seed) * 2.0 - 1.0
    g = np.arange(n + 1, dtype=np.float64) * (2.0 / n) - 1.0
    U, V = np.meshgrid(g, g, indexing="ij")
    one = np.ones_like(U)
    faces = [(one, U, V), (-one, V, U), (V, one, U), (U, -one, V), (U, V, one), (V, U, -one)]
    verts, normals, uvs, idx = [], [], [], []
    base = 0
    sx, sy, sz = size
    for f, (X, Y, Z) in enumerate(faces):
        inv = 1.0 / np.sqrt(X * X + Y * Y + Z * Z)
        x, y, z = X * inv, Y * inv, Z * inv

        def radius(x, y, z):
            return 1.0 + bump * (c[0] * x * y + c[1] * y * z + c[2] * z * x + c[3] * x * x * y + c[4] * y * y * z +
                                 c[5] * z * z * x + c[6] * x * y * z * 2.0 + 0.5 * c[7] * (x * x - y * y))
        r = radius(x, y, z)
        P = np.stack([sx * r * x + offset[0], sy * r * y + offset[1], sz * r * z + offset[2]], -1)
        # normals from finite differences on the grid (one-sided at the face border)
        dU = np.empty_like(P)
        dV = np.empty_like(P)
        dU[1:-1] = P[2:] - P[:-2]
        dU[0] = P[1] - P[0]
        dU[-1] = P[-1] - P[-2]
        dV[:, 1:-1] = P[:, 2:] - P[:, :-2]
        dV[:, 0] = P[:, 1] - P[:, 0]
        dV[:, -1] = P[:, -1] - P[:, -2]
        N = np.cross(dU, dV)
        # orient outwards
        s = np.where((N * (P - np.asarray(offset))).sum(-1) < 0, -1.0, 1.0)
        N = N * s[..., None]
        N = N / np.sqrt((N * N).sum(-1))[..., None]
        UVc = np.stack([(U + 1.0) * 0.5 * (1.0 / 3.0) + (f % 3) * (1.0 / 3.0), (V + 1.0) * 0.25 + (f // 3) * 0.5], -1)
        verts.append(P.reshape(-1, 3))
        normals.append(N.reshape(-1, 3))
        uvs.append(UVc.reshape(-1, 2))
        ii, jj = np.meshgrid(np.arange(n), np.arange(n), indexing="ij")
        a = (ii * (n + 1) + jj).ravel() + base
        b = ((ii + 1) * (n + 1) + jj).ravel() + base
        cc = ((ii + 1) * (n + 1) + jj + 1).ravel() + base
        d = (ii * (n + 1) + jj + 1).ravel() + base
        # keep the winding outward-facing: a,b,cc,d is counter-clockwise for even faces
        if f % 2 == 1:
            b, d = d, b
        grp = np.full(a.shape, (f * n_groups) // 6, np.int64)
        t1 = np.stack([a, b, cc, a, b, cc, a, b, cc, grp], -1)
        t2 = np.stack([a, cc, d, a, cc, d, a, cc, d, grp], -1)
        quad = np.empty((2 * len(a), 10), np.int64)
        quad[0::2] = t1
        quad[1::2] = t2
        idx.append(quad)
        base += (n + 1) * (n + 1)
    return (np.concatenate(verts), np.concatenate(normals), np.concatenate(uvs), np.concatenate(idx).astype(np.int32))


def grid_for_triangles(n_tris):
    """cube-sphere resolution whose 12 n^2 triangles come closest to n_tris"""
    return max(1, int(round((n_tris / 12.0) ** 0.5)))


def procedural_texture(size=1024, seed=7):
    """RGB8 camouflage-like pattern, (size, size, 3) uint8, integer arithmetic only."""
    y, x = np.meshgrid(np.arange(size, dtype=np.uint64), np.arange(size, dtype=np.uint64), indexing="ij")
    with np.errstate(over="ignore"):
        cell = (y // np.uint64(32)) * np.uint64(4096) + (x // np.uint64(32))
        a = _hash01(cell, seed)
        fine = _hash01(y * np.uint64(65536) + x, seed + 1)
    base = np.array([[70, 90, 50], [110, 100, 70], [60, 70, 60], [150, 140, 110]], np.float64)
    col = base[(a * 4).astype(np.int64)] * (0.85 + 0.3 * fine[..., None])
    return np.clip(col, 0, 255).astype(np.uint8)


def write_obj(path, verts, normals, uvs, idx10, usemtl=True):
    """Write the mesh as an OBJ the reference's reader accepts (quads are written as triangles)."""
    with open(path, "w") as f:
        for v in verts:
            f.write("v %s %s %s\n" % (repr(float(v[0])), repr(float(v[1])), repr(float(v[2]))))
        for v in normals:
            f.write("vn %s %s %s\n" % (repr(float(v[0])), repr(float(v[1])), repr(float(v[2]))))
        for v in uvs:
            f.write("vt %s %s\n" % (repr(float(v[0])), repr(float(v[1]))))
        g = -1
        for t in idx10:
            while usemtl and g < t[9]:
                g += 1
                f.write("usemtl m%d\n" % g)
            f.write("f %d/%d/%d %d/%d/%d %d/%d/%d\n" % (t[0] + 1, t[3] + 1, t[6] + 1, t[1] + 1, t[4] + 1, t[7] + 1, t[2] + 1, t[5] + 1, t[8] + 1))


# ---------------------------------------------------------------------------------------------
# scenes
# ---------------------------------------------------------------------------------------------
def _room(scene, wall_colours=None, floor_mirror=False, ceiling=False):
    cols = wall_colours or [WALL] * 4
    scene.add(Sphere((-1050, 0, 0), 970, cols[0]))
    scene.add(Sphere((1050, 0, 0), 970, cols[1]))
    scene.add(Sphere((0, 0, -1050), 940, cols[2]))
    scene.add(Sphere((0, 0, 1000), 940, cols[3]))
    scene.add(Sphere((0, -1000, 0), 990, (1, 1, 1), floor_mirror))
    if ceiling:
        scene.add(Sphere((0, 1000, 0), 990, (1, 1, 1)))


README_WALLS = [(0, 0, 1), (1, 0, 0), (0, 1, 0), (1, 0, 1)]  # trailing comments of raytracer.cpp:1174-1177


def sphere_room_deterministic(W=512):
    """cfg1: S1 diffuse / S2 mirror / S3 glass + walls + floor, point light, hard shadows, 1 spp, no
    jitter. No light sphere: with a point light objects[0] is an ordinary object."""
    from . import RT_F_NO_FRESNEL, RT_F_NO_INDIRECT, RT_F_NO_JITTER, RT_F_POINT_LIGHT
    s = Scene(I=6e9, L=(-10, 50, 40))
    s.add(Sphere((-15, 0, 0), 10, (0, 0, 1)))
    s.add(Sphere((0, 0, 0), 10, (1, 1, 1), True))
    s.add(Sphere((15, 0, 0), 10, (1, 0, 0), False, True))
    _room(s, README_WALLS)
    cam = make_camera((0, 30, 55), 60.0, -25.0, 0.0, W)
    flags = RT_F_POINT_LIGHT | RT_F_NO_INDIRECT | RT_F_NO_FRESNEL | RT_F_NO_JITTER
    return s, cam, flags


def sphere_room(W=512, dof_sigma=0.01):
    """cfg2: light sphere + S1..S3 + walls + floor with the committed estimator (area light,
    indirect bounce, stochastic Fresnel, antialiasing)."""
    s = Scene(I=6e9, L=(-10, 50, 40))
    s.add(Sphere(s.L, 5, (1, 1, 1)))
    s.add(Sphere((-15, 0, 0), 10, (0, 0, 1)))
    s.add(Sphere((0, 0, 0), 10, (1, 1, 1), True))
    s.add(Sphere((15, 0, 0), 10, (1, 0, 0), False, True))
    _room(s)
    cam = make_camera((0, 30, 55), 60.0, -25.0, 0.0, W, dof_sigma=dof_sigma)
    return s, cam, 0


def _standin(n_tris, size, z0, seed, n_tex=1, tex_size=1024, albedo=(0, 0, 1)):
    """procedural stand-in for a z-up OBJ model of the given object-space half-extents"""
    n = grid_for_triangles(n_tris)
    v, nn, uv, ix = blob_mesh(n, size=size, offset=(0.0, 0.0, z0), seed=seed, n_groups=max(1, n_tex))
    m = TriangleMesh(albedo)
    m.set_raw(v, nn, uv, ix)
    for k in range(n_tex):
        m.addTexture(procedural_texture(tex_size, seed * 10 + k))
    return m


def mesh_room(W=512, n_tris=30000, tex_size=512):
    """cfg3: the dog scene (commented code raytracer.cpp:1197-1207, :1416) with a procedural mesh of
    about n_tris triangles in the dog's place, textured, on a mirror floor."""
    s = Scene(I=6e9, L=(-10, 50, 40))
    s.add(Sphere(s.L, 5, (1, 1, 1)))
    _room(s, floor_mirror=True)
    m = _standin(n_tris, (9.0, 6.0, 8.0), 8.5, seed=3, n_tex=1, tex_size=tex_size, albedo=(0, 1, 1))
    m.zoom(0)  # zoom(0.3) truncates to 0, :1198
    m.rotate(1, 2).rotate(0, 2).translate(1, -10).invertNormals().rotateNormals(1, 2).rotateNormals(0, 2)
    m.invertNormals()  # the stand-in's normals already point outwards
    m.buildBVH()
    s.add(m)
    cam = make_camera((0, 30, 55), 60.0, -25.0, 0.0, W)
    return s, cam, 0


# (name, triangles, object-space half extents x,y and height z, zoom, textures, albedo, ops after zoom)
_BATTLE = [
    ("grenade_soldier", 60000, (0.30, 0.60, 0.72), 10, 1, (0, 0, 1),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 1, -10), ("rotateAroundAxe", 1, -30), ("translate", 0, 30), ("translate", 2, -15)]),
    ("grenade_soldier", 60000, (0.30, 0.60, 0.72), 10, 1, (0, 0, 1),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 0, 5), ("translate", 1, -10), ("rotateAroundAxe", 1, -40), ("translate", 0, 18), ("translate", 2, 13)]),
    ("rifle_soldier", 54320, (0.29, 0.58, 0.72), 4, 1, (0, 0, 1),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 1, -10), ("rotateAroundAxe", 1, -30), ("translate", 0, 25), ("translate", 2, -20)]),
    ("rifle_soldier", 54320, (0.29, 0.58, 0.72), 4, 1, (0, 0, 1),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 1, -10), ("rotateAroundAxe", 1, -10), ("translate", 0, 15), ("translate", 2, 18)]),
    ("bazooka_soldier", 56000, (0.35, 0.60, 0.72), 4, 1, (0, 0, 1),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 1, -10), ("rotateAroundAxe", 1, -30), ("translate", 0, 20), ("translate", 2, -18)]),
    ("bazooka_soldier", 56000, (0.35, 0.60, 0.72), 4, 1, (0, 0, 1),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 1, -10), ("rotateAroundAxe", 1, -10), ("translate", 0, 28), ("translate", 2, 15)]),
    ("fighter_jet", 18272, (4.74, 4.98, 1.9), 0, 0, (16. / 255, 133. / 255, 49. / 255),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 1, 28), ("translate", 0, -15), ("rotateAroundAxe", 1, 30), ("rotateAroundAxe", 0, 15)]),
    ("ki61", 40000, (6.0, 4.4, 1.8), 4, 0, (1, 0, 0),
     [("rotate", 1, 2), ("rotate", 0, 2), ("rotateAroundAxe", 1, -30), ("translate", 1, 31), ("translate", 0, 5), ("translate", 2, 25)]),
    ("t70", 70000, (1.2, 2.1, 1.0), 4, 3, (0, 0, 1),
     [("rotate", 1, 2), ("translate", 1, -11), ("rotate", 0, 2), ("translate", 2, 0), ("rotateAroundAxe", 1, 70), ("translate", 0, 20)]),
    ("t70", 70000, (1.2, 2.1, 1.0), 4, 3, (0, 0, 1),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 2, 0), ("rotateAroundAxe", 1, 95), ("translate", 0, 25), ("translate", 1, -11), ("translate", 2, -30)]),
    ("panzer3", 80000, (1.45, 2.7, 1.25), 5, 3, (1, 0, 0),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 2, 0), ("rotateAroundAxe", 1, -60), ("translate", 1, -11), ("translate", 0, -20)]),
    ("panzer3", 80000, (1.45, 2.7, 1.25), 5, 3, (1, 0, 0),
     [("rotate", 1, 2), ("rotate", 0, 2), ("translate", 2, 0), ("rotateAroundAxe", 1, -90), ("translate", 1, -11), ("translate", 0, -25), ("translate", 2, -30)]),
]

_REAL = {  # files of the reference's objects/ directory that exist there
    "rifle_soldier": ("soldiers/rifle_soldier/14070_WWII_Soldier_with_Rife_v1_L1.obj", ["soldiers/rifle_soldier/14070_WWII_Soldier_with_Rifle_diff.jpg"]),
    "fighter_jet": ("airplane/10593_Fighter_Jet_SG_v1_iterations-2.obj", []),
}


def _host_threads():
    """OMP_NUM_THREADS when set (a scheduler's share of the machine), else the CPU count"""
    try:
        t = int(os.environ.get("OMP_NUM_THREADS", "0"))
    except ValueError:
        t = 0
    return t if t > 0 else (os.cpu_count() or 1)


def battle(W=1024, tri_scale=1.0, tex_size=1024, assets_dir=None, dof_sigma=0.01):
    """cfg4: the committed scene of main(): light + 4 walls + floor + 12 meshes in the reference's
    object order with its transform chains (raytracer.cpp:1222-1355, :1401-1435). Meshes are
    procedural stand-ins unless assets_dir holds the reference's objects/ files."""
    s = Scene(I=6e9, L=(-10, 50, 40))
    s.add(Sphere(s.L, 5, (1, 1, 1)))
    _room(s)
    def make(job):
        k, (name, ntri, ext, zoom, ntex, albedo, ops) = job
        real = None
        if assets_dir and name in _REAL and os.path.exists(os.path.join(assets_dir, _REAL[name][0])):
            real = _REAL[name]
        if real:
            m = TriangleMesh(albedo)
            m.readOBJ(os.path.join(assets_dir, real[0]))
            for t in real[1]:
                m.loadTexture(os.path.join(assets_dir, t))
        else:
            m = _standin(max(12, int(ntri * tri_scale)), ext, ext[2], seed=100 + k, n_tex=ntex, tex_size=tex_size, albedo=albedo)
        m.zoom(zoom)
        for op in ops:
            getattr(m, op[0])(*op[1:])
        m.buildBVH()
        return m, dict(name=name, real=bool(real), **m.sizes())

    # the meshes do not depend on each other: they are loaded (or generated), transformed and given their BVH by a few
    # threads (the array work and the C++ calls run without the interpreter lock); the scene takes them in the reference's order
    from concurrent.futures import ThreadPoolExecutor
    workers = max(1, min(len(_BATTLE), _host_threads()))
    with ThreadPoolExecutor(max_workers=workers) as pool:
        made = list(pool.map(make, enumerate(_BATTLE)))
    meta = []
    for m, info in made:
        s.add(m)
        meta.append(info)
    cam = make_camera((0, 30, 55), 60.0, -25.0, 0.0, W, dof_sigma=dof_sigma)
    return s, cam, 0, meta


def stress(W=2048, n_tris=5_000_000, n_spheres=1000, seed=12345):
    """cfg5: one ~n_tris-triangle procedural mesh + n_spheres random spheres (70 % diffuse, 15 % mirror,
    15 % glass) + light + walls."""
    s = Scene(I=6e9, L=(-10, 50, 40))
    s.add(Sphere(s.L, 5, (1, 1, 1)))
    _room(s)
    with np.errstate(over="ignore"):
        u = _hash01(np.arange(n_spheres * 8), seed).reshape(n_spheres, 8)
    for k in range(n_spheres):
        O = (-45 + 90 * u[k, 0], -8 + 40 * u[k, 1], -60 + 90 * u[k, 2])
        R = 0.5 + 1.5 * u[k, 3]
        kind = u[k, 4]
        s.add(Sphere(O, R, (0.2 + 0.8 * u[k, 5], 0.2 + 0.8 * u[k, 6], 0.2 + 0.8 * u[k, 7]), 0.70 <= kind < 0.85, kind >= 0.85))
    n = grid_for_triangles(n_tris)
    v, nn, uv, ix = blob_mesh(n, size=(40.0, 6.0, 40.0), offset=(0.0, -14.0, -10.0), seed=seed, bump=0.3)
    m = TriangleMesh((0.8, 0.7, 0.5))
    m.set_raw(v, nn, uv, ix)
    m.buildBVH()
    s.add(m)
    cam = make_camera((0, 30, 55), 60.0, -25.0, 0.0, W)
    return s, cam, 0


def grid_mesh(n, f):
    """(n x n)-quad grid mesh whose vertex (a, b), a, b = 0..n, lies at f(a, b): verts, normals, uvs, idx10"""
    ii, jj = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    v = np.ascontiguousarray(f(ii.ravel().astype(np.float64), jj.ravel().astype(np.float64)), np.float64)
    nrm = np.tile(np.array([[0.0, 1.0, 0.0]]), (len(v), 1))
    uv = np.stack([ii.ravel() / n, jj.ravel() / n], 1).astype(np.float64)
    q = (ii[:-1, :-1] * (n + 1) + jj[:-1, :-1]).ravel()
    tris = np.concatenate([np.stack([q, q + 1, q + n + 2], 1), np.stack([q, q + n + 2, q + n + 1], 1)])
    ix = np.zeros((len(tris), 10), np.int32)
    ix[:, 0:3] = tris
    ix[:, 3:6] = tris
    ix[:, 6:9] = tris
    return v, nrm, uv, ix


def axis_aligned(W=64):
    """Geometry in coordinate planes. A leaf whose box has no thickness along an axis is never entered by the
    reference (tMax > tMin cannot hold, raytracer.cpp:184): object 2, a grid in the plane y = 2, is invisible as
    a whole; object 3, a staircase of axis-aligned faces as truncation to integers produces them
    (rotateAroundAxe, raytracer.cpp:831-834), is visible only where its leaves have a volume."""
    s = Scene(I=6e9, L=(-10, 50, 40))
    s.add(Sphere(s.L, 5, (1, 1, 1)))
    s.add(Sphere((0, -1000, 0), 990, (1, 1, 1)))
    flat = TriangleMesh((1, 0, 0))
    flat.set_raw(*grid_mesh(24, lambda a, b: np.stack([a - 12, np.full_like(a, 2.0), b - 12], 1)))
    flat.buildBVH()
    s.add(flat)
    stairs = TriangleMesh((0, 1, 0))
    stairs.set_raw(*grid_mesh(40, lambda a, b: np.stack([np.trunc(a * 0.3) - 6, np.trunc((a + b) * 0.2) - 4, b * 0.5 - 10], 1)))
    stairs.buildBVH()
    s.add(stairs)
    cam = make_camera((0, 30, 55), 60.0, -25.0, 0.0, W)
    return s, cam, 0


def rays_into_box(n, seed, org_lo, org_hi, tgt_lo, tgt_hi):
    """n seeded rays from random origins in one box towards random targets in another: (n, 6) C, u"""
    rng = np.random.default_rng(seed)
    org = rng.uniform(org_lo, org_hi, (n, 3))
    d = rng.uniform(tgt_lo, tgt_hi, (n, 3)) - org
    d /= np.sqrt((d * d).sum(1))[:, None]
    return np.ascontiguousarray(np.concatenate([org, d], 1))


def export_objects(scene):
    """Plain-data view of a scene for checkers: list of dicts with final (post-transform) arrays.
    Call it on meshes BEFORE or AFTER buildBVH alike: indices are returned in face-fan order."""
    out = []
    for o in scene.objects:
        if isinstance(o, Sphere):
            out.append(dict(kind="sphere", O=o.O, R=o.R, albedo=o.albedo, mirror=o.mirror, transp=o.transp))
        else:
            v, n, uv, ix, perm = o.arrays()
            if perm is not None:  # undo the BVH permutation
                orig = np.empty_like(ix)
                orig[perm] = ix
                ix = orig
            out.append(dict(kind="mesh", verts=v, normals=n, uvs=uv, idx=ix, albedo=o.albedo, mirror=o.mirror, transp=o.transp,
                            textures=list(o.textures)))
    return out
