#!/usr/bin/env python3
"""Random scenes against the CPU checker: python tests/fuzz_parity.py [first_seed [n_scenes [triangles per filter-tree leaf: 0 = by scene size, 4]]]

Every scene is a random mix of what the hand-written tests cover one at a time: few or many spheres (the list staged in
LDS, or the sphere tree), equal and nested spheres, zero to four meshes of 12 to 24 300 triangles, a mesh repeated in place
(equal t in two objects), triangles repeated inside a mesh (equal t in one), a mesh in a coordinate plane, mirror and
glass materials on spheres and meshes. Probes: random rays, axis-parallel rays, rays aimed exactly at vertices and edge
midpoints of the meshes (beta, gamma on the acceptance boundary), with and without distance pruning - object, triangle
and t must be the checker's bit for bit - and per-sample radiance with the path taken, bounce by bounce.
Prints one line per scene and exits non-zero at the first difference."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
rta = importlib.import_module("ray-tracing_amd")
scenes = importlib.import_module("ray-tracing_amd.scenes")
ob = importlib.import_module("oracle.bindings")


def make_scene(seed):
    rng = np.random.default_rng(seed)
    s = rta.Scene(I=float(rng.choice([6e9, 2e9])), L=(-10.0, 50.0, 40.0))
    s.add(rta.Sphere(s.L, float(rng.choice([5.0, 1.0])), (1, 1, 1)))       # objects[0]: the light
    notes = []
    n_sph = int(rng.choice([0, 2, 5, 9, 70, 150]))
    for k in range(n_sph):
        O = rng.uniform(-25, 25, 3) + np.array([0, 15, 0])
        R = float(rng.uniform(0.5, 6.0))
        kind = rng.integers(0, 6)
        sp = rta.Sphere(tuple(O), R, tuple(rng.uniform(0.1, 1.0, 3)), mirror=(kind == 0), transp=(kind == 1))
        s.add(sp)
        if k % 7 == 3:                                                     # the same sphere twice: the lower objectId must win
            s.add(rta.Sphere(tuple(O), R, tuple(rng.uniform(0.1, 1.0, 3))))
        if k % 11 == 5:                                                    # and one inside it
            s.add(rta.Sphere(tuple(O), R * 0.5, tuple(rng.uniform(0.1, 1.0, 3)), transp=True))
    if rng.random() < 0.7:                                                 # walls as huge spheres, as the reference does
        s.add(rta.Sphere((0, -2000 - 2, 0), 2000, (1, 1, 1)))
        s.add(rta.Sphere((0, 0, -2000 - 60), 2000, (0, 1, 0)))
    n_mesh = int(rng.integers(0, 5))
    verts = []
    for k in range(n_mesh):
        n = int(rng.choice([1, 2, 3, 6, 12, 20, 20, 45]))
        size = tuple(rng.uniform(2.0, 9.0, 3))
        off = tuple(rng.uniform(-18, 18, 3) + np.array([0, 12, 0]))
        v, nn, uv, ix = scenes.blob_mesh(n, size=size, offset=off, seed=int(rng.integers(1, 1000)))
        variant = int(rng.integers(0, 6))
        if variant == 0:                                                   # every triangle twice
            ix = np.concatenate([ix, ix])
            notes.append("doubled")
        if variant == 1:                                                   # squashed into a coordinate plane
            v = v.copy(); v[:, int(rng.integers(0, 3))] = off[0]
            notes.append("flat")
        kind = rng.integers(0, 8)
        reps = 2 if variant == 2 else 1                                    # the same mesh twice, as two objects
        if reps == 2:
            notes.append("twin")
        for r in range(reps):
            m = rta.TriangleMesh(tuple(rng.uniform(0.1, 1.0, 3)), mirror=(kind == 0), transp=(kind == 1))
            m.set_raw(v, nn, uv, ix)
            if rng.random() < 0.5:
                m.addTexture(scenes.procedural_texture(16, seed=int(rng.integers(1, 99))))
            m.buildBVH()
            s.add(m)
        verts.append((v, ix))
    return s, verts, n_sph, n_mesh, ",".join(notes)


def make_rays(rng, verts, n):
    org = np.stack([rng.uniform(-30, 30, n), rng.uniform(-5, 40, n), rng.uniform(-40, 50, n)], 1)
    d = rng.normal(size=(n, 3))
    d /= np.sqrt((d * d).sum(1))[:, None]
    rays = [np.concatenate([org, d], 1)]
    ax = np.zeros((60, 3)); ax[np.arange(60), np.arange(60) % 3] = np.where(np.arange(60) % 2, 1.0, -1.0)
    rays.append(np.concatenate([org[:60], ax], 1))
    for v, ix in verts:
        tri = ix[rng.integers(0, len(ix), 300)]
        A, B, Cc = v[tri[:, 0]], v[tri[:, 1]], v[tri[:, 2]]
        for tgt in (A, (A + B) / 2, (A + B + Cc) / 3, B):
            o = org[:len(tgt)]
            dd = tgt - o
            nrm = np.sqrt((dd * dd).sum(1))
            ok = nrm > 0
            rays.append(np.concatenate([o[ok], dd[ok] / nrm[ok][:, None]], 1))
    return np.ascontiguousarray(np.concatenate(rays))


def check(seed, gpu, own_leaf=0):
    """own_leaf: 0 = the filter tree's leaf size by the library's rule (two triangles at these sizes), 4 = the form large scenes get"""
    assert rta.lib.rt_debug_own_leaf(gpu._h, own_leaf) == 0
    try:
        return _check(seed, gpu)
    finally:
        rta.lib.rt_debug_own_leaf(gpu._h, 0)


def _check(seed, gpu):
    s, verts, n_sph, n_mesh, notes = make_scene(seed)
    rng = np.random.default_rng(seed + 100000)
    orc = ob.OracleScene(scenes.export_objects(s), s.I, s.L, math="portable")
    gpu.upload(s)
    rays = make_rays(rng, verts, 3000)
    o = orc.intersect_rays(rays)
    for prune in (1, 0):
        rta.lib.rt_debug_probe_prune(gpu._h, prune)
        g = gpu.intersect_rays(rays)
        for key in ("obj", "tri", "t"):
            if not np.array_equal(g[key], o[key], equal_nan=(key == "t")):
                bad = np.nonzero(g[key] != o[key])[0]
                print("seed %d: %s differs on %d rays (prune=%d); first: ray %s gpu %s checker %s" %
                      (seed, key, len(bad), prune, rays[bad[0]], {k: g[k][bad[0]] for k in ("obj", "tri", "t")}, {k: o[k][bad[0]] for k in ("obj", "tri", "t")}))
                return False
    rta.lib.rt_debug_probe_prune(gpu._h, 1)
    W = 24
    cam = rta.make_camera((0, 30, 55), 60.0, -25.0, 0.0, W, dof_sigma=0.01 if seed % 3 else 0.0)
    flags = (0, rta.RT_F_POINT_LIGHT, rta.RT_F_NO_FRESNEL, rta.RT_F_NO_INDIRECT, 0, rta.RT_F_NO_JITTER)[seed % 6]
    p = rta.make_params(W, W, 3, flags=flags)
    ijk = np.array([(i, j, k) for i in range(W) for j in range(W) for k in range(3)], np.int32)
    g, tg = gpu.sample_trace(cam, p, ijk)
    c, tc = orc.sample_trace(cam, p, ijk)
    # the same samples without the trace: rays whose answer cannot matter are then not traced at all (zero direct term,
    # beyond the depth limit, shadow ray already blocked by a sphere); the value must keep every bit
    lean = gpu.sample_radiance(cam, p, ijk)
    if not np.array_equal(lean, g, equal_nan=True):
        bad = np.nonzero((lean != g) & ~(np.isnan(lean) & np.isnan(g)))[0]
        print("seed %d: %d samples differ between the traced and the untraced form; first: %s %s vs %s" % (seed, len(bad), ijk[bad[0]], lean[bad[0]], g[bad[0]]))
        return False
    both_nan = np.isnan(g) & np.isnan(c)
    rel = np.abs(g - c) / np.maximum(np.abs(c), 1e-3)
    rel[both_nan] = 0
    rel[np.isnan(rel)] = np.inf
    worst = float(rel.max())
    paths_differ = int((tg != tc).any(axis=(1, 2)).sum())
    hits = int((o["obj"] >= 0).sum())
    print("seed %d: %d spheres, %d meshes [%s], %d objects; %d rays (%d hit, %d on triangles): equal; %d samples: worst rel %.1e, %d paths differ" %
          (seed, n_sph, n_mesh, notes, len(s.objects), len(rays), hits, int((o["tri"] >= 0).sum()), len(ijk), worst, paths_differ), flush=True)
    return worst <= 1e-9 and paths_differ == 0


def main():
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    own_leaf = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    gpu = rta.Renderer(0)
    for seed in range(first, first + n):
        if not check(seed, gpu, own_leaf):
            sys.exit(1)
    print("all %d scenes equal" % n)


if __name__ == "__main__":
    main()
