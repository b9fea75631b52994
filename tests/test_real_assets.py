"""CPU, only where the reference tree with its objects/ directory exists (the OBJ and JPG files are third-party assets and
do not travel): the real rifle soldier (twice, with the two transform chains of raytracer.cpp:1243-1261) and the real
fighter jet (:1285-1292) go through the product's whole host chain - readOBJ, loadTexture, the six transforms, buildBVH,
flatten - inside the battle scene, and what comes out (the rt_tri / rt_tri_attr / rt_bvh_node / texel arrays that
rt_upload_scene receives) is compared, by SHA-256 and element by element, with the same arrays assembled from what the
reference's own classes hold after the same calls (oracle/_ref/libref_true.so: TriangleMesh::vertices / normals / uvs /
indices, the Noeud tree, the stbi_load texels)."""
import hashlib
import importlib
import os

import numpy as np
import pytest

REF_OBJECTS = "/root/reference/objects"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF_OBJECTS), reason="the reference's assets are not on this machine")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def expected_from_reference(t, oid, mesh_index, first_tri, textured, rta):
    """rt_tri, rt_tri_attr and the tree in a numbering-free form, from the reference's arrays of object `oid`"""
    v, n, uv, ix, perm = t.mesh_arrays(oid)
    boxes, dfl = t.bvh_dump(oid)  # preorder, fg before fd: (debut, fin, is leaf)
    leaves = [k for k in range(len(dfl)) if dfl[k, 2]]
    leaves.sort(key=lambda k: -dfl[k, 0])  # the traversal pops fd before fg: leaves in descending debut (raytracer.cpp:656-665)
    visit = np.concatenate([np.arange(dfl[k, 0], dfl[k, 1]) for k in leaves])
    new_first = {int(dfl[k, 0]): first_tri + int(s) for k, s in zip(leaves, np.cumsum([0] + [dfl[k, 1] - dfl[k, 0] for k in leaves])[:-1])}
    assert len(visit) == len(ix) and len(set(visit.tolist())) == len(ix)
    tri = np.zeros(len(ix), rta.TRI_DTYPE)
    q = ix[visit]
    A = v[q[:, 0]]
    tri["A"], tri["e1"], tri["e2"] = A, v[q[:, 1]] - A, v[q[:, 2]] - A  # raytracer.cpp:673-677
    tri["orig_id"], tri["mesh"] = perm[visit], mesh_index
    at = np.zeros(len(ix), rta.ATTR_DTYPE)
    for c in range(3):
        at["n"][:, c] = n[q[:, 6 + c]]
        if textured:
            at["uv"][:, c] = uv[q[:, 3 + c]]
    at["group"] = q[:, 9]
    # the tree without record numbers: preorder over the inner Noeuds, per node both child boxes and, for a leaf child,
    # (first position in the visiting order, count), for an inner child (-1, 0)
    seq = []
    k = 0

    def walk():
        nonlocal k
        me = k
        k += 1
        if dfl[me, 2]:
            return ("leaf", me)
        a = walk()
        b = walk()
        row = []
        for kind, c in (a, b):
            row += list(boxes[c])
            row += [new_first[int(dfl[c, 0])], int(dfl[c, 1] - dfl[c, 0])] if kind == "leaf" else [-1, 0]
        seq.append((me, row))
        return ("inner", me)

    import sys
    sys.setrecursionlimit(10000)
    walk()
    seq.sort()
    return tri, at, np.array([r for _, r in seq]), boxes[0]


def canonical_tree(nodes, root):
    """the product's rt_bvh_node records of one mesh in the same numbering-free form"""
    out = []
    stack = [root]
    while stack:  # preorder, L before R
        r = nodes[stack.pop()]
        row = []
        for box, c, cnt in ((r["boxL"], r["L"], r["nL"]), (r["boxR"], r["R"], r["nR"])):
            row += list(box)
            row += [int(c), int(cnt)] if cnt > 0 else [-1, 0]
        out.append(row)
        if r["nR"] == 0:
            stack.append(int(r["R"]))
        if r["nL"] == 0:
            stack.append(int(r["L"]))
    return np.array(out)


def test_real_meshes_flatten_to_the_references_arrays(rta, scenes, ob):
    if not ob.have_trueref():
        pytest.skip("oracle/_ref not built")
    s, cam, flags, meta = scenes.battle(W=64, tri_scale=0.002, tex_size=8, assets_dir=REF_OBJECTS)  # stand-ins tiny, real files whole
    real = [k for k, m in enumerate(meta) if m["real"]]
    assert [meta[k]["name"] for k in real] == ["rifle_soldier", "rifle_soldier", "fighter_jet"]
    flat = s.flat_arrays()
    objs = []
    for k in real:
        name, _, _, zoom, ntex, albedo, ops = scenes._BATTLE[k]
        obj, texs = scenes._REAL[name]
        objs.append(dict(kind="mesh", obj_path=os.path.join(REF_OBJECTS, obj), texture_paths=[os.path.join(REF_OBJECTS, x) for x in texs],
                         ops=[("zoom", zoom)] + list(ops), albedo=albedo, mirror=False, transp=False))
    t = ob.TrueRefScene(objs, 6e9, (-10, 50, 40))
    digests = {}
    for oid, k in enumerate(real):
        me = flat["meshes"][k]
        assert me["object_id"] == 6 + k
        lo, hi = int(me["first_tri"]), int(me["first_tri"] + me["n_tris"])
        tri, at, tree, root_box = expected_from_reference(t, oid, k, lo, me["n_tex"] > 0, rta)
        assert hi - lo == len(tri) == (54320 if meta[k]["name"] == "rifle_soldier" else 18272)
        got_tri, got_at = flat["tris"][lo:hi], flat["attrs"][lo:hi]
        for f in ("A", "e1", "e2", "orig_id", "mesh"):
            assert np.array_equal(got_tri[f], tri[f]), (k, f)
        for f in ("n", "uv", "group"):
            assert np.array_equal(got_at[f], at[f]), (k, f)
        assert sha(got_tri) == sha(tri) and sha(got_at) == sha(at)
        assert np.array_equal(me["root_box"], root_box)
        got_tree = canonical_tree(flat["nodes"], int(me["root"]))
        assert got_tree.shape == tree.shape and np.array_equal(got_tree, tree), k
        assert sha(got_tree) == sha(tree)
        for j in range(int(me["n_tex"])):
            tx = flat["tex"][int(me["first_tex"]) + j]
            want = t.mesh_texture(oid, j)
            assert (tx["H"], tx["W"]) == want.shape[:2]
            got = flat["texels"][int(tx["offset"]):int(tx["offset"]) + want.size]
            assert sha(got) == sha(want)
        digests[k] = (sha(got_tri), sha(got_tree))
    # the two soldiers are one file under two chains: different arrays, same topology size
    assert digests[real[0]] != digests[real[1]]
