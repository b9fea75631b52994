"""CPU: the product's host side (C++ classes behind include/rt_amd_host.h) - OBJ reader, mesh transforms,
BVH build, camera basis, flattening - against the golden outputs of the real reference, and against the
real reference itself where it is built. No GPU call is made."""
import ctypes as C
import hashlib
import importlib
import os
import re

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLD = np.load(os.path.join(HERE, "golden", "reference_outputs.npz"))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_library_exports_every_declared_symbol(rta):
    declared = []
    for h in ("rt_amd.h", "rt_amd_host.h"):
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared += re.findall(r"\b(rt_[a-z_0-9]+|rth_[a-z_0-9]+)\s*\(", text)
    declared = sorted(set(declared))
    assert len(declared) >= 35
    assert set(declared) == set(rta.RT_SYMBOLS + rta.RTH_SYMBOLS)
    for name in declared:
        assert hasattr(rta.lib, name), name
    assert b"gfx950" in rta.lib.rt_version()


def test_pod_layouts_match_the_header(rta):
    assert C.sizeof(rta.rt_camera) == 16 * 8
    assert C.sizeof(rta.rt_params) == 4 * 4 + 8 + 4 + 4 + 3 * 8 + 16
    assert C.sizeof(rta.rt_counters) == 8 * 8


def test_camera_basis(rta):
    for key, args in (("camera_default_1024", ((0, 30, 55), 60.0, -25.0, 0.0, 1024)), ("camera_m30_10_512", ((0, 30, 55), 60.0, -30.0, 10.0, 512))):
        c = rta.make_camera(*args)
        got = np.array(list(c.right) + list(c.up) + list(c.view) + [c.zdist])
        assert np.allclose(got, GOLD[key], rtol=1e-15, atol=1e-16)  # cos/sin/tan of this machine's libm
    c = rta.make_camera()
    assert (c.focus, c.aa_sigma, c.dof_sigma) == (55.0, 0.25, 0.01)
    p = rta.make_params(64, 32, 5)
    assert (p.max_rebond, p.ior, p.epsilon, p.shadow_bias, p.seed) == (10, 1.4, 1e-5, 1e-4, 10)


def test_obj_reader(rta):
    m = rta.TriangleMesh()
    m.readOBJ(os.path.join(HERE, "golden", "reader_cases.obj"))
    v, n, uv, ix, _ = m.arrays()
    assert np.array_equal(v, GOLD["reader_verts"]) and np.array_equal(n, GOLD["reader_normals"]) and np.array_equal(uv, GOLD["reader_uvs"])
    assert np.array_equal(ix, GOLD["reader_idx"]), (ix, GOLD["reader_idx"])
    assert len(ix) == 2 + 1 + 3 + 2 + 1 + 3 + 1 + 2 + 2 + 1 + 1
    with pytest.raises(rta.RTError):
        rta.TriangleMesh().readOBJ("/nonexistent/file.obj")


def test_transform_chains_and_bvh(rta):
    mg = importlib.import_module("tests.golden.make_golden")
    for name, ops in mg.MESH_CHAINS.items():
        v, n, uv, ix = mg.chain_mesh(name)
        m = rta.TriangleMesh().set_raw(v, n, uv, ix)
        for op in ops:
            getattr(m, op[0])(*op[1:])
        m.buildBVH()
        v2, n2, uv2, ix2, perm = m.arrays()
        boxes, dfl = m.bvh_dump()
        hv, hn, hi, hp, hb, hd = GOLD["chain_%s" % name]
        assert np.array_equal(v2[:4], GOLD["chain_%s_v0" % name])
        assert sha(v2) == hv and sha(n2) == hn, name
        assert sha(ix2) == hi and sha(perm) == hp, name
        assert len(boxes) == GOLD["chain_%s_nodes" % name][0] and sha(boxes) == hb and sha(dfl) == hd, name


def test_zoom_takes_an_int(rta):
    """zoom(0.8) is zoom(0): the argument is an int (raytracer.cpp:840, :1285); zoom(k) scales by 1 + k"""
    v = np.array([[0.25, 0.5, 1.0], [1.5, -2.0, 0.125], [0.1, 0.2, 0.3]])
    n = np.array([[0, 0, 1.0]] * 3)
    ix = np.array([[0, 1, 2, 0, 0, 0, 0, 1, 2, 0]], np.int32)
    m = rta.TriangleMesh().set_raw(v, n, np.zeros((1, 2)), ix)
    m.zoom(0.8)
    assert np.array_equal(m.arrays()[0], v)
    m.zoom(4)
    assert np.array_equal(m.arrays()[0], v + 4 * v)
    m.translate(1, 0)  # distance 0 mirrors the axis (raytracer.cpp:786-792)
    assert np.array_equal(m.arrays()[0][:, 1], -(v + 4 * v)[:, 1])


def test_flatten_layout(rta, scenes):
    s, cam, flags, meta = scenes.battle(W=64, tri_scale=0.02, tex_size=16)
    info = s.flat_info()
    assert info["n_spheres"] == 6 and info["n_meshes"] == 12
    assert info["n_tris"] == sum(m["nt"] for m in meta)
    assert info["bytes"] > info["n_tris"] * (80 + 128)
    bad = rta.Scene()
    bad.add(rta.Sphere((0, 0, 0), 1, (1, 1, 1)))
    bad.add(rta.TriangleMesh())  # no triangles, no BVH
    with pytest.raises(rta.RTError):
        bad.flat_info()


def test_missing_extension_fails_loudly(tmp_path):
    """the product has no CPU fallback: without the HIP library the import raises"""
    import shutil
    import subprocess
    import sys
    dst = tmp_path / "ray-tracing_amd"
    shutil.copytree(os.path.join(ROOT, "ray-tracing_amd"), dst, ignore=shutil.ignore_patterns("*.so", "__pycache__"))
    code = "import importlib,sys; sys.path.insert(0, %r); importlib.import_module('ray-tracing_amd')" % str(tmp_path)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True)
    assert r.returncode != 0 and "HIP extension not built" in r.stderr


@pytest.mark.skipif(not os.path.isdir("/root/reference/objects"), reason="reference assets absent")
def test_host_against_the_real_reference_on_real_assets(rta, ob):
    """readOBJ + the jet's transform chain (with its int truncation) + buildBVH on the real OBJ files"""
    if not ob.have_trueref():
        pytest.skip("oracle/_ref not built")
    mg = importlib.import_module("tests.golden.make_golden")
    for rel, chain in (("airplane/10593_Fighter_Jet_SG_v1_iterations-2.obj", "fighter_jet"), ("dumbell/10499_Dumbells_v1_L3.obj", "t70"),
                       ("soldiers/rifle_soldier/14070_WWII_Soldier_with_Rife_v1_L1.obj", "rifle_soldier")):
        path = os.path.join("/root/reference/objects", rel)
        ops = mg.MESH_CHAINS[chain]
        t = ob.TrueRefScene([dict(kind="mesh", obj_path=path, ops=ops, albedo=(1, 1, 1), mirror=False, transp=False)], 1.0, (0, 0, 0))
        m = rta.TriangleMesh().readOBJ(path)
        for op in ops:
            getattr(m, op[0])(*op[1:])
        m.buildBVH()
        for a, b in zip(m.arrays(), t.mesh_arrays(0)):
            assert np.array_equal(a, b), rel
        for a, b in zip(m.bvh_dump(), t.bvh_dump(0)):
            assert np.array_equal(a, b), rel


def test_png_and_execution_time_output(rta, tmp_path):
    """the output step of main() (raytracer.cpp:1505-1512): the PNG decodes to the same pixels, also beyond one 64 KiB block"""
    from PIL import Image
    rng = np.random.default_rng(3)
    for H, W in ((1, 1), (37, 53), (300, 200)):
        img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
        path = str(tmp_path / ("o%d.png" % H))
        rta.write_png(path, img)
        with Image.open(path) as im:
            assert im.mode == "RGB" and im.size == (W, H)
            assert np.array_equal(np.asarray(im), img)
    rta.write_execution_time(str(tmp_path / "execution_time.txt"), 12.9)
    assert open(str(tmp_path / "execution_time.txt")).read() == "12"
    with pytest.raises(rta.RTError):
        rta.write_png(str(tmp_path / "no_such_dir" / "x.png"), np.zeros((2, 2, 3), np.uint8))



def _jpeg_cases(tmp_path):
    """JPEG files of every layout the decoder handles, written with PIL: sampling 4:4:4 / 4:2:2 / 4:2:0, baseline and
    progressive, restart intervals, optimised tables, grey, sizes that are not multiples of the MCU"""
    from PIL import Image
    rng = np.random.default_rng(7)
    cases = []
    for k, (w, h) in enumerate([(64, 48), (37, 23), (129, 65), (16, 16), (1, 1), (9, 200)]):
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.stack([(xx * 5 + yy * 3) % 256, (xx * xx + 7 * yy) % 256, (255 - 2 * xx + yy * yy // 3) % 256], -1).astype(np.uint8)
        img = np.where(rng.random((h, w, 1)) < 0.2, rng.integers(0, 256, (h, w, 3)), img).astype(np.uint8)
        for sub in (0, 1, 2):
            for prog in (False, True):
                for extra in ({}, {"optimize": True}, {"restart_marker_blocks": 3}):
                    name = tmp_path / ("c%d_s%d_p%d_%s.jpg" % (k, sub, prog, "_".join(extra) or "plain"))
                    Image.fromarray(img).save(name, "JPEG", quality=[35, 75, 92][(k + sub) % 3], subsampling=sub, progressive=prog, **extra)
                    cases.append(str(name))
        name = tmp_path / ("c%d_grey.jpg" % k)
        Image.fromarray(img[:, :, 0]).save(name, "JPEG", quality=80)
        cases.append(str(name))
    return cases


def test_jpeg_decoder_equals_the_reference_decoder(rta, ob, tmp_path):
    """TriangleMesh::loadTexture on JPEG files: every texel equal to what the reference's own loader (stbi_load(..., 3),
    raytracer.cpp:762) returns, on generated files of every layout"""
    if not ob.have_trueref():
        pytest.skip("oracle/_ref not built")
    stub = dict(kind="mesh", verts=np.zeros((3, 3)), normals=np.zeros((3, 3)), uvs=np.zeros((3, 2)), idx=np.zeros((1, 10), np.int32),
                albedo=(1, 1, 1), mirror=False, transp=False)
    cases = _jpeg_cases(tmp_path)
    assert len(cases) > 100
    for f in cases:
        ref = ob.TrueRefScene([dict(stub, texture_paths=[f])], 1.0, (0, 0, 0), build_bvh=False).mesh_texture(0, 0)
        got = rta.TriangleMesh().loadTexture(f).textures[0]
        assert got.shape == ref.shape and np.array_equal(got, ref), f
    with pytest.raises(rta.RTError):
        rta.decode_jpeg(open(cases[0], "rb").read()[:200])  # truncated
    with pytest.raises(rta.RTError):
        rta.decode_jpeg(b"\xff\xd8\xff\xd9")  # no frame


def test_reference_textures_decode_to_the_reference_texels(rta):
    """the 13 JPEG textures of the reference's scenes (raytracer.cpp:1223, :1244, ...): SHA-256 of the decoded image equal to
    the fixture made from the reference's own loader (tests/golden/make_golden.py). The files are third-party assets and
    exist only where the reference tree does."""
    import json
    fix = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "texture_sha256.json")))
    assert len(fix) == 13
    root = "/root/reference/objects"
    if not os.path.isdir(root):
        pytest.skip("the reference's assets are not on this machine")
    for rel, want in fix.items():
        data = open(os.path.join(root, rel), "rb").read()
        assert hashlib.sha256(data).hexdigest() == want["file_sha256"], rel
        img = rta.decode_jpeg(data)
        assert img.shape == (want["H"], want["W"], 3), rel
        assert hashlib.sha256(np.ascontiguousarray(img).tobytes()).hexdigest() == want["sha256"], rel


def test_render_devices_reports_a_device_that_cannot_be_opened(rta, scenes):
    """rt::render(scene, cam, params, image, devices) with device ids that do not exist (here: no GPU at all; on a GPU box:
    ids beyond the device count): every thread that was started is joined, every context destroyed, and the first error
    comes back as an exception with the HIP text - no crash, no hang, and the library still answers afterwards."""
    W = 16
    s, cam, flags = scenes.sphere_room(W)
    for devices in ([99], [99, 98, 97]):
        with pytest.raises(rta.RTError) as e:
            rta.render_devices(s, cam, W, W, 2, devices, flags=flags)
        assert "rt_create" in str(e.value)
    assert b"gfx950" in rta.lib.rt_version()
