"""Host code under the address and undefined-behaviour sanitizers, on the CPU: the filter-tree builder and its worker pool
(tests/builder_check.hip: structure of the trees, independence of the thread count) and the JPEG texture decoder on mutated
files (tests/jpeg_fuzz.cpp), the OBJ reader, buildBVH and flatten on mutated files (tests/obj_fuzz.cpp)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_builder_structure_and_thread_independence(tmp_path):
    exe = str(tmp_path / "builder_check")
    cmd = [HIPCC, "-O1", "-g", "-std=c++17", "-Xarch_host", "-fsanitize=address", "-Xarch_host", "-fsanitize=undefined", "-Xarch_host",
           "-fno-sanitize-recover=undefined", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "ray-tracing_amd", "csrc"),
           "--offload-arch=gfx950", os.path.join(ROOT, "tests", "builder_check.hip"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
    out = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "builder checks passed" in out.stdout


def test_jpeg_decoder_on_mutated_files(tmp_path):
    PIL = pytest.importorskip("PIL.Image")
    import numpy as np
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not installed")
    rng = np.random.default_rng(3)
    files = []
    for size in ((64, 64), (33, 17), (8, 8)):
        for mode in ("RGB", "L"):
            a = (rng.random((size[1], size[0], 3)) * 255).astype(np.uint8)
            img = PIL.fromarray(a).resize((size[0] * 2, size[1] * 2)).resize(size)
            if mode == "L":
                img = img.convert("L")
            for sub in (0, 1, 2):
                for prog in (False, True):
                    fn = str(tmp_path / ("t%d.jpg" % len(files)))
                    kw = dict(quality=30 + 20 * sub, progressive=prog, optimize=bool(len(files) % 2))
                    if mode == "RGB":
                        kw["subsampling"] = sub
                    img.save(fn, "JPEG", **kw)
                    files.append(fn)
    exe = str(tmp_path / "jpeg_fuzz")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "ray-tracing_amd", "host"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "jpeg_fuzz.cpp"), os.path.join(ROOT, "ray-tracing_amd", "host", "rt_jpeg.cpp"), "-o", exe]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", ITERS="40")
    out = subprocess.run([exe] + files, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.startswith("decoded ")
    # Fixed cases the mutations do not reach. (1) Sampling factors that do not divide the largest one: a 4:2:0 file whose Cb
    # component claims h = 3 - up-sampling by h_max / h = 1 would read a full-width row from a half-width plane.
    # (2) A short file that declares a 65535 x 65535 picture: refused before any plane is allocated.
    a = (rng.random((16, 24, 3)) * 255).astype(np.uint8)
    fn = str(tmp_path / "sub420.jpg")
    PIL.fromarray(a).save(fn, "JPEG", quality=80, subsampling=2)
    raw = bytearray(open(fn, "rb").read())
    sof = raw.index(b"\xff\xc0")
    assert raw[sof + 9] == 3 and raw[sof + 11] == 0x22 and raw[sof + 14] == 0x11  # three components, Y 2x2, Cb 1x1
    bad_factors = bytearray(raw)
    bad_factors[sof + 14] = 0x31
    huge = bytearray(raw)
    huge[sof + 5:sof + 9] = b"\xff\xff\xff\xff"
    cases = []
    for name, data in (("bad_factors.jpg", bad_factors), ("huge.jpg", huge)):
        cases.append(str(tmp_path / name))
        open(cases[-1], "wb").write(bytes(data))
    out = subprocess.run([exe] + cases, env=dict(env, EXPECT_REFUSED="1"), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert "bad sampling factors" in out.stdout and ("too large" in out.stdout or "too short" in out.stdout)


def test_obj_reader_on_mutated_files(tmp_path):
    gxx = shutil.which("g++")
    if not gxx:
        pytest.skip("g++ not installed")
    exe = str(tmp_path / "obj_fuzz")
    cmd = [gxx, "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
           "-I" + os.path.join(ROOT, "ray-tracing_amd", "host"), "-I" + os.path.join(ROOT, "ray-tracing_amd", "csrc"), "-I" + os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "obj_fuzz.cpp"), os.path.join(ROOT, "ray-tracing_amd", "host", "rt_host.cpp"),
           os.path.join(ROOT, "ray-tracing_amd", "host", "rt_jpeg.cpp"), "-o", exe, "-lpthread"]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0", ITERS="600", OBJ_TMP=str(tmp_path / "mutant.obj"))
    out = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "reader_cases.obj")], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert out.stdout.startswith("read ")
