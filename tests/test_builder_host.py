"""The filter-tree builder and its worker pool on the CPU, under the address and undefined-behaviour sanitizers
(tests/builder_check.hip: structure of the trees, independence of the thread count)."""
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
