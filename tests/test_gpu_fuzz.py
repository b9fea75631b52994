"""Random scene mixes (tests/fuzz_parity.py, which also runs on its own over any range of seeds) against the oracle: equal and nested spheres, sphere lists on either side of
the LDS limit, meshes repeated in place, triangles repeated inside a mesh, meshes in coordinate planes, all materials, rays
aimed at vertices and edge midpoints. Hit identity and t bit for bit with and without pruning; every sample within 1e-9
and on the same path."""
import importlib.util
import os

import pytest

pytestmark = pytest.mark.gpu

_spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(os.path.abspath(__file__)), "fuzz_parity.py"))
fuzz = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(fuzz)


@pytest.mark.parametrize("seed", [3, 8, 27, 30, 34, 35, 36])
def test_random_scene(gpu, seed):
    assert fuzz.check(seed, gpu)


@pytest.mark.parametrize("seed", [8, 27, 34, 36])
def test_random_scene_with_the_large_scene_tree(gpu, seed):
    """the same with filter-tree leaves of four triangles and four lanes per leaf item, the form scenes beyond 2 M triangles get"""
    assert fuzz.check(seed, gpu, own_leaf=4)
