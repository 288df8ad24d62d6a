"""SURVEY.md section 8 row a-20 on the device: the recursive-grid BVH builder as HIP kernels
(chroma_bvh_build_device, csrc/bvh_device.hip -- leaf boxes and Morton codes, device radix sort, parent unions per
layer, concatenate + offset, collapse; reference: chroma/cuda/bvh.cu:149-203,270-308,365-384,530-543 driven by
chroma/bvh/grid.py:11-95).  The node array must equal the host builder's and the NumPy restatement's bit for bit, and
the committed vectors of tests/golden/bvh_golden.json; and the device form of tools.argsort_direction."""
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    from chroma_amd import gpu as g
    ctx = g.create_cuda_context(0)
    yield g
    ctx.pop()


def _meshes():
    """The cases of tests/golden/bvh_golden.json (tools/gen_bvh_golden.py) + the 3 M-triangle C2-lite detector."""
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from gen_bvh_golden import CASES
    from chroma_amd import demo
    from chroma_amd.loader import create_geometry_from_obj
    for name, build in CASES.items():
        yield name, create_geometry_from_obj(build(), ).mesh
    yield 'lite', create_geometry_from_obj(demo.detector_lite()).mesh


def _sha(nodes):
    return hashlib.sha256(np.ascontiguousarray(nodes).tobytes()).hexdigest()


def test_device_builder_equals_the_host_builders(gpu):
    from chroma_amd.bvh.grid import make_recursive_grid_bvh
    golden = json.load(open(os.path.join(GOLDEN, 'bvh_golden.json')))
    for name, mesh in _meshes():
        for degree in (2, 3, 4):
            dev = make_recursive_grid_bvh(mesh, target_degree=degree, backend='device')
            host = make_recursive_grid_bvh(mesh, target_degree=degree, backend='native')
            assert len(dev.nodes) == len(host.nodes) and dev.layer_offsets == host.layer_offsets, (name, degree)
            assert np.array_equal(dev.nodes.view(np.uint32), host.nodes.view(np.uint32)), (name, degree)
            if len(mesh.triangles) < 500000:
                ref = make_recursive_grid_bvh(mesh, target_degree=degree, backend='numpy')
                assert np.array_equal(dev.nodes.view(np.uint32), ref.nodes.view(np.uint32)), (name, degree)
        if name in golden:                                        # the committed vectors (degree 3)
            dev = make_recursive_grid_bvh(mesh, backend='device')
            assert _sha(dev.nodes) == golden[name]['nodes_sha256'], name


def test_device_builder_is_the_default_with_a_gpu_and_refuses_bad_input(gpu):
    from chroma_amd import make, _lib
    from chroma_amd.geometry import Mesh
    from chroma_amd.bvh.grid import make_recursive_grid_bvh
    mesh = make.cube(10.0)
    a = make_recursive_grid_bvh(mesh, verbose=True)               # (prints "BVH (device)")
    b = make_recursive_grid_bvh(mesh, backend='native')
    assert np.array_equal(a.nodes.view(np.uint32), b.nodes.view(np.uint32))
    bad = Mesh(mesh.vertices, mesh.triangles.copy())
    bad.triangles[3, 1] = len(mesh.vertices)                     # a vertex index outside the mesh
    with pytest.raises(_lib.ChromaError):
        make_recursive_grid_bvh(bad, backend='device')


def test_direction_sort_is_argsort_direction_on_the_device(gpu, oracle_mod):
    """GPUPhotons.sort_by_direction == chroma_amd.tools.argsort_direction (chroma/tools.py:175-193) applied to every
    array of the set: codes non-decreasing, the same multiset of photons; equal to the NumPy order wherever the two
    arc functions give the same 16-bit angles (they differ in the last ulp for a few per million)."""
    from chroma_amd.tools import argsort_direction
    ph = oracle_mod.generate_bomb(200000, seed=3, id_base=0, wavelength_lo=300.0, wavelength_hi=700.0)
    gp = gpu.GPUPhotons(ph)
    gp.sort_by_direction()
    got = gp.get()

    def codes(d):
        maxint = 2 ** 16 - 1
        theta = (np.arccos(np.clip(d[:, 2], -1, 1)) / np.pi * maxint).astype(np.uint32)
        phi = ((np.arctan2(d[:, 1], d[:, 0]) / np.pi / 2.0 + 0.5) * maxint).astype(np.uint32)
        m = np.zeros(len(d), dtype=np.uint32)
        for i in range(16):
            bit = np.uint32(1 << i)
            m |= ((theta & bit) << np.uint32(i)) | ((phi & bit) << np.uint32(i + 1))
        return m
    c = codes(got.dir.astype(np.float64)).astype(np.int64)
    assert np.count_nonzero(np.diff(c) < 0) < 1e-4 * len(c)                      # sorted (up to last-ulp angle differences)
    order = argsort_direction(ph.dir)
    want = ph[order]
    same = (got.dir.view(np.uint32) == want.dir.view(np.uint32)).all(axis=1)
    assert same.mean() > 0.999
    # every photon is still there, whole: (wavelength, time, direction) rows as a multiset
    key_got = np.sort(np.ascontiguousarray(np.column_stack([got.wavelengths, got.dir, got.pol]).astype(np.float32)).view('V28').ravel())
    key_in = np.sort(np.ascontiguousarray(np.column_stack([ph.wavelengths, ph.dir, ph.pol]).astype(np.float32)).view('V28').ravel())
    assert np.array_equal(key_got, key_in)
