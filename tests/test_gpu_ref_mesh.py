"""Ray cast of the engine against the REFERENCE'S OWN intersect_mesh / distance_to_mesh
(chroma/cuda/mesh.h), compiled for gfx950 from the reference sources by oracle/Makefile into
oracle/_ref/libchroma_ref_mesh.so.  This pins rows a-3/a-4 of SURVEY.md section 8 on the real
reference code rather than on a restatement."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
REF_LIB = os.path.join(ROOT, 'oracle', '_ref', 'libchroma_ref_mesh.so')


@pytest.mark.skipif(not os.path.exists(REF_LIB), reason='oracle/_ref not built (needs the reference tree at build time)')
def test_engine_and_oracle_match_reference_traversal(oracle_mod, tiny_geometry, tiny_packed):
    from chroma_amd import gpu, _lib
    from chroma_amd.gpu.tools import to_gpu, GPUArray
    ref = ctypes.CDLL(REF_LIB)
    rng = np.random.default_rng(8)
    n = 40000
    o = np.zeros((n, 3), dtype=np.float32)
    o[n // 2:] = rng.uniform(-1200, 1200, (n - n // 2, 3))
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:6] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]
    mesh, bvh = tiny_geometry.mesh, tiny_geometry.bvh
    v = np.ascontiguousarray(mesh.vertices, np.float32)
    t = np.ascontiguousarray(mesh.triangles, np.uint32)
    nodes = np.ascontiguousarray(bvh.nodes.view(np.uint32).reshape(-1, 4))
    origin = (ctypes.c_float * 3)(*[float(x) for x in bvh.world_coords.world_origin])
    rdist = np.full(n, np.nan, np.float32)
    rtri = np.full(n, -2, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = ref.ref_mesh_run(p(v), len(v), p(t), len(t), p(nodes), len(nodes), origin, ctypes.c_float(float(bvh.world_coords.world_scale)),
                          n, p(o), p(d), None, p(rdist), p(rtri), 0)
    assert rc == 0
    # the reference's own distance_to_mesh kernel gives the same distances as its intersect_mesh
    kdist = np.full(n, np.nan, np.float32)
    assert ref.ref_mesh_run(p(v), len(v), p(t), len(t), p(nodes), len(nodes), origin, ctypes.c_float(float(bvh.world_coords.world_scale)),
                            n, p(o), p(d), None, p(kdist), None, 1) == 0
    assert np.array_equal(kdist.view(np.uint32), rdist.view(np.uint32))

    # CPU oracle == reference
    odist, otri, _ = oracle_mod.distance_to_mesh(tiny_packed, o, d)
    assert np.array_equal(otri, rtri)
    assert np.array_equal(odist.view(np.uint32), rdist.view(np.uint32))

    # HIP engine == reference
    ctx = gpu.create_cuda_context(0)
    gg = gpu.GPUDetector(tiny_geometry)
    dist = GPUArray(n, np.float32, ctx).fill(np.float32(np.nan))
    tri = GPUArray(n, np.int32, ctx)
    d_o, d_d = to_gpu(o.reshape(-1), ctx), to_gpu(d.reshape(-1), ctx)      # keep the device arrays alive
    _lib.check(ctx._lib.chroma_distance_to_mesh(ctx.handle, gg.handle, n, d_o.ptr, d_d.ptr, dist.ptr, tri.ptr))
    assert np.array_equal(tri.get(), rtri)
    assert np.array_equal(dist.get().view(np.uint32), rdist.view(np.uint32))
    assert (rtri >= 0).mean() > 0.9
