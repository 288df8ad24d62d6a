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


# ---- the two claims the tie-break rests on, against the compiled reference -------------------------------
def _ref_cast(ref, geometry, o, d, last_hits=None):
    mesh, bvh = geometry.mesh, geometry.bvh
    v = np.ascontiguousarray(mesh.vertices, np.float32)
    t = np.ascontiguousarray(mesh.triangles, np.uint32)
    nodes = np.ascontiguousarray(bvh.nodes.view(np.uint32).reshape(-1, 4))
    origin = (ctypes.c_float * 3)(*[float(x) for x in bvh.world_coords.world_origin])
    n = len(o)
    rdist = np.full(n, np.nan, np.float32)
    rtri = np.full(n, -2, np.int32)
    p = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
    o = np.ascontiguousarray(o, np.float32)
    d = np.ascontiguousarray(d, np.float32)
    lh = None if last_hits is None else np.ascontiguousarray(last_hits, np.int32)
    rc = ref.ref_mesh_run(p(v), len(v), p(t), len(t), p(nodes), len(nodes), origin, ctypes.c_float(float(bvh.world_coords.world_scale)),
                          n, p(o), p(d), p(lh), p(rdist), p(rtri), 0)
    assert rc == 0
    return rdist, rtri


def _engine_cast(gpu, gg, o, d, last_hits=None):
    from chroma_amd import _lib
    from chroma_amd.gpu.tools import to_gpu, GPUArray
    ctx = gpu.get_context()
    n = len(o)
    dist = GPUArray(n, np.float32, ctx).fill(np.float32(np.nan))
    tri = GPUArray(n, np.int32, ctx)
    d_o = to_gpu(np.ascontiguousarray(o, np.float32).reshape(-1), ctx)
    d_d = to_gpu(np.ascontiguousarray(d, np.float32).reshape(-1), ctx)
    d_l = None if last_hits is None else to_gpu(np.ascontiguousarray(last_hits, np.int32), ctx)
    _lib.check(ctx._lib.chroma_intersect_mesh(ctx.handle, gg.handle, n, d_o.ptr, d_d.ptr, None if d_l is None else d_l.ptr,
                                              dist.ptr, tri.ptr))
    return dist.get(), tri.get()


def _same(a, b):
    return np.array_equal(a.view(np.uint32), b.view(np.uint32))


@pytest.mark.skipif(not os.path.exists(REF_LIB), reason='oracle/_ref not built (needs the reference tree at build time)')
@pytest.mark.parametrize('which', ['tiny', 'lite'])
def test_ties_and_last_hit_against_the_compiled_reference(oracle_mod, tiny_geometry, tiny_packed, which):
    """(a) Rays aimed exactly at vertices, edge midpoints and centroids -- several triangles at one
    distance, where the ORDER of the reference's triangle tests decides -- and (b) second-step rays that
    start ON a triangle with last_hit_triangle = that triangle (mesh.h:82-101): the engine's 4-lane walk
    with its (distance, rank) tie-break and record-index exclusion, the oracle, and the reference's own
    intersect_mesh compiled for gfx950 give the same triangle ids and the same distance bits."""
    from chroma_amd import gpu, demo
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.geometry import pack_geometry
    from test_gpu_parity import _aimed_photons
    ref = ctypes.CDLL(REF_LIB)
    if which == 'tiny':
        geometry, packed = tiny_geometry, tiny_packed
    else:
        geometry = create_geometry_from_obj(demo.detector_lite())
        packed = pack_geometry(geometry)
    ctx = gpu.create_cuda_context(0)
    try:
        gg = gpu.GPUDetector(geometry, packed=packed)
        # (a) ties
        ph = _aimed_photons(geometry, (0.0, 0.0, 0.0), 40000)
        o, d = ph.pos.astype(np.float32), ph.dir.astype(np.float32)
        rdist, rtri = _ref_cast(ref, geometry, o, d)
        odist, otri, _ = oracle_mod.distance_to_mesh(packed, o, d)
        gdist, gtri = _engine_cast(gpu, gg, o, d)
        assert (rtri >= 0).mean() > 0.9
        assert np.array_equal(otri, rtri) and _same(odist, rdist), 'oracle vs reference on aimed rays'
        assert np.array_equal(gtri, rtri) and _same(gdist, rdist), 'engine vs reference on aimed rays'
        # the case is not vacuous: with the last hit excluded, a good share of the aimed rays find ANOTHER
        # triangle at exactly the same distance (the tie the test order broke)
        hit = rtri >= 0
        r2dist, r2tri = _ref_cast(ref, geometry, o[hit], d[hit], last_hits=rtri[hit])
        nties = int(np.count_nonzero((r2tri >= 0) & (r2dist.view(np.uint32) == rdist[hit].view(np.uint32))))
        assert nties > 100, 'only %d exact ties among the aimed rays' % nties
        o2dist, o2tri, _ = oracle_mod.distance_to_mesh(packed, o[hit], d[hit], last_hits=rtri[hit])
        g2dist, g2tri = _engine_cast(gpu, gg, o[hit], d[hit], last_hits=rtri[hit])
        assert np.array_equal(o2tri, r2tri) and _same(o2dist, r2dist), 'oracle vs reference, winner excluded'
        assert np.array_equal(g2tri, r2tri) and _same(g2dist, r2dist), 'engine vs reference, winner excluded'
        assert not np.array_equal(r2tri, rtri[hit])
        # (b) second-step rays: from the hit point (float arithmetic of the step, photon.h:303) into random
        # directions, last hit = the triangle they sit on
        rng = np.random.default_rng(11)
        p1 = (o[hit] + d[hit] * rdist[hit][:, None]).astype(np.float32)
        d1 = rng.normal(size=p1.shape).astype(np.float32)
        lh = rtri[hit]
        for last in (lh, None):            # with the exclusion, and without it (then many rays re-hit their own triangle)
            rd, rt = _ref_cast(ref, geometry, p1, d1, last_hits=last)
            od, ot, _ = oracle_mod.distance_to_mesh(packed, p1, d1, last_hits=last)
            gd, gt = _engine_cast(gpu, gg, p1, d1, last_hits=last)
            assert np.array_equal(ot, rt) and _same(od, rd), 'oracle vs reference, second step'
            assert np.array_equal(gt, rt) and _same(gd, rd), 'engine vs reference, second step'
            if last is not None:
                assert not (rt == lh).any()
                with_exclusion = rt
        assert (rt == lh).sum() > 10 and not np.array_equal(rt, with_exclusion)      # the exclusion matters
    finally:
        ctx.pop()
