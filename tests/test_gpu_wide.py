"""The derived 8-wide traversal tree built ON THE DEVICE (chroma_wide_build_device, csrc/wide_device.hip: the binned /
swept SAH splits one level of the tree at a time, the least-area collapse bottom-up, the wide nodes breadth first).
It must equal the host twin -- CHROMA_TREE=levels, csrc/wide_build.cpp -- bit for bit: wide nodes, record maps, ranks."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

KEYS = ('wnodes', 'tri_to_record', 'record_to_tri', 'rank')


@pytest.fixture(scope='module')
def ctx():
    from chroma_amd import gpu as g
    c = g.create_cuda_context(0)
    yield c
    c.pop()


def _host_levels(nodes, ntriangles):
    from chroma_amd import _lib
    old = os.environ.get('CHROMA_TREE')
    os.environ['CHROMA_TREE'] = 'levels'
    try:
        return _lib.wide_build(nodes, ntriangles)
    finally:
        if old is None:
            del os.environ['CHROMA_TREE']
        else:
            os.environ['CHROMA_TREE'] = old


def _same(dev, host, what):
    assert dev['depth'] == host['depth'], what
    for key in KEYS:
        assert dev[key].shape == host[key].shape, (what, key)
        if not np.array_equal(dev[key], host[key]):
            bad = np.flatnonzero((dev[key] != host[key]).reshape(len(dev[key]), -1).any(axis=1))
            raise AssertionError('%s: %s differs in %d rows, first %d: device %s host %s' % (
                what, key, len(bad), bad[0], dev[key][bad[0]].tolist(), host[key][bad[0]].tolist()))


def _meshes():
    from chroma_amd import demo, make
    from chroma_amd.geometry import Geometry, Solid, Mesh, vacuum
    from chroma_amd.loader import create_geometry_from_obj
    for name, solid in (('cube', make.cube(100.0)), ('sphere', make.sphere(50.0, 40))):
        g = Geometry()
        g.add_solid(Solid(solid, vacuum, vacuum))
        yield name, create_geometry_from_obj(g)
    yield 'tiny', create_geometry_from_obj(demo.tiny())
    yield 'lite', create_geometry_from_obj(demo.detector_lite())
    # a triangle soup with heavy overlap, exact duplicates (coincident centroids: sets that can only be halved where they
    # stand -- 5000 copies of one triangle exceed a wave's set size, so the chunked kernels take that path too) and
    # slivers along the axes (centroids equal on two axes)
    rng = np.random.default_rng(11)
    n = 60000
    centre = rng.uniform(-100, 100, size=(n, 1, 3)).astype(np.float32)
    tri = centre + rng.normal(0, 8, size=(n, 3, 3)).astype(np.float32)
    tri[:5000] = tri[0]
    tri[5000:5040] = tri[5000]
    tri[6000:9000, :, 1] = 3.0
    tri[6000:9000, :, 2] = -7.0
    vertices = tri.reshape(-1, 3)
    g = Geometry()
    g.add_solid(Solid(Mesh(vertices, np.arange(3 * n, dtype=np.int32).reshape(-1, 3), remove_null_triangles=False), vacuum, vacuum))
    yield 'soup', create_geometry_from_obj(g)


def test_device_tree_equals_the_host_twin(ctx):
    from chroma_amd import _lib
    for name, g in _meshes():
        nodes = np.ascontiguousarray(g.bvh.nodes)
        nt = len(g.mesh.triangles)
        dev = _lib.wide_build(nodes, nt, ctx=ctx)
        host = _host_levels(nodes, nt)
        _same(dev, host, name)
        assert _lib.wide_validate(dev, nt), name
        again = _lib.wide_build(nodes, nt, ctx=ctx)               # (atomics hand out list places: the result must not care)
        _same(again, dev, name + ' (second run)')


def test_device_tree_of_one_and_two_triangles(ctx):
    from chroma_amd import _lib
    from chroma_amd.geometry import Geometry, Solid, Mesh, vacuum
    from chroma_amd.loader import create_geometry_from_obj
    v = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [5, 5, 5], [6, 5, 5], [5, 6, 5]], dtype=np.float32)
    for ntri in (1, 2):
        g = Geometry()
        g.add_solid(Solid(Mesh(v[:3 * ntri], np.arange(3 * ntri, dtype=np.int32).reshape(-1, 3)), vacuum, vacuum))
        g = create_geometry_from_obj(g)
        nodes = np.ascontiguousarray(g.bvh.nodes)
        _same(_lib.wide_build(nodes, ntri, ctx=ctx), _host_levels(nodes, ntri), '%d triangles' % ntri)


def test_propagation_over_the_device_tree_matches_the_oracle(ctx, oracle_mod):
    """A geometry whose wide tree came from the device builder gives the oracle's photons (the walk's answer does not
    depend on the hierarchy above the reference's leaf boxes)."""
    from chroma_amd import demo, gpu, _lib
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.geometry import pack_geometry
    g = create_geometry_from_obj(demo.tiny())
    packed = pack_geometry(g)
    packed.attach_wide_tree(_lib.wide_build(packed.arrays['nodes'], packed.desc.ntriangles, ctx=ctx))
    gg = gpu.GPUDetector.from_packed(packed)
    ph = oracle_mod.generate_bomb(200000, seed=5)
    gp = gpu.GPUPhotons(ph)
    gp.propagate(gg, gpu.get_rng_states(64 * 1024, seed=77), max_steps=100)
    got = gp.get()
    want, counters, _ = oracle_mod.propagate(packed, ph, seed=77, max_steps=100, nthreads=8)
    for field in ('pos', 'dir', 'pol', 'wavelengths', 't', 'flags', 'last_hit_triangles'):
        assert np.array_equal(getattr(got, field).view(np.uint32), getattr(want, field).view(np.uint32)), field
    assert np.array_equal(gp.rng_counters.get(), counters)


def test_stack_need_by_passes_on_the_device_is_the_backward_sweep(ctx):
    """chroma_geometry_create works out the reference walk's stack need by passes over the uploaded node array until
    nothing changes; the value is the one backward sweep of the definition gives (mesh.h:68-110: at a node every inner
    child is pushed, the last one is walked first)."""
    from chroma_amd import demo, gpu, make
    from chroma_amd.loader import create_geometry_from_obj

    def sphere():
        return make.sphere(50.0, 60)
    for build in (demo.tiny, sphere):
        g = create_geometry_from_obj(build())
        w = np.ascontiguousarray(g.bvh.nodes).view(np.uint32).reshape(-1, 4)[:, 3]
        nchild, first = (w >> 28).astype(np.int64), (w & 0x0FFFFFFF).astype(np.int64)
        need = np.zeros(len(w), dtype=np.int64)
        for i in range(len(w) - 1, -1, -1):
            if nchild[i] == 0:
                continue
            rank = best = 0
            for c in range(first[i], first[i] + nchild[i]):
                if nchild[c]:
                    best = max(best, rank + need[c])
                    rank += 1
            need[i] = max(best, rank)
        gg = gpu.GPUGeometry(g)
        assert gg.stack_need() == max(1, int(need[0])), build.__name__


def test_device_tree_on_random_soups_around_the_kernels_size_limits(ctx):
    """Triangle soups whose sizes sit on the borders between the builder's kernels (one thread per set up to 8 triangles, one
    wave up to 32 and up to 2048, chunks of 4096 beyond), with duplicated and degenerate triangles mixed in."""
    from chroma_amd import _lib
    from chroma_amd.geometry import Geometry, Solid, Mesh, vacuum
    from chroma_amd.loader import create_geometry_from_obj
    for seed, n in enumerate((2, 3, 8, 9, 32, 33, 64, 2048, 2049, 4096, 4097, 8193, 30000)):
        rng = np.random.default_rng(100 + seed)
        centre = rng.uniform(-50, 50, size=(n, 1, 3)).astype(np.float32)
        tri = centre + rng.normal(0, 6, size=(n, 3, 3)).astype(np.float32)
        if n >= 64:
            tri[n // 2:n // 2 + n // 8] = tri[n // 2]                 # a block of identical triangles
            tri[-n // 16:, :, 0] = 1.0                                  # slivers in a plane
        g = Geometry()
        g.add_solid(Solid(Mesh(tri.reshape(-1, 3), np.arange(3 * n, dtype=np.int32).reshape(-1, 3), remove_null_triangles=False), vacuum, vacuum))
        g = create_geometry_from_obj(g)
        nodes = np.ascontiguousarray(g.bvh.nodes)
        nt = len(g.mesh.triangles)
        _same(_lib.wide_build(nodes, nt, ctx=ctx), _host_levels(nodes, nt), '%d triangles' % n)
