"""The derived 8-wide traversal tree (chroma_amd/csrc/wide_build.cpp), in both topologies -- the
collapsed reference tree and the SAH rebuild over the reference's leaf boxes: structure invariants
and the reference test-order ranks, checked against the oracle's replay of the reference loop
(chroma/cuda/mesh.h:58-110).  Host code only -- no GPU."""
import os

import numpy as np
import pytest

import oracle
from chroma_amd import _lib, demo
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.geometry import Solid, Geometry, vacuum
from chroma_amd import make

LEAF = 0x80000000
EMPTY = 0xFFFFFFFF


def _geometries():
    cube = Geometry()
    cube.add_solid(Solid(make.cube(100.0), vacuum, vacuum))
    yield 'cube', create_geometry_from_obj(cube)
    yield 'tiny', create_geometry_from_obj(demo.tiny())


def _wide(nodes, ntriangles, topology):
    old = os.environ.get('CHROMA_TREE')
    os.environ['CHROMA_TREE'] = topology
    try:
        return _lib.wide_build(nodes, ntriangles)
    finally:
        if old is None:
            del os.environ['CHROMA_TREE']
        else:
            os.environ['CHROMA_TREE'] = old


@pytest.fixture(scope='module', params=['cube-collapse', 'cube-sah', 'cube-greedy', 'cube-ploc', 'cube-levels', 'tiny-collapse', 'tiny-sah', 'tiny-greedy', 'tiny-ploc', 'tiny-levels'])
def built(request):
    want, topology = request.param.split('-')
    for name, g in _geometries():
        if name == want:
            nodes = np.ascontiguousarray(g.bvh.nodes)
            w = _wide(nodes, len(g.mesh.triangles), topology)
            w['topology'] = topology
            return g, nodes.view(np.uint32).reshape(-1, 4), w


def _lo_hi(words):
    w = words[..., :3]
    return (w & 0xFFFF).astype(np.int64), (w >> 16).astype(np.int64)


def test_every_triangle_is_one_record(built):
    g, ref, w = built
    nt = len(g.mesh.triangles)
    assert len(w['record_to_tri']) == nt
    assert np.array_equal(np.sort(w['record_to_tri']), np.arange(nt))
    assert np.array_equal(w['record_to_tri'][w['tri_to_record']], np.arange(nt))
    ent = w['wnodes'].reshape(-1, 4)
    leaf = ent[(ent[:, 3] & LEAF != 0) & (ent[:, 3] != EMPTY)]
    assert np.array_equal(np.sort(leaf[:, 3] & 0x7FFFFFFF), np.arange(nt))      # each record under exactly one entry
    if w['topology'] == 'collapse':
        # triangle children of a node are consecutive records
        for node in w['wnodes'][:2000]:
            recs = [int(e[3] & 0x7FFFFFFF) for e in node if e[3] != EMPTY and e[3] & LEAF]
            assert recs == list(range(recs[0], recs[0] + len(recs))) if recs else True


def test_leaf_boxes_are_the_reference_leaf_boxes(built, monkeypatch):
    """A leaf entry's box is the reference's leaf box (cuda/bvh.cu:149-203); with CHROMA_TIGHT_LEAVES=1 (an opt-in:
    wide_build.h) its lower bounds are moved up by the one quantum the reference pads them with -- and it still holds its
    triangle."""
    g, ref, w = built
    ent0 = w['wnodes'].reshape(-1, 4)
    leaf0 = ent0[(ent0[:, 3] & LEAF != 0) & (ent0[:, 3] != EMPTY)]
    by0 = np.zeros((len(g.mesh.triangles), 3), dtype=np.uint32)
    by0[w['record_to_tri'][leaf0[:, 3] & 0x7FFFFFFF]] = leaf0[:, :3]
    ref_leaf0 = ref[(ref[:, 3] >> 28) == 0]
    want0 = np.zeros_like(by0)
    want0[ref_leaf0[:, 3] & 0x0FFFFFFF] = ref_leaf0[:, :3]
    assert np.array_equal(by0, want0)                          # the default: the reference's boxes as they are
    monkeypatch.setenv('CHROMA_TIGHT_LEAVES', '1')
    w = dict(_wide(np.ascontiguousarray(g.bvh.nodes), len(g.mesh.triangles), w['topology']), topology=w['topology'])
    ref_leaf = ref[(ref[:, 3] >> 28) == 0]
    by_tri = np.zeros((len(g.mesh.triangles), 3), dtype=np.uint32)
    by_tri[ref_leaf[:, 3] & 0x0FFFFFFF] = ref_leaf[:, :3]
    ent = w['wnodes'].reshape(-1, 4)
    leaf = ent[(ent[:, 3] & LEAF != 0) & (ent[:, 3] != EMPTY)]
    tri = w['record_to_tri'][leaf[:, 3] & 0x7FFFFFFF]
    want = by_tri[tri]
    lo, hi = want & 0xFFFF, want >> 16
    tight = np.where((lo > 0) & (lo + 1 <= hi), want + 1, want)
    if w['topology'] == 'collapse':       # (re-uses the reference tree's own nodes where a range fits a wide node, rebuilds elsewhere)
        assert ((leaf[:, :3] == tight) | (leaf[:, :3] == want)).all()
    else:
        assert np.array_equal(leaf[:, :3], tight)
    # the triangle lies inside: every vertex coordinate between the dequantised bounds
    wc = g.bvh.world_coords
    v = g.mesh.vertices[g.mesh.triangles[tri]].astype(np.float64)                  # [leaf][3 vertices][xyz]
    blo = wc.world_origin + (leaf[:, :3] & 0xFFFF).astype(np.float64) * wc.world_scale
    bhi = wc.world_origin + (leaf[:, :3] >> 16).astype(np.float64) * wc.world_scale
    slack = 0.02 * wc.world_scale          # (float32 rounding of (v - origin) / scale: a few ulp of the world's extent)
    assert (v.min(axis=1) >= blo - slack).all() and (v.max(axis=1) <= bhi + slack).all()


def test_inner_entries_bound_their_node(built):
    g, ref, w = built
    wn = w['wnodes']
    nwide = len(wn)
    seen = np.zeros(nwide, dtype=bool)
    seen[0] = True
    for i in range(nwide):
        for e in wn[i]:
            if e[3] == EMPTY or e[3] & LEAF:
                continue
            c = int(e[3])
            assert i < c < nwide and not seen[c]        # a tree, stored parents first
            seen[c] = True
            lo, hi = _lo_hi(e[None, :])
            child = wn[c][wn[c][:, 3] != EMPTY]
            clo, chi = _lo_hi(child)
            assert (clo >= lo).all() and (chi <= hi).all()
        if i > 3000:
            break
    if nwide <= 3000:
        assert seen.all()
    # few slots stay empty
    fill = (wn[:, :, 3] != EMPTY).sum() / float(nwide)
    assert fill > (4.0 if nwide > 10 else 1.0)


def test_rank_is_the_reference_test_order(built):
    g, ref, w = built
    order = oracle.reference_test_order(ref)
    nt = len(g.mesh.triangles)
    assert len(order) == nt
    expect = np.empty(nt, dtype=np.uint32)
    expect[order] = np.arange(nt, dtype=np.uint32)
    assert np.array_equal(w['rank'], expect)


def test_unlayered_and_overwide_trees():
    # hand-made tree: root with 10 children (wider than a wide node), two of them inner, stored out of layer order
    def node(lo, hi, child, nchild):
        return [lo | hi << 16] * 3 + [nchild << 28 | child]
    nodes = [node(0, 100, 1, 10)]
    tri = 0
    for k in range(10):
        if k in (2, 7):
            nodes.append(None)
        else:
            nodes.append(node(10 * k, 10 * k + 5, tri, 0)); tri += 1
    # inner child 2 -> nodes 11..12 ; its first child is inner again -> 13..14 (not layer order: child 7's range comes last)
    nodes[3] = node(20, 29, 11, 2)
    nodes.append(node(20, 24, 13, 2)); nodes.append(node(25, 29, tri, 0)); tri += 1
    nodes.append(node(20, 21, tri, 0)); tri += 1
    nodes.append(node(22, 24, tri, 0)); tri += 1
    nodes[8] = node(70, 79, 15, 2)
    nodes.append(node(70, 72, tri, 0)); tri += 1
    nodes.append(node(75, 79, tri, 0)); tri += 1
    ref = np.array(nodes, dtype=np.uint32)
    for topology in ('sah', 'greedy', 'collapse', 'ploc', 'levels'):          # (ploc: not a tree that ends in its leaf layer -> the top-down builder)
        w = _wide(ref, tri, topology)
        ent = w['wnodes'].reshape(-1, 4)
        leaf = ent[(ent[:, 3] & LEAF != 0) & (ent[:, 3] != EMPTY)]
        assert np.array_equal(np.sort(w['record_to_tri'][leaf[:, 3] & 0x7FFFFFFF]), np.arange(tri))
    order = oracle.reference_test_order(ref)
    expect = np.empty(tri, dtype=np.uint32)
    expect[order] = np.arange(tri, dtype=np.uint32)
    assert np.array_equal(w['rank'], expect)
    ent = w['wnodes'].reshape(-1, 4)
    leaf = ent[(ent[:, 3] & LEAF != 0) & (ent[:, 3] != EMPTY)]
    assert np.array_equal(np.sort(w['record_to_tri'][leaf[:, 3] & 0x7FFFFFFF]), np.arange(tri))
    assert (w['wnodes'][0][:, 3] != EMPTY).sum() == 8          # 7 children + the tail of the over-wide range


def test_malformed_tree_is_rejected():
    bad = np.array([[0, 0, 0, 2 << 28 | 5]], dtype=np.uint32)          # children outside the array
    with pytest.raises(_lib.ChromaError):
        _lib.wide_build(bad, 1)


def test_build_is_repeatable():
    """Node order and record order do not depend on thread timing: two builds are identical."""
    for name, g in _geometries():
        if name != 'tiny':
            continue
        nodes = np.ascontiguousarray(g.bvh.nodes)
        for topology in ('sah', 'greedy', 'collapse', 'ploc', 'levels'):
            a = _wide(nodes, len(g.mesh.triangles), topology)
            b = _wide(nodes, len(g.mesh.triangles), topology)
            for key in ('wnodes', 'tri_to_record', 'record_to_tri', 'rank'):
                assert np.array_equal(a[key], b[key]), (topology, key)
            assert a['depth'] == b['depth']


def test_index_checks_reject_a_tampered_tree():
    """chroma_geometry_create uploads a derived tree only after chroma_wide_validate's checks: every
    inner child word names a later wide node, every leaf word a record inside the record table, every
    record a triangle of the mesh, every triangle a record that names it back.  (The device buffers
    are sized from these counts, so a tree that passes cannot make a kernel read past them -- the class
    of fault DESIGN.md section 4a describes.)"""
    for name, g in _geometries():
        if name != 'tiny':
            continue
        ntri = len(g.mesh.triangles)
        w = _wide(np.ascontiguousarray(g.bvh.nodes), ntri, 'sah')
        assert _lib.wide_validate(w, ntri)
        ent = w['wnodes'].reshape(-1, 4)
        inner = np.flatnonzero((ent[:, 3] & LEAF) == 0)
        leaf = np.flatnonzero(((ent[:, 3] & LEAF) != 0) & (ent[:, 3] != EMPTY))

        def tampered(**kw):
            t = {k: v.copy() for k, v in w.items() if isinstance(v, np.ndarray)}
            for k, f in kw.items():
                f(t[k])
            return t
        nwide, nrec = len(w['wnodes']), len(w['record_to_tri'])
        # a child word past the node array / pointing backwards (a cycle) / a record past the table
        assert not _lib.wide_validate(tampered(wnodes=lambda a: a.reshape(-1, 4).__setitem__((inner[5], 3), nwide)), ntri)
        assert not _lib.wide_validate(tampered(wnodes=lambda a: a.reshape(-1, 4).__setitem__((inner[-1], 3), 0)), ntri)
        assert not _lib.wide_validate(tampered(wnodes=lambda a: a.reshape(-1, 4).__setitem__((leaf[7], 3), LEAF | nrec)), ntri)
        # a record naming a triangle outside the mesh; a triangle whose record names another one
        assert not _lib.wide_validate(tampered(record_to_tri=lambda a: a.__setitem__(3, ntri)), ntri)
        assert not _lib.wide_validate(tampered(tri_to_record=lambda a: a.__setitem__(0, a[1])), ntri)
        # an EMPTY child word over a real box (the default walk tells an empty entry by its inverted box alone and
        # would take the word for leaf record 0x7FFFFFFF); a non-empty entry whose box is inverted
        empty = np.flatnonzero(ent[:, 3] == EMPTY)
        assert len(empty) and (ent[empty, :3] == 0x0000FFFF).all()
        assert not _lib.wide_validate(tampered(wnodes=lambda a: a.reshape(-1, 4).__setitem__((empty[0], 0), 0x00100000)), ntri)
        assert not _lib.wide_validate(tampered(wnodes=lambda a: a.reshape(-1, 4).__setitem__((leaf[3], 3), EMPTY)), ntri)
        assert not _lib.wide_validate(tampered(wnodes=lambda a: a.reshape(-1, 4).__setitem__((leaf[3], 1), 0x0000FFFF)), ntri)
        # fewer records than triangles (the record table would be shorter than the ids the kernels read)
        short = tampered()
        short['record_to_tri'] = short['record_to_tri'][:-1]
        assert not _lib.wide_validate(short, ntri)


def test_ploc_tree_is_independent_of_the_thread_count_and_close_to_the_sah_tree():
    """The bottom-up topology (CHROMA_TREE=ploc: data-parallel passes only, the shape a device builder would take) is a pure function of the
    reference tree's leaf layer -- same nodes with 1 thread and with all of them -- and its surface-area sum stays within
    10 % of the top-down SAH tree's (measured: +6 % on tiny, +5 % on C2-lite; the collapsed reference tree: +38 %)."""
    for name, g in _geometries():
        if name != 'tiny':
            continue
        nodes = np.ascontiguousarray(g.bvh.nodes)
        nt = len(g.mesh.triangles)
        a = _wide(nodes, nt, 'ploc')
        old = os.environ.get('CHROMA_HOST_THREADS')
        os.environ['CHROMA_HOST_THREADS'] = '1'
        try:
            b = _wide(nodes, nt, 'ploc')
        finally:
            if old is None:
                del os.environ['CHROMA_HOST_THREADS']
            else:
                os.environ['CHROMA_HOST_THREADS'] = old
        for key in ('wnodes', 'tri_to_record', 'record_to_tri', 'rank'):
            assert np.array_equal(a[key], b[key]), key
        assert _lib.wide_validate(a, nt)
        sah, collapse = _wide(nodes, nt, 'sah'), _wide(nodes, nt, 'collapse')
        assert _node_area_sum(a['wnodes']) < 1.10 * _node_area_sum(sah['wnodes'])
        assert _node_area_sum(a['wnodes']) < 0.85 * _node_area_sum(collapse['wnodes'])
        # breadth-first numbering: the children of a node follow all nodes of its level
        ent = a['wnodes'].reshape(-1, 4)
        inner = ent[((ent[:, 3] & LEAF) == 0) & (ent[:, 3] != EMPTY), 3]
        assert np.array_equal(inner, np.arange(1, len(inner) + 1))


def _node_area_sum(wnodes):
    """Sum over the inner entries of a wide tree of their box area: the surface-area estimate of node visits per ray."""
    ent = wnodes.reshape(-1, 4)
    inner = ent[(ent[:, 3] & LEAF == 0) & (ent[:, 3] != EMPTY)]
    lo, hi = _lo_hi(inner)
    d = (hi - lo).astype(np.float64)
    return float((d[:, 0] * d[:, 1] + d[:, 1] * d[:, 2] + d[:, 2] * d[:, 0]).sum())


def test_least_area_collapse_beats_the_greedy_one():
    """The default topology cuts a binary SAH tree into wide nodes by dynamic programming (least total node area);
    round 1's greedy rule is kept as CHROMA_TREE=greedy.  Same triangles, same reference ranks; fewer nodes, fuller
    nodes, and a smaller area sum."""
    for name, g in _geometries():
        if name != 'tiny':
            continue
        nodes = np.ascontiguousarray(g.bvh.nodes)
        nt = len(g.mesh.triangles)
        dp, greedy = _wide(nodes, nt, 'sah'), _wide(nodes, nt, 'greedy')
        assert np.array_equal(dp['rank'], greedy['rank'])
        assert len(dp['wnodes']) < len(greedy['wnodes'])
        fill = lambda w: float((w['wnodes'].reshape(-1, 4)[:, 3] != EMPTY).mean())
        assert fill(dp) > fill(greedy) and fill(dp) > 0.75
        assert _node_area_sum(dp['wnodes']) < _node_area_sum(greedy['wnodes'])
        for w in (dp, greedy):
            assert _lib.wide_validate(w, nt)


def test_levels_topology_does_not_depend_on_the_thread_count():
    """The default topology leaves nothing to the schedule: one thread, three threads and all of them give the same wide
    tree, record maps and ranks (the round-2 topology, 'sah', made its top part from sets whose size followed the thread
    count) -- which is what lets the device builder reproduce it bit for bit (tests/test_gpu_wide.py)."""
    g = create_geometry_from_obj(demo.tiny())
    nodes = np.ascontiguousarray(g.bvh.nodes)
    nt = len(g.mesh.triangles)
    old = os.environ.get('CHROMA_HOST_THREADS')
    trees = []
    try:
        for threads in ('1', '3', None):
            if threads is None:
                os.environ.pop('CHROMA_HOST_THREADS', None)
            else:
                os.environ['CHROMA_HOST_THREADS'] = threads
            trees.append(_wide(nodes, nt, 'levels'))
    finally:
        if old is None:
            os.environ.pop('CHROMA_HOST_THREADS', None)
        else:
            os.environ['CHROMA_HOST_THREADS'] = old
    for other in trees[1:]:
        for key in ('wnodes', 'tri_to_record', 'record_to_tri', 'rank'):
            assert np.array_equal(trees[0][key], other[key]), key
    assert _lib.wide_validate(trees[0], nt)
