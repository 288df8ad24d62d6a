"""BVH data model (port of the reference's test/test_bvh.py) and the two builders."""
import numpy as np
import pytest

from chroma_amd.bvh import (BVH, BVHLayerSlice, WorldCoords, OutOfRangeError, uint4, unpack_nodes,
                            make_recursive_grid_bvh, CHILD_BITS)
from chroma_amd import make as M


class TestWorldCoords:
    coords = WorldCoords([-1, -1, -1], 0.1)

    def test_fixed_to_world(self):
        np.testing.assert_array_max_ulp(self.coords.fixed_to_world([0, 1, 100]), [-1.0, -0.9, 9.0], dtype=np.float32)

    def test_world_to_fixed(self):
        np.testing.assert_array_equal(self.coords.world_to_fixed([-1.0, -0.9, 9.0]), [0, 1, 100])

    def test_arrays(self):
        f = [[0, 1, 100], [20, 40, 60], [210, 310, 410]]
        w = [[-1.0, -0.9, 9.0], [1.0, 3.0, 5.0], [20.0, 30.0, 40.0]]
        np.testing.assert_array_max_ulp(self.coords.fixed_to_world(f), w, dtype=np.float32)
        np.testing.assert_array_equal(self.coords.world_to_fixed(w), f)

    def test_out_of_range(self):
        with pytest.raises(OutOfRangeError):
            self.coords.world_to_fixed([-2.0, 0.0, 0.0])
        with pytest.raises(OutOfRangeError):
            self.coords.world_to_fixed([0.0, 1e9, 0.0])


def hand_built_bvh():
    """3-layer binary tree as in test/test_bvh.py:create_bvh, in the current leaf convention
    (nchild == 0 marks a leaf)."""
    wc = WorldCoords(np.array([-1.0, -1.0, -1.0]), 0.1)
    bounds = [0, 1, 3, 7]
    nodes = np.empty(bounds[-1], dtype=uint4)
    for i, (lo, hi) in enumerate([(0, 1), (1, 2), (2, 3), (3, 4)]):
        n = nodes[3 + i]
        nodes['x'][3 + i] = lo | (hi << 16); nodes['y'][3 + i] = lo | (hi << 16); nodes['z'][3 + i] = lo | (hi << 16)
        nodes['w'][3 + i] = i
    for p, (c0, lo, hi) in zip((1, 2), ((3, 0, 2), (5, 2, 4))):
        nodes['x'][p] = nodes['y'][p] = nodes['z'][p] = lo | (hi << 16)
        nodes['w'][p] = (2 << CHILD_BITS) | c0
    nodes['x'][0] = nodes['y'][0] = nodes['z'][0] = 0 | (4 << 16)
    nodes['w'][0] = (2 << CHILD_BITS) | 1
    return BVH(wc, nodes, bounds[:-1])


def test_bvh_container():
    bvh = hand_built_bvh()
    assert len(bvh) == 7 and bvh.layer_count() == 3
    assert [len(bvh.get_layer(i)) for i in range(3)] == [1, 2, 4]
    u = unpack_nodes(bvh.nodes)
    assert u['nchild'].tolist() == [2, 2, 2, 0, 0, 0, 0] and u['child'][:3].tolist() == [1, 3, 5]
    assert u['xhi'][0] == 4 and u['xlo'][0] == 0
    layer = bvh.get_layer(2)
    assert isinstance(layer, BVHLayerSlice)
    assert layer.area_fixed() == 4 * 6.0 and layer.area() == pytest.approx(4 * 6.0 * 0.01, rel=1e-6)
    assert bvh.get_layer(0).area_fixed() == 6 * 16.0


def check_tree_invariants(bvh, ntriangles):
    nodes = bvh.nodes
    u = unpack_nodes(nodes)
    n = len(nodes)
    assert bvh.layer_offsets[0] == 0 and u['nchild'].max() <= 15
    inner = np.flatnonzero(u['nchild'] > 0)
    first = u['child'][inner].astype(np.int64)
    cnt = u['nchild'][inner].astype(np.int64)
    assert (first > inner).all() and (first + cnt <= n).all()
    # every child box lies inside its parent's box
    parent_of = np.repeat(inner, cnt)
    child = np.concatenate([np.arange(f, f + c) for f, c in zip(first, cnt)]) if len(inner) else np.zeros(0, int)
    for ax in 'xyz':
        assert (u[ax + 'lo'][child] >= u[ax + 'lo'][parent_of]).all()
        assert (u[ax + 'hi'][child] <= u[ax + 'hi'][parent_of]).all()
    # every triangle sits in exactly one REACHABLE leaf
    reach = np.zeros(n, dtype=bool)
    frontier = np.array([0])
    seen_tri = []
    while len(frontier):
        reach[frontier] = True
        leaves = frontier[u['nchild'][frontier] == 0]
        seen_tri.append(u['child'][leaves])
        inn = frontier[u['nchild'][frontier] > 0]
        frontier = np.concatenate([np.arange(f, f + c) for f, c in zip(u['child'][inn].astype(np.int64), u['nchild'][inn].astype(np.int64))]) if len(inn) else np.zeros(0, int)
    tri = np.sort(np.concatenate(seen_tri))
    assert np.array_equal(tri, np.arange(ntriangles))


@pytest.mark.parametrize('mesh_fn', [lambda: M.box(100, 100, 100), lambda: M.sphere(1000.0, 24), lambda: M.torus(5, 20, 12, 8)])
def test_builders_agree_and_tree_is_valid(mesh_fn):
    mesh = mesh_fn()
    a = make_recursive_grid_bvh(mesh, backend='numpy')
    b = make_recursive_grid_bvh(mesh, backend='native')
    assert np.array_equal(a.nodes, b.nodes) and a.layer_offsets == b.layer_offsets
    assert a.world_coords.world_scale == b.world_coords.world_scale
    check_tree_invariants(b, len(mesh.triangles))


def test_tiny_detector_bvh(tiny_geometry):
    bvh = tiny_geometry.bvh
    mesh = tiny_geometry.mesh
    assert len(mesh.triangles) == 389568
    assert bvh.layer_count() == 11 and len(bvh.get_layer(10)) == 389568
    assert bvh.world_coords.world_scale == pytest.approx(0.0762963, rel=1e-5)
    ref = make_recursive_grid_bvh(mesh, backend='numpy')
    assert np.array_equal(ref.nodes, bvh.nodes)
    check_tree_invariants(bvh, len(mesh.triangles))
    # leaves contain their triangles (padded by one quantum)
    u = unpack_nodes(bvh.get_layer(10).nodes)
    tri = mesh.vertices[mesh.triangles[u['child']]]
    lo = bvh.world_coords.fixed_to_world(np.column_stack([u['xlo'], u['ylo'], u['zlo']]))
    hi = bvh.world_coords.fixed_to_world(np.column_stack([u['xhi'], u['yhi'], u['zhi']]))
    assert (tri.min(axis=1) >= lo - 1e-3).all() and (tri.max(axis=1) <= hi + 1e-3).all()


def test_single_triangle_and_degenerate_inputs():
    from chroma_amd.geometry import Mesh
    one = Mesh(np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=float), [[0, 1, 2]])
    for backend in ('numpy', 'native'):
        bvh = make_recursive_grid_bvh(one, backend=backend)
        assert len(bvh) == 1 and bvh.nodes['w'][0] == 0
    with pytest.raises(Exception):
        make_recursive_grid_bvh(Mesh(np.zeros((3, 3)), np.zeros((0, 3), dtype=int)), backend='native')


@pytest.mark.parametrize('case', ['tiny', 'box100', 'cube1000', 'sphere100_16'])
def test_bvh_golden_layers_and_node_hash(case):
    """SURVEY.md 8(c) golden 4: the full list of layer sizes, the world frame and the SHA-256 of the packed node
    array against the committed tests/golden/bvh_golden.json (tools/gen_bvh_golden.py), for BOTH builders."""
    import hashlib
    import json
    import os
    import sys
    from conftest import GOLDEN, ROOT
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    from gen_bvh_golden import CASES
    from chroma_amd.geometry import Geometry, Solid, Mesh
    from chroma_amd.loader import create_geometry_from_obj
    want = json.load(open(os.path.join(GOLDEN, 'bvh_golden.json')))[case]
    geo = create_geometry_from_obj(CASES[case]())
    for backend in ('native', 'numpy'):
        bvh = make_recursive_grid_bvh(geo.mesh, backend=backend)
        nodes = np.ascontiguousarray(bvh.nodes).view(np.uint32).reshape(-1, 4)
        lo = [int(x) for x in bvh.layer_offsets]
        assert [b - a for a, b in zip(lo, lo[1:] + [len(nodes)])] == want['layer_sizes'], backend
        assert len(geo.mesh.triangles) == want['ntriangles'] and len(nodes) == want['nnodes']
        assert float(np.float32(bvh.world_coords.world_scale)) == want['world_scale']
        assert [float(np.float32(x)) for x in bvh.world_coords.world_origin] == want['world_origin']
        assert hashlib.sha256(nodes.tobytes()).hexdigest() == want['nodes_sha256'], backend
