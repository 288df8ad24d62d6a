"""Recursive-grid BVH builder.

Algorithm of the reference (chroma/bvh/grid.py:11-95 driving the kernels make_leaves,
make_parents_detailed, copy_and_offset, collapse_child of chroma/cuda/bvh.cu): one leaf
per triangle (quantised box padded by one unit, 48-bit Morton code of the quantised
centroid), leaves sorted by Morton code, then layer by layer: shift all codes right
until the mean group size reaches ``target_degree``, make one parent per run of equal
codes, split runs longer than 15, union the child boxes.  Finally layers are
concatenated root first and single-child chains are collapsed.

The reference needs a CUDA context for this.  Here there are three builders that must
return identical node arrays: HIP kernels on the device (``backend='device'``,
csrc/bvh_device.hip: the default when a GPU is there, as in the reference), the same
algorithm on the host cores (``backend='native'``, multi-threaded C++ in libchroma_hip.so)
and a NumPy restatement (``backend='numpy'``) kept as an independent cross-check
(tests/test_bvh.py, tests/test_gpu_bvh.py).
"""
import numpy as np

from chroma_amd.bvh.bvh import BVH, WorldCoords, uint4, CHILD_BITS, MAX_CHILD


def _bounds(vertices):
    """(min, max) over the rows of an (n, 3) array.  NumPy's axis-0 reduction of a narrow array runs an inner loop of
    length 3 (3.7 s for 85 M vertices); column by column in slices on a few threads it takes 0.1 s, same values."""
    n = len(vertices)
    if n < (1 << 20):
        return vertices.min(axis=0), vertices.max(axis=0)
    import os
    from concurrent.futures import ThreadPoolExecutor
    nthreads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)))
    edges = np.linspace(0, n, 4 * nthreads + 1).astype(np.int64)

    def part(k):
        c = vertices[edges[k]:edges[k + 1]]
        return [c[:, j].min() for j in range(3)], [c[:, j].max() for j in range(3)]
    with ThreadPoolExecutor(nthreads) as pool:
        parts = list(pool.map(part, range(len(edges) - 1)))
    lo = np.min(np.array([p[0] for p in parts], dtype=vertices.dtype), axis=0)
    hi = np.max(np.array([p[1] for p in parts], dtype=vertices.dtype), axis=0)
    return lo, hi


def world_coords_for(vertices):
    """Fixed-point frame of a mesh (chroma/gpu/bvh.py:43-48)."""
    world_origin, upper = _bounds(vertices)
    world_scale = np.max(upper - world_origin) / np.float32(2 ** 16 - 2)
    return WorldCoords(world_origin=world_origin, world_scale=world_scale)


def _spread3_16(v):
    """Bits of a 16-bit value moved to every third position (cuda/bvh.cu:42-52)."""
    x = v.astype(np.uint64)
    x = (x | (x << np.uint64(16))) & np.uint64(0x00000000FF0000FF)
    x = (x | (x << np.uint64(8))) & np.uint64(0x000000F00F00F00F)
    x = (x | (x << np.uint64(4))) & np.uint64(0x00000C30C30C30C3)
    x = (x | (x << np.uint64(2))) & np.uint64(0x0000249249249249)
    return x


def _quantize(v, origin, scale):
    """Truncating float32 quantisation (cuda/bvh.cu:65-76)."""
    return ((v - origin) / scale).astype(np.uint32)


def make_leaves_numpy(vertices, triangles, world_coords):
    """make_leaves (cuda/bvh.cu:149-203) in NumPy: leaf nodes + Morton codes."""
    o = world_coords.world_origin.astype(np.float32)
    s = np.float32(world_coords.world_scale)
    tri = vertices[triangles]                              # (n,3,3) float32
    lower = tri.min(axis=1)
    upper = tri.max(axis=1)
    centroid = ((tri[:, 0] + tri[:, 1]) + tri[:, 2]) / np.float32(3.0)
    qlo = _quantize(lower, o, s)
    qlo = np.where(qlo > 0, qlo - 1, qlo).astype(np.uint32)
    qhi = (_quantize(upper, o, s) + np.uint32(1)).astype(np.uint32)
    qc = _quantize(centroid, o, s)
    morton = _spread3_16(qc[:, 0]) | (_spread3_16(qc[:, 1]) << np.uint64(1)) | (_spread3_16(qc[:, 2]) << np.uint64(2))
    nodes = np.empty(len(triangles), dtype=uint4)
    nodes['x'] = qlo[:, 0] | (qhi[:, 0] << np.uint32(16))
    nodes['y'] = qlo[:, 1] | (qhi[:, 1] << np.uint32(16))
    nodes['z'] = qlo[:, 2] | (qhi[:, 2] << np.uint32(16))
    nodes['w'] = np.arange(len(triangles), dtype=np.uint32)
    return nodes, morton


def _merge_nodes(children, first_child, nchild):
    """make_parents_detailed (cuda/bvh.cu:270-308): box union over explicit child ranges."""
    parents = np.empty(len(first_child), dtype=uint4)
    for axis in 'xyz':
        lo = np.minimum.reduceat(children[axis] & np.uint32(0xFFFF), first_child)
        hi = np.maximum.reduceat(children[axis] >> np.uint32(16), first_child)
        parents[axis] = lo | (hi << np.uint32(16))
    parents['w'] = (nchild.astype(np.uint32) << np.uint32(CHILD_BITS)) | first_child.astype(np.uint32)
    return parents


def _count_unique_sorted(a):
    return int(np.count_nonzero(a[1:] != a[:-1])) + 1


def _build_numpy(vertices, triangles, world_coords, target_degree):
    leaf_nodes, morton = make_leaves_numpy(vertices, triangles, world_coords)
    order = np.argsort(morton, kind='stable')
    leaf_nodes = leaf_nodes[order]
    morton = morton[order]

    layers = [leaf_nodes]
    while len(layers[0]) > 1:
        top = layers[0]
        nnodes = len(top)
        nunique = _count_unique_sorted(morton)
        while nnodes / float(nunique) < target_degree and nunique > 1:
            morton = morton >> np.uint64(1)
            nunique = _count_unique_sorted(morton)
        is_first = np.empty(nnodes, dtype=bool)
        is_first[0] = True
        is_first[1:] = morton[1:] != morton[:-1]
        first_child = np.flatnonzero(is_first).astype(np.int64)
        run_len = np.diff(np.append(first_child, nnodes))
        # runs longer than MAX_CHILD are cut into pieces of MAX_CHILD (grid.py:51-76)
        pieces = (run_len + MAX_CHILD - 1) // MAX_CHILD
        if (pieces > 1).any():
            rep_first = np.repeat(first_child, pieces)
            within = np.arange(len(rep_first)) - np.repeat(np.cumsum(pieces) - pieces, pieces)
            first_child = rep_first + within * MAX_CHILD
        parent_morton = morton[first_child]
        nchild = np.diff(np.append(first_child, nnodes))
        assert (nchild > 0).all() and (nchild <= MAX_CHILD).all()
        layers = [_merge_nodes(top, first_child, nchild)] + layers
        morton = parent_morton

    # concatenate_layers (gpu/bvh.py:239-267, cuda/bvh.cu:365-384)
    bounds = np.insert(np.cumsum([len(l) for l in layers]), 0, 0)
    nodes = np.empty(int(bounds[-1]), dtype=uint4)
    for lo, hi, layer in zip(bounds[:-1], bounds[1:], layers):
        nodes[lo:hi] = layer
        if hi != bounds[-1]:                       # leaves keep their triangle id
            nodes['w'][lo:hi] += np.uint32(hi)     # child index += start of the next layer
    # collapse_chains (gpu/bvh.py:114-130, cuda/bvh.cu:530-543): bottom-up over inner layers
    for lo, hi in list(zip(bounds[:-1], bounds[1:]))[:-1][::-1]:
        w = nodes['w'][lo:hi]
        single = np.flatnonzero((w >> np.uint32(CHILD_BITS)) == 1)
        if len(single):
            child = (w[single] & np.uint32(0x0FFFFFFF)).astype(np.int64)
            nodes[lo + single] = nodes[child]
    return nodes, bounds


def _device_context(cuda_device=None):
    """(context, created): the current chroma_amd.gpu context when it sits on the device asked for (``cuda_device`` None: any),
    else a NEW context on ``cuda_device`` (None: device 0) when the machine has a GPU -- the caller pops a context it had
    created, as chroma/loader.py:150-152 does with the CUDA context it makes for the build; (None, False) without a GPU."""
    try:
        from chroma_amd.gpu import tools as gtools
        cur = gtools._current
        if cur is not None and (cuda_device is None or int(cuda_device) == int(cur.device_id)):
            return cur, False
        if gtools.device_count() > 0:
            return gtools.Context(0 if cuda_device is None else int(cuda_device)), True      # (not pushed: the caller's stays current)
    except Exception:
        pass
    return None, False


def make_recursive_grid_bvh(mesh, target_degree=3, verbose=False, backend=None, cuda_device=None):
    """BVH of ``mesh`` (chroma/bvh/grid.py:11).  ``backend``: 'device' (HIP kernels, as in the reference, where these steps
    are CUDA kernels), 'native' (the same algorithm on the host cores), 'numpy' (the independent restatement); None =
    $CHROMA_BVH_BACKEND, else 'device' when a GPU is there and 'native' otherwise.  All three return the same node array bit
    for bit.  ``cuda_device``: the GPU a device build runs on (chroma/loader.py:150 builds on ``create_cuda_context(cuda_device)``)
    -- the current context when it is on that device, else a context made for the build and released afterwards, so loading
    a geometry leaves no context behind.  A device build that runs out of memory gives the pool back, and if that is not
    enough the host cores build the same tree."""
    import os
    vertices = np.ascontiguousarray(mesh.vertices, dtype=np.float32)
    triangles = np.ascontiguousarray(mesh.triangles, dtype=np.uint32)
    if len(triangles) >= 2 ** CHILD_BITS:
        raise ValueError('mesh has too many triangles for 28-bit child indices')
    world_coords = world_coords_for(vertices)
    backend = backend or os.environ.get('CHROMA_BVH_BACKEND') or 'auto'
    ctx, created = None, False
    if backend in ('auto', 'device'):
        ctx, created = _device_context(cuda_device)
        if ctx is None and backend == 'device':
            raise RuntimeError("backend='device' needs a GPU (chroma_amd.gpu.create_cuda_context)")
        backend = 'native' if ctx is None else 'device'
    if backend in ('native', 'device'):
        from chroma_amd import _lib
        try:
            try:
                nodes, bounds = _lib.bvh_build(vertices, triangles, world_coords.world_origin,
                                               world_coords.world_scale, target_degree, ctx=ctx)
            except _lib.ChromaError as exc:
                if ctx is None or 'memory' not in str(exc).lower():
                    raise
                # the builder's scratch (~60 bytes per triangle) did not fit beside what the device holds: blocks parked in
                # the pool go back first; then the host cores build the same nodes
                ctx.pool_trim()
                try:
                    nodes, bounds = _lib.bvh_build(vertices, triangles, world_coords.world_origin,
                                                   world_coords.world_scale, target_degree, ctx=ctx)
                except _lib.ChromaError:
                    backend = 'native'
                    nodes, bounds = _lib.bvh_build(vertices, triangles, world_coords.world_origin,
                                                   world_coords.world_scale, target_degree, ctx=None)
        finally:
            if created:
                ctx.shutdown()
    elif backend == 'numpy':
        nodes, bounds = _build_numpy(vertices, triangles, world_coords, target_degree)
    else:
        raise ValueError('unknown backend %r' % backend)
    if verbose:
        print('BVH (%s): %d nodes, layers %s' % (backend, len(nodes), list(np.diff(bounds))))
    return BVH(world_coords, nodes, [int(b) for b in bounds[:-1]])
