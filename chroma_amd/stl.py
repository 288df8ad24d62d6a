"""STL meshes (binary and ASCII, optionally bz2-compressed) -> chroma_amd.geometry.Mesh.

Takes the place of chroma/stl.py:7-103 (``mesh_from_stl`` and its two readers): same result -- vertices
merged when they are exactly equal, triangles in file order -- but read with NumPy in one go instead of a
Python loop per vertex, so a multi-million-triangle detector file loads in seconds.
"""
import bz2
import re

import numpy as np

from chroma_amd.geometry import Mesh


def _read(filename):
    opener = bz2.BZ2File if filename.endswith('.bz2') else open
    with opener(filename, 'rb') as f:
        return f.read()


def _merge(corners):
    """[ntri][3][3] float32 corner coordinates -> Mesh with exactly-equal vertices merged; vertex numbers follow
    first appearance in the file, as the reference's dict-based readers give them."""
    flat = np.ascontiguousarray(corners.reshape(-1, 3))
    keys = flat.view([('x', flat.dtype), ('y', flat.dtype), ('z', flat.dtype)]).reshape(-1)
    _, first, inverse = np.unique(keys, return_index=True, return_inverse=True)
    order = np.argsort(first, kind='stable')             # unique vertex k (sorted order) -> rank by first appearance
    rank = np.empty(len(order), dtype=np.int64)
    rank[order] = np.arange(len(order))
    vertices = flat[first[order]].astype(np.float64)
    triangles = rank[inverse.reshape(-1)].reshape(-1, 3).astype(np.uint32)
    return Mesh(vertices, triangles)


def mesh_from_binary_stl(filename):
    "Return a mesh from a binary stl file."
    data = _read(filename)
    ntriangles = int(np.frombuffer(data, dtype='<u4', count=1, offset=80)[0])
    rec = np.dtype([('normal', '<f4', 3), ('corners', '<f4', (3, 3)), ('attr', '<u2')])
    if len(data) < 84 + ntriangles * rec.itemsize:
        raise ValueError('%s: truncated binary STL (%d triangles announced)' % (filename, ntriangles))
    body = np.frombuffer(data, dtype=rec, count=ntriangles, offset=84)
    return _merge(np.array(body['corners'], dtype=np.float32))


_VERTEX = re.compile(rb'vertex\s+(\S+)\s+(\S+)\s+(\S+)')


def mesh_from_ascii_stl(filename):
    "Return a mesh from an ascii stl file."
    found = _VERTEX.findall(_read(filename))
    if len(found) % 3:
        raise ValueError('%s: ASCII STL with %d vertex lines (not a multiple of 3)' % (filename, len(found)))
    corners = np.array(found, dtype=np.float64).reshape(-1, 3, 3)
    return _merge(corners)


def mesh_from_stl(filename):
    "Returns a `chroma_amd.geometry.Mesh` from an STL file (binary or ASCII, plain or .bz2)."
    head = _read(filename)[:512]
    is_ascii = head.lstrip().startswith(b'solid') and b'facet' in head
    return mesh_from_ascii_stl(filename) if is_ascii else mesh_from_binary_stl(filename)
