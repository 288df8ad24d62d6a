"""BVH data model: packed 16-byte nodes in 16-bit fixed-point world coordinates.

Node layout (chroma/cuda/geometry_types.h:57-67, chroma/bvh/bvh.py:8-36): a record of
four uint32 ``x, y, z, w``; ``x = lower_x | upper_x << 16`` (same for y, z);
``w = nchild << 28 | child``.  ``nchild == 0`` marks a leaf whose ``child`` is a
triangle id; otherwise ``child`` is the index of the first of ``nchild`` contiguous
children.  Layers are stored root first (chroma/bvh/bvh.py:106-195).
"""
import numpy as np

uint4 = np.dtype([('x', np.uint32), ('y', np.uint32), ('z', np.uint32), ('w', np.uint32)])

CHILD_BITS = 28
NCHILD_MASK = np.uint32(0xF0000000)    # (0xFFFF << 28) truncated to 32 bits
MAX_CHILD = 2 ** (32 - CHILD_BITS) - 1

_UNPACKED = np.dtype([('xlo', np.uint16), ('xhi', np.uint16), ('ylo', np.uint16), ('yhi', np.uint16),
                      ('zlo', np.uint16), ('zhi', np.uint16), ('child', np.uint32), ('nchild', np.uint16)])


def unpack_nodes(nodes):
    """Record array with the bounds, child index and child count of packed nodes."""
    out = np.empty(len(nodes), dtype=_UNPACKED)
    for axis in 'xyz':
        out[axis + 'lo'] = nodes[axis] & 0xFFFF
        out[axis + 'hi'] = nodes[axis] >> 16
    out['child'] = nodes['w'] & ~NCHILD_MASK
    out['nchild'] = nodes['w'] >> CHILD_BITS
    return out


class OutOfRangeError(Exception):
    """World coordinates do not fit the unsigned 16-bit fixed-point range."""


class WorldCoords(object):
    """world = fixed * world_scale + world_origin."""
    MAX_INT = 2 ** 16 - 1

    def __init__(self, world_origin, world_scale):
        self.world_origin = np.array(world_origin, dtype=np.float32)
        self.world_scale = np.float32(world_scale)

    def world_to_fixed(self, world):
        """Round-to-nearest conversion; raises OutOfRangeError outside [0, 65535]."""
        fixed = ((np.asarray(world, dtype=float) - self.world_origin) / self.world_scale).round()
        if int(fixed.max()) > WorldCoords.MAX_INT or fixed.min() < 0:
            raise OutOfRangeError('range = (%f, %f)' % (fixed.min(), fixed.max()))
        return fixed.astype(np.uint16)

    def fixed_to_world(self, fixed):
        return np.asarray(fixed) * self.world_scale + self.world_origin


def node_areas(nodes):
    """Surface area of each node's box in fixed-point units."""
    u = unpack_nodes(nodes)
    dx = u['xhi'].astype(float) - u['xlo']
    dy = u['yhi'].astype(float) - u['ylo']
    dz = u['zhi'].astype(float) - u['zlo']
    return 2.0 * (dx * dy + dy * dz + dz * dx)


class BVHLayerSlice(object):
    """One layer of a BVH; a view into the parent's node array."""

    def __init__(self, world_coords, nodes):
        self.world_coords = world_coords
        self.nodes = nodes

    def __len__(self):
        return len(self.nodes)

    def areas_fixed(self):
        return node_areas(self.nodes)

    def area_fixed(self):
        return node_areas(self.nodes).sum()

    def areas(self):
        return self.areas_fixed() * self.world_coords.world_scale ** 2

    def area(self):
        return self.area_fixed() * self.world_coords.world_scale ** 2


class BVH(object):
    """world_coords + packed node array + offset of each layer (root layer first)."""

    def __init__(self, world_coords, nodes, layer_offsets):
        self.world_coords = world_coords
        self.nodes = nodes
        self.layer_offsets = layer_offsets
        self.layer_bounds = list(layer_offsets) + [len(nodes)]

    def get_layer(self, layer_number):
        lo, hi = self.layer_bounds[layer_number], self.layer_bounds[layer_number + 1]
        return BVHLayerSlice(world_coords=self.world_coords, nodes=self.nodes[lo:hi])

    def layer_count(self):
        return len(self.layer_offsets)

    def __len__(self):
        return len(self.nodes)
