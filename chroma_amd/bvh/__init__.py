"""Bounding volume hierarchy: data model and the recursive-grid builder."""
from chroma_amd.bvh.bvh import (BVH, BVHLayerSlice, WorldCoords, OutOfRangeError, uint4,
                                unpack_nodes, node_areas, CHILD_BITS, NCHILD_MASK, MAX_CHILD)
from chroma_amd.bvh.grid import make_recursive_grid_bvh
