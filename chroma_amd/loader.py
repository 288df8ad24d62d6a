"""Geometry loading helpers (the part of chroma/loader.py the propagate path needs).

``load_bvh`` (chroma/loader.py:131-160) in the reference needs a CUDA context because its
BVH builder runs on the GPU; here the builder is host code, so no device is touched.
The on-disk cache of the reference (pickles under ~/.chroma) is not reproduced yet.
"""
from chroma_amd.bvh import make_recursive_grid_bvh
from chroma_amd.log import logger


def load_bvh(geometry, bvh_name="default", auto_build_bvh=True, read_bvh_cache=True,
             update_bvh_cache=True, cache_dir=None, cuda_device=None, target_degree=3):
    """Attach a BVH to a flattened geometry and return it."""
    if not hasattr(geometry, 'mesh'):
        geometry.flatten()
    logger.info('Building new BVH using recursive grid algorithm.')
    return make_recursive_grid_bvh(geometry.mesh, target_degree=target_degree)


def create_geometry_from_obj(obj, bvh_name="default", auto_build_bvh=True, read_bvh_cache=True,
                             update_bvh_cache=True, cache_dir=None, cuda_device=None):
    """Flatten a Geometry/Detector (or wrap a Solid/Mesh in one) and build its BVH
    (chroma/loader.py:46-88)."""
    from chroma_amd.geometry import Geometry, Solid, Mesh, vacuum
    if isinstance(obj, Geometry):
        geometry = obj
    elif isinstance(obj, Solid):
        geometry = Geometry()
        geometry.add_solid(obj)
    elif isinstance(obj, Mesh):
        geometry = Geometry()
        geometry.add_solid(Solid(obj, vacuum, vacuum, color=0x33ffffff))
    else:
        raise TypeError('cannot build a geometry from %r' % type(obj))
    geometry.flatten()
    if geometry.bvh is None:
        geometry.bvh = load_bvh(geometry, auto_build_bvh=auto_build_bvh)
    return geometry
