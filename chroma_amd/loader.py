"""Geometry loading helpers (the part of chroma/loader.py the propagate path needs).

``load_bvh`` (chroma/loader.py:131-160) in the reference makes a CUDA context on ``cuda_device`` because its BVH
builder runs on the GPU, and pops it afterwards.  Same here: the BVH is built on the device when there is one -- on the
current context if that sits on ``cuda_device``, else on a context made for the build and released again -- and on the
host cores otherwise (same nodes bit for bit: chroma_amd/bvh/grid.py).
BVHs are cached as .npz files (chroma_amd/cache.py) when a ``cache_dir`` is given; unlike the
reference the cache is OFF by default, because a build takes seconds.
``load_geometry_from_string`` resolves the "@module.function" form of chroma/loader.py:90-112.
"""
import importlib
import os
import sys

from chroma_amd.bvh import make_recursive_grid_bvh
from chroma_amd.log import logger


def load_bvh(geometry, bvh_name="default", auto_build_bvh=True, read_bvh_cache=True,
             update_bvh_cache=True, cache_dir=None, cuda_device=None, target_degree=3):
    """Return a BVH for a flattened geometry: from the cache if present, otherwise built (and
    cached when ``update_bvh_cache`` and a ``cache_dir`` are given)."""
    if not hasattr(geometry, 'mesh'):
        geometry.flatten()
    cache = None
    mesh_hash = None
    if cache_dir is not None:
        from chroma_amd.cache import Cache
        cache = Cache(cache_dir)
        mesh_hash = geometry.mesh.md5()
        if read_bvh_cache and cache.exist_bvh(mesh_hash, bvh_name):
            logger.info('Loading BVH "%s" from cache.' % bvh_name)
            return cache.load_bvh(mesh_hash, bvh_name)
    if not auto_build_bvh:
        raise Exception('BVH "%s" not found in cache and auto_build_bvh is off' % bvh_name)
    logger.info('Building new BVH using recursive grid algorithm.')
    bvh = make_recursive_grid_bvh(geometry.mesh, target_degree=target_degree, cuda_device=cuda_device)
    if cache is not None and update_bvh_cache:
        logger.info('Saving BVH (%s:%s) to cache.' % (mesh_hash, bvh_name))
        cache.save_bvh(bvh, mesh_hash, bvh_name)
    return bvh


def create_geometry_from_obj(obj, bvh_name="default", auto_build_bvh=True, read_bvh_cache=True,
                             update_bvh_cache=True, cache_dir=None, cuda_device=None):
    """Flatten a Geometry/Detector (or wrap a Solid/Mesh in one) and attach its BVH
    (chroma/loader.py:46-88).  A callable is called first (chroma/loader.py:166: ``if callable(obj): obj = obj()``),
    so a module-level Geometry / Solid / Mesh works as well as a function that returns one."""
    from chroma_amd.geometry import Geometry, Solid, Mesh, vacuum
    if callable(obj):
        obj = obj()
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
        geometry.bvh = load_bvh(geometry, bvh_name=bvh_name, auto_build_bvh=auto_build_bvh, read_bvh_cache=read_bvh_cache,
                                update_bvh_cache=update_bvh_cache, cache_dir=cache_dir, cuda_device=cuda_device)
    return geometry


def load_geometry_from_string(geometry_str, auto_build_bvh=True, read_bvh_cache=True, update_bvh_cache=True,
                              cache_dir=None, cuda_device=None):
    """A flattened geometry with its BVH from one of the reference's geometry strings (chroma/loader.py:13-137):
    ``"file.stl[.bz2][:bvh]"`` (a mesh on disk, vacuum inside and out), ``"@module.name[:bvh]"`` (a Geometry, Solid or Mesh, or a
    function that returns one when called without arguments; the current directory is importable too),
    ``"name[:bvh]"`` (a geometry saved in the cache under that name) and ``""`` (the cache's default geometry).
    ``cuda_device``: the GPU the BVH is built on (load_bvh)."""
    geometry_id, _, bvh_name = geometry_str.partition(':')
    bvh_name = bvh_name or 'default'
    kw = dict(bvh_name=bvh_name, auto_build_bvh=auto_build_bvh, read_bvh_cache=read_bvh_cache,
              update_bvh_cache=update_bvh_cache, cache_dir=cache_dir, cuda_device=cuda_device)
    if geometry_id.startswith('@'):
        module_name, _, function_name = geometry_id[1:].rpartition('.')
        saved = list(sys.path)
        try:
            sys.path.append('.')
            module = importlib.import_module(module_name)
        finally:
            sys.path[:] = saved
        return create_geometry_from_obj(getattr(module, function_name), **kw)
    if os.path.exists(geometry_id) and geometry_id.lower().endswith(('.stl', '.bz2')):
        from chroma_amd.stl import mesh_from_stl
        from chroma_amd.geometry import Geometry, Solid, vacuum
        geometry = Geometry()
        geometry.add_solid(Solid(mesh_from_stl(geometry_id), vacuum, vacuum, color=0x33ffffff))
        return create_geometry_from_obj(geometry, **kw)
    from chroma_amd.cache import Cache
    cache = Cache(cache_dir)
    geometry = cache.load_default_geometry() if geometry_id == '' else cache.load_geometry(geometry_id)
    return create_geometry_from_obj(geometry, **kw)
