"""On-disk cache of BVHs keyed by the mesh MD5.

The reference keeps pickles of ``BVH`` / ``Geometry`` objects under ``~/.chroma/{bvh,geo}``
(chroma/cache.py:64-88,178-234, key = ``Mesh.md5()``, chroma/geometry.py:105-110).  Here a BVH is
one ``.npz`` file -- ``world_origin``, ``world_scale``, ``nodes`` ([n][4] uint32), ``layer_offsets`` --
so nothing is unpickled and the file can be read without this package.  Building the BVH of the
29k-PMT geometry takes ~10 s on the GPU box's cores, so the cache matters less than in the
reference (whose builder needs a CUDA context), but ``chroma-sim``-style drivers expect it.
"""
import os

import numpy as np

from chroma_amd.bvh.bvh import BVH, WorldCoords, uint4


class BVHNotFoundError(Exception):
    pass


def default_cache_dir():
    return os.path.join(os.path.expanduser('~'), '.chroma_amd')


class GeometryNotFoundError(Exception):
    pass


class Cache(object):
    def __init__(self, cache_dir=None):
        self.cache_dir = default_cache_dir() if cache_dir is None else cache_dir
        self.bvh_dir = os.path.join(self.cache_dir, 'bvh')
        self.geo_dir = os.path.join(self.cache_dir, 'geo')
        os.makedirs(self.bvh_dir, exist_ok=True)
        os.makedirs(self.geo_dir, exist_ok=True)

    # ---- named geometries (chroma/cache.py:90-176): pickles of flattened Geometry objects WITHOUT their BVH, which
    # is looked up by mesh hash.  A pickle runs code when loaded: the cache directory is the user's own, as in the
    # reference; never point cache_dir at files from somebody else.
    def get_geometry_filename(self, name):
        if not name or not all(c.isalnum() or c in '._-' for c in name) or name.startswith('.'):
            raise ValueError('invalid geometry name %r' % name)
        return os.path.join(self.geo_dir, name)

    def list_geometry(self):
        return sorted(n for n in os.listdir(self.geo_dir) if n != 'default')

    def save_geometry(self, name, geometry):
        import copy
        import pickle
        g = copy.copy(geometry)
        g.bvh = None
        path = self.get_geometry_filename(name)
        with open(path + '.tmp', 'wb') as f:
            pickle.dump(g, f, pickle.HIGHEST_PROTOCOL)
        os.replace(path + '.tmp', path)
        return path

    def load_geometry(self, name):
        import pickle
        path = self.get_geometry_filename(name)
        if not os.path.exists(path):
            raise GeometryNotFoundError(name)
        with open(path, 'rb') as f:
            return pickle.load(f)

    def remove_geometry(self, name):
        path = self.get_geometry_filename(name)
        if not os.path.exists(path):
            raise GeometryNotFoundError(name)
        os.remove(path)

    def set_default_geometry(self, name):
        if not os.path.exists(self.get_geometry_filename(name)):
            raise GeometryNotFoundError(name)
        link = os.path.join(self.geo_dir, 'default')
        if os.path.lexists(link):
            os.remove(link)
        os.symlink(name, link)

    def load_default_geometry(self):
        link = os.path.join(self.geo_dir, 'default')
        if not os.path.exists(link):
            raise GeometryNotFoundError('default')
        return self.load_geometry(os.path.basename(os.path.realpath(link)))

    def get_bvh_path(self, mesh_hash, name='default'):
        if not name or not all(c.isalnum() or c in '._-' for c in name) or name.startswith('.'):
            raise ValueError('invalid BVH name %r' % name)
        return os.path.join(self.bvh_dir, '%s_%s.npz' % (mesh_hash, name))

    def list_bvh(self, mesh_hash):
        prefix = mesh_hash + '_'
        return sorted(f[len(prefix):-4] for f in os.listdir(self.bvh_dir) if f.startswith(prefix) and f.endswith('.npz'))

    def exist_bvh(self, mesh_hash, name='default'):
        return os.path.exists(self.get_bvh_path(mesh_hash, name))

    def save_bvh(self, bvh, mesh_hash, name='default'):
        path = self.get_bvh_path(mesh_hash, name)
        tmp = path + '.tmp.npz'
        np.savez(tmp, world_origin=bvh.world_coords.world_origin, world_scale=np.float32(bvh.world_coords.world_scale),
                 nodes=np.ascontiguousarray(bvh.nodes).view(np.uint32).reshape(-1, 4),
                 layer_offsets=np.asarray(bvh.layer_offsets, dtype=np.int64))
        os.replace(tmp, path)
        return path

    def load_bvh(self, mesh_hash, name='default'):
        path = self.get_bvh_path(mesh_hash, name)
        if not os.path.exists(path):
            raise BVHNotFoundError(path)
        with np.load(path) as f:
            nodes = np.ascontiguousarray(f['nodes'], dtype=np.uint32).reshape(-1, 4).view(uint4).reshape(-1)
            return BVH(WorldCoords(f['world_origin'], f['world_scale']), nodes, [int(x) for x in f['layer_offsets']])

    def remove_bvh(self, mesh_hash, name='default'):
        path = self.get_bvh_path(mesh_hash, name)
        if not os.path.exists(path):
            raise BVHNotFoundError(path)
        os.remove(path)
