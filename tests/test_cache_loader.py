"""BVH cache (.npz) and the "@module.function" geometry loader."""
import numpy as np
import pytest

from chroma_amd.cache import Cache, BVHNotFoundError
from chroma_amd.loader import load_geometry_from_string, create_geometry_from_obj, load_bvh
from chroma_amd import make


def test_bvh_cache_round_trip(tmp_path):
    geo = create_geometry_from_obj(make.sphere(100.0, 16))
    cache = Cache(str(tmp_path))
    h = geo.mesh.md5()
    assert not cache.exist_bvh(h) and cache.list_bvh(h) == []
    with pytest.raises(BVHNotFoundError):
        cache.load_bvh(h)
    cache.save_bvh(geo.bvh, h, 'default')
    assert cache.list_bvh(h) == ['default']
    back = cache.load_bvh(h)
    assert np.array_equal(back.nodes, geo.bvh.nodes) and back.layer_offsets == list(geo.bvh.layer_offsets)
    assert back.world_coords.world_scale == geo.bvh.world_coords.world_scale
    assert np.array_equal(back.world_coords.world_origin, geo.bvh.world_coords.world_origin)
    cache.remove_bvh(h)
    with pytest.raises(BVHNotFoundError):
        cache.remove_bvh(h)
    with pytest.raises(ValueError):
        cache.get_bvh_path(h, '../evil')


def test_loader_uses_and_fills_the_cache(tmp_path):
    geo = load_geometry_from_string('@chroma_amd.demo.tiny', cache_dir=str(tmp_path))
    assert geo.bvh is not None and len(geo.mesh.triangles) == 389568
    cache = Cache(str(tmp_path))
    assert cache.exist_bvh(geo.mesh.md5())
    again = load_bvh(geo, cache_dir=str(tmp_path), auto_build_bvh=False)       # must come from the cache
    assert np.array_equal(again.nodes, geo.bvh.nodes)
    with pytest.raises(Exception):
        load_bvh(geo, bvh_name='other', cache_dir=str(tmp_path), auto_build_bvh=False)
    with pytest.raises(ValueError):
        load_geometry_from_string('detector.stl')
