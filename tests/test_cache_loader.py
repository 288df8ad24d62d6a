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
    with pytest.raises(Exception):
        load_geometry_from_string('no_such_geometry', cache_dir=str(tmp_path))


def test_loader_takes_module_level_objects_as_well_as_functions(tmp_path, monkeypatch):
    """chroma/loader.py:166 calls the named attribute only ``if callable(obj)``: a module-level Mesh, Solid or
    Geometry is a valid "@module.name" too."""
    (tmp_path / 'my_geometry_module.py').write_text(
        'from chroma_amd.make import cube\n'
        'from chroma_amd.geometry import Solid, vacuum\n'
        'a_mesh = cube(10.0)\n'
        'a_solid = Solid(cube(20.0), vacuum, vacuum)\n'
        'def a_function():\n'
        '    return cube(30.0)\n')
    monkeypatch.chdir(tmp_path)                      # (the current directory is importable, chroma/loader.py:152)
    sizes = {}
    for name in ('a_mesh', 'a_solid', 'a_function'):
        geo = load_geometry_from_string('@my_geometry_module.' + name, cache_dir=str(tmp_path / 'cache'))
        assert geo.bvh is not None and len(geo.mesh.triangles) > 0
        sizes[name] = float(np.ptp(geo.mesh.vertices[:, 0]))
    assert sizes == {'a_mesh': 10.0, 'a_solid': 20.0, 'a_function': 30.0}


def _write_stl(path, mesh, binary):
    import bz2
    import struct
    tri = mesh.vertices[mesh.triangles].astype(np.float32)
    if binary:
        data = b'binary stl'.ljust(80, b' ') + struct.pack('<I', len(tri))
        for t in tri:
            data += struct.pack('<fff', 0, 0, 0) + t.tobytes() + b'\0\0'
    else:
        lines = ['solid test']
        for t in tri:
            lines += [' facet normal 0 0 0', '  outer loop'] + ['   vertex %r %r %r' % tuple(float(x) for x in v) for v in t] + ['  endloop', ' endfacet']
        data = ('\n'.join(lines + ['endsolid test']) + '\n').encode()
    if path.endswith('.bz2'):
        data = bz2.compress(data)
    with open(path, 'wb') as f:
        f.write(data)


@pytest.mark.parametrize('name,binary', [('cube.stl', True), ('cube_ascii.stl', False), ('cube.stl.bz2', True), ('cube_ascii.stl.bz2', False)])
def test_stl_files(tmp_path, name, binary):
    """"file.stl[.bz2][:bvh]" (chroma/loader.py:82-88, chroma/stl.py): binary and ASCII, plain and compressed; vertices
    merged when equal, numbered by first appearance, triangles in file order -- what the reference's readers give."""
    from chroma_amd.stl import mesh_from_stl
    mesh = make.cube(100.0)
    path = str(tmp_path / name)
    _write_stl(path, mesh, binary)
    back = mesh_from_stl(path)
    nv = len(np.unique(mesh.vertices[mesh.triangles].reshape(-1, 3).astype(np.float32), axis=0))
    assert back.triangles.shape == mesh.triangles.shape and len(back.vertices) == nv
    assert np.array_equal(back.vertices[back.triangles].astype(np.float32), mesh.vertices[mesh.triangles].astype(np.float32))
    # first-appearance numbering: walking the corners in file order meets the vertex numbers in increasing order
    first = np.unique(back.triangles.reshape(-1), return_index=True)[1]
    assert np.array_equal(np.argsort(first), np.arange(nv))
    geo = load_geometry_from_string(path + ':mine', cache_dir=str(tmp_path))
    assert geo.bvh is not None and len(geo.mesh.triangles) == len(mesh.triangles)
    assert Cache(str(tmp_path)).list_bvh(geo.mesh.md5()) == ['mine']


def test_named_geometries_in_the_cache(tmp_path):
    """"name[:bvh]" and "" (chroma/loader.py:114-124, chroma/cache.py:90-176)."""
    from chroma_amd.cache import GeometryNotFoundError
    from chroma_amd.geometry import Geometry, Solid, vacuum
    from chroma_amd.demo.optics import water
    cache = Cache(str(tmp_path))
    g = Geometry(water)
    g.add_solid(Solid(make.sphere(50.0, 12), water, vacuum))
    g.flatten()
    with pytest.raises(GeometryNotFoundError):
        cache.load_default_geometry()
    cache.save_geometry('ball', g)
    assert cache.list_geometry() == ['ball']
    cache.set_default_geometry('ball')
    for s in ('ball', 'ball:fine', ''):
        geo = load_geometry_from_string(s, cache_dir=str(tmp_path))
        assert geo.bvh is not None and np.array_equal(geo.mesh.triangles, g.mesh.triangles)
        assert geo.unique_materials[0].name == g.unique_materials[0].name
    assert sorted(cache.list_bvh(g.mesh.md5())) == ['default', 'fine']
    cache.remove_geometry('ball')
    with pytest.raises(GeometryNotFoundError):
        cache.load_geometry('ball')
    with pytest.raises(ValueError):
        cache.get_geometry_filename('../x')
