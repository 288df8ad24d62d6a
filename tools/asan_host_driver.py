"""Driver of tools/asan_host.sh: the host-side builders (vertex de-duplication, BVH, wide tree in both
topologies, malformed input) under AddressSanitizer + UBSan, on a cube and on demo.tiny()."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from ctypes import c_void_p, c_uint64, c_uint32, c_float, POINTER
lib = ctypes.CDLL(os.environ['CHROMA_ASAN_LIB'])
from chroma_amd import demo, make
from chroma_amd.geometry import Geometry, Solid, vacuum
from chroma_amd.bvh.grid import world_coords_for

def ptr(a): return a.ctypes.data_as(c_void_p)

def run(name, geo):
    geo.flatten() if hasattr(geo, 'flatten') and getattr(geo, 'mesh', None) is None else None
    mesh = geo.mesh
    v = np.ascontiguousarray(mesh.vertices, dtype=np.float32)
    t = np.ascontiguousarray(mesh.triangles, dtype=np.uint32)
    # dedupe on a copy with duplicated vertices appended
    v2 = np.concatenate([v, v[: len(v) // 3]])
    t2 = t.copy(); uniq = np.empty_like(v2); nu = c_uint64()
    rc = lib.chroma_dedupe_vertices(ptr(v2), c_uint64(len(v2)), ptr(t2), c_uint64(t2.size), ptr(uniq), ctypes.byref(nu))
    assert rc == 0 and nu.value <= len(v)
    wc = world_coords_for(v)
    origin = (c_float * 3)(*[float(x) for x in wc.world_origin])
    for degree in (2, 3, 4):
        handle, nnodes, nlayers = c_void_p(), c_uint64(), c_uint32()
        rc = lib.chroma_bvh_build(ptr(v), c_uint32(len(v)), ptr(t), c_uint32(len(t)), origin, c_float(float(wc.world_scale)), degree,
                                  ctypes.byref(handle), ctypes.byref(nnodes), ctypes.byref(nlayers))
        assert rc == 0, rc
        pn, pb = c_void_p(), c_void_p()
        lib.chroma_bvh_data(handle, ctypes.byref(pn), ctypes.byref(pb))
        nodes = np.array((ctypes.c_uint32 * (4 * nnodes.value)).from_address(pn.value), dtype=np.uint32).reshape(-1, 4)
        for topo in ('sah', 'greedy', 'collapse'):
            os.environ['CHROMA_TREE'] = topo
            wh, nw, nr, dp = c_void_p(), c_uint64(), c_uint64(), c_uint32()
            rc = lib.chroma_wide_build(ptr(nodes), c_uint64(len(nodes)), c_uint32(len(t)), ctypes.byref(wh), ctypes.byref(nw), ctypes.byref(nr), ctypes.byref(dp))
            assert rc == 0, rc
            lib.chroma_wide_free(wh)
        # malformed input must be refused, not crash
        bad = nodes.copy(); bad[0, 3] = (3 << 28) | 0x0FFFFFF0
        wh = c_void_p()
        rc = lib.chroma_wide_build(ptr(bad), c_uint64(len(bad)), c_uint32(len(t)), ctypes.byref(wh), None, None, None)
        assert rc != 0
        lib.chroma_bvh_free(handle)
    print(name, 'ok:', len(t), 'triangles')

for fn in (lib.chroma_dedupe_vertices, lib.chroma_bvh_build, lib.chroma_bvh_data, lib.chroma_bvh_free, lib.chroma_wide_build, lib.chroma_wide_free):
    fn.restype = ctypes.c_int32
cube = Geometry(); cube.add_solid(Solid(make.cube(100.0), vacuum, vacuum)); cube.flatten()
run('cube', cube)
tiny = demo.tiny(); tiny.flatten()
run('tiny', tiny)
