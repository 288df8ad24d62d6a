"""ctypes binding of libchroma_hip.so (C ABI: include/chroma_hip.h).

This is the only place the package touches native code.  There is deliberately no
CPU fallback: if the shared library is missing or cannot be loaded, importing a GPU
entry point raises immediately and says how to build it.
"""
import ctypes
import os
import sys
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int32, c_size_t, c_uint32,
                    c_uint64, c_void_p)

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# CHROMA_HIP_LIBRARY points at an alternative build (kernel A/B experiments); default: in-tree
LIBRARY_PATH = os.environ.get('CHROMA_HIP_LIBRARY') or os.path.join(_HERE, 'libchroma_hip.so')

_f32p = POINTER(c_float)
_u32p = POINTER(c_uint32)
_i32p = POINTER(c_int32)


class GeometryDesc(Structure):
    """chroma_geometry_desc"""
    _fields_ = [
        ('vertices', c_void_p), ('triangles', c_void_p), ('material_codes', c_void_p),
        ('solid_id_map', c_void_p), ('colors', c_void_p),
        ('nvertices', c_uint32), ('ntriangles', c_uint32),
        ('nodes', c_void_p), ('nnodes', c_uint32),
        ('world_origin', c_float * 3), ('world_scale', c_float),
        ('wavelength_n', c_uint32), ('wavelength_start', c_float), ('wavelength_step', c_float),
        ('time_n', c_uint32), ('time_start', c_float), ('time_step', c_float),
        ('nmaterials', c_uint32),
        ('mat_refractive_index', c_void_p), ('mat_absorption_length', c_void_p),
        ('mat_scattering_length', c_void_p), ('mat_num_comp', c_void_p), ('mat_comp_offset', c_void_p),
        ('ncomp_total', c_uint32),
        ('comp_reemission_prob', c_void_p), ('comp_reemission_wvl_cdf', c_void_p),
        ('comp_absorption_length', c_void_p), ('comp_reemission_time_cdf', c_void_p),
        ('nsurfaces', c_uint32),
        ('surf_detect', c_void_p), ('surf_absorb', c_void_p), ('surf_reemit', c_void_p),
        ('surf_reflect_diffuse', c_void_p), ('surf_reflect_specular', c_void_p),
        ('surf_eta', c_void_p), ('surf_k', c_void_p), ('surf_reemission_cdf', c_void_p),
        ('surf_model', c_void_p), ('surf_transmissive', c_void_p), ('surf_thickness', c_void_p),
        ('surf_dichroic_index', c_void_p),
        ('ndichroic', c_uint32),
        ('dichroic_nangles', c_void_p), ('dichroic_offset', c_void_p),
        ('ndichroic_angles_total', c_uint32),
        ('dichroic_angles', c_void_p), ('dichroic_reflect', c_void_p), ('dichroic_transmit', c_void_p),
        ('solid_id_to_channel_index', c_void_p),
        ('nsolids', c_uint32), ('nchannels', c_uint32),
        ('wide_nodes', c_void_p), ('wide_tri_to_record', c_void_p), ('wide_record_to_tri', c_void_p),
        ('wide_rank', c_void_p), ('nwide', c_uint64), ('nrecords', c_uint64),
    ]


class PhotonArrays(Structure):
    """chroma_photon_arrays (device or host pointers, depending on the callee)"""
    _fields_ = [('pos', c_void_p), ('dir', c_void_p), ('pol', c_void_p), ('wavelengths', c_void_p),
                ('t', c_void_p), ('flags', c_void_p), ('last_hit_triangles', c_void_p),
                ('weights', c_void_p), ('evidx', c_void_p), ('rng_counters', c_void_p)]


class Rng(Structure):
    """chroma_rng"""
    _fields_ = [('seed', c_uint64), ('photon_id_base', c_uint64)]


class PropagateStats(Structure):
    """chroma_propagate_stats"""
    _fields_ = [('photon_steps', c_uint64), ('nodes_visited', c_uint64), ('triangles_tested', c_uint64),
                ('launches', c_uint64), ('stack_overflows', c_uint64), ('kernel_ms', c_double),
                ('raycast_ms', c_double), ('raycast_launches', c_uint64), ('stack_spills', c_uint64),
                ('physics_ms', c_double), ('physics_launches', c_uint64), ('packet_ms', c_double), ('packet_launches', c_uint64),
                ('packet_rays', c_uint64), ('packet_nodes_visited', c_uint64), ('packet_triangles_tested', c_uint64),
                ('reordered', c_uint64)]

    def as_dict(self):
        return {name: getattr(self, name) for name, _ in self._fields_}


class HitsRequest(Structure):
    """chroma_hits_request"""
    _fields_ = [('detection_state', c_uint32), ('capacity', c_uint32), ('dst', POINTER(PhotonArrays)), ('d_channels', c_void_p),
                ('d_hit_count', c_void_p), ('d_earliest_time_bits', c_void_p), ('nhits', c_uint32)]


class PropagateOptions(Structure):
    """chroma_propagate_options: what ONE call does (-1: the context's setting)"""
    _fields_ = [('max_steps', c_int32), ('use_weights', c_int32), ('scatter_first', c_int32), ('time_kernels', c_int32),
                ('walk', c_int32), ('tail', c_int32), ('counting', c_int32), ('reserved', c_int32 * 5)]

    def __init__(self, max_steps=10, use_weights=False, scatter_first=0, time_kernels=False, walk=-1, tail=-1, counting=-1):
        super().__init__(int(max_steps), int(bool(use_weights)), int(scatter_first), int(bool(time_kernels)), int(walk), int(tail), int(counting))


class DaqTables(Structure):
    """chroma_daq_tables"""
    _fields_ = [('d_time_cdf_x', c_void_p), ('d_time_cdf_y', c_void_p), ('time_cdf_len', c_int32),
                ('d_charge_cdf_x', c_void_p), ('d_charge_cdf_y', c_void_p), ('charge_cdf_len', c_int32),
                ('charge_unit', c_float)]


# name -> (restype, argtypes); every symbol include/chroma_hip.h declares
SIGNATURES = {
    'chroma_last_error': (c_char_p, []),
    'chroma_version': (c_char_p, []),
    'chroma_device_count': (c_int32, [POINTER(c_int32)]),
    'chroma_init': (c_int32, [c_int32, POINTER(c_void_p)]),
    'chroma_shutdown': (c_int32, [c_void_p]),
    'chroma_synchronize': (c_int32, [c_void_p]),
    'chroma_mem_info': (c_int32, [c_void_p, POINTER(c_size_t), POINTER(c_size_t)]),
    'chroma_device_name': (c_int32, [c_void_p, c_char_p, c_size_t]),
    'chroma_malloc': (c_int32, [c_void_p, c_size_t, POINTER(c_void_p)]),
    'chroma_free': (c_int32, [c_void_p, c_void_p]),
    'chroma_memcpy_htod': (c_int32, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'chroma_upload': (c_int32, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'chroma_pool_trim': (c_int32, [c_void_p]),
    'chroma_pool_stats': (c_int32, [c_void_p, POINTER(c_uint64), POINTER(c_uint64), POINTER(c_uint64)]),
    'chroma_memcpy_dtoh': (c_int32, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'chroma_memcpy_dtod': (c_int32, [c_void_p, c_void_p, c_void_p, c_size_t]),
    'chroma_memset32': (c_int32, [c_void_p, c_void_p, c_uint32, c_size_t]),
    'chroma_geometry_create': (c_int32, [c_void_p, POINTER(GeometryDesc), POINTER(c_void_p)]),
    'chroma_geometry_destroy': (c_int32, [c_void_p]),
    'chroma_geometry_device_ptr': (c_int32, [c_void_p, c_char_p, POINTER(c_void_p), POINTER(c_size_t)]),
    'chroma_geometry_stack_need': (c_int32, [c_void_p, POINTER(c_uint32)]),
    'chroma_propagate_step': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, Rng,
                                        POINTER(PhotonArrays), c_int32, c_int32, c_int32]),
    'chroma_photon_duplicate': (c_int32, [c_void_p, c_int32, c_int32, POINTER(PhotonArrays), c_int32, c_int32]),
    'chroma_count_photons': (c_int32, [c_void_p, c_int32, c_int32, c_uint32, c_void_p, POINTER(c_uint32)]),
    'chroma_copy_photons': (c_int32, [c_void_p, c_int32, c_int32, c_uint32, POINTER(PhotonArrays),
                                      POINTER(PhotonArrays), POINTER(c_uint32)]),
    'chroma_copy_photon_queue': (c_int32, [c_void_p, c_int32, c_int32, c_void_p, POINTER(PhotonArrays),
                                           POINTER(PhotonArrays)]),
    'chroma_count_photon_hits': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_uint32,
                                           POINTER(PhotonArrays), POINTER(c_uint32)]),
    'chroma_copy_photon_hits': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_uint32, POINTER(PhotonArrays),
                                          POINTER(PhotonArrays), c_void_p, POINTER(c_uint32)]),
    'chroma_distance_to_mesh': (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p]),
    'chroma_intersect_mesh': (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    'chroma_propagate': (c_int32, [c_void_p, c_void_p, POINTER(PhotonArrays), c_uint64, c_uint32, Rng, c_int32,
                                   c_int32, c_int32, c_int32, POINTER(PropagateStats), POINTER(c_int32)]),
    'chroma_propagate_hits': (c_int32, [c_void_p, c_void_p, POINTER(PhotonArrays), c_uint64, c_uint32, Rng, c_int32,
                                        c_int32, c_int32, c_int32, POINTER(PropagateStats), POINTER(c_int32), POINTER(HitsRequest)]),
    'chroma_propagate_opt': (c_int32, [c_void_p, c_void_p, POINTER(PhotonArrays), c_uint64, c_uint32, Rng, POINTER(PropagateOptions),
                                       POINTER(PropagateStats), POINTER(c_int32), POINTER(HitsRequest)]),
    'chroma_channel_hits': (c_int32, [c_void_p, c_void_p, c_uint64, c_uint32, POINTER(PhotonArrays),
                                      c_void_p, c_void_p]),
    'chroma_daq_reset': (c_int32, [c_void_p, c_float, c_uint32, c_void_p, c_void_p, c_void_p]),
    'chroma_daq_acquire': (c_int32, [c_void_p, c_void_p, POINTER(DaqTables), c_int32, c_int32, c_uint32, POINTER(PhotonArrays),
                                     Rng, c_uint32, c_float, c_void_p, c_void_p, c_void_p]),
    'chroma_daq_acquire_many': (c_int32, [c_void_p, c_void_p, POINTER(DaqTables), c_int32, c_int32, c_uint32, POINTER(PhotonArrays),
                                          Rng, c_uint32, c_float, c_int32, c_int32, c_void_p, c_void_p, c_void_p]),
    'chroma_daq_convert': (c_int32, [c_void_p, c_uint32, c_float, c_void_p, c_void_p, c_void_p, c_void_p]),
    'chroma_generate_bomb': (c_int32, [c_void_p, POINTER(PhotonArrays), c_uint64, c_uint64, c_uint64,
                                       POINTER(c_float), c_float, c_float]),
    'chroma_render': (c_int32, [c_void_p, c_void_p, c_int32, c_void_p, c_void_p, c_uint32, c_void_p, c_void_p, c_void_p, c_void_p, c_uint32]),
    'chroma_points_translate': (c_int32, [c_void_p, c_int32, c_void_p, POINTER(c_float)]),
    'chroma_points_rotate': (c_int32, [c_void_p, c_int32, c_void_p, c_float, POINTER(c_float)]),
    'chroma_color_solids': (c_int32, [c_void_p, c_void_p, c_int32, c_int32, c_void_p, c_void_p, c_uint32]),
    'chroma_points_rotate_around_point': (c_int32, [c_void_p, c_int32, c_void_p, c_float, POINTER(c_float), POINTER(c_float)]),
    'chroma_comm_unique_id': (c_int32, [c_void_p]),
    'chroma_comm_init': (c_int32, [c_void_p, c_int32, c_int32, c_void_p]),
    'chroma_comm_destroy': (c_int32, [c_void_p]),
    'chroma_allreduce_hits': (c_int32, [c_void_p, c_void_p, c_void_p, c_uint32]),
    'chroma_allreduce_daq': (c_int32, [c_void_p, c_void_p, c_void_p, c_void_p, c_uint32]),
    'chroma_probe': (c_int32, [c_void_p, c_int32, c_uint64, c_void_p, c_void_p, c_void_p, c_uint32, c_float, c_float, c_void_p]),
    'chroma_bvh_build': (c_int32, [c_void_p, c_uint32, c_void_p, c_uint32, POINTER(c_float), c_float, c_int32,
                                   POINTER(c_void_p), POINTER(c_uint64), POINTER(c_uint32)]),
    'chroma_bvh_build_device': (c_int32, [c_void_p, c_void_p, c_uint32, c_void_p, c_uint32, POINTER(c_float), c_float, c_int32,
                                          POINTER(c_void_p), POINTER(c_uint64), POINTER(c_uint32)]),
    'chroma_hits_sort': (c_int32, [c_void_p, POINTER(PhotonArrays), c_void_p, c_uint64]),
    'chroma_photons_sort_direction': (c_int32, [c_void_p, POINTER(PhotonArrays), c_uint64]),
    'chroma_bvh_fetch': (c_int32, [c_void_p, c_void_p, c_void_p]),
    'chroma_bvh_data': (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_void_p)]),
    'chroma_bvh_free': (c_int32, [c_void_p]),
    'chroma_wide_build': (c_int32, [c_void_p, c_uint64, c_uint32, POINTER(c_void_p), POINTER(c_uint64), POINTER(c_uint64),
                                    POINTER(c_uint32)]),
    'chroma_wide_build_device': (c_int32, [c_void_p, c_void_p, c_uint64, c_uint32, POINTER(c_void_p), POINTER(c_uint64),
                                           POINTER(c_uint64), POINTER(c_uint32)]),
    'chroma_wide_data': (c_int32, [c_void_p, POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p), POINTER(c_void_p)]),
    'chroma_wide_free': (c_int32, [c_void_p]),
    'chroma_wide_validate': (c_int32, [c_void_p, c_uint64, c_void_p, c_uint32, c_void_p, c_uint64]),
    'chroma_dedupe_vertices': (c_int32, [c_void_p, c_uint64, c_void_p, c_uint64, c_void_p, POINTER(c_uint64)]),
    'chroma_propagate_stats_read': (c_int32, [c_void_p, POINTER(PropagateStats)]),
    'chroma_set_counting': (c_int32, [c_void_p, c_int32]),
    'chroma_set_walk': (c_int32, [c_void_p, c_int32]),
    'chroma_set_tail': (c_int32, [c_void_p, c_int32]),
    'chroma_set_packet': (c_int32, [c_void_p, c_int32]),
    'chroma_set_autosort': (c_int32, [c_void_p, c_int32]),
}

_lib = None
_variants = {}


class ChromaError(RuntimeError):
    pass


def _bind(path):
    if not os.path.exists(path):
        raise ChromaError(
            'libchroma_hip.so is not built (%s missing).  Build it with '
            '`python -c "import __graft_entry__ as g; g.build()"` or `make -C chroma_amd/csrc`.  '
            'There is no CPU fallback.' % path)
    # torch ships its own libamdhip64 under the same SONAME; if it is going to be used in this
    # process it has to be loaded first so that both sides share one HIP runtime.
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_GLOBAL)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)       # AttributeError = symbol missing: also loud
        fn.restype = restype
        fn.argtypes = argtypes
    return lib


def load(path=None):
    """Load libchroma_hip.so once; fail loudly when it is not there.  ``path`` names another build of
    the same sources (build_variants/, kernel A/B experiments and the shallow-stack test variant): it
    is loaded beside the product library and used by the contexts created with it."""
    global _lib
    if path is not None and os.path.abspath(path) != os.path.abspath(LIBRARY_PATH):
        key = os.path.abspath(path)
        if key not in _variants:
            _variants[key] = _bind(key)
        return _variants[key]
    if _lib is None:
        _lib = _bind(LIBRARY_PATH)
    return _lib


def check(rc, lib=None):
    if rc != 0:
        msg = (lib or load()).chroma_last_error()
        if not msg and lib is None:       # the call may have gone to a variant library (its own message buffer)
            for v in _variants.values():
                msg = msg or v.chroma_last_error()
        raise ChromaError('libchroma_hip error %d: %s' % (rc, msg.decode() if msg else '?'))


def ptr(arr):
    """c_void_p of a C-contiguous NumPy array (or None)."""
    if arr is None:
        return None
    return arr.ctypes.data_as(c_void_p)


def bvh_build(vertices, triangles, world_origin, world_scale, target_degree=3, ctx=None):
    """BVH build: on the host cores (``ctx`` None: chroma_bvh_build) or on the device of the chroma_amd.gpu context
    ``ctx`` (chroma_bvh_build_device) -- the same node array bit for bit.  Returns (nodes as structured uint4
    array, layer bounds)."""
    from chroma_amd.bvh.bvh import uint4
    lib = load() if ctx is None else ctx._lib
    vertices = np.ascontiguousarray(vertices, dtype=np.float32)
    triangles = np.ascontiguousarray(triangles, dtype=np.uint32)
    origin = (c_float * 3)(*[float(x) for x in world_origin])
    handle = c_void_p()
    nnodes = c_uint64()
    nlayers = c_uint32()
    if ctx is None:
        check(lib.chroma_bvh_build(ptr(vertices), len(vertices), ptr(triangles), len(triangles), origin,
                                   c_float(float(world_scale)), int(target_degree),
                                   ctypes.byref(handle), ctypes.byref(nnodes), ctypes.byref(nlayers)))
    else:
        check(lib.chroma_bvh_build_device(ctx.handle, ptr(vertices), len(vertices), ptr(triangles), len(triangles), origin,
                                          c_float(float(world_scale)), int(target_degree),
                                          ctypes.byref(handle), ctypes.byref(nnodes), ctypes.byref(nlayers)), lib)
    # view the builder's own buffer (no copy); it is released when the array is garbage-collected
    p_nodes, p_bounds = c_void_p(), c_void_p()
    check(lib.chroma_bvh_data(handle, ctypes.byref(p_nodes), ctypes.byref(p_bounds)))
    raw = (ctypes.c_uint32 * (4 * nnodes.value)).from_address(p_nodes.value)
    nodes = np.frombuffer(raw, dtype=uint4)
    bounds = np.array((ctypes.c_uint64 * (nlayers.value + 1)).from_address(p_bounds.value), dtype=np.uint64)
    import weakref
    weakref.finalize(raw, lib.chroma_bvh_free, handle)
    return nodes, bounds.astype(np.int64)


def wide_build(nodes, ntriangles, ctx=None):
    """The derived 8-wide traversal tree of a reference-format BVH (what chroma_geometry_create uploads).
    ``ctx`` None: built on the host cores (chroma_wide_build; topology by $CHROMA_TREE); a chroma_amd.gpu context:
    built by HIP kernels on its device (chroma_wide_build_device: the "levels" topology).  Returns a dict of
    arrays (views of the builder's buffers): ``wnodes`` [nwide][8][4] uint32, ``tri_to_record``, ``record_to_tri``, ``rank`` and ``depth``."""
    lib = load() if ctx is None else ctx._lib
    raw = np.ascontiguousarray(nodes).view(np.uint32).reshape(-1, 4)
    handle = c_void_p()
    nwide, nrec, depth = c_uint64(), c_uint64(), c_uint32()
    if ctx is None:
        rc = lib.chroma_wide_build(ptr(raw), len(raw), int(ntriangles), ctypes.byref(handle), ctypes.byref(nwide),
                                   ctypes.byref(nrec), ctypes.byref(depth))
        if rc != 0:
            raise ChromaError('chroma_wide_build failed (%d): malformed BVH' % rc)
    else:
        check(lib.chroma_wide_build_device(ctx.handle, ptr(raw), len(raw), int(ntriangles), ctypes.byref(handle),
                                           ctypes.byref(nwide), ctypes.byref(nrec), ctypes.byref(depth)))
    # views of the builder's own buffers (no copies: 4 GB of wide nodes at 170 M triangles); the handle is released when
    # the last of them is garbage-collected
    class _Owner(object):
        def __init__(self, free, h):
            self.free, self.h = free, h

        def __del__(self):
            self.free(self.h)
    owner = _Owner(lib.chroma_wide_free, handle)
    p = [c_void_p() for _ in range(4)]
    check(lib.chroma_wide_data(handle, *[ctypes.byref(x) for x in p]))

    def view(pp, n):
        if n == 0:
            return np.zeros(0, dtype=np.uint32)
        raw = (ctypes.c_uint32 * n).from_address(pp.value)
        raw._owner = owner
        return np.frombuffer(raw, dtype=np.uint32)
    return {'wnodes': view(p[0], 32 * nwide.value).reshape(-1, 8, 4),
            'tri_to_record': view(p[1], int(ntriangles)), 'record_to_tri': view(p[2], nrec.value),
            'rank': view(p[3], int(ntriangles)), 'depth': depth.value}


def wide_validate(wide, ntriangles):
    """True when the index checks chroma_geometry_create applies to a derived tree pass for ``wide``
    (a dict as returned by wide_build, possibly tampered with by a test)."""
    lib = load()
    wn = np.ascontiguousarray(wide['wnodes'], dtype=np.uint32)
    t2r = np.ascontiguousarray(wide['tri_to_record'], dtype=np.uint32)
    r2t = np.ascontiguousarray(wide['record_to_tri'], dtype=np.uint32)
    return lib.chroma_wide_validate(ptr(wn), wn.size // 32, ptr(t2r), int(ntriangles), ptr(r2t), len(r2t)) == 0


def dedupe_vertices(vertices, triangles):
    """Native Mesh.remove_duplicate_vertices.  ``triangles`` (a C-contiguous 32-bit integer array)
    is remapped IN PLACE; returns the unique vertices (a view of a buffer of the original size)."""
    lib = load()
    vertices = np.ascontiguousarray(vertices, dtype=np.float32)
    if not (triangles.flags['C_CONTIGUOUS'] and triangles.dtype.itemsize == 4 and triangles.dtype.kind in 'iu'):
        raise ValueError('triangles must be a C-contiguous int32/uint32 array')
    unique = np.empty_like(vertices)
    nunique = c_uint64()
    check(lib.chroma_dedupe_vertices(ptr(vertices), len(vertices), ptr(triangles), triangles.size, ptr(unique), ctypes.byref(nunique)))
    return unique[:nunique.value]
