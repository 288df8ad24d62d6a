"""GPUGeometry: resample the optics tables, pack the mesh + BVH and upload them.

Reference: chroma/gpu/geometry.py:14-259.  The reference builds pointer-linked
Material / Surface / DichroicProps structs byte by byte; here the same numbers go into
flat row-major tables (``pack_geometry`` -> chroma_geometry_desc, include/chroma_hip.h)
and the library lays them out in HBM.  ``pack_geometry`` needs no GPU and is also what
the CPU oracle consumes in the tests.
"""
import ctypes

import numpy as np

from chroma_amd import _lib
from chroma_amd.geometry import standard_wavelengths
from chroma_amd.gpu.tools import GPUArray, vec, get_context, format_array, format_size
from chroma_amd.log import logger


def _uniform_step(values, what):
    d = np.unique(np.diff(values))
    if len(d) != 1:
        raise ValueError('%s must be equally spaced apart.' % what)
    return d.item()


def interp_material_property(wavelengths, prop):
    """Linear resampling onto the common grid (chroma/gpu/geometry.py:41-45).  Linear on
    purpose: interpolated surface probabilities still sum to one."""
    prop = np.asarray(prop)
    return np.interp(wavelengths, prop[:, 0], prop[:, 1]).astype(np.float32)


class PackedGeometry(object):
    """Flat host arrays of one geometry + the ctypes descriptor pointing at them."""

    def __init__(self):
        self.arrays = {}
        self.desc = _lib.GeometryDesc()

    def put(self, name, array, dtype):
        if isinstance(array, np.ndarray) and array.dtype == np.int32 and dtype == np.uint32 and array.flags['C_CONTIGUOUS']:
            array = array.view(np.uint32)             # (the same bits: no 2-GB copy of a detector's triangle indices)
        a = np.ascontiguousarray(array, dtype=dtype)
        self.arrays[name] = a
        setattr(self.desc, name, a.ctypes.data if a.size else None)
        return a

    def attach_wide_tree(self, wide=None, ctx='auto'):
        """Carry the derived 8-wide traversal tree along (``wide``: what chroma_amd._lib.wide_build returned
        for these nodes; None: build it now), so that chroma_geometry_create uploads it instead of deriving
        it again -- once per cache file, or once per node when several processes drive one GPU each.
        Built on the device of ``ctx`` ('auto': the current chroma_amd.gpu context if there is one and the
        topology is the default; None: on the host cores) -- the same tree either way."""
        if wide is None:
            if ctx == 'auto':
                import os
                from chroma_amd.gpu import tools as gtools
                default = os.environ.get('CHROMA_TREE', 'levels') == 'levels' and os.environ.get('CHROMA_WIDE_BUILD') != 'host'
                ctx = gtools._current if default else None
            try:
                wide = _lib.wide_build(self.arrays['nodes'], self.desc.ntriangles, ctx=ctx)
            except _lib.ChromaError as exc:
                if ctx is None or 'out of memory' not in str(exc).lower():
                    raise
                # (the builder's scratch, ~150 bytes per triangle, did not fit beside what is on the card: the host cores
                #  build the same tree, bit for bit)
                logger.warning('no room on the device for the tree builder: building the wide tree on the host cores')
                ctx.pool_trim()
                wide = _lib.wide_build(self.arrays['nodes'], self.desc.ntriangles)
        self.put('wide_nodes', wide['wnodes'], np.uint32)
        self.put('wide_tri_to_record', wide['tri_to_record'], np.uint32)
        self.put('wide_record_to_tri', wide['record_to_tri'], np.uint32)
        self.put('wide_rank', wide['rank'], np.uint32)
        self.desc.nwide = self.arrays['wide_nodes'].size // 32
        self.desc.nrecords = len(self.arrays['wide_record_to_tri'])
        return self

    # ---- as a directory of .npy files (memory-mapped on load: processes of one node share the pages) ----
    def save(self, path):
        import json
        import os
        os.makedirs(path, exist_ok=True)
        scalars = {}
        for name, ctype in self.desc._fields_:
            if ctype is ctypes.c_void_p:
                continue
            v = getattr(self.desc, name)
            scalars[name] = [float(x) for x in v] if hasattr(v, '__len__') else v
        for name, a in self.arrays.items():
            np.save(os.path.join(path, name + '.npy'), a)
        with open(os.path.join(path, 'desc.json.tmp'), 'w') as f:
            json.dump({'scalars': scalars, 'arrays': sorted(self.arrays)}, f)
        os.replace(os.path.join(path, 'desc.json.tmp'), os.path.join(path, 'desc.json'))      # (the last file written)
        return path

    @classmethod
    def load(cls, path, mmap=True):
        import json
        import os
        with open(os.path.join(path, 'desc.json')) as f:
            meta = json.load(f)
        pk = cls()
        for name, v in meta['scalars'].items():
            if isinstance(v, list):
                for k, x in enumerate(v):
                    getattr(pk.desc, name)[k] = x
            else:
                setattr(pk.desc, name, v)
        for name in meta['arrays']:
            a = np.load(os.path.join(path, name + '.npy'), mmap_mode='r' if mmap else None)
            pk.arrays[name] = a
            setattr(pk.desc, name, a.ctypes.data if a.size else None)
        return pk


def pack_geometry(geometry, wavelengths=None, times=None):
    """Everything chroma_geometry_create needs, as host arrays.  ``geometry`` must be
    flattened and carry a ``bvh``."""
    if wavelengths is None:
        wavelengths = standard_wavelengths
    wavelengths = np.asarray(wavelengths)
    wavelength_step = _uniform_step(wavelengths, 'wavelengths')
    if times is None:
        time_step = 0.05
        times = np.arange(0, 1000, time_step)
    else:
        times = np.asarray(times)
        time_step = _uniform_step(times, 'times')
    if geometry.bvh is None:
        raise ValueError('geometry has no BVH: use chroma_amd.loader.build_bvh or make_recursive_grid_bvh')

    pk = PackedGeometry()
    d = pk.desc
    mesh = geometry.mesh
    pk.put('vertices', mesh.vertices, np.float32)
    pk.put('triangles', mesh.triangles, np.uint32)
    d.nvertices, d.ntriangles = len(mesh.vertices), len(mesh.triangles)
    # 8-bit two's-complement indices, -1 = no surface (chroma/gpu/geometry.py:203-205)
    def byte_of(index, shift):          # low byte of a (possibly negative) index, in place in one 32-bit temporary
        b = np.asarray(index).astype(np.uint32)
        b &= np.uint32(0xff)
        b <<= np.uint32(shift)
        return b
    codes = byte_of(geometry.inner_material_index, 24)
    codes |= byte_of(geometry.outer_material_index, 16)
    codes |= byte_of(geometry.surface_index, 8)
    pk.put('material_codes', codes, np.uint32)
    pk.put('solid_id_map', geometry.solid_id, np.uint32)
    pk.put('colors', geometry.colors, np.uint32)

    nodes = geometry.bvh.nodes
    pk.put('nodes', nodes.view(np.uint32).reshape(-1, 4), np.uint32)
    d.nnodes = len(nodes)
    for k in range(3):
        d.world_origin[k] = float(geometry.bvh.world_coords.world_origin[k])
    d.world_scale = float(geometry.bvh.world_coords.world_scale)

    d.wavelength_n = len(wavelengths)
    d.wavelength_start = float(wavelengths[0])
    d.wavelength_step = float(wavelength_step)
    d.time_n = len(times)
    d.time_start = float(times[0])
    d.time_step = float(time_step)

    # ---- materials (chroma/gpu/geometry.py:47-103)
    mats = geometry.unique_materials
    if len(mats) > 127:
        raise ValueError('at most 127 materials (8-bit signed indices)')
    for m in mats:
        if m is None:
            raise Exception('one or more triangles is missing a material.')
    d.nmaterials = len(mats)
    wl = wavelengths
    pk.put('mat_refractive_index', [interp_material_property(wl, m.refractive_index) for m in mats], np.float32)
    pk.put('mat_absorption_length', [interp_material_property(wl, m.absorption_length) for m in mats], np.float32)
    pk.put('mat_scattering_length', [interp_material_property(wl, m.scattering_length) for m in mats], np.float32)
    num_comp, comp_offset = [], []
    prob, wcdf, tcdf, cabs = [], [], [], []
    for m in mats:
        n = len(m.comp_reemission_prob)
        assert n == len(m.comp_reemission_wvl_cdf) == len(m.comp_reemission_time_cdf) == len(m.comp_absorption_length), \
            'component arrays must be same length'
        comp_offset.append(len(prob))
        num_comp.append(n)
        prob += [interp_material_property(wl, c) for c in m.comp_reemission_prob]
        wcdf += [interp_material_property(wl, c) for c in m.comp_reemission_wvl_cdf]
        tcdf += [interp_material_property(times, c) for c in m.comp_reemission_time_cdf]
        cabs += [interp_material_property(wl, c) for c in m.comp_absorption_length]
    pk.put('mat_num_comp', num_comp, np.uint32)
    pk.put('mat_comp_offset', comp_offset, np.uint32)
    d.ncomp_total = len(prob)
    pk.put('comp_reemission_prob', np.array(prob, dtype=np.float32).reshape(len(prob), len(wl)), np.float32)
    pk.put('comp_reemission_wvl_cdf', np.array(wcdf, dtype=np.float32).reshape(len(prob), len(wl)), np.float32)
    pk.put('comp_absorption_length', np.array(cabs, dtype=np.float32).reshape(len(prob), len(wl)), np.float32)
    pk.put('comp_reemission_time_cdf', np.array(tcdf, dtype=np.float32).reshape(len(prob), len(times)), np.float32)

    # ---- surfaces (chroma/gpu/geometry.py:108-189); None keeps its slot with zero tables
    surfs = geometry.unique_surfaces
    if len(surfs) > 127:
        raise ValueError('at most 127 surfaces (8-bit signed indices)')
    d.nsurfaces = len(surfs)
    zero_row = np.zeros(len(wl), dtype=np.float32)

    def surf_table(attr):
        return [zero_row if s is None else interp_material_property(wl, getattr(s, attr)) for s in surfs]
    for attr in ('detect', 'absorb', 'reemit', 'reflect_diffuse', 'reflect_specular', 'eta', 'k', 'reemission_cdf'):
        pk.put('surf_' + attr, np.array(surf_table(attr), dtype=np.float32).reshape(len(surfs), len(wl)), np.float32)
    pk.put('surf_model', [0 if s is None else int(s.model) for s in surfs], np.uint32)
    pk.put('surf_transmissive', [0 if s is None else int(s.transmissive) for s in surfs], np.uint32)
    pk.put('surf_thickness', [0.0 if s is None else float(s.thickness) for s in surfs], np.float32)
    dich_index, nangles, offsets, angles, refl, trans = [], [], [], [], [], []
    for s in surfs:
        props = None if s is None else s.dichroic_props
        if not props:
            dich_index.append(-1)
            continue
        dich_index.append(len(nangles))
        offsets.append(len(angles))
        nangles.append(len(props.angles))
        angles += [float(a) for a in props.angles]
        refl += [interp_material_property(wl, t) for t in props.dichroic_reflect]
        trans += [interp_material_property(wl, t) for t in props.dichroic_transmit]
    pk.put('surf_dichroic_index', dich_index, np.int32)
    d.ndichroic = len(nangles)
    d.ndichroic_angles_total = len(angles)
    pk.put('dichroic_nangles', nangles, np.uint32)
    pk.put('dichroic_offset', offsets, np.uint32)
    pk.put('dichroic_angles', angles, np.float32)
    pk.put('dichroic_reflect', np.array(refl, dtype=np.float32).reshape(len(angles), len(wl)), np.float32)
    pk.put('dichroic_transmit', np.array(trans, dtype=np.float32).reshape(len(angles), len(wl)), np.float32)

    # ---- detector part (chroma/gpu/detector.py:17-20)
    if hasattr(geometry, 'num_channels'):
        s2c = pk.put('solid_id_to_channel_index', geometry.solid_id_to_channel_index, np.int32)
        d.nsolids = len(s2c)
        d.nchannels = geometry.num_channels()
    else:
        d.nsolids = 0
        d.nchannels = 0
    return pk


class GPUGeometry(object):
    def __init__(self, geometry, wavelengths=None, times=None, print_usage=False, min_free_gpu_mem=300e6, packed=None):
        # min_free_gpu_mem controlled the reference's spill of BVH nodes to host memory
        # (chroma/gpu/geometry.py:211-230); with 288 GB of HBM everything stays on the device.
        # ``packed``: a pack_geometry() result the caller already holds (and keeps): it is uploaded as
        # it is and left whole instead of being packed a second time.
        self.ctx = get_context()
        keep_host_arrays = packed is not None
        self.packed = packed if packed is not None else pack_geometry(geometry, wavelengths=wavelengths, times=times)
        handle = ctypes.c_void_p()
        _lib.check(self.ctx._lib.chroma_geometry_create(self.ctx.handle, ctypes.byref(self.packed.desc), ctypes.byref(handle)))
        self.handle = handle
        self.gpudata = handle      # what kernels take in place of the reference's Geometry*
        self.geometry = geometry
        self.world_origin = vec.make_float3(*geometry.bvh.world_coords.world_origin)
        self.world_scale = np.float32(geometry.bvh.world_coords.world_scale)

        self.vertices = self._device_array('vertices', vec.float3)
        self.triangles = self._device_array('triangles', vec.uint3)
        self.nodes = self._device_array('nodes', vec.uint4)
        self.extra_nodes = GPUArray.from_pointer(self.nodes.ptr, 0, vec.uint4, self, ctx=self.ctx)
        self.material_codes = self._device_array('material_codes', np.uint32)
        self.colors = self._device_array('colors', np.uint32)
        self.solid_id_map = self._device_array('solid_id_map', np.uint32)
        # host tables are not needed once uploaded, except the small ones tests look at
        # (and the descriptor must not keep pointing at arrays that are gone)
        for big in ('vertices', 'triangles', 'nodes', 'material_codes', 'solid_id_map', 'colors'):
            if keep_host_arrays:
                break
            if self.packed.arrays.pop(big, None) is not None:
                setattr(self.packed.desc, big, None)
        if print_usage:
            self.print_device_usage()
        logger.info(self.device_usage_str())

    @classmethod
    def from_packed(cls, packed):
        """Upload a PackedGeometry without the Geometry object it was made from (a rank that loaded the
        arrays another process of its node packed): everything the propagate / hit path needs."""
        self = cls.__new__(cls)
        self.ctx = get_context()
        self.packed = packed
        handle = ctypes.c_void_p()
        _lib.check(self.ctx._lib.chroma_geometry_create(self.ctx.handle, ctypes.byref(packed.desc), ctypes.byref(handle)))
        self.handle = handle
        self.gpudata = handle
        self.geometry = None
        self.world_origin = vec.make_float3(*[packed.desc.world_origin[k] for k in range(3)])
        self.world_scale = np.float32(packed.desc.world_scale)
        self.vertices = self._device_array('vertices', vec.float3)
        self.triangles = self._device_array('triangles', vec.uint3)
        self.nodes = self._device_array('nodes', vec.uint4)
        self.extra_nodes = GPUArray.from_pointer(self.nodes.ptr, 0, vec.uint4, self, ctx=self.ctx)
        self.material_codes = self._device_array('material_codes', np.uint32)
        self.colors = self._device_array('colors', np.uint32)
        self.solid_id_map = self._device_array('solid_id_map', np.uint32)
        if packed.desc.nsolids:
            self.solid_id_to_channel_index_gpu = self._device_array('solid_id_to_channel_index', np.int32)
        self.nchannels = int(packed.desc.nchannels)
        self.detector_gpu = handle
        return self

    def _device_array(self, name, dtype):
        p, nbytes = ctypes.c_void_p(), ctypes.c_size_t()
        _lib.check(self.ctx._lib.chroma_geometry_device_ptr(self.handle, name.encode(), ctypes.byref(p), ctypes.byref(nbytes)))
        return GPUArray.from_pointer(p.value or 0, nbytes.value // np.dtype(dtype).itemsize, dtype, self, ctx=self.ctx)

    def stack_need(self):
        n = ctypes.c_uint32()
        _lib.check(self.ctx._lib.chroma_geometry_stack_need(self.handle, ctypes.byref(n)))
        return n.value

    def device_usage_str(self):
        s = 'device usage:\n' + '-' * 10 + '\n'
        s += format_array('nodes', self.nodes) + '\n'
        s += '%-15s %6s %6s' % ('total', '', format_size(self.nodes.nbytes)) + '\n' + '-' * 10 + '\n'
        free, total = self.ctx.mem_get_info()
        s += '%-15s %6s %6s' % ('device total', '', format_size(total)) + '\n'
        s += '%-15s %6s %6s' % ('device used', '', format_size(total - free)) + '\n'
        s += '%-15s %6s %6s' % ('device free', '', format_size(free)) + '\n'
        return s

    def print_device_usage(self):
        print(self.device_usage_str())
        print()

    def color_solids(self, solid_hit, colors, nblocks_per_thread=64, max_blocks=1024):
        """The triangles of every solid marked in ``solid_hit`` (one bool per solid) take that solid's entry of ``colors``
        (chroma/gpu/geometry.py:283-298, kernel color_solids of chroma/cuda/mesh.h:153-166); ``reset_colors`` undoes it.
        The launch-shape arguments are accepted and ignored."""
        solid_hit = np.ascontiguousarray(np.asarray(solid_hit, dtype=bool)).view(np.uint8)
        colors = np.ascontiguousarray(colors, dtype=np.uint32)
        if len(solid_hit) != len(colors):
            raise ValueError('solid_hit and colors must have one entry per solid (%d, %d)' % (len(solid_hit), len(colors)))
        hit_gpu = GPUArray(len(solid_hit), np.uint8, self.ctx).set(solid_hit)
        colors_gpu = GPUArray(len(colors), np.uint32, self.ctx).set(colors)
        _lib.check(self.ctx._lib.chroma_color_solids(self.ctx.handle, self.handle, 0, self.triangles.size, hit_gpu.ptr, colors_gpu.ptr,
                                                     len(colors)))
        self.ctx.synchronize()          # (the two small arrays go out of scope here)

    def reset_colors(self):
        self.colors.set(self.geometry.colors.astype(np.uint32))

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.ctx._lib.chroma_geometry_destroy(self.handle)
                self.handle = None
        except Exception:
            pass
