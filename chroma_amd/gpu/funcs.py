"""``get_cu_module`` / ``GPUFuncs``: the reference's kernels by NAME, as callables over the C ABI.

The reference JIT-compiles ``chroma/cuda/*.cu`` with PyCUDA and looks kernels up by name
(chroma/gpu/tools.py:14-54); host code then calls e.g.
``gpu_funcs.propagate(np.int32(first), np.int32(n), input_queue[1:], output_queue, rng_states, pos, dir,
wavelengths, pol, t, flags, last_hit_triangles, weights, evidx, np.int32(nsteps), np.int32(use_weights),
np.int32(scatter_first), gpu_geometry.gpudata, block=(64, 1, 1), grid=(blocks, 1))``.  There is no JIT
here: ``get_cu_module(name)`` returns a table of the kernels that source file defines, and each entry
takes the reference kernel's POSITIONAL arguments (same order and meaning; ``block`` / ``grid`` are
accepted and ignored -- launch shapes are the library's) and calls the entry point of
include/chroma_hip.h that replaces it.  Device arrays are chroma_amd.gpu.GPUArray objects, the
``Geometry*`` / ``Detector*`` argument is ``GPUGeometry.gpudata`` (the geometry handle), the
``curandState*`` argument is what ``get_rng_states`` returned.

The per-photon Philox draw counters that replace curandState live beside the photon arrays; for arrays that
do not come from a GPUPhotons object (which owns ``rng_counters``) they are created on first use and kept
per position array.
"""
import ctypes
import weakref

import numpy as np

from chroma_amd import _lib
from chroma_amd.gpu.tools import GPUArray, RNGStates, get_context, zeros

_counters = {}      # pos.ptr -> (rng_counters GPUArray, photon id base)


def _scalar(x):
    return int(np.asarray(x).reshape(-1)[0]) if not isinstance(x, (int, float)) else x


def _photons(pos, dir, wavelengths, pol, t, flags, last_hit_triangles, weights, evidx, rng_counters=None):
    s = _lib.PhotonArrays()
    s.pos, s.dir, s.pol, s.wavelengths, s.t = pos.ptr, dir.ptr, pol.ptr, wavelengths.ptr, t.ptr
    s.flags, s.last_hit_triangles, s.weights, s.evidx = flags.ptr, last_hit_triangles.ptr, weights.ptr, evidx.ptr
    s.rng_counters = rng_counters.ptr if rng_counters is not None else None
    return s


def _stream_of(pos, rng_states):
    """(rng_counters, chroma_rng) of the photon set whose position array is ``pos``."""
    key = pos.ptr
    if key not in _counters:
        base = rng_states.reserve(len(pos)) if isinstance(rng_states, RNGStates) else 0
        _counters[key] = (zeros(len(pos), np.uint32, pos.ctx), base)
        weakref.finalize(pos._alloc if hasattr(pos, '_alloc') else pos, _counters.pop, key, None)
    counters, base = _counters[key]
    seed = rng_states.seed if isinstance(rng_states, RNGStates) else int(getattr(rng_states, 'seed', 0))
    if isinstance(rng_states, _lib.Rng):
        return counters, rng_states
    return counters, _lib.Rng(seed, base)


# ---- chroma/cuda/propagate.cu ---------------------------------------------------------------------------------
def _photon_duplicate(first_photon, nthreads, pos, dir, wavelengths, pol, t, flags, last_hit_triangles, weights, evidx,
                      copies, stride, block=None, grid=None):
    ctx = pos.ctx
    s = _photons(pos, dir, wavelengths, pol, t, flags, last_hit_triangles, weights, evidx)
    _lib.check(ctx._lib.chroma_photon_duplicate(ctx.handle, _scalar(first_photon), _scalar(nthreads), ctypes.byref(s),
                                                _scalar(copies), _scalar(stride)))


def _count_photons(first_photon, nthreads, target_flag, index_counter, histories, block=None, grid=None):
    ctx = histories.ctx
    n = ctypes.c_uint32()
    _lib.check(ctx._lib.chroma_count_photons(ctx.handle, _scalar(first_photon), _scalar(nthreads), _scalar(target_flag),
                                             histories.ptr, ctypes.byref(n)))
    index_counter.set(index_counter.get() + np.uint32(n.value))          # the kernel ADDS to the counter (propagate.cu:74-78)


def _copy_photons(first_photon, nthreads, target_flag, index_counter, *arrays, **kw):
    src, dst = arrays[:9], arrays[9:18]
    ctx = src[0].ctx
    n = ctypes.c_uint32()
    a, b = _photons(*src), _photons(*dst)
    _lib.check(ctx._lib.chroma_copy_photons(ctx.handle, _scalar(first_photon), _scalar(nthreads), _scalar(target_flag),
                                            ctypes.byref(a), ctypes.byref(b), ctypes.byref(n)))
    index_counter.set(index_counter.get() + np.uint32(n.value))


def _copy_photon_queue(first_photon, nthreads, queue, *arrays, **kw):
    src, dst = arrays[:9], arrays[9:18]
    ctx = src[0].ctx
    a, b = _photons(*src), _photons(*dst)
    _lib.check(ctx._lib.chroma_copy_photon_queue(ctx.handle, _scalar(first_photon), _scalar(nthreads), queue.ptr,
                                                 ctypes.byref(a), ctypes.byref(b)))


def _count_photon_hits(first_photon, nphotons, detection_state, histories, solid_map, last_hit_triangles, detector,
                       index_counter, block=None, grid=None):
    ctx = histories.ctx
    s = _lib.PhotonArrays()
    for name in ('pos', 'dir', 'pol', 'wavelengths', 't', 'weights', 'evidx'):       # (only flags and last hits are read)
        setattr(s, name, histories.ptr)
    s.flags, s.last_hit_triangles = histories.ptr, last_hit_triangles.ptr
    n = ctypes.c_uint32()
    _lib.check(ctx._lib.chroma_count_photon_hits(ctx.handle, detector, _scalar(first_photon), _scalar(nphotons),
                                                 _scalar(detection_state), ctypes.byref(s), ctypes.byref(n)))
    index_counter.set(index_counter.get() + np.uint32(n.value))


def _copy_photon_hits(first_photon, nphotons, detection_state, solid_map, detector, index_counter, *arrays, **kw):
    src, dst, channels = arrays[:9], arrays[9:18], arrays[18]
    ctx = src[0].ctx
    a, b = _photons(*src), _photons(*dst)
    n = ctypes.c_uint32()
    _lib.check(ctx._lib.chroma_copy_photon_hits(ctx.handle, detector, _scalar(first_photon), _scalar(nphotons),
                                                _scalar(detection_state), ctypes.byref(a), ctypes.byref(b), channels.ptr,
                                                ctypes.byref(n)))
    index_counter.set(index_counter.get() + np.uint32(n.value))


def _propagate(first_photon, nthreads, input_queue, output_queue, rng_states, pos, dir, wavelengths, pol, t, flags,
               last_hit_triangles, weights, evidx, max_steps, use_weights, scatter_first, geometry, block=None, grid=None):
    ctx = pos.ctx
    counters, rng = _stream_of(pos, rng_states)
    s = _photons(pos, dir, wavelengths, pol, t, flags, last_hit_triangles, weights, evidx, counters)
    _lib.check(ctx._lib.chroma_propagate_step(ctx.handle, geometry, _scalar(first_photon), _scalar(nthreads),
                                              input_queue.ptr if input_queue is not None else None, output_queue.ptr,
                                              rng, ctypes.byref(s), _scalar(max_steps), _scalar(use_weights),
                                              _scalar(scatter_first)))


# ---- chroma/cuda/mesh.h ------------------------------------------------------------------------------------------
def _distance_to_mesh(nthreads, origin, direction, geometry, distance, block=None, grid=None):
    ctx = origin.ctx
    _lib.check(ctx._lib.chroma_distance_to_mesh(ctx.handle, geometry, _scalar(nthreads), origin.ptr, direction.ptr,
                                                distance.ptr, None))


def _color_solids(first_triangle, nthreads, solid_id_map, solid_hit, solid_colors, geometry, block=None, grid=None):
    # (solid_id_map is the geometry's own array; the kernel reads it from the handle)
    ctx = solid_colors.ctx
    _lib.check(ctx._lib.chroma_color_solids(ctx.handle, geometry, _scalar(first_triangle), _scalar(nthreads), solid_hit.ptr,
                                            solid_colors.ptr, min(len(solid_hit), len(solid_colors))))


_MODULES = {
    'propagate.cu': {'photon_duplicate': _photon_duplicate, 'count_photons': _count_photons, 'copy_photons': _copy_photons,
                     'copy_photon_queue': _copy_photon_queue, 'count_photon_hits': _count_photon_hits,
                     'copy_photon_hits': _copy_photon_hits, 'propagate': _propagate},
    'mesh.h': {'distance_to_mesh': _distance_to_mesh, 'color_solids': _color_solids},
}


class KernelModule(object):
    """What ``get_cu_module`` returns: ``get_function(name)`` like a pycuda SourceModule."""

    def __init__(self, name):
        if name not in _MODULES:
            raise KeyError('no kernel table for %r: this engine provides %s (the DAQ kernels of daq.cu are reached '
                           'through chroma.gpu.GPUDaq, the BVH kernels of bvh.cu through chroma.bvh)' % (name, sorted(_MODULES)))
        self.name = name
        self._table = _MODULES[name]

    def get_function(self, name):
        try:
            return self._table[name]
        except KeyError:
            raise AttributeError('%s has no kernel %r (has: %s)' % (self.name, name, sorted(self._table)))


def get_cu_module(name, options=None, include_source_directory=True):
    """The kernels of the reference's ``chroma/cuda/<name>`` as a name -> callable table (chroma/gpu/tools.py:14-32).
    ``options`` (nvcc flags) is accepted and ignored: nothing is compiled at run time."""
    if options is not None and not isinstance(options, tuple):
        raise TypeError('`options` must be a tuple.')
    get_context()
    return KernelModule(name)


class GPUFuncs(object):
    """Simple container class for GPU functions as attributes (chroma/gpu/tools.py:42-54)."""

    def __init__(self, module):
        self.module = module
        self.funcs = {}

    def __getattr__(self, name):
        try:
            return self.funcs[name]
        except KeyError:
            f = self.module.get_function(name)
            self.funcs[name] = f
            return f
