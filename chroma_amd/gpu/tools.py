"""Device context, device arrays and launch helpers over the C ABI.

This module takes the place of chroma/gpu/tools.py (PyCUDA): ``create_cuda_context``
(:121-142), ``get_rng_states`` (:75-84), ``chunk_iterator`` (:98-119), ``to_float3`` /
``to_uint3`` (:86-96) keep their names and meaning; ``GPUArray`` is the small part of
``pycuda.gpuarray.GPUArray`` the propagate path uses (get / set / fill / slicing / size).
"""
import ctypes
import os
import threading
import weakref

import numpy as np

from chroma_amd import _lib

# kept for signature compatibility; there is no JIT compiler behind this engine
cuda_options = ('--use_fast_math',)


class vec(object):
    """Vector dtypes with the field names pycuda.gpuarray.vec uses."""
    float3 = np.dtype([('x', np.float32), ('y', np.float32), ('z', np.float32)])
    uint3 = np.dtype([('x', np.uint32), ('y', np.uint32), ('z', np.uint32)])
    uint4 = np.dtype([('x', np.uint32), ('y', np.uint32), ('z', np.uint32), ('w', np.uint32)])

    @staticmethod
    def make_float3(x, y, z):
        return np.array((x, y, z), dtype=vec.float3)


def to_float3(arr):
    """(N,3) array -> (N,) float3 array."""
    arr = np.ascontiguousarray(arr, dtype=np.float32)
    return arr.view(vec.float3)[:, 0]


def to_uint3(arr):
    """(N,3) array -> (N,) uint3 array."""
    arr = np.ascontiguousarray(arr, dtype=np.uint32)
    return arr.view(vec.uint3)[:, 0]


# ---- context ----------------------------------------------------------------------------------
_current = None
_bound = threading.local()      # a context bound to THIS thread (Context.bound()) goes before the process-wide current one


class Context(object):
    """One HIP device + stream (chroma_ctx).  ``pop``/``push`` exist for API compatibility:
    a HIP context is not a stack object, so pop() only synchronises and makes the context
    no longer current for this module."""

    def __init__(self, device_id=None, library=None):
        lib = _lib.load(library)
        handle = ctypes.c_void_p()
        _lib.check(lib.chroma_init(-1 if device_id is None else int(device_id), ctypes.byref(handle)))
        self.handle = handle
        self.device_id = 0 if device_id is None else int(device_id)
        self._lib = lib

    def synchronize(self):
        _lib.check(self._lib.chroma_synchronize(self.handle))

    def push(self):
        global _current
        _current = self

    def pop(self):
        global _current
        self.synchronize()
        if _current is self:
            _current = None

    def bound(self):
        """``with ctx.bound():`` -- this context is what get_context() returns IN THIS THREAD for the duration of the block,
        whatever the process-wide current context is: several host threads, each driving its own context on the one GPU
        (Simulation(lanes=K)), without touching each other's idea of "the current context"."""
        return _Bound(self)

    def detach(self):
        self.pop()

    def shutdown(self):
        """pop() and give the device side of this context back (chroma_shutdown: stream, pool, staging buffers).  The
        context cannot be used afterwards; device arrays made on it must be gone by then."""
        if self.handle is not None and self.handle.value:
            self.pop()
            _lib.check(self._lib.chroma_shutdown(self.handle))
            self.handle = ctypes.c_void_p()

    def mem_get_info(self):
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        _lib.check(self._lib.chroma_mem_info(self.handle, ctypes.byref(free), ctypes.byref(total)))
        return free.value, total.value

    def device_name(self):
        buf = ctypes.create_string_buffer(256)
        _lib.check(self._lib.chroma_device_name(self.handle, buf, 256))
        return buf.value.decode()

    def set_counting(self, enabled):
        _lib.check(self._lib.chroma_set_counting(self.handle, 1 if enabled else 0))

    WALKS = {'reference': 0, 'wide': 1, 'coop': 2, 'quad': 3, 'pair': 4, 'literal': 5, 'literal_lane': 6}

    def set_walk(self, mode):
        """How the per-step ray cast walks.  'quad' (default), 'pair', 'coop', 'wide': the fast walks over the derived
        8-wide tree -- equal to each other, and to the reference wherever the winning hit lies inside its triangle's
        leaf box (every geometrically real hit).  'literal' (alias 'exact'): chroma/cuda/mesh.h:42-118 as it stands for
        every ray -- the reference's answer on EVERY ray, several times slower.  'reference': the reference's tree and
        order with postponed triangle tests (a cross-check).  See include/chroma_hip.h, chroma_set_walk."""
        mode = 'literal' if mode == 'exact' else mode
        _lib.check(self._lib.chroma_set_walk(self.handle, self.WALKS[mode]))
        self._walk = mode

    @property
    def walk(self):
        """The current walk mode (the library's default, or CHROMA_WALK, until set_walk is called)."""
        w = getattr(self, '_walk', None)
        if w is None:
            w = os.environ.get('CHROMA_WALK', 'quad')
            w = 'literal' if w == 'exact' else w
            self._walk = w = w if w in self.WALKS else 'quad'
        return w

    def set_packet(self, mode):
        """'off' (default), 'on' or 'auto': whether the FIRST step of a propagate call goes to the packet ray cast (64
        rays per wavefront as one packet: same results; measured no faster than the default walk even for a
        direction-sorted bomb, much slower for unrelated rays: an opt-in experiment).  'auto' decides per call from the
        photons themselves."""
        _lib.check(self._lib.chroma_set_packet(self.handle, {'off': 0, 'on': 1, 'auto': 2}[mode]))

    def set_autosort(self, mode):
        """'off' (default), 'auto' or 'on': whether a large propagate call takes its photons up in direction order by
        itself (an index sort on the device; same results photon by photon).  'auto' looks at a sample of the input: one
        origin and scattered directions are ordered, coherent photons and photons from many places are not.  Measured
        SLOWER than taking a bomb as it comes (the gather through the order costs more than the coherent first launches
        win): an opt-in experiment -- sort_by_direction() before the clock is what pays, as in chroma/benchmark.py:80-82."""
        _lib.check(self._lib.chroma_set_autosort(self.handle, {'off': 0, 'on': 1, 'auto': 2}[mode]))

    def set_tail(self, mode):
        """'coop' (default), 'split' or 'fused': how propagate() finishes -- or, with 'fused', runs -- a batch."""
        _lib.check(self._lib.chroma_set_tail(self.handle, {'coop': 0, 'split': 1, 'fused': 2}[mode]))

    def pool_stats(self):
        """(bytes parked in the device-memory pool, allocations served from it, allocations that went to hipMalloc)."""
        a, b, c = ctypes.c_uint64(), ctypes.c_uint64(), ctypes.c_uint64()
        _lib.check(self._lib.chroma_pool_stats(self.handle, ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)))
        return a.value, b.value, c.value

    def pool_trim(self):
        _lib.check(self._lib.chroma_pool_trim(self.handle))

    def read_stats(self):
        stats = _lib.PropagateStats()
        _lib.check(self._lib.chroma_propagate_stats_read(self.handle, ctypes.byref(stats)))
        return stats.as_dict()


class _Bound(object):
    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        self.previous = getattr(_bound, 'ctx', None)
        _bound.ctx = self.ctx
        return self.ctx

    def __exit__(self, *exc):
        _bound.ctx = self.previous
        return False


def create_cuda_context(device_id=None, library=None):
    """Initialise the device and return the (now current) context.  ``library``: path of another build
    of libchroma_hip.so to run this context's calls through (tests, A/B experiments)."""
    ctx = Context(device_id, library)
    ctx.push()
    return ctx


def get_context():
    """The current context; created on device 0 on first use."""
    global _current
    ctx = getattr(_bound, 'ctx', None)
    if ctx is not None:
        return ctx
    if _current is None:
        create_cuda_context(None)
    return _current


def device_count():
    n = ctypes.c_int32()
    lib = _lib.load()
    lib.chroma_device_count(ctypes.byref(n))
    return n.value


# ---- device arrays ------------------------------------------------------------------------------
class _Allocation(object):
    """Owns one chroma_malloc'd block; freed when the last array/view drops it."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        p = ctypes.c_void_p()
        _lib.check(ctx._lib.chroma_malloc(ctx.handle, nbytes, ctypes.byref(p)))
        self.ptr = p.value
        self._finalizer = weakref.finalize(self, _free_block, ctx._lib, ctx.handle, self.ptr)

    def free(self):
        self._finalizer()


def _free_block(lib, ctx_handle, ptr):
    try:
        lib.chroma_free(ctx_handle, ctypes.c_void_p(ptr))
    except Exception:   # interpreter shutdown
        pass


class GPUArray(object):
    """1-D device array (or a view of one).  ``gpudata`` is the raw device pointer."""

    def __init__(self, shape, dtype, ctx=None, _base=None, _ptr=None):
        if isinstance(shape, (tuple, list)):
            size = int(np.prod(shape)) if len(shape) else 1
        else:
            size = int(shape)
        self.dtype = np.dtype(dtype)
        self.size = size
        self.shape = (size,)
        self.ctx = ctx if ctx is not None else get_context()
        if _base is None:
            self._alloc = _Allocation(self.ctx, max(size * self.dtype.itemsize, 4))
            self.ptr = self._alloc.ptr
        else:
            self._alloc = _base      # keeps the owner alive
            self.ptr = _ptr

    @classmethod
    def from_pointer(cls, ptr, size, dtype, owner, ctx=None):
        """Wrap memory owned by something else (``owner`` is kept alive)."""
        return cls(size, dtype, ctx=ctx, _base=owner, _ptr=ptr)

    @property
    def gpudata(self):
        return self.ptr

    @property
    def nbytes(self):
        return self.size * self.dtype.itemsize

    def __len__(self):
        return self.size

    def get(self):
        out = np.empty(self.size, dtype=self.dtype)
        if self.size:
            _lib.check(self.ctx._lib.chroma_memcpy_dtoh(self.ctx.handle, _lib.ptr(out), ctypes.c_void_p(self.ptr), self.nbytes))
        return out

    def set(self, ary, upload=False):
        """Host -> device.  ``upload=True``: on the context's second stream (chroma_upload), not ordered with the work
        queued on the main one -- for an array nothing queued is using, e.g. one just allocated (Simulation's
        prefetching batch loop)."""
        ary = np.ascontiguousarray(ary)
        if ary.dtype != self.dtype:
            if ary.dtype.itemsize * ary.size == self.nbytes and (ary.dtype.fields or self.dtype.fields):
                ary = ary.view(self.dtype).reshape(-1)
            else:
                ary = ary.astype(self.dtype)
        if ary.size != self.size:
            raise ValueError('size mismatch: %d vs %d' % (ary.size, self.size))
        if self.size:
            copy = self.ctx._lib.chroma_upload if upload else self.ctx._lib.chroma_memcpy_htod
            _lib.check(copy(self.ctx.handle, ctypes.c_void_p(self.ptr), _lib.ptr(ary), self.nbytes))
        return self

    def fill(self, value):
        """Fill with a scalar (4-byte element types and vectors of them)."""
        if self.size == 0:
            return self
        if self.dtype.fields or self.dtype.itemsize != 4:
            self.set(np.full(self.size, value, dtype=self.dtype))
            return self
        word = int(np.array(value, dtype=self.dtype).view(np.uint32))
        _lib.check(self.ctx._lib.chroma_memset32(self.ctx.handle, ctypes.c_void_p(self.ptr), word, self.size))
        return self

    def copy_from_device(self, other):
        if other.nbytes != self.nbytes:
            raise ValueError('size mismatch')
        _lib.check(self.ctx._lib.chroma_memcpy_dtod(self.ctx.handle, ctypes.c_void_p(self.ptr), ctypes.c_void_p(other.ptr), self.nbytes))
        return self

    def __getitem__(self, key):
        if not isinstance(key, slice):
            raise TypeError('GPUArray supports contiguous slices only')
        start, stop, step = key.indices(self.size)
        if step != 1:
            raise ValueError('GPUArray supports contiguous slices only')
        n = max(0, stop - start)
        return GPUArray(n, self.dtype, ctx=self.ctx, _base=self._alloc if isinstance(self._alloc, _Allocation) else self._alloc,
                        _ptr=self.ptr + start * self.dtype.itemsize)


def empty(shape, dtype, ctx=None):
    return GPUArray(shape, dtype, ctx=ctx)


def zeros(shape, dtype, ctx=None):
    a = GPUArray(shape, dtype, ctx=ctx)
    if a.size:
        _lib.check(a.ctx._lib.chroma_memset32(a.ctx.handle, ctypes.c_void_p(a.ptr), 0, (a.nbytes + 3) // 4))
    return a


def to_gpu(ary, ctx=None):
    ary = np.ascontiguousarray(ary)
    a = GPUArray(ary.size, ary.dtype, ctx=ctx)
    a.set(ary.reshape(-1))
    return a


# ---- RNG ------------------------------------------------------------------------------------------
class RNGStates(object):
    """What ``get_rng_states`` returns.  The reference allocates one XORWOW state per thread
    slot (chroma/gpu/tools.py:75-84); here a photon's stream is Philox4x32-10 keyed by
    ``seed`` and indexed by a global photon id, so the object only carries the seed and the
    next unused photon id.  ``size`` is accepted and remembered for API compatibility."""

    def __init__(self, size, seed=1):
        self.size = int(size)
        self.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.next_photon_id = 0

    def reserve(self, nphotons):
        """Hand out a block of ``nphotons`` fresh photon ids; returns the first."""
        base = self.next_photon_id
        self.next_photon_id += int(nphotons)
        return base


def get_rng_states(size, seed=1):
    return RNGStates(size, seed)


# ---- launch helpers ---------------------------------------------------------------------------------
def chunk_iterator(nelements, nthreads_per_block=64, max_blocks=1024):
    """Yield (first_index, elements_this_iteration, nblocks_this_iteration) covering
    ``nelements`` in passes of at most ``max_blocks`` blocks.

    >>> list(chunk_iterator(300, 32, 2))
    [(0, 64, 2), (64, 64, 2), (128, 64, 2), (192, 64, 2), (256, 44, 2)]
    """
    first = 0
    while first < nelements:
        left = nelements - first
        blocks = min(max_blocks, -(-left // nthreads_per_block))
        this_round = min(left, blocks * nthreads_per_block)
        yield (first, this_round, blocks)
        first += this_round


def format_size(size):
    if size < 1e3:
        return '%d' % size
    if size < 1e6:
        return '%1.1fK' % (size / 1e3)
    if size < 1e9:
        return '%1.1fM' % (size / 1e6)
    return '%1.1fG' % (size / 1e9)


def format_array(name, array):
    return '%-15s %6s %6s' % (name, format_size(len(array)), format_size(array.nbytes))
