"""GPUPhotons: photon arrays on the device and the propagate / select / hit-extraction calls.

Public interface as chroma/gpu/photon.py (GPUPhotons :13-348, GPUPhotonsSlice :351-381):
same constructor arguments, attribute names (pos, dir, pol, wavelengths, t,
last_hit_triangles, flags, weights, evidx, true_nphotons, ncopies) and method
signatures.  ``nthreads_per_block`` / ``max_blocks`` are accepted and ignored: launch
shapes are chosen by the library.  One array is new: ``rng_counters``, the number of
uniforms each photon has consumed, which makes the Philox stream of a photon continue
across repeated propagate() calls.
"""
import ctypes
import sys

import numpy as np

from chroma_amd import _lib, event
from chroma_amd.tools import profile_if_possible
from chroma_amd.gpu.tools import (GPUArray, vec, to_float3, get_context, empty, zeros, RNGStates)

_FIELDS = ('pos', 'dir', 'pol', 'wavelengths', 't', 'flags', 'last_hit_triangles', 'weights', 'evidx')


def _structure(p):
    """chroma_photon_arrays of device pointers for a GPUPhotons-like object."""
    s = _lib.PhotonArrays()
    for name in _FIELDS:
        setattr(s, name, getattr(p, name).ptr)
    rc = getattr(p, 'rng_counters', None)
    s.rng_counters = rc.ptr if rc is not None else None
    return s


def _alloc_fields(n, ctx):
    return dict(pos=empty(n, vec.float3, ctx), dir=empty(n, vec.float3, ctx), pol=empty(n, vec.float3, ctx),
                wavelengths=empty(n, np.float32, ctx), t=empty(n, np.float32, ctx),
                last_hit_triangles=empty(n, np.int32, ctx), flags=empty(n, np.uint32, ctx),
                weights=empty(n, np.float32, ctx), evidx=empty(n, np.uint32, ctx))


def _vec3_to_rows(a):
    return a.view(np.float32).reshape((len(a), 3))


class GPUPhotons(object):
    def __init__(self, photons, ncopies=1, copy_flags=True, copy_triangles=True, copy_weights=True, upload=False):
        """Load ``photons`` (chroma_amd.event.Photons) onto the device, ``ncopies`` times.  ``upload=True``: the
        copies go through the context's second stream (GPUArray.set(upload=True)), so that this constructor can run in
        another thread while a previous photon set propagates (Simulation's batch loop)."""
        self.ctx = get_context()
        # ``photons`` may also be a LIST of Photons objects (the events of a batch): each is copied straight into its
        # slice of the device arrays, without concatenating them on the host first
        parts = list(photons) if isinstance(photons, (list, tuple)) else [photons]
        nphotons = sum(len(p) for p in parts)
        n = nphotons * ncopies
        f = _alloc_fields(n, self.ctx)
        self.pos, self.dir, self.pol = f['pos'], f['dir'], f['pol']
        self.wavelengths, self.t = f['wavelengths'], f['t']
        self.last_hit_triangles, self.flags, self.weights = f['last_hit_triangles'], f['flags'], f['weights']
        # the reference allocates evidx for nphotons only although photon_duplicate writes
        # all copies; allocate the full length
        self.evidx = f['evidx']
        self.rng_counters = zeros(n, np.uint32, self.ctx)
        if not copy_triangles:
            self.last_hit_triangles.fill(-1)
        if not copy_flags:
            self.flags.fill(0)
        if not copy_weights:
            self.weights.fill(1.0)

        # (np.asarray: no host-side copy when the array already has the device type, as event.Photons' arrays do)
        lo = 0
        for part in parts:
            hi = lo + len(part)
            if hi > lo:
                self.pos[lo:hi].set(to_float3(part.pos), upload)
                self.dir[lo:hi].set(to_float3(part.dir), upload)
                self.pol[lo:hi].set(to_float3(part.pol), upload)
                self.wavelengths[lo:hi].set(np.asarray(part.wavelengths, dtype=np.float32), upload)
                self.t[lo:hi].set(np.asarray(part.t, dtype=np.float32), upload)
                if copy_triangles:
                    self.last_hit_triangles[lo:hi].set(np.asarray(part.last_hit_triangles, dtype=np.int32), upload)
                if copy_flags:
                    self.flags[lo:hi].set(np.asarray(part.flags, dtype=np.uint32), upload)
                if copy_weights:
                    self.weights[lo:hi].set(np.asarray(part.weights, dtype=np.float32), upload)
                self.evidx[lo:hi].set(np.asarray(part.evidx, dtype=np.uint32), upload)
            lo = hi

        self.true_nphotons = nphotons
        self.ncopies = ncopies
        self._rng_base = None
        if ncopies > 1 and nphotons > 0:
            s = _structure(self)
            _lib.check(self.ctx._lib.chroma_photon_duplicate(self.ctx.handle, 0, nphotons, ctypes.byref(s),
                                                             ncopies - 1, nphotons))

    # ---- host copies -------------------------------------------------------------------------
    def get(self):
        return event.Photons(_vec3_to_rows(self.pos.get()), _vec3_to_rows(self.dir.get()),
                             _vec3_to_rows(self.pol.get()), self.wavelengths.get(), self.t.get(),
                             self.last_hit_triangles.get(), self.flags.get(), self.weights.get(),
                             self.evidx.get())

    def __len__(self):
        return self.pos.size

    # ---- propagation ---------------------------------------------------------------------------
    def _rng(self, rng_states):
        """chroma_rng for this photon set: the id block is reserved on first use.  A slice made by
        iterate_copies() uses its parent's block, shifted by its offset, so a photon keeps its stream."""
        parent = getattr(self, '_rng_parent', None)
        if parent is not None:
            base = parent[0]._rng(rng_states)
            if isinstance(rng_states, RNGStates):
                return _lib.Rng(base.seed, base.photon_id_base + parent[1])
        if isinstance(rng_states, RNGStates):
            if self._rng_base is None or getattr(self, '_rng_owner', None) is not rng_states:
                self._rng_base = rng_states.reserve(len(self))
                self._rng_owner = rng_states
            return _lib.Rng(rng_states.seed, self._rng_base)
        if isinstance(rng_states, _lib.Rng):
            return rng_states
        raise TypeError('rng_states must come from chroma_amd.gpu.get_rng_states()')

    @profile_if_possible
    def propagate(self, gpu_geometry, rng_states, nthreads_per_block=64, max_blocks=1024, max_steps=10,
                  use_weights=False, scatter_first=0, track=False, stats=None, time_kernels=False, exact=False, counting=None):
        """Propagate to termination or ``max_steps``, whichever comes first.  May be called
        repeatedly to single-step.  With ``track=True`` returns (step_photon_ids, step_photons)
        like the reference (chroma/gpu/photon.py:218-238,258-259).

        ``exact=True``: every ray takes the reference's own traversal loop (chroma/cuda/mesh.h:42-118 literally; the
        walk 'literal') for this call -- the reference's triangle on EVERY ray, including the numerically erratic
        Moeller-Trumbore hits the default nearest-first walk does not reproduce (~2e-6 of rays aimed exactly at mesh
        features, none in 2.4e8 random photons; include/chroma_hip.h at chroma_set_walk), at several times the cost.
        (``track=True`` always runs the literal loop: one lane per photon, chroma_propagate_step.)
        ``counting``: whether the kernels count node visits and triangle tests for ``stats`` (None: the context's setting).
        Both travel with the call (chroma_propagate_options): they do not change the context, so two threads may use one."""
        nphotons = self.pos.size
        lib, ctx = self.ctx._lib, self.ctx
        rng = self._rng(rng_states)
        s = _structure(self)
        if not track:
            st = _lib.PropagateStats()
            aborted = ctypes.c_int32(0)
            # (what the call does travels WITH the call -- chroma_propagate_options -- not as a setting of the context)
            opt = _lib.PropagateOptions(max_steps, use_weights, scatter_first, time_kernels, walk=ctx.WALKS['literal'] if exact else -1,
                                        counting=-1 if counting is None else int(bool(counting)))
            _lib.check(lib.chroma_propagate_opt(ctx.handle, gpu_geometry.handle, ctypes.byref(s), nphotons, self.ncopies, rng,
                                                ctypes.byref(opt), ctypes.byref(st), ctypes.byref(aborted), None))
            if stats is not None:
                for k, v in st.as_dict().items():
                    stats[k] = stats.get(k, 0) + v
            if aborted.value:
                print("WARNING: ABORTED PHOTONS", file=sys.stderr)
            return None

        # tracking mode: one step per launch, queues kept on the Python side
        input_queue = np.empty(nphotons + 1, dtype=np.uint32)
        input_queue[0] = 0
        for copy in range(self.ncopies):
            input_queue[1 + copy::self.ncopies] = np.arange(self.true_nphotons, dtype=np.uint32) + copy * self.true_nphotons
        in_q = GPUArray(nphotons + 1, np.uint32, ctx).set(input_queue)
        out_init = np.zeros(nphotons + 1, dtype=np.uint32)
        out_init[0] = 1
        out_q = GPUArray(nphotons + 1, np.uint32, ctx).set(out_init)
        step_photon_ids = [in_q[1:nphotons + 1].get()]
        step_photons = [self.copy_queue(in_q[1:], nphotons).get()]
        step = 0
        while step < max_steps:
            _lib.check(lib.chroma_propagate_step(ctx.handle, gpu_geometry.handle, 0, nphotons, in_q[1:].ptr, out_q.ptr,
                                                 rng, ctypes.byref(s), 1, int(bool(use_weights)), int(scatter_first)))
            step_photon_ids.append(in_q[1:nphotons + 1].get())
            step_photons.append(self.copy_queue(in_q[1:], nphotons).get())
            step += 1
            scatter_first = 0
            if step < max_steps:
                in_q, out_q = out_q, in_q
                out_q[:1].set(np.ones(1, dtype=np.uint32))
                nphotons = int(in_q[:1].get()[0]) - 1
                if nphotons == 0:
                    break
        ctx.synchronize()
        if int(np.bitwise_or.reduce(self.flags.get(), initial=0)) & (1 << 31):
            print("WARNING: ABORTED PHOTONS", file=sys.stderr)
        return step_photon_ids, step_photons

    @profile_if_possible
    def propagate_hits(self, gpu_detector, rng_states, max_steps=10, use_weights=False, scatter_first=0, target_flag=(0x1 << 2),
                       capacity=None, channel_arrays=None, stats=None, time_kernels=False, exact=False,
                       nthreads_per_block=64, max_blocks=1024, sort=False):
        """``propagate`` followed by ``get_flat_hits`` as ONE library call (chroma_propagate_hits): the pass that finishes
        the propagation also counts and compacts the detected photons with their channels (chroma/gpu/photon.py:96-175) and,
        when ``channel_arrays=(counts, earliest)`` (device arrays of nchannels uint32) is given, accumulates the per-channel
        hit count and earliest time into them.  Returns what ``get_flat_hits`` would return after ``propagate`` (the same set of photons;
        their order is unspecified, as in the reference).  ``capacity``: room for that many flat hits (default: a quarter of the photons, at least
        65 536); should more be detected, the full set is fetched with ``get_flat_hits`` afterwards."""
        nphotons = self.pos.size
        lib, ctx = self.ctx._lib, self.ctx
        if capacity is None:
            capacity = min(nphotons, max(65536, nphotons // 4))
        capacity = int(max(capacity, 1))
        out = GPUPhotonsSlice(**_alloc_fields(capacity, ctx))
        channels = empty(capacity, np.int32, ctx)
        dst = _structure(out)
        s = _structure(self)
        req = _lib.HitsRequest()
        req.detection_state = int(target_flag)
        req.capacity = capacity
        req.dst = ctypes.pointer(dst)
        req.d_channels = channels.ptr
        if channel_arrays is not None:
            req.d_hit_count = channel_arrays[0].ptr
            req.d_earliest_time_bits = channel_arrays[1].ptr if channel_arrays[1] is not None else None
        st = _lib.PropagateStats()
        aborted = ctypes.c_int32(0)
        opt = _lib.PropagateOptions(max_steps, use_weights, scatter_first, time_kernels, walk=ctx.WALKS['literal'] if exact else -1)
        _lib.check(lib.chroma_propagate_opt(ctx.handle, gpu_detector.handle, ctypes.byref(s), nphotons, self.ncopies,
                                            self._rng(rng_states), ctypes.byref(opt), ctypes.byref(st), ctypes.byref(aborted), ctypes.byref(req)))
        if stats is not None:
            for k, v in st.as_dict().items():
                stats[k] = stats.get(k, 0) + v
            stats['nhits'] = stats.get('nhits', 0) + int(req.nhits)
        if aborted.value:
            print("WARNING: ABORTED PHOTONS", file=sys.stderr)
        n = int(req.nhits)
        if n > capacity:
            return self.get_flat_hits(gpu_detector, target_flag=target_flag, sort=sort)
        if sort and n > 1:
            # (event, channel) order on the device (chroma_hits_sort): the caller's split by event and channel is slicing
            _lib.check(lib.chroma_hits_sort(ctx.handle, ctypes.byref(dst), channels.ptr, n))
        w = slice(0, n)
        p = GPUPhotonsSlice(pos=out.pos[w], dir=out.dir[w], pol=out.pol[w], wavelengths=out.wavelengths[w], t=out.t[w],
                            last_hit_triangles=out.last_hit_triangles[w], flags=out.flags[w], weights=out.weights[w],
                            evidx=out.evidx[w], rng_counters=out.rng_counters[w]).get()
        p.channel = channels[w].get().astype(np.uint32) if n else np.zeros(0, dtype=np.uint32)
        return p

    @profile_if_possible
    def copy_queue(self, queue_gpu, nphotons, nthreads_per_block=64, max_blocks=1024, start_photon=0):
        """Gather the photons listed in ``queue_gpu`` (tracking mode, photon.py:261-285)."""
        f = _alloc_fields(nphotons, self.ctx)
        out = GPUPhotonsSlice(**f)
        if nphotons > 0:
            src, dst = _structure(self), _structure(out)
            _lib.check(self.ctx._lib.chroma_copy_photon_queue(self.ctx.handle, start_photon, nphotons, queue_gpu.ptr,
                                                              ctypes.byref(src), ctypes.byref(dst)))
        return out

    @profile_if_possible
    def select(self, target_flag, nthreads_per_block=64, max_blocks=1024, start_photon=None, nphotons=None):
        """New photon set with the photons whose history has ``target_flag`` set."""
        if start_photon is None:
            start_photon = 0
        if nphotons is None:
            nphotons = self.pos.size - start_photon
        lib, ctx = self.ctx._lib, self.ctx
        count = ctypes.c_uint32()
        _lib.check(lib.chroma_count_photons(ctx.handle, start_photon, nphotons, int(target_flag), self.flags.ptr,
                                            ctypes.byref(count)))
        out = GPUPhotonsSlice(**_alloc_fields(count.value, ctx))
        if count.value > 0:
            src, dst = _structure(self), _structure(out)
            ncopied = ctypes.c_uint32()
            _lib.check(lib.chroma_copy_photons(ctx.handle, start_photon, nphotons, int(target_flag), ctypes.byref(src),
                                               ctypes.byref(dst), ctypes.byref(ncopied)))
            assert ncopied.value == count.value
        return out

    def get_hits(self, *args, **kwargs):
        """dict channel -> Photons detected on that channel."""
        # the hits put in channel order on the device, then slices (chroma/gpu/photon.py:96-105 masks all hits once per
        # channel): the same photons per channel.  (Copies of one photon set share evidx 0 .. so the device order by
        # (evidx, channel) is re-grouped by channel alone here when there is more than one event index.)
        flat = self.get_flat_hits(*args, sort=True, **kwargs)
        if len(flat) and int(flat.evidx.max()) != int(flat.evidx.min()):
            flat = flat[np.argsort(flat.channel, kind='stable')]
        ch = flat.channel
        first = np.flatnonzero(np.concatenate(([True], ch[1:] != ch[:-1]))) if len(ch) else np.zeros(0, np.intp)
        last = np.append(first[1:], len(ch))
        return {c: flat[a:b] for c, a, b in zip(ch[first].tolist(), first.tolist(), last.tolist())}

    def get_flat_hits(self, gpu_detector, target_flag=(0x1 << 2), nthreads_per_block=64, max_blocks=1024,
                      start_photon=None, nphotons=None, no_map=False, sort=False):
        """Photons with ``target_flag`` set whose last hit triangle belongs to a channel, plus
        that channel (chroma/gpu/photon.py:107-175).  Order is unspecified, as in the reference; ``sort=True``: in
        (evidx, channel) order, made on the device (chroma_hits_sort)."""
        if start_photon is None:
            start_photon = 0
        if nphotons is None:
            nphotons = self.pos.size - start_photon
        lib, ctx = self.ctx._lib, self.ctx
        src = _structure(self)
        count = ctypes.c_uint32()
        _lib.check(lib.chroma_count_photon_hits(ctx.handle, gpu_detector.handle, start_photon, nphotons, int(target_flag),
                                                ctypes.byref(src), ctypes.byref(count)))
        n = count.value
        out = GPUPhotonsSlice(**_alloc_fields(n, ctx))
        channels = empty(n, np.int32, ctx)
        if n > 0:
            dst = _structure(out)
            ncopied = ctypes.c_uint32()
            _lib.check(lib.chroma_copy_photon_hits(ctx.handle, gpu_detector.handle, start_photon, nphotons, int(target_flag),
                                                   ctypes.byref(src), ctypes.byref(dst), channels.ptr, ctypes.byref(ncopied)))
            assert ncopied.value == n
            if sort and n > 1:
                _lib.check(lib.chroma_hits_sort(ctx.handle, ctypes.byref(dst), channels.ptr, n))
        p = out.get()
        p.channel = channels.get().astype(np.uint32) if n else np.zeros(0, dtype=np.uint32)
        return p

    def channel_hits(self, gpu_detector, target_flag=(0x1 << 2)):
        """Per-channel (hit count, earliest hit time) of this photon set, reduced on the device.
        This is the array that is all-reduced across GPUs."""
        lib, ctx = self.ctx._lib, self.ctx
        nch = gpu_detector.nchannels
        counts = zeros(nch, np.uint32, ctx)
        earliest = GPUArray(nch, np.uint32, ctx).fill(np.uint32(0x7f800000))
        src = _structure(self)
        _lib.check(lib.chroma_channel_hits(ctx.handle, gpu_detector.handle, self.pos.size, int(target_flag),
                                           ctypes.byref(src), counts.ptr, earliest.ptr))
        return counts, earliest

    def sort_by_direction(self):
        """Put the photons in the order of tools.argsort_direction (chroma/tools.py:175-193: a Morton code of theta
        and phi of their directions), on the device -- what the reference's benchmark does to its photons before it
        starts the clock (chroma/benchmark.py:80-82), so that neighbouring photons take neighbouring paths.  Every
        array of the set is reordered; slot i then holds the photon of rank i (random streams stay keyed by slot)."""
        s = _structure(self)
        _lib.check(self.ctx._lib.chroma_photons_sort_direction(self.ctx.handle, ctypes.byref(s), self.pos.size))
        return self

    def iterate_copies(self):
        """GPUPhotonsSlice views of the ``ncopies`` replicas."""
        for i in range(self.ncopies):
            w = slice(self.true_nphotons * i, self.true_nphotons * (i + 1))
            view = GPUPhotonsSlice(pos=self.pos[w], dir=self.dir[w], pol=self.pol[w],
                                   wavelengths=self.wavelengths[w], t=self.t[w],
                                   last_hit_triangles=self.last_hit_triangles[w], flags=self.flags[w],
                                   weights=self.weights[w], evidx=self.evidx[w],
                                   rng_counters=self.rng_counters[w])
            view._rng_parent = (self, self.true_nphotons * i)
            yield view


def generate_bomb(nphotons, seed, id_base=0, pos=(0.0, 0.0, 0.0), wavelength_lo=400.0, wavelength_hi=0.0, ctx=None):
    """An isotropic photon bomb made ON THE DEVICE (chroma/benchmark.py:77-83 with the formulas of
    chroma/sample.py:16-30; photon ``i`` draws from the Philox stream of ``0xB0B0... + id_base + i`` under
    ``seed``): the source for batches too large to upload.  Returns a GPUPhotonsSlice; propagate it with
    ``_lib.Rng(engine_seed, id_base)`` to give every photon the stream of its global id."""
    ctx = ctx or get_context()
    out = GPUPhotonsSlice(rng_counters=empty(nphotons, np.uint32, ctx), **_alloc_fields(nphotons, ctx))
    s = _structure(out)
    p = (ctypes.c_float * 3)(*[float(x) for x in pos])
    _lib.check(ctx._lib.chroma_generate_bomb(ctx.handle, ctypes.byref(s), int(nphotons), int(seed), int(id_base), p,
                                             float(wavelength_lo), float(wavelength_hi)))
    return out


class GPUPhotonsSlice(GPUPhotons):
    """A view of (or a set of freshly gathered) device photon arrays; same methods as
    GPUPhotons (chroma/gpu/photon.py:351-381)."""

    def __init__(self, pos, dir, pol, wavelengths, t, last_hit_triangles, flags, weights, evidx, rng_counters=None):
        self.ctx = pos.ctx
        self.pos, self.dir, self.pol = pos, dir, pol
        self.wavelengths, self.t = wavelengths, t
        self.last_hit_triangles, self.flags, self.weights, self.evidx = last_hit_triangles, flags, weights, evidx
        self.rng_counters = rng_counters if rng_counters is not None else zeros(len(pos), np.uint32, self.ctx)
        self.true_nphotons = len(pos)
        self.ncopies = 1
        self._rng_base = None
