"""Simulation driver: batches events, propagates them on the GPU and extracts hits.

Reference: chroma/sim.py:21-186 (``Simulation.__init__``, ``_simulate_batch``, ``simulate``).
Out of scope here, as in SURVEY.md section 8: GEANT4 photon generation (``geant4_processes``
is accepted; Event/Vertex inputs need a generator that this package does not provide), the
PDF / likelihood entry points.
"""
import os
import time
from timeit import default_timer as timer

import numpy as np

from chroma_amd import event, gpu, itertoolset


def pick_seed():
    """A seed mixed from the time and the process id (chroma/sim.py:16-19)."""
    return int(time.time()) ^ (os.getpid() << 16) & 2 ** 32 - 1


class Simulation(object):
    def __init__(self, detector, seed=None, cuda_device=None, particle_tracking=False, photon_tracking=False,
                 geant4_processes=0, nthreads_per_block=64, max_blocks=1024, exact=False, prefetch=True, lanes=1):
        # ``exact``: propagate with the reference's own traversal loop for every ray (GPUPhotons.propagate(exact=True)):
        # the reference's hit triangle on every ray, several times slower than the default walk
        self.exact = bool(exact)
        # ``prefetch``: simulate() uploads the photons of the NEXT batch (another host thread, the context's second
        # stream, pinned staging buffers) while the current batch propagates; the batches' device arrays come from the
        # library's pool, so after the first two batches nothing is allocated.  The user's iterable is read one batch
        # ahead of the events that are yielded.
        self.prefetch = bool(prefetch)
        # ``lanes`` > 1: simulate() keeps that many batches in flight at once, each on a context of its own (its own streams,
        # queues and working sets; the geometry is uploaded once per lane) driven by a host thread of its own.  On the DEVICE
        # a batch of 1e4-1e6 photons is launch- and latency-bound -- a step under ~1e5 rays lasts as long as its longest ray,
        # the last 8192 photons as long as the longest photon -- and several in flight fill the chip: 2.8 x the photons per
        # second at 1e4 photons per batch, 2 x at 1e6, 1.3 x at 1e7, nothing at 1e8 (device photons, tools/concurrency_probe.py,
        # profiles/r04/concurrency_probe.txt).  END TO END, with host photons in and Python event objects out, the host side
        # of a batch (staging, event objects, hit dictionaries) is most of its time and the lanes share the interpreter: 1.0-1.8 x
        # with four lanes, 0.9-1.3 x with two (profiles/r04/sim_lanes_probe.txt) -- so the default is 1.  Results are those of lanes=1 bit for bit (a photon's random
        # stream is keyed by its global id, handed out in batch order) and are yielded in order.  Not used with photon
        # tracking, keep_photons_beg or run_daq (those take the one-batch-at-a-time loop).
        self.nlanes = max(1, int(lanes))
        self.detector = detector
        self.nthreads_per_block = nthreads_per_block
        self.max_blocks = max_blocks
        self.photon_tracking = photon_tracking
        self.seed = pick_seed() if seed is None else seed
        np.random.seed(self.seed % (2 ** 32))
        self.photon_generator = None      # GEANT4 generation is outside the propagate path

        self.context = gpu.create_cuda_context(cuda_device)
        if getattr(detector, 'bvh', None) is None:
            from chroma_amd.loader import load_bvh
            detector.flatten()
            detector.bvh = load_bvh(detector)
        make = gpu.GPUDetector if hasattr(detector, 'num_channels') else gpu.GPUGeometry
        packed = None
        if self.nlanes > 1:
            from chroma_amd.gpu.geometry import pack_geometry
            packed = pack_geometry(detector)          # packed once, uploaded once per lane
        self.gpu_geometry = make(detector, packed=packed)
        if hasattr(detector, 'num_channels'):
            self.gpu_daq = gpu.GPUDaq(self.gpu_geometry)
        self._lanes = [(self.context, self.gpu_geometry)]
        for _ in range(1, self.nlanes):
            ctx = gpu.tools.Context(self.context.device_id)
            with ctx.bound():
                self._lanes.append((ctx, make(detector, packed=packed)))
        packed = None
        self.rng_states = gpu.get_rng_states(self.nthreads_per_block * self.max_blocks, seed=self.seed)
        self.pdf_config = None

    def _upload_batch(self, batch_events, upload=True):
        """The photons of all ``batch_events`` as one GPUPhotons (chroma/sim.py:66-72) + the events' bounds in it.
        With ``upload`` the copies use the context's second stream: safe to run while another batch propagates."""
        # (large events go to their slice of the device arrays directly; many small ones are joined on the host first:
        #  a copy call per array per event would cost more than the concatenation)
        sizes = [len(ev.photons_beg) for ev in batch_events]
        if len(batch_events) == 1 or min(sizes) >= 1_000_000:
            batch_photons = [ev.photons_beg for ev in batch_events]
        else:
            batch_photons = event.Photons.join([ev.photons_beg for ev in batch_events])
        bounds = np.cumsum(np.concatenate([[0], [len(ev.photons_beg) for ev in batch_events]]))
        return gpu.GPUPhotons(batch_photons, copy_triangles=False, copy_weights=False, upload=upload), bounds

    def _simulate_batch(self, batch_events, keep_photons_beg=False, keep_photons_end=False, keep_hits=True,
                        keep_flat_hits=True, run_daq=False, max_steps=100, verbose=False, uploaded=None, gpu_geometry=None):
        """Propagate the photons of all ``batch_events`` in one go and split the results
        back per event (by evidx).  Yields the events.  ``uploaded``: what _upload_batch returned for them;
        ``gpu_geometry``: the geometry of the lane this batch runs on (default: the simulation's own)."""
        t_start = timer()
        geometry = gpu_geometry if gpu_geometry is not None else self.gpu_geometry
        if uploaded is None:
            uploaded = self._upload_batch(batch_events, upload=False)
        gpu_photons, bounds = uploaded
        t_copy = timer()
        is_detector = hasattr(self.detector, 'num_channels')
        want_hits = is_detector and (keep_hits or keep_flat_hits)
        batch_hits = tracking = None
        if want_hits and not self.photon_tracking:
            # propagate + get_flat_hits as one library call (chroma_propagate_hits): the same set of hits
            batch_hits = gpu_photons.propagate_hits(geometry, self.rng_states, max_steps=max_steps, exact=self.exact, sort=True)
        else:
            tracking = gpu_photons.propagate(geometry, self.rng_states,
                                             nthreads_per_block=self.nthreads_per_block, max_blocks=self.max_blocks,
                                             max_steps=max_steps, track=self.photon_tracking, exact=self.exact)
        t_prop = timer()
        if verbose:
            print('GPU copy took %0.2f s' % (t_copy - t_start))
            print('GPU propagate took %0.2f s' % (t_prop - t_copy))

        batch_end = gpu_photons.get() if keep_photons_end else None
        if want_hits and batch_hits is None:
            batch_hits = gpu_photons.get_flat_hits(geometry, sort=True)

        # The hits of each event, and within an event of each channel, are SLICES: the library hands the batch's hits over in
        # (event, channel) order (chroma_hits_sort: a radix sort and a gather on the device), where the reference masks all
        # hits once per event and once per channel (chroma/sim.py:118-123: hits x events + hits x channels element tests --
        # 2e11 for one 1e8-photon batch on a 29k-channel detector).
        per_event_hits = None
        if batch_hits is not None:
            if len(batch_events) == 1:
                per_event_hits = [batch_hits]
            else:
                cuts = np.searchsorted(batch_hits.evidx, np.arange(len(batch_events) + 1))
                per_event_hits = [batch_hits[a:b] for a, b in zip(cuts[:-1], cuts[1:])]
        for i, (ev, lo, hi) in enumerate(zip(batch_events, bounds[:-1], bounds[1:])):
            if not keep_photons_beg:
                ev.photons_beg = None
            if self.photon_tracking:
                step_ids_list, step_photons_list = tracking
                tracks = [[] for _ in range(hi - lo)]
                for step_ids, step_photons in zip(step_ids_list, step_photons_list):
                    mask = np.logical_and(step_ids >= lo, step_ids < hi)
                    if np.count_nonzero(mask) == 0:
                        break
                    ids = step_ids[mask] - lo
                    photons = step_photons[mask]
                    for k, pid in enumerate(ids):
                        tracks[pid].append(photons[k])
                ev.photon_tracks = [event.Photons.join(t, concatenate=False) if len(t) > 0 else event.Photons()
                                    for t in tracks]
            if keep_photons_end:
                ev.photons_end = batch_end[lo:hi]
            if batch_hits is not None:
                ev_hits = per_event_hits[i]
                if keep_hits:
                    # (the event's hits are in channel order already: a channel's hits are one slice)
                    ch = ev_hits.channel
                    first = np.flatnonzero(np.concatenate(([True], ch[1:] != ch[:-1]))) if len(ch) else np.zeros(0, np.intp)
                    last = np.append(first[1:], len(ch))
                    ev.hits = {c: ev_hits[a:b] for c, a, b in zip(ch[first].tolist(), first.tolist(), last.tolist())}
                if keep_flat_hits:
                    ev.flat_hits = ev_hits
            if hasattr(self, 'gpu_daq') and run_daq:
                # one acquisition per event, as the reference (chroma/sim.py:128-137)
                self.gpu_daq.begin_acquire()
                self.gpu_daq.acquire(gpu_photons, self.rng_states, start_photon=int(lo), nphotons=int(hi - lo),
                                     nthreads_per_block=self.nthreads_per_block, max_blocks=self.max_blocks)
                ev.channels = self.gpu_daq.end_acquire().get()
            yield ev

    def simulate(self, iterable, keep_photons_beg=False, keep_photons_end=False, keep_hits=True,
                 keep_flat_hits=True, run_daq=False, max_steps=1000, photons_per_batch=1000000, evid_start=0):
        """Simulate Photons objects (or Events that already carry ``photons_beg``); events are
        batched until ``photons_per_batch`` photons are collected (chroma/sim.py:141-186).

        With ``Simulation(prefetch=True)`` (the default) the NEXT batch is taken from ``iterable`` and uploaded by a second
        thread while the current one propagates: the iterable is consumed one batch ahead of the events this generator
        yields.  An iterable whose items share buffers, or depend on results already yielded, needs ``prefetch=False``;
        ``keep_photons_beg=True`` turns the read-ahead off by itself (the reference's loop is lazy, chroma/sim.py:141-186)."""
        if isinstance(iterable, event.Photons):
            first, iterable = iterable, [iterable]
        else:
            first, iterable = itertoolset.peek(iterable)
        if isinstance(first, event.Photons):
            iterable = (event.Event(photons_beg=x) for x in iterable)
        elif isinstance(first, event.Event):
            if first.photons_beg is None:
                raise NotImplementedError('events without photons need the GEANT4 generator, which is out of scope')
        else:
            raise NotImplementedError('Vertex input needs the GEANT4 generator, which is out of scope')

        kwargs = dict(keep_photons_beg=keep_photons_beg, keep_photons_end=keep_photons_end, keep_hits=keep_hits,
                      keep_flat_hits=keep_flat_hits, run_daq=run_daq, max_steps=max_steps)
        def batches():
            nphotons = 0
            batch = []
            evid = evid_start
            for ev in iterable:
                ev.id = evid
                evid += 1
                ev.nphotons = len(ev.photons_beg)
                ev.photons_beg.evidx[:] = len(batch)
                nphotons += ev.nphotons
                batch.append(ev)
                if nphotons >= photons_per_batch:
                    yield batch
                    nphotons = 0
                    batch = []
            if batch:
                yield batch

        # (read-ahead contract: with prefetch the iterable is pulled one batch ahead of the events being yielded, and `evidx` is
        #  written into the photons of that next batch early; a generator that reuses its buffers, or a caller who wants
        #  `photons_beg` back untouched, gets the reference's lazy loop instead)
        if self.nlanes > 1 and not (self.photon_tracking or keep_photons_beg or run_daq):
            yield from self._simulate_lanes(batches(), kwargs)
            return
        if not self.prefetch or self.photon_tracking or keep_photons_beg:
            for batch in batches():
                yield from self._simulate_batch(batch, **kwargs)
            return
        # one batch ahead: while batch k propagates (the library call releases the GIL), a second thread stages and
        # uploads batch k + 1 on the context's second stream
        from concurrent.futures import ThreadPoolExecutor
        it = batches()
        with ThreadPoolExecutor(max_workers=1) as pool:
            cur = next(it, None)
            fut = pool.submit(self._upload_batch, cur) if cur is not None else None
            while cur is not None:
                uploaded = fut.result()
                nxt = next(it, None)
                fut = pool.submit(self._upload_batch, nxt) if nxt is not None else None
                yield from self._simulate_batch(cur, uploaded=uploaded, **kwargs)
                uploaded = None
                cur = nxt

    def _simulate_lanes(self, batches, kwargs):
        """``lanes`` batches in flight at once, each on its own context from its own host thread (the library calls
        release the GIL); events come back in the order of the iterable.  The photon-id block of a batch (its random
        streams) is reserved HERE, in batch order, so the results are those of the one-batch-at-a-time loop."""
        from collections import deque
        from concurrent.futures import ThreadPoolExecutor

        def work(lane, batch, rng_base):
            ctx, geometry = self._lanes[lane]
            with ctx.bound():
                uploaded = self._upload_batch(batch, upload=False)
                uploaded[0]._rng_base, uploaded[0]._rng_owner = rng_base, self.rng_states
                return list(self._simulate_batch(batch, uploaded=uploaded, gpu_geometry=geometry, **kwargs))

        with ThreadPoolExecutor(max_workers=self.nlanes) as pool:
            free, pending = deque(range(self.nlanes)), deque()
            for batch in batches:
                if not free:
                    lane, fut = pending.popleft()
                    events = fut.result()
                    free.append(lane)
                    yield from events
                lane = free.popleft()
                base = self.rng_states.reserve(sum(len(ev.photons_beg) for ev in batch))
                pending.append((lane, pool.submit(work, lane, batch, base)))
            while pending:
                lane, fut = pending.popleft()
                yield from fut.result()

    def __del__(self):
        try:
            for ctx, geometry in self._lanes[1:]:
                ctx.synchronize()
            self.context.pop()
        except Exception:
            pass
