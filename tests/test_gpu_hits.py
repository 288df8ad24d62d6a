"""chroma_propagate_hits -- propagate and hit extraction as one library call -- against the separate calls it replaces.

The fused call lets photons that end in k_physics travel as 64-byte records and fills the caller's arrays, counts and
compacts the hits and bumps the per-channel arrays in ONE pass (k_finalize_hits).  Everything must come out as from
GPUPhotons.propagate + get_flat_hits + channel_hits (chroma/gpu/photon.py:96-175,193-259; chroma/cuda/propagate.cu:147-214):
the photon arrays bit for bit, the same SET of flat hits with the same channels, the channel arrays.
"""
import numpy as np
import pytest

from chroma_amd import event
from conftest import bomb, make_stress_geometry
from test_gpu_parity import FIELDS, assert_bit_exact, _edge_photons

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    from chroma_amd import gpu as g
    ctx = g.create_cuda_context(0)
    yield g
    ctx.pop()


def canonical(hits):
    keys = [hits.dir[:, 0].view(np.uint32), hits.dir[:, 1].view(np.uint32), hits.pol[:, 0].view(np.uint32), hits.pos[:, 0].view(np.uint32),
            hits.pos[:, 1].view(np.uint32), hits.wavelengths.view(np.uint32), hits.t.view(np.uint32), hits.last_hit_triangles, hits.evidx, hits.channel]
    return hits[np.lexsort(keys)]


def both_ways(g, gg, photons, max_steps, ncopies=1, seed=7, capacity=None, **kw):
    from chroma_amd.gpu.tools import zeros, GPUArray
    ctx = g.get_context()
    # the separate calls
    a = g.GPUPhotons(photons, ncopies=ncopies)
    a.propagate(gg, g.get_rng_states(64, seed=seed), max_steps=max_steps, **kw)
    want = a.get()
    want_counters = a.rng_counters.get()
    want_hits = a.get_flat_hits(gg)
    wc, we = a.channel_hits(gg)
    # the fused call
    b = g.GPUPhotons(photons, ncopies=ncopies)
    counts = zeros(gg.nchannels, np.uint32, ctx)
    earliest = GPUArray(gg.nchannels, np.uint32, ctx).fill(np.uint32(0x7f800000))
    stats = {}
    got_hits = b.propagate_hits(gg, g.get_rng_states(64, seed=seed), max_steps=max_steps, capacity=capacity,
                                channel_arrays=(counts, earliest), stats=stats, **kw)
    got = b.get()
    assert_bit_exact(got, want, 'photon arrays after the fused call')
    assert np.array_equal(b.rng_counters.get(), want_counters)
    assert len(got_hits) == len(want_hits) == stats['nhits']
    # (the order of the flat hits is unspecified, in the reference -- one atomic per hit -- and here -- one per block of 4096
    #  photons, blocks landing in the order their atomics do: compare as sets, in a canonical order)
    got_hits, want_hits = canonical(got_hits), canonical(want_hits)
    assert_bit_exact(got_hits, want_hits, 'flat hits')
    assert np.array_equal(got_hits.channel, want_hits.channel)
    assert np.array_equal(counts.get(), wc.get()) and np.array_equal(earliest.get(), we.get())
    assert int(counts.get().sum()) == len(want_hits)
    return want, want_hits


def test_large_batch_mixed_wavelengths(gpu, tiny_geometry):
    gg = gpu.GPUDetector(tiny_geometry)
    end, hits = both_ways(gpu, gg, bomb(60000, 3, wavelength=400.0, wavelength_hi=800.0), 30)
    assert 200 < len(hits) < 6000 and (hits.flags & event.SURFACE_DETECT).all()


def test_small_batch_takes_the_tail_kernel(gpu, tiny_geometry):
    gg = gpu.GPUDetector(tiny_geometry)
    both_ways(gpu, gg, bomb(3000, 4), 100)


def test_photons_still_alive_at_max_steps(gpu, tiny_geometry):
    gg = gpu.GPUDetector(tiny_geometry)
    end, hits = both_ways(gpu, gg, bomb(40000, 5), 1)
    assert np.count_nonzero((end.flags & event.TERMINAL_MASK) == 0) > 100


def test_edge_inputs_and_photons_that_were_terminal_before(gpu, tiny_geometry):
    gg = gpu.GPUDetector(tiny_geometry)
    ph = _edge_photons()
    both_ways(gpu, gg, ph, 20)
    # detected before the call: counted as hits although the call never touches them
    ph2 = bomb(20000, 8)
    ph2.flags[::7] = event.SURFACE_DETECT
    ph2.last_hit_triangles[::7] = np.arange(len(ph2))[::7] * 17 % 380000
    end, hits = both_ways(gpu, gg, ph2, 20)


def test_copies(gpu, tiny_geometry):
    gg = gpu.GPUDetector(tiny_geometry)
    both_ways(gpu, gg, bomb(9000, 9), 30, ncopies=3)


def test_capacity_too_small_falls_back(gpu, tiny_geometry):
    gg = gpu.GPUDetector(tiny_geometry)
    both_ways(gpu, gg, bomb(60000, 3), 30, capacity=10)


def test_every_surface_model_and_the_exact_walk(gpu):
    gg = gpu.GPUDetector(make_stress_geometry())
    both_ways(gpu, gg, bomb(40000, 6, wavelength=350.0), 100, seed=11)
    both_ways(gpu, gg, bomb(20000, 7, wavelength=350.0), 100, seed=11, exact=True)


def test_weights(gpu, tiny_geometry):
    gg = gpu.GPUDetector(tiny_geometry)
    both_ways(gpu, gg, bomb(30000, 12), 30, use_weights=True)


def test_hits_in_event_and_channel_order_on_the_device(gpu, tiny_geometry):
    """chroma_hits_sort (get_flat_hits(sort=True), propagate_hits(sort=True)): the same SET of flat hits as without, in
    (evidx, channel) order -- what lets Simulation and get_hits split a batch's hits by slicing where the reference masks all
    hits once per event and per channel (chroma/sim.py:118-123, chroma/gpu/photon.py:96-105).  Also with a capacity too small
    (the fall-back path), with no hit at all, and through get_hits."""
    gg = gpu.GPUDetector(tiny_geometry)
    ph = bomb(120000, 41)
    ph.evidx[:] = np.random.default_rng(3).integers(0, 7, len(ph)).astype(np.uint32)
    a = gpu.GPUPhotons(ph)
    a.propagate(gg, gpu.get_rng_states(64, seed=9), max_steps=100)
    plain = a.get_flat_hits(gg)
    ordered = a.get_flat_hits(gg, sort=True)
    assert len(plain) == len(ordered) > 1000 and len(np.unique(ordered.evidx)) == 7
    key = ordered.evidx.astype(np.uint64) << np.uint64(32) | ordered.channel.astype(np.uint64)
    assert (np.diff(key.astype(np.int64)) >= 0).all(), 'not in (evidx, channel) order'
    for f in FIELDS + ('evidx', 'channel'):
        x, y = getattr(canonical(plain), f), getattr(canonical(ordered), f)
        assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), f
    for capacity in (None, 100):
        b = gpu.GPUPhotons(ph)
        fused = b.propagate_hits(gg, gpu.get_rng_states(64, seed=9), max_steps=100, capacity=capacity, sort=True)
        k2 = fused.evidx.astype(np.uint64) << np.uint64(32) | fused.channel.astype(np.uint64)
        assert len(fused) == len(plain) and (np.diff(k2.astype(np.int64)) >= 0).all()
        assert np.array_equal(canonical(fused).t.view(np.uint32), canonical(plain).t.view(np.uint32))
    # get_hits: per channel the photons the reference's mask selects
    hitmap = a.get_hits(gg)
    assert sorted(hitmap) == sorted(int(c) for c in np.unique(plain.channel))
    for ch in list(hitmap)[:50]:
        want = plain[plain.channel == ch]
        got = hitmap[ch]
        assert len(got) == len(want) and (got.channel == ch).all()
        assert np.array_equal(np.sort(got.t.view(np.uint32)), np.sort(want.t.view(np.uint32)))
    # nothing detected: nothing to sort
    none = gpu.GPUPhotons(bomb(1000, 42))
    assert len(none.get_flat_hits(gg, sort=True)) == 0 and none.get_hits(gg) == {}
