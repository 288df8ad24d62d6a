"""Simulation's batch loop with host photons (reference semantics chroma/sim.py:58-139,
chroma/gpu/photon.py:13-94): the device arrays of consecutive batches come from the library's pool (no hipMalloc per
batch after the first two), the next batch is uploaded by a second thread on the context's second stream while the
current one propagates, and none of that changes a single result."""
import numpy as np
import pytest

from test_gpu_parity import assert_bit_exact

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def gpu():
    from chroma_amd import gpu as g
    ctx = g.create_cuda_context(0)
    yield g
    ctx.pop()


def test_pool_reuses_the_blocks_of_a_dropped_photon_set(gpu, oracle_mod):
    ctx = gpu.get_context()
    ctx.synchronize()
    ctx.pool_trim()
    ph = oracle_mod.generate_bomb(300000, seed=1)
    gp = gpu.GPUPhotons(ph)
    ptrs = sorted(a.ptr for a in (gp.pos, gp.dir, gp.pol, gp.wavelengths, gp.t, gp.flags, gp.last_hit_triangles, gp.weights, gp.evidx, gp.rng_counters))
    _, reused0, allocated0 = ctx.pool_stats()
    del gp
    ctx.synchronize()
    parked, _, _ = ctx.pool_stats()
    assert parked >= 300000 * 64
    gp2 = gpu.GPUPhotons(ph)                                  # same sizes: every array comes back from the pool
    _, reused1, allocated1 = ctx.pool_stats()
    assert allocated1 == allocated0 and reused1 - reused0 == 10
    ptrs2 = sorted(a.ptr for a in (gp2.pos, gp2.dir, gp2.pol, gp2.wavelengths, gp2.t, gp2.flags, gp2.last_hit_triangles, gp2.weights, gp2.evidx, gp2.rng_counters))
    assert ptrs2 == ptrs
    assert_bit_exact(gp2.get(), ph, 'photons through reused blocks')
    del gp2
    ctx.pool_trim()
    assert ctx.pool_stats()[0] == 0


def test_upload_stream_and_staged_copies_deliver_the_bytes(gpu):
    """chroma_memcpy_htod / chroma_upload above the staging threshold (8 MB: pinned ring, parallel staging) and below."""
    from chroma_amd.gpu.tools import GPUArray
    ctx = gpu.get_context()
    rng = np.random.default_rng(5)
    for n in (1000, (8 << 20) // 4 + 3, 30_000_001):
        a = rng.integers(0, 2 ** 32, size=n, dtype=np.uint32)
        for upload in (False, True):
            d = GPUArray(n, np.uint32, ctx)
            d.set(a, upload)
            assert np.array_equal(d.get(), a), (n, upload)


def test_simulation_with_prefetch_equals_simulation_without(gpu, tiny_geometry, oracle_mod):
    """Five event batches through Simulation.simulate with the prefetching loop and without: the same hits per event,
    bit for bit (photon ids -- hence random streams -- are handed out per batch in the same order), and the pool
    serves every batch after the second."""
    from chroma_amd import event
    from chroma_amd.sim import Simulation

    def events():
        for k in range(10):
            yield oracle_mod.generate_bomb(40000, seed=100 + k)

    results = {}
    for prefetch in (False, True):
        sim = Simulation(tiny_geometry, seed=77, prefetch=prefetch)
        ctx = gpu.get_context()
        out = []
        for ev in sim.simulate(events(), keep_photons_end=True, photons_per_batch=80000, max_steps=100):
            out.append((ev.id, ev.photons_end, ev.flat_hits))
        results[prefetch] = out
        if prefetch:
            parked, reused, allocated = ctx.pool_stats()
            assert reused >= 30                # (five batches of ten arrays: two sets allocated at most, the rest come from the pool)
    assert [r[0] for r in results[False]] == list(range(10)) == [r[0] for r in results[True]]
    for (ida, enda, hitsa), (idb, endb, hitsb) in zip(results[False], results[True]):
        assert_bit_exact(enda, endb, 'event %d with and without prefetch' % ida)
        assert len(hitsa) == len(hitsb) > 0
        oa, ob = np.lexsort((hitsa.t, hitsa.channel)), np.lexsort((hitsb.t, hitsb.channel))
        assert np.array_equal(hitsa.channel[oa], hitsb.channel[ob]) and np.array_equal(hitsa.t[oa].view(np.uint32), hitsb.t[ob].view(np.uint32))


def test_simulation_with_lanes_equals_one_batch_at_a_time(gpu, tiny_geometry, oracle_mod):
    """Simulation(lanes=3): three batches in flight at once, each on its own context from its own host thread.  Twelve
    batches of events of uneven sizes: the events come back in order with the photons and hits of the one-batch-at-a-time
    loop, bit for bit (a batch's photon-id block is reserved in batch order whatever lane runs it); the contexts of the
    lanes leave the process-wide current context alone; and small batches go through faster."""
    import time
    from chroma_amd.sim import Simulation

    sizes = [7000, 15000, 3000, 12000, 9000, 20000] * 6

    def events():
        for k, n in enumerate(sizes):
            yield oracle_mod.generate_bomb(n, seed=300 + k)

    results, seconds = {}, {}
    for lanes in (1, 3):
        sim = Simulation(tiny_geometry, seed=78, lanes=lanes)
        current = gpu.get_context()
        assert current is sim.context and len(sim._lanes) == lanes
        list(sim.simulate(events(), photons_per_batch=30000, max_steps=100))        # (warm the lanes' pools and working sets)
        sim.rng_states.next_photon_id = 0
        t0 = time.perf_counter()
        out = [(ev.id, ev.photons_end, ev.flat_hits) for ev in sim.simulate(events(), keep_photons_end=True, photons_per_batch=30000, max_steps=100)]
        seconds[lanes] = time.perf_counter() - t0
        results[lanes] = out
        assert gpu.get_context() is current
        del sim
    assert [r[0] for r in results[1]] == list(range(len(sizes))) == [r[0] for r in results[3]]
    for (ida, enda, hitsa), (idb, endb, hitsb) in zip(results[1], results[3]):
        assert len(enda) == sizes[ida]
        assert_bit_exact(enda, endb, 'event %d with one lane and with three' % ida)
        assert len(hitsa) == len(hitsb)
        oa, ob = np.lexsort((hitsa.t, hitsa.channel)), np.lexsort((hitsb.t, hitsb.channel))
        assert np.array_equal(hitsa.channel[oa], hitsb.channel[ob]) and np.array_equal(hitsa.t[oa].view(np.uint32), hitsb.t[ob].view(np.uint32))
    print('Simulation over %d photons in batches of 3e4: %.1f ms with one lane, %.1f ms with three' % (sum(sizes), 1e3 * seconds[1], 1e3 * seconds[3]))
