"""Parity of the HIP engine with the CPU oracle, through the C ABI, on a real MI355X.

Bar: history flags, hit triangle ids, draw counters and every float field BIT-EXACT against
the contract-math oracle (both sides execute the arithmetic of include/chroma_math.h); the
north star's looser float tolerance (1e-5 relative) is what separates the contract oracle from
the libm oracle, checked on the CPU in tests/test_oracle.py.
"""
import ctypes

import numpy as np
import pytest

from chroma_amd import event
from chroma_amd.event import Photons
from conftest import make_box_geometry, make_stress_geometry, bomb

pytestmark = pytest.mark.gpu

FIELDS = ('flags', 'last_hit_triangles', 'pos', 'dir', 'pol', 't', 'wavelengths', 'weights', 'evidx')


@pytest.fixture(scope='module')
def gpu():
    from chroma_amd import gpu as g
    ctx = g.create_cuda_context(0)
    yield g
    ctx.pop()


def assert_bit_exact(got, want, what=''):
    for name in FIELDS:
        a, b = getattr(got, name), getattr(want, name)
        same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
        assert same.all(), '%s: %s differs for %d of %d photons (first at %s)' % (
            what, name, np.count_nonzero(~same.reshape(len(a), -1).all(axis=1)), len(a), np.argwhere(~same)[0])


def run_both(gpu, oracle_mod, geometry, photons, seed=12345, max_steps=100, **kw):
    from chroma_amd.gpu.geometry import pack_geometry
    gg = gpu.GPUDetector(geometry) if hasattr(geometry, 'num_channels') else gpu.GPUGeometry(geometry)
    rng_states = gpu.get_rng_states(64 * 1024, seed=seed)
    gp = gpu.GPUPhotons(photons)
    stats = {}
    gpu.get_context().set_counting(True)
    gp.propagate(gg, rng_states, max_steps=max_steps, stats=stats, **kw)
    gpu.get_context().set_counting(False)
    got = gp.get()
    want, counters, ostats = oracle_mod.propagate(pack_geometry(geometry), photons, seed=seed, max_steps=max_steps,
                                                  nthreads=8, **kw)
    return gg, gp, got, want, counters, stats, ostats


def test_tiny_detector_full_histories(gpu, oracle_mod, tiny_geometry):
    """demo.tiny(), 1e4 isotropic 400 nm photons from the centre."""
    ph = oracle_mod.generate_bomb(10000, seed=20240502)
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, tiny_geometry, ph)
    assert_bit_exact(got, want, 'tiny')
    assert np.array_equal(gp.rng_counters.get(), counters)
    # the wide walk fetches 8 child entries (one 128-B line) per visit and tests fewer triangles
    # than the reference's walk; the roofline's algorithmic bytes use the engine's own counts
    assert stats['photon_steps'] == ostats['photon_steps']
    assert stats['launches'] == ostats['launches']
    assert 0 < stats['nodes_visited'] <= 2.0 * ostats['nodes_visited']
    assert 0 < stats['triangles_tested'] <= 1.3 * ostats['triangles_tested']
    assert (got.flags & event.TERMINAL_MASK != 0).all()
    assert 50 < np.count_nonzero(got.flags & event.SURFACE_DETECT) < 1000


def test_large_batch_uses_per_step_launches(gpu, oracle_mod, tiny_geometry):
    """> 8192 survivors: one step per launch with queue compaction (photon.py:225-252)."""
    ph = bomb(60000, 3, wavelength=400.0, wavelength_hi=800.0)
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, tiny_geometry, ph, max_steps=30)
    assert_bit_exact(got, want, 'tiny 60k')
    assert stats['launches'] == ostats['launches'] and stats['launches'] >= 2
    assert stats['photon_steps'] == ostats['photon_steps']
    # the wide walk fetches 8 child entries (one 128-B line) per visit and tests fewer triangles
    assert stats['nodes_visited'] <= 2.0 * ostats['nodes_visited']
    assert stats['triangles_tested'] <= 1.3 * ostats['triangles_tested']
    # the same batch with the other two ray casts -- one lane per ray over the wide tree, and the
    # reference tree in the reference's order: identical results
    for mode in ('pair', 'coop', 'wide', 'reference'):
        gpu.get_context().set_walk(mode)
        try:
            gp2 = gpu.GPUPhotons(ph)
            stats2 = {}
            gpu.get_context().set_counting(True)
            gp2.propagate(gg, gpu.get_rng_states(64 * 1024, seed=12345), max_steps=30, stats=stats2)
            gpu.get_context().set_counting(False)
        finally:
            gpu.get_context().set_walk('quad')
        assert_bit_exact(gp2.get(), want, 'tiny 60k, %s walk' % mode)
        assert stats2['photon_steps'] == ostats['photon_steps']
        if mode == 'reference':
            assert ostats['nodes_visited'] <= stats2['nodes_visited'] <= 1.3 * ostats['nodes_visited']


def test_every_surface_model_and_bulk_reemission(gpu, oracle_mod):
    """BASELINE.md C5 in miniature: scintillator + thin film + WLS + dichroic + default surface."""
    geo = make_stress_geometry()
    ph = bomb(40000, 6, wavelength=350.0)
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, geo, ph, seed=11)
    assert_bit_exact(got, want, 'stress')
    assert np.array_equal(gp.rng_counters.get(), counters)
    assert int(np.bitwise_or.reduce(got.flags)) & 0x3FE == 0x3FE
    assert (got.flags & event.NAN_ABORT).sum() == 0


@pytest.mark.parametrize('use_weights,scatter_first', [(True, 0), (True, 1), (False, 1), (False, -1)])
def test_weights_and_forced_scatter(gpu, oracle_mod, use_weights, scatter_first):
    geo = make_box_geometry(100.0)
    ph = bomb(20000, 8, wavelength=400.0)
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, geo, ph, seed=5, max_steps=5,
                                                          use_weights=use_weights, scatter_first=scatter_first)
    assert_bit_exact(got, want, 'weights=%s scatter_first=%d' % (use_weights, scatter_first))


def test_edge_inputs(gpu, oracle_mod, tiny_geometry):
    gg = gpu.GPUDetector(tiny_geometry)
    rng_states = gpu.get_rng_states(64, seed=1)
    # empty batch
    gp = gpu.GPUPhotons(Photons())
    gp.propagate(gg, rng_states)
    assert len(gp.get()) == 0 and len(gp.get_flat_hits(gg)) == 0
    # one photon; already-terminal photons are left untouched (propagate.cu:258)
    ph = bomb(5, 1)
    ph.flags[:] = [0, event.BULK_ABSORB, 0, event.NO_HIT, event.SURFACE_DETECT]
    ph.dir[1] *= 3.0                                      # not re-normalised for terminal photons
    gp = gpu.GPUPhotons(ph)
    gp.propagate(gg, rng_states, max_steps=10)
    out = gp.get()
    assert np.array_equal(out.dir[1], ph.dir[1]) and out.flags[1] == event.BULK_ABSORB
    assert out.flags[0] & event.TERMINAL_MASK and out.flags[2] & event.TERMINAL_MASK
    # NaN input -> NO_HIT | NAN_ABORT (propagate.cu:270-273)
    ph = bomb(4, 2)
    ph.pos[2, 1] = np.nan
    gp = gpu.GPUPhotons(ph)
    gp.propagate(gg, rng_states, max_steps=3)
    assert gp.get().flags[2] == (event.NO_HIT | event.NAN_ABORT)
    # photons outside the world pointing away: NO_HIT
    ph = Photons(np.full((3, 3), 1e6), np.tile([1.0, 0, 0], (3, 1)), np.tile([0, 1.0, 0], (3, 1)), np.full(3, 400.0))
    gp = gpu.GPUPhotons(ph)
    gp.propagate(gg, rng_states, max_steps=3)
    assert (gp.get().flags == event.NO_HIT).all()


def test_repeated_propagate_single_steps(gpu, oracle_mod):
    """Re-entrancy: 6 calls of max_steps=1 == oracle driven the same way (counters carried over)."""
    from chroma_amd.gpu.geometry import pack_geometry
    geo = make_stress_geometry()
    pk = pack_geometry(geo)
    ph = bomb(3000, 9, wavelength=350.0)
    gg = gpu.GPUDetector(geo)
    rng_states = gpu.get_rng_states(64, seed=21)
    gp = gpu.GPUPhotons(ph)
    cur, ctr = ph, None
    for _ in range(6):
        gp.propagate(gg, rng_states, max_steps=1)
        cur, ctr, _ = oracle_mod.propagate(pk, cur, seed=21, max_steps=1, rng_counters=ctr)
    assert_bit_exact(gp.get(), cur, 'single steps')
    assert np.array_equal(gp.rng_counters.get(), ctr)


def test_distance_to_mesh_vs_oracle(gpu, oracle_mod, tiny_geometry, tiny_packed):
    from chroma_amd.gpu.tools import to_gpu, GPUArray
    from chroma_amd.tools import from_film
    rng = np.random.default_rng(4)
    n = 50000
    o = rng.uniform(-1500, 1500, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:6] = [[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]]      # 1/d = inf paths
    o[:6] = 0
    gg = gpu.GPUDetector(tiny_geometry)
    ctx = gpu.get_context()
    dist = GPUArray(n, np.float32, ctx).fill(np.float32(np.nan))
    tri = GPUArray(n, np.int32, ctx)
    from chroma_amd import _lib
    d_o, d_d = to_gpu(o.reshape(-1), ctx), to_gpu(d.reshape(-1), ctx)      # keep the device arrays alive
    _lib.check(ctx._lib.chroma_distance_to_mesh(ctx.handle, gg.handle, n, d_o.ptr, d_d.ptr, dist.ptr, tri.ptr))
    wd, wt, _ = oracle_mod.distance_to_mesh(tiny_packed, o, d)
    gd, gt = dist.get(), tri.get()
    assert np.array_equal(gt, wt)
    assert np.array_equal(gd.view(np.uint32), wd.view(np.uint32))
    assert (gt >= 0).mean() > 0.9


def test_flat_hits_select_duplicate(gpu, oracle_mod, tiny_geometry):
    ph = oracle_mod.generate_bomb(20000, seed=77)
    gg = gpu.GPUDetector(tiny_geometry)
    rng_states = gpu.get_rng_states(64, seed=3)
    gp = gpu.GPUPhotons(ph)
    gp.propagate(gg, rng_states, max_steps=100)
    end = gp.get()
    # reference semantics of count/copy_photon_hits (propagate.cu:147-214), by hand
    det = (end.flags & event.SURFACE_DETECT) != 0
    tri = end.last_hit_triangles
    chan = np.full(len(end), -1)
    ok = det & (tri > -1)
    chan[ok] = tiny_geometry.solid_id_to_channel_index[tiny_geometry.solid_id[tri[ok]]]
    expect = np.flatnonzero(chan >= 0)
    hits = gp.get_flat_hits(gg)
    assert len(hits) == len(expect) > 0
    # order is unspecified (atomics): compare as sets keyed by position+time
    key = lambda p, idx: sorted(zip(p.pos[idx, 0].tolist(), p.pos[idx, 1].tolist(), p.t[idx].tolist()))
    assert key(hits, np.arange(len(hits))) == key(end, expect)
    lookup = {(x, y, t): c for x, y, t, c in zip(end.pos[expect, 0].tolist(), end.pos[expect, 1].tolist(), end.t[expect].tolist(), chan[expect].tolist())}
    assert all(lookup[(x, y, t)] == c for x, y, t, c in zip(hits.pos[:, 0].tolist(), hits.pos[:, 1].tolist(), hits.t.tolist(), hits.channel.tolist()))
    hitmap = gp.get_hits(gg)
    assert sum(len(v) for v in hitmap.values()) == len(hits)
    # per-channel reduction
    counts, earliest = gp.channel_hits(gg)
    assert np.array_equal(counts.get(), np.bincount(chan[expect], minlength=gg.nchannels).astype(np.uint32))
    e = earliest.get().view(np.float32)
    for c in np.unique(chan[expect])[:10]:
        assert e[c] == end.t[expect][chan[expect] == c].min()
    # select
    sel = gp.select(event.SURFACE_ABSORB).get()
    assert len(sel) == np.count_nonzero(end.flags & event.SURFACE_ABSORB)
    assert sorted(sel.t.tolist()) == sorted(end.t[(end.flags & event.SURFACE_ABSORB) != 0].tolist())
    # ncopies: clones get the same inputs but their own random streams
    small = ph[:500]
    gp2 = gpu.GPUPhotons(small, ncopies=3)
    assert len(gp2) == 1500
    before = gp2.get()
    for k in range(3):
        assert np.array_equal(before.dir[500 * k:500 * (k + 1)], small.dir)
    gp2.propagate(gg, rng_states, max_steps=20)
    after = gp2.get()
    assert not np.array_equal(after.flags[:500], after.flags[500:1000]) or not np.array_equal(after.t[:500], after.t[500:1000])
    assert len(list(gp2.iterate_copies())) == 3


def test_tracking_mode(gpu, oracle_mod):
    geo = make_stress_geometry()
    ph = bomb(300, 12, wavelength=350.0)
    gg = gpu.GPUDetector(geo)
    rng_states = gpu.get_rng_states(64, seed=2)
    gp = gpu.GPUPhotons(ph)
    ids, steps = gp.propagate(gg, rng_states, max_steps=8, track=True)
    assert len(ids) == len(steps) and len(ids[0]) == 300
    assert all(len(i) == len(s) for i, s in zip(ids, steps))
    assert len(ids[-1]) <= len(ids[0])
    # step by step against the oracle driven the same way (one launch per step, chroma/gpu/photon.py:218-238):
    # entry k holds the photons that ENTERED step k, in queue order, as they are AFTER it
    from chroma_amd.gpu.geometry import pack_geometry
    pk = pack_geometry(geo)
    cur, ctr = ph, None
    assert_bit_exact(steps[0], cur[ids[0]], 'track, before the first step')
    for k in range(1, len(ids)):
        alive_before = np.flatnonzero((cur.flags & event.TERMINAL_MASK) == 0) if k > 1 else np.arange(len(ph))
        assert sorted(ids[k].tolist()) == alive_before.tolist(), 'track: step %d did not get exactly the live photons' % k
        cur, ctr, _ = oracle_mod.propagate(pk, cur, seed=2, max_steps=1, rng_counters=ctr)
        assert_bit_exact(steps[k], cur[ids[k]], 'track, after step %d' % k)
    assert_bit_exact(gp.get(), cur, 'track, final arrays')


def test_simulation_api_ports_of_reference_tests(gpu):
    """Ports of test/test_propagation.py and test/test_rayleigh.py through Simulation."""
    from chroma_amd.sim import Simulation
    from chroma_amd.geometry import Geometry, Solid, vacuum
    from chroma_amd.make import box
    from chroma_amd.demo.optics import water
    from scipy import stats as sstats
    cube = Geometry(vacuum)
    cube.add_solid(Solid(box(100, 100, 100), vacuum, vacuum))
    sim = Simulation(cube, geant4_processes=0, seed=1)
    n = 10000
    rng = np.random.default_rng(0)
    d = np.zeros((n, 3)); axis = rng.integers(0, 3, n); d[np.arange(n), axis] = rng.choice([-1.0, 1.0], n)
    pol = np.zeros((n, 3)); ang = rng.uniform(0, 2 * np.pi, n)
    pol[np.arange(n), (axis + 1) % 3] = np.cos(ang); pol[np.arange(n), (axis + 2) % 3] = np.sin(ang)
    photons = Photons(np.zeros((n, 3)), d, pol, np.full(n, 400.0))
    end = next(sim.simulate([photons], keep_photons_end=True, max_steps=1)).photons_end
    for a in (end.pos, end.dir, end.pol, end.t, end.wavelengths):
        assert not np.isnan(a).any()
    end = next(sim.simulate([photons], keep_photons_end=True, max_steps=10)).photons_end
    assert (end.flags & event.NAN_ABORT).sum() == 0

    wbox = Geometry(water)
    wbox.add_solid(Solid(box(100, 100, 100), water, water))
    sim = Simulation(wbox, geant4_processes=0, seed=2)
    n = 2000000
    photons = Photons(np.zeros((n, 3)), np.tile([0.0, 0.0, 1.0], (n, 1)), np.tile([1.0, 0.0, 0.0], (n, 1)), np.full(n, 400.0))
    end = next(sim.simulate([photons], keep_photons_end=True, max_steps=1)).photons_end
    m = (end.flags & event.RAYLEIGH_SCATTER) != 0
    assert m.sum() > 1000
    theta = np.arccos(np.clip(end.dir[m, 2], -1, 1))
    hist, edges = np.histogram(theta, bins=20, range=(0, np.pi))
    cdf = lambda t: (4.0 / 3.0 - np.cos(t) - np.cos(t) ** 3 / 3.0) / (8.0 / 3.0)
    expect = m.sum() * np.diff(cdf(edges))
    assert sstats.chi2.sf(((hist - expect) ** 2 / expect).sum(), len(hist) - 1) > 1e-3


def test_simulation_batching_and_per_event_split(gpu, tiny_geometry):
    from chroma_amd.sim import Simulation
    sim = Simulation(tiny_geometry, geant4_processes=0, seed=5)
    events = [bomb(n, 100 + n) for n in (700, 1500, 300)]
    out = list(sim.simulate(events, keep_photons_end=True, photons_per_batch=2000, max_steps=50))
    assert [len(ev.photons_end) for ev in out] == [700, 1500, 300] and [ev.id for ev in out] == [0, 1, 2]
    for ev in out:
        det = (ev.photons_end.flags & event.SURFACE_DETECT) != 0
        assert len(ev.flat_hits) <= det.sum()
        assert sum(len(v) for v in ev.hits.values()) == len(ev.flat_hits)
        assert (ev.flat_hits.flags & event.SURFACE_DETECT != 0).all()
        # ev.hits comes from ONE sort of the event's hits by channel: per channel the photons the reference's mask
        # (chroma/sim.py:122-123: flat_hits[flat_hits.channel == channel]) selects, in the same order
        assert sorted(ev.hits) == sorted(int(c) for c in np.unique(ev.flat_hits.channel))
        for ch, got in ev.hits.items():
            want = ev.flat_hits[ev.flat_hits.channel == ch]
            assert len(got) == len(want) > 0 and (got.channel == ch).all()
            assert np.array_equal(got.t.view(np.uint32), want.t.view(np.uint32)) and np.array_equal(got.pos.view(np.uint32), want.pos.view(np.uint32))


def test_simulation_exact_switch_and_per_event_hits(gpu, oracle_mod, tiny_geometry):
    """Simulation(exact=True) propagates with the reference's own traversal loop for every ray (the walk 'literal') and
    leaves the context's walk as it was; on photons without erratic hits it gives what the default walk gives, photon
    for photon.  The events' hits come from ONE sort of the batch's hits by event index: the same hits, event by event,
    as masking the batch once per event (chroma/sim.py:118-121)."""
    from chroma_amd.sim import Simulation

    def events():
        return [oracle_mod.generate_bomb(n, seed=300 + n) for n in (9000, 15000, 4000)]
    results = {}
    for exact in (False, True):
        sim = Simulation(tiny_geometry, geant4_processes=0, seed=11, exact=exact)
        assert gpu.get_context().walk == 'quad'
        results[exact] = list(sim.simulate(events(), keep_photons_end=True, photons_per_batch=40000, max_steps=100))
        assert gpu.get_context().walk == 'quad'
    for a, b in zip(results[False], results[True]):
        assert_bit_exact(a.photons_end, b.photons_end, 'Simulation exact vs default, event %d' % a.id)
        assert len(a.flat_hits) == len(b.flat_hits) > 0
    # one batch, three events: each event's hits are exactly the detected photons of ITS photons_end that sit on a channel
    for ev in results[False]:
        end = ev.photons_end
        det = (end.flags & event.SURFACE_DETECT) != 0
        tri = end.last_hit_triangles
        chan = np.full(len(end), -1, dtype=np.int64)
        ok = det & (tri > -1)
        chan[ok] = tiny_geometry.solid_id_to_channel_index[tiny_geometry.solid_id[tri[ok]]]
        want_t = np.sort(end.t[chan >= 0])
        assert np.array_equal(np.sort(ev.flat_hits.t), want_t)
        assert (ev.flat_hits.evidx == ev.id).all()


def test_daq_matches_oracle_and_reference_test(gpu, oracle_mod, tiny_geometry, tiny_packed):
    """GPUDaq (chroma/gpu/daq.py + cuda/daq.cu) against the oracle's run_daq, bit for bit, and the
    intent of the reference's test/test_detector.py (time spread 1.2 ns, unit charge +- 0.1)."""
    from chroma_amd.gpu.geometry import pack_geometry
    ph = oracle_mod.generate_bomb(60000, seed=31)
    ph.t[:] = 100.0
    gg = gpu.GPUDetector(tiny_geometry)
    rng_states = gpu.get_rng_states(64, seed=9)
    gp = gpu.GPUPhotons(ph)
    gp.propagate(gg, rng_states, max_steps=100)
    end = gp.get()
    daq = gpu.GPUDaq(gg)
    for acquisition in range(2):
        daq.begin_acquire()
        daq.acquire(gp, rng_states)
        ch = daq.end_acquire().get()
        t, q, hist, hit = oracle_mod.run_daq(tiny_packed, end, daq._tables_host, daq.charge_unit, seed=9, acquisition=acquisition)
        assert hit.sum() > 10 and np.array_equal(ch.hit, hit)
        assert np.array_equal(ch.t.view(np.uint32), t.view(np.uint32))
        assert np.array_equal(ch.q.view(np.uint32), q.view(np.uint32))
        assert np.array_equal(ch.flags, hist)
        if acquisition == 0:
            first = ch.t.copy()
    assert not np.array_equal(first, ch.t)                 # a new acquisition draws new smearing
    # ndaq > 1: run_daq_many, copies side by side (chroma/gpu/daq.py:85-99, daq.cu:88-150)
    daq4 = gpu.GPUDaq(gg, ndaq=4)
    daq4.begin_acquire()
    daq4.acquire(gp, rng_states)
    many = daq4.end_acquire()
    t, q, hist, hit = oracle_mod.run_daq_many(tiny_packed, end, daq4._tables_host, daq4.charge_unit, seed=9, ndaq=4)
    ch4 = many.get()
    assert np.array_equal(ch4.hit, hit) and np.array_equal(ch4.t.view(np.uint32), t.view(np.uint32))
    assert np.array_equal(ch4.q.view(np.uint32), q.view(np.uint32)) and np.array_equal(ch4.flags, hist)
    copies = [c.get() for c in many.iterate_copies()]
    assert len(copies) == 4 and all(len(c.t) == gg.nchannels for c in copies)
    assert np.array_equal(copies[0].hit, copies[3].hit) and not np.array_equal(copies[0].t, copies[3].t)

    # test/test_detector.py: one photocathode box, single-photon events through Simulation(run_daq=True)
    from chroma_amd.sim import Simulation
    from chroma_amd.detector import Detector
    from chroma_amd.geometry import Solid, vacuum
    from chroma_amd.make import box
    from chroma_amd.demo.optics import r7081hqe_photocathode
    from chroma_amd.loader import create_geometry_from_obj
    cube = Detector(vacuum)
    cube.add_pmt(Solid(box(10.0, 10, 10), vacuum, vacuum, surface=r7081hqe_photocathode))
    cube.set_time_dist_gaussian(1.2, -6.0, 6.0)
    cube.set_charge_dist_gaussian(1.0, 0.1, 0.5, 1.5)
    sim = Simulation(create_geometry_from_obj(cube), geant4_processes=0, seed=4)
    n = 1
    photons = Photons(np.zeros((n, 3)), np.tile([0, 0, 1.0], (n, 1)), np.tile([1.0, 0, 0], (n, 1)), np.full(n, 400.0),
                      t=np.full(n, 100.0))
    times, charges = [], []
    for ev in sim.simulate((photons for _ in range(1500)), run_daq=True, max_steps=10):
        if ev.channels.hit[0]:
            times.append(ev.channels.t[0])
            charges.append(ev.channels.q[0])
    assert len(times) > 200
    assert abs(np.std(times) - 1.2) < 0.15
    assert abs(np.mean(charges) - 1.0) < 0.1 and np.std(charges) < 0.2


def test_reference_benchmark_harness(gpu, tiny_geometry):
    """chroma/benchmark.py's intersect / load_photons / propagate run and return sane rates."""
    from chroma_amd import benchmark
    gg = gpu.GPUDetector(tiny_geometry)
    for fn, args in ((benchmark.intersect, (gg, 3, 100000)), (benchmark.load_photons, (3, 100000)),
                     (benchmark.propagate, (gg, 3, 100000))):
        mean, std = fn(*args)
        assert mean > 1e5 and std >= 0


def test_chroma_sim_driver(gpu, tmp_path):
    """bin/chroma-sim: detector string -> events -> npz with flat hits and DAQ channels."""
    from chroma_amd.cli import main
    out = tmp_path / 'hits.npz'
    assert main(['@chroma_amd.demo.tiny', '-n', '3', '--nphotons', '20000', '-s', '7', '--run-daq', '-o', str(out)]) == 0
    f = np.load(out)
    assert int(f['nevents']) == 3
    for i in range(3):
        ch = f['ev%d/channel' % i]
        assert len(ch) > 50 and ch.max() < 53 and len(f['ev%d/t' % i]) == len(ch)
        assert f['ev%d/daq_hit' % i].sum() == len(np.unique(ch))       # every hit channel fired (weight 1)
    # the photons themselves (bin/chroma-sim:51-56) and their step-by-step tracks
    out2 = tmp_path / 'photons.npz'
    assert main(['@chroma_amd.demo.tiny', '-n', '2', '--nphotons', '1500', '-s', '7', '--max-steps', '6',
                 '--save-photons-beg', '--save-photons-end', '--track', '-o', str(out2)]) == 0
    f = np.load(out2)
    for i in range(2):
        beg, end = f['ev%d/photons_beg/pos' % i], f['ev%d/photons_end/pos' % i]
        assert beg.shape == (1500, 3) and end.shape == (1500, 3) and (beg == 0).all() and (end != 0).any()
        who, tpos, tt = f['ev%d/track/photon' % i], f['ev%d/track/pos' % i], f['ev%d/track/t' % i]
        assert len(who) == len(tpos) == len(tt) and set(np.unique(who)) == set(range(1500))
        first = np.r_[True, who[1:] != who[:-1]]
        assert (tpos[first] == 0).all()                                 # a track starts where the photon did
        last = np.r_[who[1:] != who[:-1], True]
        assert np.array_equal(tpos[last], end[who[last]])               # ... and ends where it ended
        assert (np.diff(tt)[~first[1:]] >= 0).all()                     # time runs forward along a track


def test_c_abi_rejects_bad_input(gpu, tiny_geometry):
    """Error behaviour at the boundary: a non-zero status + message instead of a crash."""
    import ctypes
    from chroma_amd import _lib
    from chroma_amd.gpu.geometry import pack_geometry
    ctx = gpu.get_context()
    lib = ctx._lib
    # geometry whose BVH points outside the node array
    pk = pack_geometry(make_box_geometry(50.0))
    nodes = pk.arrays['nodes']
    saved = nodes[0, 3]
    nodes[0, 3] = (3 << 28) | 5000
    handle = ctypes.c_void_p()
    rc = lib.chroma_geometry_create(ctx.handle, ctypes.byref(pk.desc), ctypes.byref(handle))
    assert rc != 0 and b'child range' in lib.chroma_last_error()
    nodes[0, 3] = saved
    # triangle index out of range
    tri = pk.arrays['triangles']
    saved = tri[0, 0]
    tri[0, 0] = 10 ** 6
    rc = lib.chroma_geometry_create(ctx.handle, ctypes.byref(pk.desc), ctypes.byref(handle))
    assert rc != 0 and b'vertex' in lib.chroma_last_error()
    tri[0, 0] = saved
    # null photon arrays
    gg = gpu.GPUDetector(tiny_geometry)
    empty = _lib.PhotonArrays()
    rc = lib.chroma_propagate(ctx.handle, gg.handle, ctypes.byref(empty), 10, 1, _lib.Rng(1, 0), 5, 0, 0, 0, None, None)
    assert rc != 0 and b'null' in lib.chroma_last_error()
    with pytest.raises(_lib.ChromaError):
        _lib.check(rc)
    # hits on a geometry without a channel map
    plain = gpu.GPUGeometry(make_box_geometry(50.0))
    gp = gpu.GPUPhotons(bomb(10, 1))
    with pytest.raises(_lib.ChromaError, match='channel map'):
        gp.get_flat_hits(plain)
    # copies keep their stream when propagated through a slice view
    small = bomb(300, 5)
    rng_a, rng_b = gpu.get_rng_states(1, seed=3), gpu.get_rng_states(1, seed=3)
    a = gpu.GPUPhotons(small, ncopies=2)
    a.propagate(gg, rng_a, max_steps=20)
    b = gpu.GPUPhotons(small, ncopies=2)
    for view in b.iterate_copies():
        view.propagate(gg, rng_b, max_steps=20)
    assert np.array_equal(a.get().flags, b.get().flags) and np.array_equal(a.get().t, b.get().t)


def _aimed_photons(geometry, origin, n_min):
    """Photons from ``origin`` aimed exactly at mesh vertices, edge midpoints and triangle centroids:
    a ray through a vertex or an edge meets several triangles at (often bit-for-bit) the same
    distance, which is where the ORDER of the triangle tests decides the reference's answer."""
    m = geometry.mesh
    v = m.vertices.astype(np.float64)
    t = m.triangles
    rng = np.random.default_rng(5)
    pick = rng.choice(len(t), size=min(len(t), 6000), replace=False)
    tri = v[t[pick]]                                           # [k][3][3]
    targets = np.concatenate([tri.reshape(-1, 3), 0.5 * (tri[:, 0] + tri[:, 1]), 0.5 * (tri[:, 1] + tri[:, 2]),
                              0.5 * (tri[:, 2] + tri[:, 0]), tri.mean(axis=1)])
    d = targets - np.asarray(origin, dtype=np.float64)
    d = d[np.linalg.norm(d, axis=1) > 1e-9]
    reps = int(np.ceil(n_min / float(len(d))))
    d = np.tile(d, (reps, 1))
    d /= np.linalg.norm(d, axis=1)[:, None]
    pol = np.cross(d, np.roll(d, 1, axis=1) + 1e-3)
    pol /= np.linalg.norm(pol, axis=1)[:, None]
    n = len(d)
    return Photons(np.tile(np.asarray(origin, dtype=float), (n, 1)), d, pol, np.full(n, 400.0))


@pytest.mark.parametrize('count', ['large', 'small'])
def test_exact_ties_follow_the_reference_test_order(gpu, oracle_mod, tiny_geometry, count):
    """Rays through vertices and edges: every walk (8 lanes per ray over the SAH tree, one lane per
    ray, the reference tree in the reference's order; per-step launches and the fused tail) returns
    the triangle the reference's test order picks."""
    ph = _aimed_photons(tiny_geometry, (0.0, 0.0, 0.0), 20000 if count == 'large' else 3000)
    if count == 'small':
        ph = ph[:3000]                                        # fewer than 8192: the fused tail kernel from step 0
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, tiny_geometry, ph, max_steps=4)
    assert_bit_exact(got, want, 'aimed rays')
    # the case is not vacuous: some of these rays do hit several triangles at one distance
    from chroma_amd.gpu.geometry import pack_geometry
    dist, tri, _ = oracle_mod.distance_to_mesh(pack_geometry(tiny_geometry), ph.pos[:4000], ph.dir[:4000])
    assert (tri >= 0).mean() > 0.9
    for mode in ('pair', 'coop', 'wide', 'reference'):
        gpu.get_context().set_walk(mode)
        try:
            gp2 = gpu.GPUPhotons(ph)
            gp2.propagate(gg, gpu.get_rng_states(64 * 1024, seed=12345), max_steps=4)
        finally:
            gpu.get_context().set_walk('quad')
        assert_bit_exact(gp2.get(), want, 'aimed rays, %s walk' % mode)


def _edge_photons():
    """Terminal, NaN, exactly axis-parallel, outside-the-world and repeated-last-hit photons in a batch of 24000."""
    ph = bomb(24000, 17)
    n = len(ph)
    ph.flags[100:200] = event.BULK_ABSORB                      # terminal: untouched
    ph.dir[100:200] *= 2.5
    ph.pos[300, 2] = np.nan                                    # NaN guard
    ph.dir[301, 0] = np.nan
    axes = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [-1, 0, 0], [0, -1, 0], [0, 0, -1],
                     [1, 1, 0], [0, 1, 1], [1, 0, -1]], dtype=float)
    ph.dir[400:400 + 9 * 40] = np.tile(axes, (40, 1))         # exactly parallel to one or two axes (1/d = inf)
    ph.pos[400:400 + 9 * 40] += np.repeat(np.linspace(-300, 300, 40), 9)[:, None] * np.array([0.3, 0.7, -0.2])
    ph.pos[1000:1100] = 1e6                                    # outside the world box
    ph.dir[1000:1050] = [1.0, 0, 0]                            # ... pointing away
    ph.dir[1050:1100] = [-1.0, -1.0, -1.0]                     # ... pointing at it
    ph.last_hit_triangles[2000:2200] = np.arange(200)          # a last hit that is not on the ray: no effect
    return ph


def test_edge_inputs_in_a_large_batch(gpu, oracle_mod, tiny_geometry):
    """Terminal, NaN, exactly axis-parallel, outside-the-world and repeated-last-hit photons inside a
    batch large enough for per-step launches (>= 8192 alive): same as the oracle, bit for bit."""
    ph = _edge_photons()
    n = len(ph)
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, tiny_geometry, ph, max_steps=20)
    assert_bit_exact(got, want, 'edge inputs, large batch')
    assert np.array_equal(gp.rng_counters.get(), counters)
    assert (got.flags[100:200] == event.BULK_ABSORB).all() and np.array_equal(got.dir[100:200], ph.dir[100:200])
    assert got.flags[300] == (event.NO_HIT | event.NAN_ABORT) and got.flags[301] == (event.NO_HIT | event.NAN_ABORT)
    assert (got.flags[1000:1050] == event.NO_HIT).all()
    assert stats['launches'] == ostats['launches'] and stats['photon_steps'] == ostats['photon_steps']


def test_axis_plane_photons_stay_out_of_the_fast_walk_step_after_step(gpu, oracle_mod):
    """A mirror box in vacuum: photons that start inside a coordinate plane (one direction component exactly 0, so
    1/d is infinite: not a ray for the fast walk) are reflected inside that plane at every step.  At every step,
    then, their ray record -- written by k_load_working for the first step, by k_physics for the later ones -- says
    "not to be cast", and k_raycast_quad has to settle the slot (hit entry + retry list) when it meets it.
    20 000 photons keep every step a per-step launch.  (Exactly axis-parallel photons would not do: the reference's
    specular reflection at normal incidence divides 0 by 0, photon.h:620-624, and they end as NaN aborts.)"""
    from chroma_amd.geometry import Surface, vacuum
    mirror = Surface('mirror')
    mirror.set('reflect_specular', 1.0)
    geometry = make_box_geometry(200.0, material=vacuum, surface=mirror)
    ph = bomb(20000, 5)
    planes = np.array([[1, 1, 0], [0, 1, -1], [1, 0, 1], [-1, 2, 0]], dtype=float)
    ph.dir[:8000] = np.tile(planes / np.linalg.norm(planes, axis=1)[:, None], (2000, 1))
    ph.pos[:8000] = np.random.RandomState(1).uniform(-90, 90, (8000, 3))
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, geometry, ph, max_steps=12)
    assert_bit_exact(got, want, 'mirror box')
    assert np.array_equal(gp.rng_counters.get(), counters)
    assert stats['launches'] == ostats['launches'] == 12 and stats['photon_steps'] == ostats['photon_steps']
    terminal = event.NO_HIT | event.BULK_ABSORB | event.SURFACE_DETECT | event.SURFACE_ABSORB | event.NAN_ABORT
    alive = (got.flags[:8000] & terminal) == 0
    assert alive.mean() > 0.95 and (got.flags[:8000][alive] & event.REFLECT_SPECULAR).all()
    assert (np.abs(got.dir[:8000][alive]).min(axis=1) == 0).mean() > 0.99  # still inside their plane after 12 reflections


@pytest.mark.parametrize('tail', ['split', 'fused'])
def test_tail_modes_give_the_same_photons(gpu, oracle_mod, tiny_geometry, tail):
    """chroma_set_tail: per-step launch sets to the end, and the reference's own launch shape (the
    lane-per-photon kernel for every launch), against the oracle on the edge-input batch -- run to the
    end, and cut off after 3 steps with photons still alive (they go back to the caller's arrays)."""
    ph = _edge_photons()
    gpu.get_context().set_tail(tail)
    try:
        for max_steps in (20, 3):
            gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, tiny_geometry, ph, max_steps=max_steps)
            assert_bit_exact(got, want, 'edge inputs, %s tail, max_steps %d' % (tail, max_steps))
            assert np.array_equal(gp.rng_counters.get(), counters)
            assert stats['launches'] == ostats['launches'] and stats['photon_steps'] == ostats['photon_steps']
        assert ((got.flags & event.TERMINAL_MASK) == 0).sum() > 100
    finally:
        gpu.get_context().set_tail('coop')


def test_large_batch_properties(gpu, oracle_mod, tiny_geometry):
    """2e6 photons (too many for the oracle inside a test; tools/parity_sweep.py does that offline):
    every ray cast -- 4, 8 or 1 lane per ray over the SAH tree, the reference tree in the reference's
    order -- gives the same photons bit for bit, a repeated run is identical, and a 1e5 sample of the
    batch, propagated alone by the ORACLE with the same per-photon streams, has the same ray-cast
    answer for its first step."""
    n = 2_000_000
    ph = oracle_mod.generate_bomb(n, seed=77, wavelength_lo=400.0, wavelength_hi=650.0)
    gg = gpu.GPUDetector(tiny_geometry)
    results = {}
    for mode in ('quad', 'quad', 'pair', 'coop', 'wide', 'reference'):
        gpu.get_context().set_walk(mode)
        try:
            gp = gpu.GPUPhotons(ph)
            gp.propagate(gg, gpu.get_rng_states(64, seed=9), max_steps=100)
            out = gp.get()
        finally:
            gpu.get_context().set_walk('quad')
        if mode in results:
            assert_bit_exact(out, results[mode], 'repeat of %s' % mode)
        results[mode] = out
    for mode in ('pair', 'coop', 'wide', 'reference'):
        assert_bit_exact(results[mode], results['quad'], '%s vs quad' % mode)
    out = results['quad']
    assert ((out.flags & event.TERMINAL_MASK) != 0).mean() > 0.999
    # first step of a sample against the oracle: same triangle for the same ray
    sample = ph[:100_000]
    gp = gpu.GPUPhotons(sample)
    gp.propagate(gg, gpu.get_rng_states(64, seed=9), max_steps=1)
    want, _, _ = oracle_mod.propagate(pack_geometry_cached(tiny_geometry), sample, seed=9, max_steps=1)
    assert_bit_exact(gp.get(), want, 'first step of the sample')


_packed_cache = {}


def pack_geometry_cached(geometry):
    from chroma_amd.gpu.geometry import pack_geometry
    if id(geometry) not in _packed_cache:
        _packed_cache[id(geometry)] = pack_geometry(geometry)
    return _packed_cache[id(geometry)]


def test_first_launch_is_decided_on_the_array_size(gpu, oracle_mod, tiny_geometry):
    """A batch of 20 000 photons of which only 5 000 are still alive: the reference decides its FIRST launch on
    pos.size (terminal photons included: one step, chroma/gpu/photon.py:207,227) and the next one on the
    survivors (< 8192: all remaining steps), and every launch re-normalises dir/pol on load
    (propagate.cu:248,250) -- two re-normalisations, visible in the last bits.  Per-step launches, the fused
    tail and the reference's own launch shape agree with the oracle's restatement of that loop."""
    ph = bomb(20000, 23)
    ph.dir *= np.linspace(0.7, 1.3, len(ph))[:, None]          # not normalised: every re-normalisation shows
    ph.flags[5000:] = event.BULK_ABSORB
    for tail in ('coop', 'split', 'fused'):
        gpu.get_context().set_tail(tail)
        try:
            gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, tiny_geometry, ph, max_steps=20)
        finally:
            gpu.get_context().set_tail('coop')
        assert_bit_exact(got, want, 'mostly terminal batch, %s tail' % tail)
        assert np.array_equal(gp.rng_counters.get(), counters)
        assert stats['launches'] == ostats['launches'] >= 2, (tail, stats['launches'], ostats['launches'])
    assert (got.flags[5000:] == event.BULK_ABSORB).all() and np.array_equal(got.dir[5000:], ph.dir[5000:].astype(np.float32))


def test_kernels_by_name_and_the_chroma_import_name(oracle_mod, tiny_geometry):
    """Host code written against the reference: ``import chroma``, kernels looked up by name through
    get_cu_module / GPUFuncs (chroma/gpu/tools.py:14-54) and called with the reference kernels' positional
    arguments (chroma/gpu/photon.py:58-63,232-241).  Same photons as GPUPhotons.propagate step by step."""
    import chroma
    from chroma import gpu as cgpu
    from chroma.gpu import get_cu_module, GPUFuncs, cuda_options, GPUPhotons, to_float3
    from chroma.gpu.tools import GPUArray
    import chroma_amd.gpu
    assert cgpu is chroma_amd.gpu and chroma.event.Photons is Photons
    ctx = cgpu.create_cuda_context(0)
    try:
        gg = cgpu.GPUDetector(tiny_geometry)
        funcs = GPUFuncs(get_cu_module('propagate.cu', options=cuda_options))
        ph = oracle_mod.generate_bomb(3000, seed=4)
        n = len(ph)
        f3 = lambda a: cgpu.to_gpu(to_float3(a), ctx)
        pos, dir_, pol = f3(ph.pos), f3(ph.dir), f3(ph.pol)
        wl, t, w = (cgpu.to_gpu(ph.wavelengths.astype(np.float32), ctx), cgpu.to_gpu(ph.t.astype(np.float32), ctx),
                    cgpu.to_gpu(ph.weights.astype(np.float32), ctx))
        flags, last, evidx = (cgpu.to_gpu(ph.flags.astype(np.uint32), ctx), cgpu.to_gpu(ph.last_hit_triangles.astype(np.int32), ctx),
                              cgpu.to_gpu(ph.evidx.astype(np.uint32), ctx))
        rng_states = cgpu.get_rng_states(64 * 1024, seed=6)
        # the reference's own launch loop (gpu/photon.py:225-252) for a batch below 8192: one launch, all steps
        in_q = cgpu.to_gpu(np.arange(n + 1, dtype=np.uint32), ctx)          # slot 0 unused, then ids 0..n-1 shifted by the [1:] view
        in_q.set(np.r_[0, np.arange(n)].astype(np.uint32))
        out_q = cgpu.to_gpu(np.r_[1, np.zeros(n)].astype(np.uint32), ctx)
        funcs.propagate(np.int32(0), np.int32(n), in_q[1:], out_q, rng_states, pos, dir_, wl, pol, t, flags, last, w, evidx,
                        np.int32(20), np.int32(0), np.int32(0), gg.gpudata, block=(64, 1, 1), grid=(n // 64 + 1, 1))
        want, _, _ = oracle_mod.propagate(pack_geometry_cached(tiny_geometry), ph, seed=6, max_steps=20)
        assert np.array_equal(flags.get(), want.flags) and np.array_equal(last.get(), want.last_hit_triangles)
        assert np.array_equal(t.get().view(np.uint32), want.t.view(np.uint32))
        # count_photons adds to its counter, like the kernel
        counter = cgpu.to_gpu(np.array([5], dtype=np.uint32), ctx)
        funcs.count_photons(np.int32(0), np.int32(n), np.uint32(event.SURFACE_ABSORB), counter, flags, block=(64, 1, 1), grid=(n // 64 + 1, 1))
        assert int(counter.get()[0]) == 5 + np.count_nonzero(want.flags & event.SURFACE_ABSORB)
        with pytest.raises(AttributeError):
            funcs.no_such_kernel
        with pytest.raises(KeyError):
            get_cu_module('pdf.cu')
        mesh = GPUFuncs(get_cu_module('mesh.h'))
        o = cgpu.to_gpu(np.zeros(3 * 100, dtype=np.float32), ctx)
        d = cgpu.to_gpu(np.tile([0.0, 0.0, 1.0], 100).astype(np.float32), ctx)
        dist = GPUArray(100, np.float32, ctx).fill(np.float32(-1))
        mesh.distance_to_mesh(np.int32(100), o, d, gg.gpudata, dist, block=(64, 1, 1), grid=(2, 1))
        assert (dist.get() > 1000).all()
    finally:
        ctx.pop()


@pytest.mark.gpu
def test_every_ray_of_a_launch_is_cast_once_at_the_sizes_where_the_work_claims_change(gpu, oracle_mod, tiny_geometry, tiny_packed):
    """A persistent ray-cast wave owns part of its rays by position and takes the rest from a counter (WorkClaim,
    csrc/kernel_step_control.h); what it owns depends on the launch size against the grid (6144 waves of 16 rays) and on the
    claim size (16 rays, 64 above 4 * 64 * 6144).  One pool of rays checked against the oracle once; then launches of every
    size around those boundaries, for the default and the exact walk, must return the pool's first n results -- every ray
    cast, none twice (a ray cast by nobody keeps the NaN its slot was filled with)."""
    from chroma_amd.gpu.tools import to_gpu, GPUArray
    from chroma_amd import _lib
    ctx = gpu.get_context()
    waves = 6144
    sizes = [1, 15, 16, 17, 1000, 16 * waves - 1, 16 * waves, 16 * waves + 1, 16 * waves + 16, 2 * 16 * waves + 5, 300001,
             4 * 64 * waves, 4 * 64 * waves + 1, 4 * 64 * waves + 64 * waves + 77]
    nmax = max(sizes)
    rng = np.random.default_rng(41)
    o = rng.uniform(-1500, 1500, (nmax, 3)).astype(np.float32)
    d = rng.normal(size=(nmax, 3)).astype(np.float32)
    gg = gpu.GPUDetector(tiny_geometry)
    d_o, d_d = to_gpu(o.reshape(-1), ctx), to_gpu(d.reshape(-1), ctx)
    check = np.unique(np.concatenate([np.arange(0, 40000), np.arange(nmax - 40000, nmax), rng.integers(0, nmax, 40000),
                                      np.concatenate([np.arange(max(0, s - 40), min(nmax, s + 40)) for s in sizes])]))
    wd, wt, _ = oracle_mod.distance_to_mesh(tiny_packed, o[check], d[check])
    full = None
    for n in sorted(sizes, reverse=True):
        dist = GPUArray(n, np.float32, ctx).fill(np.float32(np.nan))
        tri = GPUArray(n, np.int32, ctx).fill(np.int32(-7))
        _lib.check(ctx._lib.chroma_distance_to_mesh(ctx.handle, gg.handle, n, d_o.ptr, d_d.ptr, dist.ptr, tri.ptr))
        gd, gt = dist.get(), tri.get()
        if full is None:
            full = (gd, gt)
            assert np.array_equal(gt[check], wt)
            assert np.array_equal(gd[check].view(np.uint32), wd.view(np.uint32))
            assert (gt != -7).all()
        else:
            assert np.array_equal(gt, full[1][:n]), '%d rays' % n
            assert np.array_equal(gd.view(np.uint32), full[0][:n].view(np.uint32)), '%d rays' % n
    # the exact walk's kernel takes its rays the same way: one step of n photons, exact against default (equal on this geometry
    # for random photons: the walks differ on aimed rays of one class only, tests/test_gpu_literal.py)
    for n in (16 * waves - 1, 16 * waves + 1, 2 * 16 * waves + 5, 4 * 64 * waves + 1):
        ph = oracle_mod.generate_bomb(n, seed=n)
        ends = []
        for exact in (False, True):
            gp = gpu.GPUPhotons(ph)
            gp.propagate(gg, gpu.get_rng_states(64, seed=3), max_steps=1, exact=exact)      # (the same draws for both)
            ends.append(gp.get())
        assert_bit_exact(ends[0], ends[1], 'one step of %d photons, exact walk against default' % n)
        assert (ends[1].last_hit_triangles >= 0).mean() > 0.5
