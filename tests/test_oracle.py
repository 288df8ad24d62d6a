"""The CPU oracle on its own: the reference's statistical tests restated (the only tests the
reference holds for the physics), determinism, re-entrancy, and the libm cross-check."""
import numpy as np
import pytest
from scipy import stats

from chroma_amd import event
from chroma_amd.event import Photons
from chroma_amd.gpu.geometry import pack_geometry
from conftest import make_box_geometry, make_stress_geometry, bomb


@pytest.fixture(scope='module')
def water_box():
    return pack_geometry(make_box_geometry(100.0))


@pytest.fixture(scope='module')
def vacuum_box():
    from chroma_amd.geometry import vacuum
    return pack_geometry(make_box_geometry(100.0, material=vacuum))


def axis_photons(n, rng):
    """test/test_propagation.py:27-35: photons from the centre along +-x, +-y, +-z with random
    transverse polarisation."""
    d = np.zeros((n, 3))
    axis = rng.integers(0, 3, n)
    d[np.arange(n), axis] = rng.choice([-1.0, 1.0], n)
    pol = np.zeros((n, 3))
    ang = rng.uniform(0, 2 * np.pi, n)
    pol[np.arange(n), (axis + 1) % 3] = np.cos(ang)
    pol[np.arange(n), (axis + 2) % 3] = np.sin(ang)
    return Photons(np.zeros((n, 3)), d, pol, np.full(n, 400.0))


def test_propagation_no_nan_no_abort(oracle_mod, vacuum_box):
    """Restates test/test_propagation.py:12-57 (testAbort / no NaN)."""
    rng = np.random.default_rng(0)
    ph = axis_photons(10000, rng)
    out, _, _ = oracle_mod.propagate(vacuum_box, ph, seed=1, max_steps=1)
    for a in (out.pos, out.dir, out.pol, out.t, out.wavelengths):
        assert not np.isnan(a).any()
    out, _, _ = oracle_mod.propagate(vacuum_box, ph, seed=1, max_steps=10)
    assert (out.flags & event.NAN_ABORT).sum() == 0


def test_rayleigh_angular_distribution(oracle_mod, water_box):
    """Restates test/test_rayleigh.py:13-54 with SciPy instead of ROOT: the angle between the
    initial and final direction of Rayleigh-scattered photons follows (1 + cos^2) sin."""
    n = 100000
    ph = Photons(np.zeros((n, 3)), np.tile([0.0, 0.0, 1.0], (n, 1)), np.tile([1.0, 0.0, 0.0], (n, 1)), np.full(n, 400.0))
    # water scatters once per ~40 m, so a 100 mm box gives only ~80 natural scatters per 1e5
    # photons (what the reference's test fits); scatter_first=1 forces one for every photon
    out, _, _ = oracle_mod.propagate(water_box, ph, seed=3, max_steps=1, scatter_first=1)
    m = (out.flags & event.RAYLEIGH_SCATTER) != 0
    assert m.sum() > 0.4 * n      # the forced-scatter loop gives up after 1000 redraws (photon.h:211)
    cos_t = np.clip(out.dir[m] @ np.array([0.0, 0.0, 1.0]), -1, 1)
    theta = np.arccos(cos_t)
    hist, edges = np.histogram(theta, bins=25, range=(0, np.pi))
    cdf = lambda t: (4.0 / 3.0 - np.cos(t) - np.cos(t) ** 3 / 3.0) / (8.0 / 3.0)     # integral of (1+cos^2) sin
    expect = m.sum() * np.diff(cdf(edges))
    chi2 = ((hist - expect) ** 2 / expect).sum()
    assert stats.chi2.sf(chi2, len(hist) - 1) > 1e-3
    # polarisation stays transverse and unit
    assert np.allclose(np.linalg.norm(out.pol[m], axis=1), 1, atol=1e-5)
    assert np.abs((out.pol[m] * out.dir[m]).sum(1)).max() < 1e-4


def test_deterministic_and_thread_independent(oracle_mod, tiny_packed):
    ph = oracle_mod.generate_bomb(3000, seed=9)
    a, ca, sa = oracle_mod.propagate(tiny_packed, ph, seed=12345, max_steps=100, nthreads=1)
    b, cb, sb = oracle_mod.propagate(tiny_packed, ph, seed=12345, max_steps=100, nthreads=4)
    for name in ('pos', 'dir', 'pol', 't', 'wavelengths', 'flags', 'last_hit_triangles', 'weights'):
        assert np.array_equal(getattr(a, name), getattr(b, name)), name
    assert np.array_equal(ca, cb) and sa['nodes_visited'] == sb['nodes_visited']
    # the photon-id base selects the stream: shifting ids == shifting photons
    c, _, _ = oracle_mod.propagate(tiny_packed, ph[1000:2000], seed=12345, photon_id_base=1000, max_steps=100)
    assert np.array_equal(c.flags, a.flags[1000:2000]) and np.array_equal(c.pos, a.pos[1000:2000])
    d, _, _ = oracle_mod.propagate(tiny_packed, ph, seed=54321, max_steps=100)
    assert not np.array_equal(d.flags, a.flags)


def test_the_launch_policy_ties_the_last_photons_of_a_batch_to_their_batch(oracle_mod):
    """Why the bits of a job sharded over GPUs can depend on the sharding, although photons do not interact and every photon
    has the random stream of its global id: the reference runs one launch per step while at least 8192 photons of the BATCH are
    alive and one launch for all remaining steps after that (chroma/gpu/photon.py:227-230), and every launch re-normalises
    dir / pol on load (chroma/cuda/propagate.cu:248,250).  So the photons that outlive the switch are re-normalised or not
    depending on how many OTHER photons their batch still holds.  The oracle restates that policy (the engine follows it:
    tests/test_gpu_parity.py); here: a batch against its two halves.  Batches under 8192 photons never switch
    (test_deterministic_and_thread_independent), and with weights every step runs in one launch whatever the count."""
    pk = pack_geometry(make_stress_geometry())
    n = 40000
    ph = oracle_mod.generate_bomb(n, seed=21, wavelength_lo=350.0)
    kw = dict(seed=77, max_steps=100, nthreads=4)
    whole, cw, _ = oracle_mod.propagate(pk, ph, **kw)
    lo, cl, _ = oracle_mod.propagate(pk, ph[:n // 2], photon_id_base=0, **kw)
    hi, ch, _ = oracle_mod.propagate(pk, ph[n // 2:], photon_id_base=n // 2, **kw)
    halves_pos = np.concatenate([lo.pos, hi.pos]).view(np.uint32)
    halves_flags = np.concatenate([lo.flags, hi.flags])
    same = (halves_pos == whole.pos.view(np.uint32)).all(axis=1) & (halves_flags == whole.flags)
    # most photons end before either batch switches: the same bits; the stragglers differ in the last places, hardly ever in history
    assert 0.5 < same.mean() < 1.0, same.mean()
    assert (halves_flags == whole.flags).mean() > 0.995
    m = halves_flags == whole.flags
    assert np.allclose(np.concatenate([lo.pos, hi.pos])[m], whole.pos[m], rtol=1e-4, atol=1e-2)
    # with weights the policy does not look at the count: a batch and its halves give the same bits
    kw = dict(seed=77, max_steps=10, nthreads=4, use_weights=True)
    whole, cw, _ = oracle_mod.propagate(pk, ph, **kw)
    lo, cl, _ = oracle_mod.propagate(pk, ph[:n // 2], photon_id_base=0, **kw)
    hi, ch, _ = oracle_mod.propagate(pk, ph[n // 2:], photon_id_base=n // 2, **kw)
    for name in ('pos', 'dir', 'pol', 't', 'wavelengths', 'weights'):
        assert np.array_equal(np.concatenate([getattr(lo, name), getattr(hi, name)]).view(np.uint32), getattr(whole, name).view(np.uint32)), name
    assert np.array_equal(np.concatenate([lo.flags, hi.flags]), whole.flags) and np.array_equal(np.concatenate([cl, ch]), cw)


def test_single_stepping_equals_one_call(oracle_mod):
    """propagate() may be called repeatedly (chroma/gpu/photon.py:199-200): ten calls of one
    step give the histories of one call of ten steps, thanks to the per-photon draw counter.
    Floats agree to rounding only: every call re-normalises dir/pol on load (propagate.cu:248,250)."""
    pk = pack_geometry(make_stress_geometry())
    ph = bomb(4000, 5, wavelength=350.0)
    full, cfull, _ = oracle_mod.propagate(pk, ph, seed=7, max_steps=10)
    cur, ctr = ph, None
    for _ in range(10):
        cur, ctr, _ = oracle_mod.propagate(pk, cur, seed=7, max_steps=1, rng_counters=ctr)
    assert np.array_equal(cur.flags, full.flags) and np.array_equal(ctr, cfull)
    assert np.allclose(cur.pos, full.pos, rtol=1e-5, atol=1e-3) and np.allclose(cur.t, full.t, rtol=1e-5, atol=1e-4)


def test_every_physics_branch_is_reached(oracle_mod):
    pk = pack_geometry(make_stress_geometry())
    ph = bomb(60000, 6, wavelength=350.0)
    out, _, st = oracle_mod.propagate(pk, ph, seed=11, max_steps=100, nthreads=4)
    assert int(np.bitwise_or.reduce(out.flags)) & 0x3FE == 0x3FE        # bits 1..9 (the world box catches escapes)
    assert (out.flags & event.NAN_ABORT).sum() == 0
    assert len(np.unique(out.flags)) > 40
    assert (out.flags & event.TERMINAL_MASK != 0).all()
    re = (out.flags & (event.BULK_REEMIT | event.SURFACE_REEMIT)) != 0
    assert out.wavelengths[re].mean() > 380                               # re-emission shifts to 400-500 nm
    assert st['stack_overflows'] == 0


def test_weights_and_scatter_first(oracle_mod, water_box):
    ph = bomb(20000, 8, wavelength=400.0)
    out, _, _ = oracle_mod.propagate(water_box, ph, seed=5, max_steps=5, use_weights=True, scatter_first=1)
    assert ((out.flags & event.RAYLEIGH_SCATTER) != 0).mean() > 0.4       # forced scatter (1000-redraw cap)
    assert (out.weights < 1).all() and (out.weights > 0).all()
    assert (out.flags & event.BULK_ABSORB).sum() == 0                      # weights mode never absorbs in the bulk
    out2, _, _ = oracle_mod.propagate(water_box, ph, seed=5, max_steps=1, scatter_first=-1)
    assert ((out2.flags & event.RAYLEIGH_SCATTER) != 0).sum() == 0        # scatter prevented


def test_libm_build_agrees(oracle_mod, tiny_packed):
    """Independent check that nothing physical hinges on the contract's polynomials."""
    ph = oracle_mod.generate_bomb(20000, seed=4)
    a, _, _ = oracle_mod.propagate(tiny_packed, ph, seed=12345, max_steps=100, nthreads=4)
    b, _, _ = oracle_mod.propagate(tiny_packed, ph, seed=12345, max_steps=100, nthreads=4, variant='libm')
    same = a.flags == b.flags
    assert same.mean() > 0.999
    assert np.array_equal(a.last_hit_triangles[same], b.last_hit_triangles[same])
    scale = np.maximum(np.linalg.norm(a.pos[same], axis=1), 1.0)[:, None]
    assert (np.abs(a.pos[same] - b.pos[same]) / scale).max() < 1e-5
    assert np.allclose(a.t[same], b.t[same], rtol=1e-5, atol=1e-5)


def test_ray_cast_basics(oracle_mod, vacuum_box):
    o = np.zeros((6, 3), dtype=np.float32)
    d = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], dtype=np.float32)
    dist, tri, _ = oracle_mod.distance_to_mesh(vacuum_box, o, d)         # axis-aligned: 1/d = inf on two axes
    assert np.allclose(dist, 50.0) and (tri >= 0).all()
    dist, tri, _ = oracle_mod.distance_to_mesh(vacuum_box, o + np.float32(200.0), d[:1])   # outside, pointing away
    assert tri[0] == -1 and np.isnan(dist[0])


def test_daq_oracle_time_and_charge_spread(oracle_mod):
    """Intent of the reference's test/test_detector.py on the CPU: a detected photon's channel time is
    smeared by the time CDF (1.2 ns here) and its charge follows the charge CDF (1.0 +- 0.1)."""
    from chroma_amd.detector import Detector
    from chroma_amd.geometry import Solid, vacuum
    from chroma_amd.make import box
    from chroma_amd.demo.optics import r7081hqe_photocathode
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.daq import _padded_cdf
    cube = Detector(vacuum)
    cube.add_pmt(Solid(box(10.0, 10, 10), vacuum, vacuum, surface=r7081hqe_photocathode))
    cube.set_time_dist_gaussian(1.2, -6.0, 6.0)
    cube.set_charge_dist_gaussian(1.0, 0.1, 0.5, 1.5)
    geo = create_geometry_from_obj(cube)
    pk = pack_geometry(geo)
    n = 4000
    ph = Photons(np.zeros((n, 3)), np.tile([0, 0, 1.0], (n, 1)), np.tile([1.0, 0, 0], (n, 1)), np.full(n, 400.0), t=np.full(n, 100.0))
    end, _, _ = oracle_mod.propagate(pk, ph, seed=2, max_steps=10)
    detected = np.flatnonzero(end.flags & event.SURFACE_DETECT)
    assert len(detected) > 800                          # QE 31.8 % at 400 nm
    tables = _padded_cdf(*geo.time_cdf) + _padded_cdf(*geo.charge_cdf)
    unit = np.float32(geo.charge_cdf[0][-1] / 2 ** 16)
    times, charges = [], []
    for i in detected[:800]:                            # one photon per acquisition: the channel's min is that photon's time
        t, q, hist, hit = oracle_mod.run_daq(pk, end, tables, unit, seed=2, start_photon=int(i), nphotons=1)
        assert hit[0] and hist[0] == end.flags[i]
        times.append(t[0]); charges.append(q[0])
    assert abs(np.std(times) - 1.2) < 0.15 and abs(np.mean(times) - end.t[detected[0]]) < 0.4
    assert abs(np.mean(charges) - 1.0) < 0.05 and 0.05 < np.std(charges) < 0.2
    # nothing detected -> nothing hit; weight 0 -> nothing hit
    t, q, hist, hit = oracle_mod.run_daq(pk, end, tables, unit, seed=2, weight=0.0)
    assert not hit.any() and q.sum() == 0
    # run_daq_many (daq.cu:88-150): copies of the acquisition side by side, each with a unit normal jitter
    ndaq = 64
    times = []
    for i in detected[:60]:
        t, q, hist, hit = oracle_mod.run_daq_many(pk, end, tables, unit, seed=2, ndaq=ndaq, start_photon=int(i), nphotons=1)
        assert t.shape == (ndaq,) and hit.all() and (hist == end.flags[i]).all()
        times.append(t)
    times = np.concatenate(times)
    # the jitter adds in quadrature to the 1.2 ns of the time CDF; the copies are independent draws
    assert abs(np.std(times) - np.hypot(1.2, 1.0)) < 0.1 and abs(np.mean(times) - end.t[detected[0]]) < 0.5
    assert len(np.unique(times)) > 0.99 * len(times)
    t1, q1, _, _ = oracle_mod.run_daq_many(pk, end, tables, unit, seed=2, ndaq=1)
    t4, q4, _, hit4 = oracle_mod.run_daq_many(pk, end, tables, unit, seed=2, ndaq=4)
    assert np.array_equal(t4[:1], t1) and np.array_equal(q4[:1], q1)              # copy 0 does not depend on ndaq
    assert hit4.all() and abs(q4.mean() / q1[0] - 1.0) < 0.05                     # every copy sees all the photons


def test_numpy_restatement_agrees_with_the_c_oracle(oracle_mod, tiny_packed):
    """oracle/numpy_propagate.py (the "pure NumPy" figure of BASELINE.md section 5, C1): its ray cast
    returns the C oracle's triangles and distances, and its histories agree statistically (its random
    numbers are numpy's, so not photon by photon)."""
    from oracle import numpy_propagate as npp
    from chroma_amd import event
    ph = oracle_mod.generate_bomb(10000, seed=20240502)
    tab = npp.Tables(tiny_packed)
    d = (ph.dir / np.linalg.norm(ph.dir, axis=1)[:, None]).astype(np.float32)
    tri, dist = npp.intersect_mesh(tab, ph.pos.astype(np.float32), d, np.full(len(ph), -1))
    odist, otri, _ = oracle_mod.distance_to_mesh(tiny_packed, ph.pos, ph.dir)
    assert np.array_equal(tri, otri)
    hit = otri >= 0
    assert np.array_equal(dist[hit].view(np.uint32), odist[hit].view(np.uint32))
    out = npp.propagate(tiny_packed, ph, seed=1, max_steps=100, tables=tab)
    want, _, _ = oracle_mod.propagate(tiny_packed, ph, seed=12345, max_steps=100)
    assert ((out.flags & event.TERMINAL_MASK) != 0).all()
    n = float(len(ph))
    for name in ('BULK_ABSORB', 'SURFACE_DETECT', 'SURFACE_ABSORB', 'RAYLEIGH_SCATTER', 'REFLECT_DIFFUSE', 'REFLECT_SPECULAR'):
        bit = getattr(event, name)
        a, b = ((out.flags & bit) != 0).sum(), ((want.flags & bit) != 0).sum()
        sigma = np.sqrt(a + b + 1.0)                          # two independent Poisson-like counts
        assert abs(a - b) < 5 * sigma, (name, a, b)
    assert not ((out.flags & (event.NO_HIT | event.NAN_ABORT)) != 0).any()


def test_bulk_reemission_spectrum(oracle_mod):
    """The intent of the reference's test/test_reemission.py:15-80 (skipped there, and written against
    attributes Material no longer has): monoenergetic photons started inside a wavelength-shifting sphere
    are absorbed and re-emitted, and the wavelengths that reach the detecting sphere follow the
    component's re-emission CDF.  Here the material absorbs below 400 nm only, so a photon is
    shifted once and then leaves."""
    import scipy.stats
    from chroma_amd.geometry import Geometry, Solid, Material, Surface, vacuum, standard_wavelengths
    from chroma_amd.make import sphere
    from chroma_amd.loader import create_geometry_from_obj
    wl = standard_wavelengths.astype(float)
    scint = Material('scint')
    scint.set('refractive_index', 1.0)
    absorb = np.where(wl < 400.0, 1.0, 1e7)      # the material's total; its one component takes all of it (photon.h:205-214)
    scint.set('absorption_length', absorb)
    scint.set('scattering_length', 1e7)
    norm = scipy.stats.norm(loc=600.0, scale=50.0)
    cdf = norm.cdf(wl)
    cdf[wl <= 400.0] = 0.0                       # nothing re-emitted where the component still absorbs
    cdf = (cdf - cdf.min()) / (cdf.max() - cdf.min())
    tgrid = np.arange(0, 100, 0.05)
    tcdf = 1.0 - np.exp(-tgrid / 5.0)
    tcdf /= tcdf[-1]
    for name, value in (('comp_reemission_prob', 1.0), ('comp_reemission_wvl_cdf', cdf),
                        ('comp_absorption_length', absorb)):
        tmp = Material('tmp'); tmp.set('x', value)
        getattr(scint, name).append(tmp.x)
    scint.comp_reemission_time_cdf.append(np.column_stack([tgrid, tcdf]).astype(np.float32))
    detector = Surface('detector')
    detector.set('detect', 1.0)
    world = Geometry(vacuum)
    world.add_solid(Solid(sphere(1000.0), vacuum, vacuum, surface=detector))
    world.add_solid(Solid(sphere(500.0), scint, vacuum))
    pk = pack_geometry(create_geometry_from_obj(world))
    ph = bomb(40000, 21, wavelength=250.0)
    out, _, _ = oracle_mod.propagate(pk, ph, seed=5, max_steps=20, nthreads=4)
    hit = (out.flags & event.SURFACE_DETECT) != 0
    assert hit.mean() > 0.95 and ((out.flags[hit] & event.BULK_REEMIT) != 0).all()
    got = out.wavelengths[hit].astype(float)
    assert got.min() >= 400.0 and abs(got.mean() - 600.0) < 2.0 and abs(got.std() - 50.0) < 2.0
    # Kolmogorov-Smirnov against the tabulated CDF the engine samples (linear between the grid points)
    ks = scipy.stats.kstest(got, lambda x: np.interp(x, wl, cdf))
    assert ks.pvalue > 1e-3, ks
    # and the re-emission is delayed by the component's decay time (5 ns) on top of the 1000 mm flight
    flight = 1000.0 / 299.792458
    assert abs((out.t[hit] - flight).mean() - 5.0) < 0.3


def test_sample_cdf_reproduces_a_binned_gaussian(oracle_mod):
    """The intent of the reference's test/test_sample_cdf.py:10-67 for the non-uniform CDF sampler
    (chroma/cuda/random.h:28-31 over interpolate.h:33-58), which only the DAQ uses: draws through a
    tabulated unit-Gaussian CDF follow that CDF (Kolmogorov probability > 0.01)."""
    import scipy.stats
    from chroma_amd.detector import Detector
    from chroma_amd.geometry import Solid, vacuum
    from chroma_amd.make import box
    from chroma_amd.demo.optics import r7081hqe_photocathode
    from chroma_amd.loader import create_geometry_from_obj
    cube = Detector(vacuum)
    cube.add_pmt(Solid(box(10.0, 10, 10), vacuum, vacuum, surface=r7081hqe_photocathode))
    pk = pack_geometry(create_geometry_from_obj(cube))
    n = 200
    # (t = 100: a channel time below zero would lose the atomicMin on its bit pattern, daq.cu:5-12)
    ph = Photons(np.zeros((n, 3)), np.tile([0, 0, 1.0], (n, 1)), np.tile([1.0, 0, 0], (n, 1)), np.full(n, 400.0), t=np.full(n, 100.0))
    end, _, _ = oracle_mod.propagate(pk, ph, seed=2, max_steps=10)
    one = int(np.flatnonzero(end.flags & event.SURFACE_DETECT)[0])
    edges = np.linspace(-5.0, 5.0, 101).astype(np.float32)
    cdf = scipy.stats.norm.cdf(edges).astype(np.float32)
    cdf = (cdf - cdf[0]) / (cdf[-1] - cdf[0])
    flat_x, flat_y = np.array([0.0, 1.0], dtype=np.float32), np.array([0.0, 1.0], dtype=np.float32)
    draws = []
    for k in range(3000):                         # one photon, a new acquisition (= a new stream) per draw
        t, q, hist, hit = oracle_mod.run_daq(pk, end, (edges, cdf, flat_x, flat_y), 1.0 / 1024, seed=4, acquisition=k,
                                             start_photon=one, nphotons=1)
        draws.append(t[0] - end.t[one])
    draws = np.asarray(draws, dtype=float)
    assert abs(draws.mean()) < 0.06 and abs(draws.std() - 1.0) < 0.05
    ks = scipy.stats.kstest(draws, lambda x: np.interp(x, edges, cdf))
    assert ks.pvalue > 0.01, ks


def test_render_restatement_invariants(oracle_mod):
    """oracle_render (chroma/cuda/render.cu:37-181) on two nested cubes: a ray through the middle crosses four faces;
    the per-ray lists are sorted, capped at alpha_depth (the nearest ones survive), misses show the background, and a
    second call with the first call's lists (keep_last_render) merges instead of starting over."""
    from chroma_amd.geometry import Geometry, Solid, vacuum
    from chroma_amd.make import box
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.geometry import pack_geometry
    from chroma_amd.tools import from_film
    g = Geometry(vacuum)
    g.add_solid(Solid(box(100.0, 100.0, 100.0), vacuum, vacuum, color=0x80FF0000))
    g.add_solid(Solid(box(40.0, 40.0, 40.0), vacuum, vacuum, color=0x4000FF00))
    pk = pack_geometry(create_geometry_from_obj(g))
    pos, d = from_film(position=(0.0, -400.0, 0.0), size=(64, 48), width=35.0, focal_length=50.0)
    pix, (dx, n, col) = oracle_mod.render(pk, pos, d, alpha_depth=10, bg_color=0x00000000)
    # rays through both cubes, away from shared triangle edges: four hits at four different distances
    four = np.flatnonzero((n == 4) & (dx[:, 0] < dx[:, 1]) & (dx[:, 1] < dx[:, 2]) & (dx[:, 2] < dx[:, 3]))
    assert len(four) > 50 and n.max() <= 8 and (n == 0).any()
    assert (pix[n == 0] == 0).all() and (pix[n > 0] >> 24 > 0).all()
    for k in range(1, 4):
        assert (dx[four, k] >= dx[four, k - 1]).all()
    k0 = four[0]
    assert 349.0 < dx[k0, 0] < 360.0 and 379.0 < dx[k0, 1] < 392.0 and 449.0 < dx[k0, 3] < 465.0
    assert (col[four, 0, 0] > 200).all() and (col[four, 1, 1] > 200).all() and (col[four, 2, 1] > 200).all() and (col[four, 3, 0] > 200).all()
    # alpha_depth 2 keeps the two nearest
    pix2, (dx2, n2, _) = oracle_mod.render(pk, pos, d, alpha_depth=2)
    assert n2.max() == 2 and np.array_equal(dx2[four], dx[four][:, :2])
    assert (pix2[n2 == 2] >> 24 == 255).all()                              # a full list is opaque (render.cu:169-172)
    # keep_last_render: the same rays again -> every distance twice, still sorted
    pix3, (dx3, n3, _) = oracle_mod.render(pk, pos, d, alpha_depth=10, state=(dx, n, col))
    assert (n3[four] == 8).all() and np.array_equal(dx3[four][:, 0:8:2], dx[four][:, :4]) and np.array_equal(dx3[four][:, 1:8:2], dx[four][:, :4])
