"""The exact walk (CHROMA_WALK_LITERAL, k_raycast_literal: chroma/cuda/mesh.h:42-118 for every ray, four lanes
per ray) against the oracle and against its own lane-per-ray cross-check (LITERAL_LANE: intersect_mesh_strict).

Besides every output field bit for bit, the COUNTS must be the oracle's: the number of child boxes tested and the number
of triangles tested per batch are functions of the exact order of box and triangle tests, so equal counts say the walk
made the reference's tests, no more and no fewer (the fast walks only promise a superset).
"""
import os

import numpy as np
import pytest

from conftest import ROOT, bomb, make_stress_geometry
from test_gpu_parity import assert_bit_exact, _aimed_photons, _edge_photons

pytestmark = pytest.mark.gpu

VARIANT = os.path.join(ROOT, 'build_variants', 'libchroma_hip_stack4.so')


def run_walk(g, gg, photons, walk, max_steps, seed=12345, **kw):
    ctx = g.get_context()
    ctx.set_walk(walk)
    try:
        gp = g.GPUPhotons(photons)
        stats = {}
        ctx.set_counting(True)
        gp.propagate(gg, g.get_rng_states(64 * 1024, seed=seed), max_steps=max_steps, stats=stats, **kw)
        ctx.set_counting(False)
    finally:
        ctx.set_walk('quad')
    return gp, gp.get(), stats


def check(g, oracle_mod, geometry, photons, what, max_steps=30, seed=12345, lane=True, parallel_rays=False, **kw):
    """`parallel_rays`: the batch holds rays exactly parallel to an axis.  The reference skips the slab of such an axis
    (intersect.h:115,124,133) and walks every box in the ray's plane; the engine's strict loop tests containment instead
    (box_tmin, propagate_device.h: same hits, far fewer boxes), so its counts are BELOW the oracle's there and the two
    literal walks are compared with each other."""
    from chroma_amd.gpu.geometry import pack_geometry
    gg = g.GPUDetector(geometry) if hasattr(geometry, 'num_channels') else g.GPUGeometry(geometry)
    want, counters, ostats = oracle_mod.propagate(pack_geometry(geometry), photons, seed=seed, max_steps=max_steps, nthreads=8, **kw)
    gp, got, stats = run_walk(g, gg, photons, 'literal', max_steps, seed, **kw)
    assert_bit_exact(got, want, what + ', literal')
    assert np.array_equal(gp.rng_counters.get(), counters)
    assert stats['photon_steps'] == ostats['photon_steps'] and stats['launches'] == ostats['launches']
    if parallel_rays:
        assert stats['nodes_visited'] <= ostats['nodes_visited'] and stats['triangles_tested'] <= ostats['triangles_tested']
    else:
        assert stats['nodes_visited'] == ostats['nodes_visited'], (stats['nodes_visited'], ostats['nodes_visited'])
        assert stats['triangles_tested'] == ostats['triangles_tested'], (stats['triangles_tested'], ostats['triangles_tested'])
    if lane or parallel_rays:
        gp2, got2, stats2 = run_walk(g, gg, photons, 'literal_lane', max_steps, seed, **kw)
        assert_bit_exact(got2, want, what + ', literal_lane')
        assert stats2['nodes_visited'] == stats['nodes_visited'] and stats2['triangles_tested'] == stats['triangles_tested']
    return stats


@pytest.fixture(scope='module')
def gpu():
    from chroma_amd import gpu as g
    ctx = g.create_cuda_context(0)
    yield g
    ctx.pop()


def test_bomb_large_and_small_batches(gpu, oracle_mod, tiny_geometry):
    """Per-step launches (>= 8192 alive) and the small-batch policy, mixed wavelengths."""
    check(gpu, oracle_mod, tiny_geometry, bomb(60000, 3, wavelength=400.0, wavelength_hi=800.0), 'tiny 60k')
    check(gpu, oracle_mod, tiny_geometry, bomb(3000, 4), 'tiny 3k', max_steps=100)


def test_aimed_rays_and_ties(gpu, oracle_mod, tiny_geometry):
    """Rays through vertices, edge midpoints and centroids: exact ties, decided by the reference's test order."""
    ph = _aimed_photons(tiny_geometry, (0.0, 0.0, 0.0), 20000)
    check(gpu, oracle_mod, tiny_geometry, ph, 'aimed rays', max_steps=4, parallel_rays=True)
    ph = _aimed_photons(tiny_geometry, (150.0, -420.0, 310.0), 20000)
    check(gpu, oracle_mod, tiny_geometry, ph, 'aimed rays, off-centre', max_steps=4)


def test_edge_inputs(gpu, oracle_mod, tiny_geometry):
    """Axis-parallel rays (1/d infinite: the strict loop takes them), NaN photons, terminal photons."""
    check(gpu, oracle_mod, tiny_geometry, _edge_photons(), 'edge inputs', max_steps=20, parallel_rays=True)


def test_every_surface_model(gpu, oracle_mod):
    check(gpu, oracle_mod, make_stress_geometry(), bomb(40000, 6, wavelength=350.0), 'stress', seed=11, max_steps=100)


def test_random_soups(gpu, oracle_mod):
    """Triangle soups (overlapping boxes, slivers, zero-area triangles), twin spheres (every hit a tie), nested boxes that
    share face planes: where the order of tests matters most.  The fast walks are allowed two erratic rays here; the exact
    walk is allowed none."""
    from test_gpu_fuzz import _geometries
    from chroma_amd.loader import create_geometry_from_obj
    rng = np.random.default_rng(5)
    for k, (name, geo) in enumerate(_geometries()):
        ph = bomb(30000, 21 + k)
        ph.pos[:] = rng.uniform(-300, 300, (len(ph), 3))
        check(gpu, oracle_mod, create_geometry_from_obj(geo), ph, name, max_steps=30, seed=5, lane=(k == 0))


def test_through_the_spill_area(oracle_mod, tiny_geometry):
    """The same sources with 4 stack words per ray in LDS: pushes and pops through global memory."""
    from chroma_amd import gpu as g
    if not os.path.exists(VARIANT):
        pytest.fail('%s is not built' % VARIANT)
    ctx = g.create_cuda_context(0, library=VARIANT)
    try:
        stats = check(g, oracle_mod, tiny_geometry, bomb(60000, 3, wavelength=400.0, wavelength_hi=800.0), 'stack4, tiny 60k', lane=False)
        assert stats['stack_spills'] > 10000, stats
        stats = check(g, oracle_mod, tiny_geometry, _aimed_photons(tiny_geometry, (0.0, 0.0, 0.0), 20000), 'stack4, aimed', max_steps=4, lane=False, parallel_rays=True)
        assert stats['stack_spills'] > 1000, stats
    finally:
        ctx.pop()


def test_two_threads_on_one_handle_one_exact_one_default(gpu, oracle_mod, tiny_geometry):
    """What a call does travels with the call (chroma_propagate_options), not as a setting of the context: two threads
    sharing one context -- one asking for the exact walk, one for the default -- each get the oracle's photons, and the
    counts say each got the walk it asked for (the exact walk makes the oracle's very tests, the default walk fewer
    triangle tests over a different tree)."""
    import threading
    from chroma_amd.gpu.geometry import pack_geometry
    gg = gpu.GPUDetector(tiny_geometry)
    packed = pack_geometry(tiny_geometry)
    jobs = {'exact': (bomb(60000, 31, wavelength=400.0, wavelength_hi=800.0), True), 'default': (bomb(50000, 32), False)}
    want = {k: oracle_mod.propagate(packed, ph, seed=77, max_steps=30, nthreads=8) for k, (ph, _) in jobs.items()}
    failures = []

    def work(name):
        try:
            ph, exact = jobs[name]
            for rep in range(4):
                gp = gpu.GPUPhotons(ph)
                stats = {}
                gp.propagate(gg, gpu.get_rng_states(64, seed=77), max_steps=30, exact=exact, counting=True, stats=stats)
                end, counters, ostats = want[name]
                assert_bit_exact(gp.get(), end, '%s thread, call %d' % (name, rep))
                assert np.array_equal(gp.rng_counters.get(), counters)
                if exact:
                    assert stats['nodes_visited'] == ostats['nodes_visited'] and stats['triangles_tested'] == ostats['triangles_tested']
                else:
                    assert stats['nodes_visited'] != ostats['nodes_visited'] and 0 < stats['triangles_tested'] < ostats['triangles_tested']
        except BaseException as exc:      # noqa: B902 (reported by the main thread)
            failures.append((name, exc))

    threads = [threading.Thread(target=work, args=(k,)) for k in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not failures, failures
    assert gpu.get_context().walk == 'quad'
