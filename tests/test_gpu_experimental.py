"""The experiments that did not pay and therefore live OUTSIDE the product library (chroma_amd/csrc/experimental/, built into
build_variants/libchroma_hip_experimental.so by `make variants`): the packet ray cast and the engine-side direction sort.
They stay parity-green -- tested here through the variant library -- and the product library refuses to switch them on.

k_raycast_packet: the first step of a propagate call with 64 rays per wavefront walking the wide tree as one packet.
Whatever the rays look like, every photon must come out as the default (quad) walk and the oracle give it, bit for bit;
only the speed depends on coherence (and it is not better than the default walk's: profiles/r03/ab_packet_first_step.txt
-- the kernel is an opt-in, off by default).  'auto' must pick the packet kernel for a direction-sorted bomb and leave an
unsorted one to the quad walk."""
import os

import numpy as np
import pytest

from chroma_amd import event
from conftest import ROOT, bomb
from test_gpu_parity import assert_bit_exact, _edge_photons

pytestmark = pytest.mark.gpu

VARIANT = os.path.join(ROOT, 'build_variants', 'libchroma_hip_experimental.so')


@pytest.fixture(scope='module')
def gpu():
    from chroma_amd import gpu as g
    if not os.path.exists(VARIANT):
        pytest.fail('%s is not built: run `make -C chroma_amd/csrc variants` (build() does)' % VARIANT)
    ctx = g.create_cuda_context(0, library=VARIANT)
    yield g
    ctx.set_packet('off')
    ctx.pop()


def _propagate(gpu, gg, photons, mode, seed=4242, max_steps=100, sort=False, counting=True):
    ctx = gpu.get_context()
    ctx.set_packet(mode)
    gp = gpu.GPUPhotons(photons)
    if sort:
        gp.sort_by_direction()
    stats = {}
    ctx.set_counting(counting)
    try:
        gp.propagate(gg, gpu.get_rng_states(64, seed=seed), max_steps=max_steps, stats=stats)
    finally:
        ctx.set_counting(False)
        ctx.set_packet('off')
    return gp, gp.get(), stats


@pytest.mark.parametrize('geometry_name', ['tiny', 'lite'])
def test_packet_walk_equals_quad_walk_and_oracle(gpu, oracle_mod, tiny_geometry, geometry_name):
    from chroma_amd import demo
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.geometry import pack_geometry
    geo = tiny_geometry if geometry_name == 'tiny' else create_geometry_from_obj(demo.detector_lite())
    packed = pack_geometry(geo)
    gg = gpu.GPUDetector(geo, packed=packed)
    n = 2000000             # (enough photons for 64 neighbours of the sorted bomb to lie within the probe's 50 mrad)
    ph = oracle_mod.generate_bomb(n, seed=11, id_base=0)
    # (a) coherent rays: the bomb in direction order -- sorted on the device, read back, so that oracle and engine see the same photons
    gps = gpu.GPUPhotons(ph)
    gps.sort_by_direction()
    sorted_ph = gps.get()
    want, counters, ostats = oracle_mod.propagate(packed, sorted_ph, seed=4242, max_steps=100, nthreads=8)
    for mode in ('on', 'off', 'auto'):
        gp, got, stats = _propagate(gpu, gg, sorted_ph, mode)
        assert_bit_exact(got, want, '%s: sorted bomb, packet %s' % (geometry_name, mode))
        assert np.array_equal(gp.rng_counters.get(), counters)
        assert stats['photon_steps'] == ostats['photon_steps'] and stats['launches'] == ostats['launches']
        took = stats['packet_rays']
        assert (took == n) if mode in ('on', 'auto') else (took == 0), (mode, took)
        if mode == 'on':
            # a lane counts the nodes ITS ray entered and the triangles it tested: the same kind of numbers as the quad walk's
            assert 0 < stats['packet_nodes_visited'] <= stats['nodes_visited'] and 0 < stats['packet_triangles_tested'] <= stats['triangles_tested']
    # (b) unrelated rays in one packet: the same photons in generation order.  Forced on: correct all the same; auto: left to the quad walk
    want, counters, ostats = oracle_mod.propagate(packed, ph, seed=4242, max_steps=100, nthreads=8)
    for mode in ('on', 'auto'):
        gp, got, stats = _propagate(gpu, gg, ph, mode)
        assert_bit_exact(got, want, '%s: unsorted bomb, packet %s' % (geometry_name, mode))
        assert np.array_equal(gp.rng_counters.get(), counters)
        assert stats['packet_rays'] == (n if mode == 'on' else 0)


def test_packet_walk_on_awkward_photons(gpu, oracle_mod, tiny_geometry):
    """Terminal, NaN, axis-parallel, outside-the-world photons and photons with a last hit, mixed into the packets
    (the slots the ray cast must settle instead of casting), and a second propagate call whose first step starts ON
    triangles with last hits set."""
    from chroma_amd.gpu.geometry import pack_geometry
    packed = pack_geometry(tiny_geometry)
    gg = gpu.GPUDetector(tiny_geometry, packed=packed)
    ph = _edge_photons()
    want, counters, _ = oracle_mod.propagate(packed, ph, seed=4242, max_steps=20, nthreads=8)
    gp, got, stats = _propagate(gpu, gg, ph, 'on', max_steps=20)
    assert_bit_exact(got, want, 'edge inputs through the packet walk')
    assert np.array_equal(gp.rng_counters.get(), counters)
    assert stats['packet_rays'] > 20000
    # photons that have taken one step (they sit on triangles, last_hit_triangles set) start a NEW call: its first step is a packet step
    ph2 = bomb(100000, 23)
    one, ctr1, _ = oracle_mod.propagate(packed, ph2, seed=4242, max_steps=1, nthreads=8)
    two, ctr2, _ = oracle_mod.propagate(packed, one, seed=4242, max_steps=100, nthreads=8, rng_counters=ctr1)
    ctx = gpu.get_context()
    gp = gpu.GPUPhotons(ph2)
    rs = gpu.get_rng_states(64, seed=4242)
    ctx.set_packet('on')
    try:
        gp.propagate(gg, rs, max_steps=1)
        assert_bit_exact(gp.get(), one, 'first call')
        gp.propagate(gg, rs, max_steps=100)
    finally:
        ctx.set_packet('off')
    assert_bit_exact(gp.get(), two, 'second call: rays starting on their last hit')
    assert np.array_equal(gp.rng_counters.get(), ctr2)


def test_a_large_unsorted_bomb_is_taken_up_in_direction_order_with_the_same_results(gpu, oracle_mod, tiny_geometry):
    """chroma_set_autosort: a call of >= 2^21 photons from one origin in generation order is ordered by direction cell
    on the device (stats['reordered']); every photon ends exactly as with the photons taken as they come, and as the
    oracle says; a direction-sorted bomb and photons from many origins are left alone."""
    import numpy as np
    n = (1 << 21) + 12345
    ph = oracle_mod.generate_bomb(n, seed=99)
    gg = gpu.GPUDetector(tiny_geometry)
    ctx = gpu.get_context()
    results = {}
    for mode in ('off', 'auto', 'on'):
        ctx.set_autosort(mode)
        gp = gpu.GPUPhotons(ph)
        stats = {}
        gp.propagate(gg, gpu.get_rng_states(64 * 1024, seed=5), max_steps=100, stats=stats)
        results[mode] = (gp.get(), gp.rng_counters.get(), stats.get('reordered', 0))
    assert results['off'][2] == 0 and results['auto'][2] == n and results['on'][2] == n
    ctx.set_autosort('auto')
    for mode in ('auto', 'on'):
        for field in ('pos', 'dir', 'pol', 'wavelengths', 't', 'flags', 'last_hit_triangles', 'weights'):
            assert np.array_equal(getattr(results[mode][0], field).view(np.uint32), getattr(results['off'][0], field).view(np.uint32)), (mode, field)
        assert np.array_equal(results[mode][1], results['off'][1]), mode
    # against the oracle (the whole call: the launch policy, and with it the arithmetic, depends on how many photons live)
    from chroma_amd.gpu.geometry import pack_geometry
    want, counters, _ = oracle_mod.propagate(pack_geometry(tiny_geometry), ph, seed=5, max_steps=100, nthreads=8)
    assert_bit_exact(results['auto'][0], want, 'autosort')
    assert np.array_equal(results['auto'][1], counters)
    # already sorted: left alone; many origins: left alone
    gp = gpu.GPUPhotons(ph)
    gp.sort_by_direction()
    stats = {}
    gp.propagate(gg, gpu.get_rng_states(64 * 1024, seed=5), max_steps=5, stats=stats)
    assert stats.get('reordered', 0) == 0
    moved = oracle_mod.generate_bomb(n, seed=100)
    moved.pos[:] = np.random.default_rng(1).uniform(-100.0, 100.0, size=moved.pos.shape).astype(np.float32)
    gp = gpu.GPUPhotons(moved)
    stats = {}
    gp.propagate(gg, gpu.get_rng_states(64 * 1024, seed=5), max_steps=5, stats=stats)
    assert stats.get('reordered', 0) == 0
    ctx.set_autosort('off')


def test_the_product_library_does_not_carry_the_experiments():
    from chroma_amd import gpu as g
    from chroma_amd._lib import ChromaError
    ctx = g.create_cuda_context(0)
    try:
        for switch in (ctx.set_packet, ctx.set_autosort):
            switch('off')
            with pytest.raises(ChromaError, match='experiment'):
                switch('on')
    finally:
        ctx.pop()
