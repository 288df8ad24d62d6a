"""The deep-stack path of the fast walks, on purpose.

A ray of k_raycast_quad / k_raycast_coop / k_tail_coop keeps 24 stack entries in LDS and pushes and
pops deeper ones through a per-ray slice of global memory.  Only the largest trees need that (C3: 60
entries in the worst case), and a small geometry never gets there -- so this module runs the SAME
sources built with 4 entries in LDS (build_variants/libchroma_hip_stack4.so, `make variants`; loaded
beside the product library) on demo.tiny(), where most rays then go through the spill area, and
compares with the oracle bit for bit: ordinary bomb, rays aimed at vertices and edges (ties), the
edge-input batch, small batches (fused tail kernel) and large ones (per-step launches), and the
8-lane walk as well as the 4-lane one.  The counting build reports how many entries were spilled,
so the test cannot pass vacuously.
"""
import os

import numpy as np
import pytest

from chroma_amd import event
from conftest import ROOT, bomb
from test_gpu_parity import assert_bit_exact, run_both, _aimed_photons, _edge_photons

pytestmark = pytest.mark.gpu

VARIANT = os.path.join(ROOT, 'build_variants', 'libchroma_hip_stack4.so')


@pytest.fixture(scope='module')
def gpu():
    from chroma_amd import gpu as g
    if not os.path.exists(VARIANT):
        pytest.fail('%s is not built: run `make -C chroma_amd/csrc variants` (build() does)' % VARIANT)
    ctx = g.create_cuda_context(0, library=VARIANT)
    yield g
    ctx.pop()


def test_large_batch_through_the_spill_area(gpu, oracle_mod, tiny_geometry):
    """Per-step launches (>= 8192 alive): 4-lane and 8-lane walks with a 4-entry LDS stack."""
    ph = bomb(60000, 3, wavelength=400.0, wavelength_hi=800.0)
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, tiny_geometry, ph, max_steps=30)
    assert_bit_exact(got, want, 'stack4, tiny 60k')
    assert np.array_equal(gp.rng_counters.get(), counters)
    assert stats['photon_steps'] == ostats['photon_steps'] and stats['launches'] == ostats['launches']
    assert stats['stack_spills'] > 10000, 'the variant did not spill: %r' % (stats,)
    for walk in ('coop', 'pair'):
        gpu.get_context().set_walk(walk)
        try:
            gp2 = gpu.GPUPhotons(ph)
            stats2 = {}
            gpu.get_context().set_counting(True)
            gp2.propagate(gg, gpu.get_rng_states(64 * 1024, seed=12345), max_steps=30, stats=stats2)
            gpu.get_context().set_counting(False)
        finally:
            gpu.get_context().set_walk('quad')
        assert_bit_exact(gp2.get(), want, 'stack4, tiny 60k, %s walk' % walk)
        assert stats2['stack_spills'] > 10000, walk


@pytest.mark.parametrize('count', ['large', 'small'])
def test_ties_through_the_spill_area(gpu, oracle_mod, tiny_geometry, count):
    """Rays through vertices and edges: the tie-break survives entries that travel through global memory
    (per-step launches, and the fused tail kernel for the small batch)."""
    ph = _aimed_photons(tiny_geometry, (0.0, 0.0, 0.0), 20000 if count == 'large' else 3000)
    if count == 'small':
        ph = ph[:3000]
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, tiny_geometry, ph, max_steps=4)
    assert_bit_exact(got, want, 'stack4, aimed rays (%s)' % count)
    assert stats['stack_spills'] > 1000


def test_edge_inputs_through_the_spill_area(gpu, oracle_mod, tiny_geometry):
    ph = _edge_photons()
    gg, gp, got, want, counters, stats, ostats = run_both(gpu, oracle_mod, tiny_geometry, ph, max_steps=20)
    assert_bit_exact(got, want, 'stack4, edge inputs')
    assert np.array_equal(gp.rng_counters.get(), counters)
    assert stats['launches'] == ostats['launches'] and stats['photon_steps'] == ostats['photon_steps']
    assert stats['stack_spills'] > 1000


def test_product_library_does_not_spill_on_tiny(oracle_mod, tiny_geometry):
    """The other half of the claim: with the product's 24 entries demo.tiny() (need 11) never spills."""
    from chroma_amd import gpu as g
    ctx = g.create_cuda_context(0)
    try:
        ph = bomb(20000, 3)
        gg, gp, got, want, counters, stats, ostats = run_both(g, oracle_mod, tiny_geometry, ph, max_steps=10)
        assert_bit_exact(got, want, 'product library')
        assert stats['stack_spills'] == 0
    finally:
        ctx.pop()
