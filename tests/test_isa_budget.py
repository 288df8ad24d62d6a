"""Compile-time properties of the hot kernels that the measured performance rests on (no GPU needed: hipcc
cross-compiles gfx950): the ray cast keeps its state in registers -- no scratch -- at 7 waves per SIMD, the
streaming kernels fit full occupancy.  tools/isa_report.sh is the same table for profiles/."""
import os
import shutil
import subprocess

import pytest

from conftest import ROOT


@pytest.fixture(scope='module')
def isa_table():
    if not os.path.exists('/opt/rocm/bin/hipcc') or shutil.which('c++filt') is None:
        pytest.skip('hipcc / c++filt not available')
    out = subprocess.run([os.path.join(ROOT, 'tools', 'isa_report.sh')], check=True, capture_output=True, text=True, timeout=900).stdout
    table = {}
    for line in out.splitlines():
        if line.startswith('#') or not line.strip():
            continue
        name, vgpr, sgpr, scratch, lds, waves, code = [x.strip() for x in line.rsplit(',', 6)]
        table[name.replace('void ', '')] = dict(vgpr=int(vgpr), scratch=int(scratch), lds=int(lds), waves=int(waves), code=int(code))
    return table


@pytest.mark.timeout(1000)
def test_ray_cast_kernels_hold_their_state_in_registers(isa_table):
    k = isa_table['k_raycast_coop<false>']
    assert k['scratch'] == 0 and k['waves'] == 7 and k['vgpr'] <= 72, k
    # the product walk runs at the full 8 waves per SIMD: 64 VGPRs at most, and nothing in scratch but the base pointer
    # of the global spill area (12 bytes, reloaded in the deep-stack push only: see the kernel's QUAD_WAVES_PER_EU)
    k = isa_table['k_raycast_quad<false>']
    assert k['scratch'] <= 16 and k['waves'] == 8 and k['vgpr'] <= 64, k
    # the counting build (untimed: one pass per bench run, and the parity tests) may keep a few more words there
    k = isa_table['k_raycast_quad<true>']
    assert k['scratch'] <= 48 and k['waves'] == 8, k
    # exactly the 16 per-ray areas (24 stack entries of two words, a ring of 16 postponed triangles, origin and
    # direction, one word of padding): a struct the compiler cannot keep in registers is "promoted" to LDS silently
    # (768 bytes per wave until RayFast::a became three scalars)
    assert isa_table['k_raycast_quad<false>']['lds'] == 16 * (2 * 24 + 16 + 6 + 1) * 4
    k = isa_table['k_raycast_pair<false>']
    assert k['scratch'] == 0 and k['waves'] == 5 and k['lds'] == 32 * (2 * 18 + 16 + 2) * 4, k


def test_streaming_kernels_run_at_full_occupancy(isa_table):
    for name in ('k_ray_setup', 'k_load_working', 'k_store_working', 'k_step_begin', 'k_channel_hits', 'k_count_hits',
                 'k_run_daq', 'k_run_daq_many', 'k_generate_bomb'):
        k = isa_table[name]
        assert k['scratch'] == 0 and k['waves'] == 8, (name, k)


def test_every_kernel_of_the_path_is_in_the_table(isa_table):
    # the exact walk: full occupancy, its state in registers
    k = isa_table['k_raycast_literal<false>']
    assert k['scratch'] == 0 and k['waves'] == 8 and k['vgpr'] <= 64 and k['lds'] == 16 * 33 * 4, k
    for name in ('k_physics<false>', 'k_physics<true>', 'k_tail_coop<false, false>', 'k_tail_coop<false, true>', 'k_finalize_hits', 'k_propagate<24, false>', 'k_raycast_retry<false, false>', 'k_raycast_retry<false, true>', 'k_raycast_wide<false>',
                 'k_raycast_persistent<false>', 'k_distance_to_mesh<24, false>', 'k_copy_hits', 'k_daq_reset', 'k_daq_convert'):
        assert name in isa_table, name
    assert isa_table['k_physics<true>']['waves'] >= 4
    # the build for plain optics (no re-emission, default surface model only: configs C1-C4) runs in 96 registers since
    # round 3: five waves per SIMD (blocks of four waves, PHYS_PLAIN_BLOCK).  The allocator sits on that limit: with the
    # geometry view's last field (slab_grow) it parks two register pairs in scratch -- four scratch instructions per
    # round of ~4 200, one store/load pair of them back to back -- which the measured numbers of round 3 include.
    k = isa_table['k_physics<false>']
    assert k['scratch'] <= 24 and k['waves'] >= 5 and k['vgpr'] <= 96 and k['code'] < 40000, k
    # (the all-models build: 128 VGPRs at 4 waves and a spill area -- 136 bytes in round 3, 176 with the final-record store of
    #  chroma_propagate_hits beside the array stores)
    assert isa_table['k_physics<true>']['scratch'] <= 192
