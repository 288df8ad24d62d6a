"""The one JSON line `bench.py` prints is a contract with the driver (metric, value, unit, n_gpus, steps, warmup,
ms_per_step, higher_is_better, scaling, vs_baseline, dtype, data, config.workload + the `roofline` and `cpu_baseline`
objects): run it as the driver does -- a child process -- on the smallest configuration and check the line."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(*extra):
    env = dict(os.environ)
    env.pop('CHROMA_BENCH_GEOMETRY_CACHE', None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'tiny', '--steps', '2', '--warmup', '1'] + list(extra),
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-2000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip()]
    assert len(lines) == 1, 'exactly one line on stdout: %r' % lines
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_line_carries_the_contract():
    j = _run()
    assert j['metric'].startswith('photons/sec') and j['unit'] == 'photons/s' and j['higher_is_better'] is True
    assert j['n_gpus'] == 1 and j['steps'] == 2 and j['warmup'] == 1 and j['scaling'] == 'weak'
    assert j['vs_baseline'] is None and j['dtype'] == 'f32' and j['data'] == 'synthetic'
    assert 'workload' in j['config'] and 'model' not in j['config']
    assert j['value'] > 1e7 and abs(j['value'] * j['ms_per_step'] * 1e-3 / j['config']['photons_per_gpu_per_step'] - 1.0) < 1e-6
    r = j['roofline']
    assert r['bound'] in ('hbm', 'mfma') and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-9 and 0.0 < r['frac'] < 1.0
    assert 'traffic' in r and 'traffic_source' in r and r['launches'] > 0 and r['avg_launch_ms'] > 0
    # algorithmic bytes per launch over the measured launch time IS `achieved`
    assert abs(r['algorithmic_bytes_per_launch'] / (r['avg_launch_ms'] * 1e-3) / 1e9 - r['achieved']) < 1e-6 * r['achieved']
    # the exact walk's rate on the same batch, and which reduction path the run took
    assert j['config']['reduction'].startswith('none')
    # the three legs, each over as many batches as the headline, with their spreads: `value`'s own input order
    # (generation order: SURVEY.md 8d), the pre-sorted input of chroma/benchmark.py:80-82 with what the sort costs, the exact walk
    cfg = j['config']
    assert cfg['value_is'] == 'generation_order' and 'generation order' in cfg['inputs']
    for leg in ('generation_order', 'presorted', 'exact_walk'):
        assert cfg[leg]['batches'] == 2 and cfg[leg]['value'] > 0 and cfg[leg]['std'] >= 0 and cfg[leg]['ms_per_batch'] > 0, leg
    assert abs(cfg['generation_order']['value'] / j['value'] - 1.0) < 0.2          # (the headline loop's own per-step times)
    assert cfg['sort']['ms_per_batch'] > 0 and cfg['exact_walk']['kernel'] == 'k_raycast_literal'
    assert 0 < cfg['exact_walk_photons_per_s'] == cfg['exact_walk']['value'] < 1.5 * j['value']
    c = j['cpu_baseline']
    assert c['kind'] in ('port', 'reference') and c['unit'] == 'photons/s' and c['value'] > 0 and c['cores'] >= 1 and c['sample']


@pytest.mark.timeout(900)
def test_bench_refuses_a_world_size_that_is_not_gpus():
    env = dict(os.environ, WORLD_SIZE='2', RANK='0', LOCAL_RANK='0')
    p = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--config', 'tiny', '--gpus', '1', '--steps', '1', '--warmup', '0'],
                       cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert p.returncode != 0 and b'WORLD_SIZE' in p.stderr + p.stdout


@pytest.mark.timeout(900)
def test_two_rank_flow_of_bench_rehearsed_over_gloo():
    """The whole multi-rank flow of bench.py -- rendezvous, geometry built once by local rank 0 and mapped by the other, the
    vote on the library communicator, the timed loop between barriers, the max over ranks, one JSON line from rank 0 -- with
    two ranks sharing this box's one GPU and gloo as the transport (CHROMA_BENCH_BACKEND=gloo: a REHEARSAL, its number means
    nothing).  Every collective is entered by both ranks or the run hangs into the timeout: this is the test that catches a
    rank-0-only step drifting into a collective (round 3's first rehearsal did)."""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, CHROMA_BENCH_BACKEND='gloo', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('CHROMA_BENCH_GEOMETRY_CACHE', None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--config', 'tiny', '--steps', '2', '--warmup', '1']
    p = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    assert p.returncode == 0, p.stderr.decode()[-3000:]
    lines = [l for l in p.stdout.decode().splitlines() if l.strip().startswith('{')]
    assert len(lines) == 1, 'exactly one JSON line (rank 0): %r' % lines
    j = json.loads(lines[0])
    assert j['n_gpus'] == 2 and j['steps'] == 2 and j['scaling'] == 'weak'
    assert 'torch.distributed fallback (gloo)' in j['config']['reduction']
    assert j['config']['exact_walk'] is None and j['cpu_baseline'] is None          # single-GPU legs stay out of multi-rank runs
    assert j['value'] > 0 and abs(j['value'] * j['ms_per_step'] * 1e-3 / (2 * j['config']['photons_per_gpu_per_step']) - 1.0) < 1e-6
