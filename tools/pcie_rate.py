"""The path with HOST photons on both ends (what chroma.sim.Simulation does per event batch): GPUPhotons(host arrays) ->
propagate -> get_flat_hits back to the host.  bench.py's `value` has its inputs resident in HBM; this is the
PCIe-inclusive figure DESIGN.md section 7 quotes next to it.  usage: pcie_rate.py [c3|detector|lite|tiny] [photons]  (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chroma_amd import demo, gpu
from chroma_amd.event import Photons
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.gpu.geometry import pack_geometry

config = sys.argv[1] if len(sys.argv) > 1 else 'c3'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
geo = create_geometry_from_obj({'tiny': demo.tiny, 'lite': demo.detector_lite, 'detector': demo.detector, 'c3': demo.detector29k}[config]())
from chroma_amd.sim import Simulation
sim0 = Simulation(geo, seed=5, prefetch=False)          # (creates the context and uploads the geometry once)
ctx, gg = sim0.context, sim0.gpu_geometry
rng = np.random.default_rng(1)
d = rng.standard_normal((n, 3), dtype=np.float32); d /= np.linalg.norm(d, axis=1)[:, None]
pol = np.cross(d, np.roll(d, 1, axis=1)).astype(np.float32); pol /= np.linalg.norm(pol, axis=1)[:, None]
host = Photons(np.zeros((n, 3), np.float32), d, pol, np.full(n, 400.0, np.float32))
rs = gpu.get_rng_states(64, seed=7)
for rep in range(3):
    t0 = time.perf_counter()
    gp = gpu.GPUPhotons(host)
    ctx.synchronize(); t1 = time.perf_counter()
    gp.propagate(gg, rs, max_steps=100)
    ctx.synchronize(); t2 = time.perf_counter()
    hits = gp.get_flat_hits(gg)
    t3 = time.perf_counter()
    print('%s, %d host photons: upload %.3f s (%.1f GB/s), propagate %.3f s, flat hits to host (%d) %.3f s -> %.3g photons/s '
          'end to end, %.3g with resident inputs' % (config, n, t1 - t0, n * 64e-9 / (t1 - t0), t2 - t1, len(hits), t3 - t2,
                                                     n / (t3 - t0), n / (t3 - t1)), flush=True)
    del gp, hits

# the whole batch loop of Simulation (chroma/sim.py:58-139): several batches of host photons, hits back to the host,
# with and without the prefetching upload (second thread + second stream + pooled device arrays)
nb = 4
for prefetch in (False, True):
    sim = sim0
    sim.prefetch = prefetch
    list(sim.simulate([host], keep_hits=False, photons_per_batch=n, max_steps=100))            # warm-up: pool, staging buffers
    t0 = time.perf_counter()
    nhits = sum(len(ev.flat_hits) for ev in sim.simulate([host] * nb, keep_hits=False, photons_per_batch=n, max_steps=100))
    dt = time.perf_counter() - t0
    print('Simulation.simulate, %d batches of %d host photons, prefetch=%s: %.3f s -> %.3g photons/s end to end (%d flat hits on the host)' % (
        nb, n, prefetch, dt, nb * n / dt, nhits), flush=True)
# ... and with the per-channel dictionaries of the DEFAULT call (keep_hits=True: ev.hits, chroma/sim.py:122-123), which the
# reference builds with one mask over all hits per channel and this package from one ordering of the hits (chroma_amd/sim.py)
sim0.prefetch = True
t0 = time.perf_counter()
nch = sum(len(ev.hits) for ev in sim0.simulate([host] * 2, photons_per_batch=n, max_steps=100))
dt = time.perf_counter() - t0
print('Simulation.simulate, 2 batches of %d host photons, keep_hits=True (default): %.3f s -> %.3g photons/s end to end (%d channel entries)' % (
    n, dt, 2 * n / dt, nch), flush=True)
