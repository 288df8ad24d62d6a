"""Where Simulation(prefetch=True) spends a batch: how long the main thread waits for the next upload, how long propagate and
the hit download take while the second thread uploads, against the same steps run one after the other.
usage: sim_overlap_probe.py [config] [photons]  (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from concurrent.futures import ThreadPoolExecutor
from chroma_amd import demo, gpu, event
from chroma_amd.event import Photons
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.sim import Simulation

config = sys.argv[1] if len(sys.argv) > 1 else 'c3'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 50_000_000
geo = create_geometry_from_obj({'tiny': demo.tiny, 'lite': demo.detector_lite, 'detector': demo.detector, 'c3': demo.detector29k}[config]())
sim = Simulation(geo, seed=5, prefetch=True)
ctx, gg = sim.context, sim.gpu_geometry
rng = np.random.default_rng(1)
d = rng.standard_normal((n, 3), dtype=np.float32); d /= np.linalg.norm(d, axis=1)[:, None]
pol = np.cross(d, np.roll(d, 1, axis=1)).astype(np.float32); pol /= np.linalg.norm(pol, axis=1)[:, None]
host = Photons(np.zeros((n, 3), np.float32), d, pol, np.full(n, 400.0, np.float32))
ev = event.Event(photons_beg=host)
T = time.perf_counter


def upload():
    t0 = T()
    up = sim._upload_batch([ev])
    return up, t0, T()


def work(up):
    t0 = T()
    gp, bounds = up
    st = {}
    gp.propagate(gg, sim.rng_states, max_steps=100, stats=st, time_kernels=True)
    ctx.synchronize()
    t1 = T()
    hits = gp.get_flat_hits(gg)
    t2 = T()
    return t0, t1, t2, st.get('kernel_ms', 0.0)


for label, overlapped in (('one after the other', False), ('upload of the next batch on a second thread', True)):
    up, _, _ = upload(); work(up)                 # warm-up
    t_begin = T()
    rows = []
    with ThreadPoolExecutor(max_workers=1) as pool:
        fut = pool.submit(upload) if overlapped else None
        for k in range(5):
            tw0 = T()
            if overlapped:
                up, u0, u1 = fut.result()
                fut = pool.submit(upload) if k < 4 else None
            else:
                up, u0, u1 = upload()
            tw1 = T()
            t0, t1, t2, nh = work(up)
            rows.append((tw1 - tw0, u1 - u0, t1 - t0, nh * 1e-3, t2 - t1))
            up = None
    total = T() - t_begin
    print('%s: %.3f s for 5 batches of %d -> %.3g photons/s' % (label, total, n, 5 * n / total))
    for r in rows:
        print('   waited for the upload %.3f s (the upload itself took %.3f s), propagate %.3f s (%.3f s inside its kernels), flat hits %.3f s' % r)
