"""photons/s against batch size (one propagate + hit extraction per batch, device-resident photons)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chroma_amd import demo, gpu, event
from chroma_amd.loader import create_geometry_from_obj

config = sys.argv[1] if len(sys.argv) > 1 else 'lite'
geo = create_geometry_from_obj({'tiny': demo.tiny, 'lite': demo.detector_lite}[config]())
ctx = gpu.create_cuda_context(0)
gg = gpu.GPUDetector(geo)
rs = gpu.get_rng_states(64, seed=5)
print('# %s: photons per batch, ms per batch, photons/s' % config)
for n in (1_000, 10_000, 100_000, 1_000_000, 10_000_000):
    times = []
    for rep in range(6):
        gp = gpu.GPUPhotons.bomb(n, seed=100 + rep, wavelength=400.0) if hasattr(gpu.GPUPhotons, 'bomb') else None
        if gp is None:
            from chroma_amd.event import Photons
            rng = np.random.default_rng(rep)
            th = rng.uniform(0, 2 * np.pi, n); u = rng.uniform(-1, 1, n); c = np.sqrt(1 - u * u)
            d = np.column_stack([c * np.cos(th), c * np.sin(th), u])
            pol = np.cross(d, [0.3, 0.5, 0.81]); pol /= np.linalg.norm(pol, axis=1)[:, None]
            gp = gpu.GPUPhotons(Photons(np.zeros((n, 3)), d, pol, np.full(n, 400.0)))
        ctx.synchronize()
        t0 = time.perf_counter()
        gp.propagate(gg, rs, max_steps=100)
        nh = gp.get_flat_hits(gg) if n <= 100_000 else None
        ctx.synchronize()
        times.append(time.perf_counter() - t0)
    t = np.median(times[1:])
    print('%10d  %9.3f ms  %.3g photons/s' % (n, 1e3 * t, n / t), flush=True)
