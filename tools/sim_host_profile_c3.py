import sys, os, cProfile, pstats, io, time
sys.path.insert(0, os.getcwd())
import numpy as np
from chroma_amd import demo, gpu
from chroma_amd.event import Photons
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.sim import Simulation
geo = create_geometry_from_obj(demo.detector29k())
n = 50_000_000
rng = np.random.default_rng(1)
d = rng.standard_normal((n, 3), dtype=np.float32); d /= np.linalg.norm(d, axis=1)[:, None]
pol = np.cross(d, np.roll(d, 1, axis=1)).astype(np.float32); pol /= np.linalg.norm(pol, axis=1)[:, None]
host = Photons(np.zeros((n, 3), np.float32), d, pol, np.full(n, 400.0, np.float32))
sim = Simulation(geo, seed=5)
list(sim.simulate([host], photons_per_batch=n, max_steps=100))
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter(); k = sum(len(ev.hits) for ev in sim.simulate([host] * 2, photons_per_batch=n, max_steps=100)); dt = time.perf_counter() - t0
pr.disable()
print('%.3f s per batch' % (dt / 2))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(16); print(s.getvalue())
