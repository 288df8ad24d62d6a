import sys, time, os
sys.path.insert(0, os.getcwd())
import numpy as np, oracle
from chroma_amd import demo, gpu
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.sim import Simulation
cfg = sys.argv[1]; nev = int(sys.argv[2]); per_ev = int(float(sys.argv[3])); per_batch = int(float(sys.argv[4]))
geo = create_geometry_from_obj({'tiny': demo.tiny, 'lite': demo.detector_lite, 'detector': demo.detector}[cfg]())
evs = [oracle.generate_bomb(per_ev, seed=500 + k) for k in range(8)]
def events():
    for k in range(nev):
        yield evs[k % 8]
for lanes in (1, 2, 4):
    sim = Simulation(geo, seed=5, lanes=lanes)
    n = sum(len(ev.flat_hits) for ev in sim.simulate(events(), photons_per_batch=per_batch, max_steps=100))
    t0 = time.perf_counter()
    n = sum(len(ev.flat_hits) for ev in sim.simulate(events(), photons_per_batch=per_batch, max_steps=100))
    dt = time.perf_counter() - t0
    print('%s: %d events of %d photons, batches of %d, lanes %d: %.1f ms -> %.3g photons/s end to end (%d hits)' % (cfg, nev, per_ev, per_batch, lanes, 1e3 * dt, nev * per_ev / dt, n), flush=True)
    del sim
