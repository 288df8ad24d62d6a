"""cProfile of the HOST side of Simulation.simulate (1e6-photon events on demo.detector_lite(), one lane): where an end-to-end batch
spends its time beside the ~4 ms the library call takes.  usage (GPU box): python tools/sim_host_profile.py"""
import sys, os, cProfile, pstats, io
sys.path.insert(0, os.getcwd())
import numpy as np, oracle
from chroma_amd import demo
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.sim import Simulation
geo = create_geometry_from_obj(demo.detector_lite())
evs = [oracle.generate_bomb(1000000, seed=500 + k) for k in range(4)]
sim = Simulation(geo, seed=5)
def run(n):
    return sum(len(ev.flat_hits) for ev in sim.simulate((evs[k % 4] for k in range(n)), photons_per_batch=1000000, max_steps=100))
run(4)
pr = cProfile.Profile(); pr.enable(); run(16); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(28); print(s.getvalue())
