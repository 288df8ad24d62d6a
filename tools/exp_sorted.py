"""Experiment: how much faster is the first ray cast when neighbouring photons have neighbouring directions?"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chroma_amd import demo, gpu
from chroma_amd.event import Photons
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.tools import argsort_direction

cfg = sys.argv[1] if len(sys.argv) > 1 else 'detector_lite'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
geo = create_geometry_from_obj(getattr(demo, cfg)())
ctx = gpu.create_cuda_context(0)
gg = gpu.GPUDetector(geo)
rng = np.random.default_rng(1)
theta = rng.uniform(0, 2 * np.pi, n); u = rng.uniform(-1, 1, n); c = np.sqrt(1 - u * u)
d = np.column_stack([c * np.cos(theta), c * np.sin(theta), u]).astype(np.float32)
pol = np.cross(d, [0, 0, 1.0]).astype(np.float32); pol /= np.linalg.norm(pol, axis=1)[:, None]
for label, order in (('random', np.arange(n)), ('sorted by direction', argsort_direction(d))):
    ph = Photons(np.zeros((n, 3), np.float32), d[order], pol[order], np.full(n, 400.0, np.float32))
    for steps in (1, 2, 3, 100):
        gp = gpu.GPUPhotons(ph)
        rs = gpu.get_rng_states(1, seed=5)
        st = {}
        gp.propagate(gg, rs, max_steps=steps, stats=st, time_kernels=True)
        print('%-20s max_steps=%d: kernels %.1f ms (ray cast %.1f ms) in %d launches' % (label, steps, st['kernel_ms'], st['raycast_ms'], st['launches']), flush=True)
        del gp
