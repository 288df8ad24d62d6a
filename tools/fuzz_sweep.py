"""More seeds of tests/test_gpu_fuzz.py's generators than the test suite runs: random optical tables (every surface
model, bulk re-emission; plain and weighted) and random triangle soups, engine vs oracle bit for bit.
usage: fuzz_sweep.py [first seed] [count]   (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import importlib.util
spec = importlib.util.spec_from_file_location('fz', os.path.join(ROOT, 'tests', 'test_gpu_fuzz.py'))
fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
import oracle
from conftest import bomb
from chroma_amd import gpu
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.gpu.geometry import pack_geometry
from chroma_amd.geometry import Geometry, Solid
from chroma_amd.demo.optics import water, glass

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
gpu.create_cuda_context(0)
FIELDS = fz.FIELDS + ('weights',)
total_bad = 0
for seed in range(first, first + count):
    for kind in ('optics', 'optics-weights', 'soup'):
        if kind == 'soup':
            g = Geometry(water); g.add_solid(Solid(fz._soup(seed, 3000), glass, water))
            geometry = create_geometry_from_obj(g); gg = gpu.GPUGeometry(geometry); kw = {}
            ph = bomb(30000, seed); ph.pos[:] = np.random.default_rng(seed).uniform(-300, 300, (len(ph), 3))
        else:
            geometry = create_geometry_from_obj(fz._random_optics(seed)); gg = gpu.GPUDetector(geometry)
            kw = dict(use_weights=True, scatter_first=1) if kind.endswith('weights') else {}
            ph = bomb(30000, seed, wavelength=300.0, wavelength_hi=700.0)
        packed = pack_geometry(geometry)
        gp = gpu.GPUPhotons(ph)
        gp.propagate(gg, gpu.get_rng_states(64, seed=seed), max_steps=50, **kw)
        got = gp.get()
        want, counters, _ = oracle.propagate(packed, ph, seed=seed, max_steps=50, nthreads=16, **kw)
        bad = ~(gp.rng_counters.get() == counters)
        for f in FIELDS:
            a, b = getattr(got, f), getattr(want, f)
            same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
            bad |= ~same.reshape(len(a), -1).all(axis=1)
        total_bad += int(bad.sum())
        if bad.any():
            print('seed %d %s: %d of %d photons differ' % (seed, kind, int(bad.sum()), len(ph)), flush=True)
print('seeds %d..%d x (optics, optics-weights, soup): %d photons differ in total' % (first, first + count - 1, total_bad))
