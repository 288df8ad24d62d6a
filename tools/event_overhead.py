"""What do the HIP events that chroma_propagate records and reads for `time_kernels` cost?  Same batch, with and without.
usage: event_overhead.py [c5|lite|tiny] [photons]  (GPU box)"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chroma_amd import demo, gpu, _lib
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.gpu.photon import generate_bomb
config = sys.argv[1] if len(sys.argv) > 1 else 'c5'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
geo = create_geometry_from_obj({'tiny': demo.tiny, 'lite': demo.detector_lite, 'c5': demo.scintillator_stress}[config]())
ctx = gpu.create_cuda_context(0)
gg = gpu.GPUDetector(geo)
rs = gpu.get_rng_states(64, seed=3)
for timed in (0, 1, 0, 1):
    ts = []
    for rep in range(4):
        gp = generate_bomb(n, seed=11 + rep, wavelength_lo=350.0 if config == 'c5' else 400.0, ctx=ctx)
        ctx.synchronize()
        t0 = time.perf_counter()
        gp.propagate(gg, rs, max_steps=100, time_kernels=bool(timed))
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    print('%s %d photons, time_kernels %d: %.2f ms per propagate (min of 4: %.2f)' % (config, n, timed, 1e3 * np.mean(ts[1:]), 1e3 * min(ts)))
