"""Large-sample parity sweep on the GPU box: engine vs oracle, bit for bit, on random bombs and on rays
aimed at mesh vertices/edges from several origins.  usage: parity_sweep.py [tiny|lite|detector|c3|c5] [photons per batch] [batches] [exact]
(`exact`: through the exact walk, GPUPhotons.propagate(exact=True) -- then the aimed rays must not differ either)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from chroma_amd import demo, gpu
from chroma_amd.event import Photons
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.gpu.geometry import pack_geometry

config = sys.argv[1] if len(sys.argv) > 1 else 'tiny'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 5_000_000
batches = int(sys.argv[3]) if len(sys.argv) > 3 else 4
exact = len(sys.argv) > 4 and sys.argv[4] == 'exact'
geo = create_geometry_from_obj({'tiny': demo.tiny, 'lite': demo.detector_lite, 'detector': demo.detector, 'c3': demo.detector29k,
                                'c5': demo.scintillator_stress}[config]())
pk = pack_geometry(geo)
gpu.create_cuda_context(0)
gg = gpu.GPUDetector(geo, packed=pk)
wl0 = 350.0 if config == 'c5' else 400.0
FIELDS = ('flags', 'last_hit_triangles', 'pos', 'dir', 'pol', 't', 'wavelengths')


def compare(ph, seed, what):
    rs = gpu.get_rng_states(64, seed=seed)
    gp = gpu.GPUPhotons(ph)
    t0 = time.time(); gp.propagate(gg, rs, max_steps=100, exact=exact); got = gp.get(); t1 = time.time()
    want, _, _ = oracle.propagate(pk, ph, seed=seed, max_steps=100, nthreads=min(64, len(os.sched_getaffinity(0))))
    t2 = time.time()
    bad = np.zeros(len(ph), dtype=bool)
    for f in FIELDS:
        a, b = getattr(got, f), getattr(want, f)
        same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
        bad |= ~same.reshape(len(a), -1).all(axis=1)
    print('%-28s %9d photons: %d differ (engine %.1f s, oracle %.1f s)' % (what, len(ph), int(bad.sum()), t1 - t0, t2 - t1), flush=True)
    return int(bad.sum())


total = 0
for b in range(batches):
    ph = oracle.generate_bomb(n, seed=1000 + b, wavelength_lo=wl0, wavelength_hi=700.0 if b % 2 else 0.0)
    total += compare(ph, 500 + b, 'bomb seed %d' % (1000 + b))
# aimed rays from a few origins
m = geo.mesh
v = m.vertices.astype(np.float64); t = m.triangles
rng = np.random.default_rng(3)
for origin in ([0, 0, 0], [300.0, -200.0, 150.0], [0.0, 0.0, 1200.0]):
    pick = rng.choice(len(t), size=min(len(t), 60000), replace=False)
    tri = v[t[pick]]
    targets = np.concatenate([tri.reshape(-1, 3), 0.5 * (tri[:, 0] + tri[:, 1]), 0.5 * (tri[:, 1] + tri[:, 2]), tri.mean(axis=1)])
    d = targets - np.asarray(origin, dtype=np.float64)
    d = d[np.linalg.norm(d, axis=1) > 1e-9]
    d /= np.linalg.norm(d, axis=1)[:, None]
    pol = np.cross(d, np.roll(d, 1, axis=1) + 1e-3); pol /= np.linalg.norm(pol, axis=1)[:, None]
    ph = Photons(np.tile(np.asarray(origin, dtype=float), (len(d), 1)), d, pol, np.full(len(d), wl0))
    total += compare(ph, 77, 'aimed from %s' % (origin,))
print('TOTAL differing photons (%s walk):' % ('exact' if exact else 'default'), total)
