"""Step one photon of the aimed-ray test through engine and oracle side by side (GPU box)."""
import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from chroma_amd import demo, event, gpu
from chroma_amd.event import Photons
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.gpu.geometry import pack_geometry
g = create_geometry_from_obj(demo.tiny())
pk = pack_geometry(g)
src = open(os.path.join(ROOT, 'tests', 'test_gpu_parity.py')).read()
ns = {'np': np, 'Photons': Photons}
exec(src[src.index('def _aimed_photons'):src.index("@pytest.mark.parametrize('count'")], ns)
ph = ns['_aimed_photons'](g, (0.0, 0.0, 0.0), 20000)
which = [int(x) for x in sys.argv[1:]] or [3385]
gpu.create_cuda_context(0)
gg = gpu.GPUDetector(g)
for mode in ('coop', 'wide', 'reference'):
    gpu.get_context().set_walk(mode)
    # whole batch, stepwise, so that the photon takes the same kernels as in the test
    gp = gpu.GPUPhotons(ph)
    rs = gpu.get_rng_states(64 * 1024, seed=12345)
    cur, ctr = ph, None
    for step in range(4):
        gp.propagate(gg, rs, max_steps=1)
        cur, ctr, _ = oracle.propagate(pk, cur, seed=12345, max_steps=1, rng_counters=ctr)
        got = gp.get()
        bad = np.nonzero((got.last_hit_triangles != cur.last_hit_triangles) | (got.flags != cur.flags))[0]
        print(mode, 'step', step, 'mismatching photons:', bad[:10], 'of', len(ph))
        for i in list(bad[:3]):
            print('   photon', i, 'engine tri', got.last_hit_triangles[i], hex(got.flags[i]), 'pos', got.pos[i], '| oracle tri', cur.last_hit_triangles[i], hex(cur.flags[i]), 'pos', cur.pos[i])
        if len(bad):
            break
