"""Diagnostic: time the first propagate launch of several freshly generated batches and show
where their arrays live (looking for an address-dependent slowdown)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from chroma_amd import demo, gpu, _lib
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.gpu.photon import _structure
from chroma_amd.gpu.tools import vec, empty

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
geo = create_geometry_from_obj(demo.detector_lite())
ctx = gpu.create_cuda_context(0)
gg = gpu.GPUDetector(geo)
lib = ctx._lib

class Batch(object):
    def __init__(self):
        self.pos, self.dir, self.pol = empty(n, vec.float3, ctx), empty(n, vec.float3, ctx), empty(n, vec.float3, ctx)
        self.wavelengths, self.t, self.weights = empty(n, np.float32, ctx), empty(n, np.float32, ctx), empty(n, np.float32, ctx)
        self.flags, self.evidx, self.rng_counters = empty(n, np.uint32, ctx), empty(n, np.uint32, ctx), empty(n, np.uint32, ctx)
        self.last_hit_triangles = empty(n, np.int32, ctx)
        self.struct = _structure(self)
    def gen(self, id_base):
        pos = (ctypes.c_float * 3)(0.0, 0.0, 0.0)
        _lib.check(lib.chroma_generate_bomb(ctx.handle, ctypes.byref(self.struct), n, 12345, id_base, pos, 400.0, 0.0))
        ctx.synchronize()
    def step(self, id_base, max_steps=1):
        st = _lib.PropagateStats(); ab = ctypes.c_int32()
        t = time.perf_counter()
        _lib.check(lib.chroma_propagate(ctx.handle, gg.handle, ctypes.byref(self.struct), n, 1, _lib.Rng(12345, id_base), max_steps, 0, 0, 1, ctypes.byref(st), ctypes.byref(ab)))
        return 1e3 * (time.perf_counter() - t), st.kernel_ms

batches = [Batch() for _ in range(5)]
for rnd in range(3):
    for i, b in enumerate(batches):
        base = (rnd * 5 + i) * n
        b.gen(base)
        wall, k = b.step(base)
        print('round %d batch %d id_base %12d pos=%x dir=%x flags=%x : first step %.1f ms (kernel %.1f)' % (rnd, i, base, b.pos.ptr, b.dir.ptr, b.flags.ptr, wall, k), flush=True)
# same batch, same ids, repeated
b = batches[0]
for rep in range(3):
    b.gen(0)
    print('repeat id_base 0: %.1f ms' % b.step(0)[0], flush=True)
for rep in range(3):
    b.gen(3 * n)
    print('repeat id_base 3n: %.1f ms' % b.step(3 * n)[0], flush=True)
