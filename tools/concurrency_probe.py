"""Small batches are launch- and latency-bound (a step under ~1e5 rays lasts as long as its longest ray): how much of the chip do
SEVERAL batches in flight at once win back?  K contexts on the one GPU, K host threads, each propagating its own batches of n photons
(propagate + per-channel hit arrays, as bench.py's step), against one context doing them one after the other.
usage: concurrency_probe.py [tiny|lite|detector|c3] [photons per batch] [batches per context] [most contexts]"""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chroma_amd import demo, gpu, _lib
from chroma_amd.gpu.tools import Context
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.gpu.geometry import pack_geometry

cfg = sys.argv[1] if len(sys.argv) > 1 else 'detector'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 1_000_000
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
builder = {'tiny': demo.tiny, 'detector': demo.detector, 'lite': demo.detector_lite, 'c3': demo.detector29k}[cfg]
ctx0 = gpu.create_cuda_context(0)
packed = pack_geometry(create_geometry_from_obj(builder())).attach_wide_tree()
KMAX = int(sys.argv[4]) if len(sys.argv) > 4 else 8
ctxs = [ctx0] + [Context(0) for _ in range(KMAX - 1)]
ggs, batches = [], []
for k, c in enumerate(ctxs):
    c.push()
    ggs.append(gpu.GPUDetector.from_packed(packed))
    batches.append([gpu.generate_bomb(n, 12345, id_base=(k * reps + r) * n, ctx=c) for r in range(2)])     # two buffers per context, refilled
    c.synchronize()


def worker(k, count, out):
    ctx, gg = ctxs[k], ggs[k]
    ctx.push()
    hits = 0
    for r in range(count):
        base = (k * reps + r) * n
        gp = batches[k][r & 1]
        s = gpu.photon._structure(gp)
        pos = (3 * __import__('ctypes').c_float)(0.0, 0.0, 0.0)
        _lib.check(ctx._lib.chroma_generate_bomb(ctx.handle, __import__('ctypes').byref(s), n, 12345, base, pos, 400.0, 0.0))
        gp.propagate(gg, _lib.Rng(12345, base), max_steps=100)
        c, e = gp.channel_hits(gg)
        hits += int(c.get().sum())
    ctx.synchronize()
    out[k] = hits


for K in [k for k in (1, 2, 3, 4, 8) if k <= KMAX]:
    for trial in range(2):
        out = {}
        th = [threading.Thread(target=worker, args=(k, reps, out)) for k in range(K)]
        t0 = time.perf_counter()
        for t in th: t.start()
        for t in th: t.join()
        dt = time.perf_counter() - t0
    print('%s, %d photons per batch, %d contexts x %d batches: %.1f ms -> %.3g photons/s (%.2f ms per batch per context), hits %d' % (
        cfg, n, K, reps, 1e3 * dt, K * reps * n / dt, 1e3 * dt / reps, sum(out.values())), flush=True)
