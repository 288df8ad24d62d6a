"""Is there anything to win from running the ray cast of one part of a batch beside the physics of another?  Two contexts on
the one GPU, two host threads, each propagating its own half batch at the same time, against one context doing the whole batch.
usage: overlap_probe.py [c3|detector|lite] [photons]"""
import os, sys, time, threading, ctypes
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from chroma_amd import demo, gpu, _lib
from chroma_amd.gpu.tools import Context
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.gpu.geometry import pack_geometry

cfg = sys.argv[1] if len(sys.argv) > 1 else 'lite'
n = int(float(sys.argv[2])) if len(sys.argv) > 2 else 10_000_000
builder = {'c3': demo.detector29k, 'detector': demo.detector, 'lite': demo.detector_lite}[cfg]
ctx0 = gpu.create_cuda_context(0)
packed = pack_geometry(create_geometry_from_obj(builder())).attach_wide_tree()
ctxs = [ctx0, Context(0)]
ggs = []
for c in ctxs:
    c.push()
    ggs.append(gpu.GPUDetector.from_packed(packed))
ctx0.push()


def batch(ctx, count, base):
    ctx.push()
    return gpu.generate_bomb(count, 12345, id_base=base, ctx=ctx)


def run(ctx, gg, gp, base, out, key):
    t0 = time.perf_counter()
    gp.propagate(gg, _lib.Rng(12345, base), max_steps=100)
    ctx.synchronize()
    out[key] = time.perf_counter() - t0


for rep in range(3):
    whole = batch(ctxs[0], n, 0)
    out = {}
    run(ctxs[0], ggs[0], whole, 0, out, 'whole')
    del whole
    halves = [batch(ctxs[k], n // 2, k * (n // 2)) for k in range(2)]
    for c in ctxs:
        c.synchronize()
    th = [threading.Thread(target=run, args=(ctxs[k], ggs[k], halves[k], k * (n // 2), out, 'half%d' % k)) for k in range(2)]
    t0 = time.perf_counter()
    for t in th: t.start()
    for t in th: t.join()
    both = time.perf_counter() - t0
    one = {}
    h = batch(ctxs[0], n // 2, 0)
    run(ctxs[0], ggs[0], h, 0, one, 'half_alone')
    print('%s %d photons: whole batch %.1f ms | two halves at once %.1f ms (%.1f, %.1f) | one half alone %.1f ms' % (
        cfg, n, 1e3 * out['whole'], 1e3 * both, 1e3 * out['half0'], 1e3 * out['half1'], 1e3 * one['half_alone']), flush=True)
    del halves, h
