"""The reference's own timing harness for this path (chroma/benchmark.py:22-96): ``intersect``
(ray intersections per second through distance_to_mesh), ``load_photons`` (host -> device photon
upload rate) and ``propagate`` (photons per second, default max_steps=10, Morton-sorted isotropic
bomb of U(400, 800) nm photons).  Returns (mean, std) of the rate instead of an ``uncertainties.ufloat``
(that package is not a dependency here).  ``pdf`` / ``pdf_eval`` belong to the likelihood layer (out of scope).
"""
import ctypes
import time

import numpy as np

from chroma_amd import _lib, event, gpu, sample, tools
from chroma_amd.transform import normalize


def _rate(nphotons, run_times):
    t = np.asarray(run_times)
    mean = nphotons / t.mean()
    return mean, mean * (t.std() / t.mean())


def intersect(gpu_geometry, number=100, nphotons=500000, nthreads_per_block=64, max_blocks=1024):
    "Average number of ray intersections per second (mean, std)."
    ctx = gpu_geometry.ctx
    distances = gpu.empty(nphotons, np.float32, ctx)
    run_times = []
    for i in range(number):
        pos = gpu.zeros(3 * nphotons, np.float32, ctx)
        d = sample.uniform_sphere(nphotons)
        d = gpu.to_gpu(np.ascontiguousarray(d[tools.argsort_direction(d)], dtype=np.float32).reshape(-1), ctx)
        ctx.synchronize()
        t0 = time.time()
        _lib.check(ctx._lib.chroma_distance_to_mesh(ctx.handle, gpu_geometry.handle, nphotons, pos.ptr, d.ptr, distances.ptr, None))
        ctx.synchronize()
        if i > 0:       # the first call pays one-off costs
            run_times.append(time.time() - t0)
    return _rate(nphotons, run_times)


def _bomb(nphotons):
    pos = np.zeros((nphotons, 3))
    d = sample.uniform_sphere(nphotons)
    d = d[tools.argsort_direction(d)]
    pol = normalize(np.cross(sample.uniform_sphere(nphotons), d))
    return event.Photons(pos, d, pol, np.random.uniform(400, 800, size=nphotons))


def load_photons(number=100, nphotons=500000):
    "Average number of photons moved to device memory per second (mean, std)."
    photons = _bomb(nphotons)
    ctx = gpu.get_context()
    run_times = []
    for i in range(number):
        t0 = time.time()
        gp = gpu.GPUPhotons(photons)
        ctx.synchronize()
        if i > 0:
            run_times.append(time.time() - t0)
        del gp
    return _rate(nphotons, run_times)


def propagate(gpu_detector, number=10, nphotons=500000, nthreads_per_block=64, max_blocks=1024):
    "Average number of photons propagated per second (mean, std); photons resident when the clock starts."
    rng_states = gpu.get_rng_states(nthreads_per_block * max_blocks)
    ctx = gpu_detector.ctx
    run_times = []
    for i in range(number):
        gp = gpu.GPUPhotons(_bomb(nphotons))
        ctx.synchronize()
        t0 = time.time()
        gp.propagate(gpu_detector, rng_states, nthreads_per_block, max_blocks)
        ctx.synchronize()
        if i > 0:
            run_times.append(time.time() - t0)
        del gp
    return _rate(nphotons, run_times)
