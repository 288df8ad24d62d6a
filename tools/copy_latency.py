import sys, time, numpy as np
sys.path.insert(0, '.')
from chroma_amd import gpu
from chroma_amd.gpu.tools import GPUArray, to_gpu
ctx = gpu.create_cuda_context(0)
for n in (1, 1000, 30000, 1000000):
    a = to_gpu(np.arange(n, dtype=np.uint32), ctx)
    a.get()
    t0 = time.perf_counter()
    for _ in range(200): a.get()
    t1 = time.perf_counter()
    for _ in range(200): ctx.synchronize()
    t2 = time.perf_counter()
    print('get() of %8d uint32: %.1f us each; synchronize alone %.1f us' % (n, (t1 - t0) / 200 * 1e6, (t2 - t1) / 200 * 1e6))
