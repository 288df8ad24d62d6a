"""The library's own RCCL path (chroma_comm_* + chroma_allreduce_hits / chroma_allreduce_daq) on the one
GPU of the test box: a one-rank communicator, so every collective is the identity -- what is exercised
is finding RCCL (dlopen), creating a communicator from a unique id, the grouped all-reduces and the
all-gather + OR kernel on the library's stream, and that the device arrays come back unchanged.  (More
ranks need more GPUs: RCCL refuses two ranks on one device.  The arithmetic of the reduction across
ranks is covered on the CPU over gloo in tests/test_dist_cpu.py.)"""
import ctypes

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_reductions_are_the_identity(oracle_mod, tiny_geometry):
    from chroma_amd import gpu, _lib, event
    from chroma_amd.gpu.tools import to_gpu
    ctx = gpu.create_cuda_context(0)
    try:
        lib = ctx._lib
        # without a communicator: identity, no RCCL needed
        a = to_gpu(np.arange(100, dtype=np.uint32), ctx)
        b = to_gpu(np.arange(100, dtype=np.uint32)[::-1].copy(), ctx)
        _lib.check(lib.chroma_allreduce_hits(ctx.handle, a.ptr, b.ptr, 100))
        assert np.array_equal(a.get(), np.arange(100))
        ident = (ctypes.c_uint8 * 128)()
        _lib.check(lib.chroma_comm_unique_id(ident))
        assert any(bytes(ident))
        _lib.check(lib.chroma_comm_init(ctx.handle, 1, 0, ident))
        assert lib.chroma_comm_init(ctx.handle, 1, 0, ident) != 0           # one communicator per context
        # the per-channel arrays of a real batch
        gg = gpu.GPUDetector(tiny_geometry)
        gp = gpu.GPUPhotons(oracle_mod.generate_bomb(50000, seed=5))
        gp.propagate(gg, gpu.get_rng_states(64, seed=2), max_steps=100)
        counts, earliest = gp.channel_hits(gg)
        c0, e0 = counts.get(), earliest.get()
        assert c0.sum() > 100
        from chroma_amd.dist import allreduce_channel_hits_device
        allreduce_channel_hits_device(ctx, counts, earliest)
        assert np.array_equal(counts.get(), c0) and np.array_equal(earliest.get(), e0)
        # DAQ accumulators: min, sum, OR (all-gather + OR kernel)
        rng = np.random.default_rng(3)
        t = rng.integers(0, 2 ** 30, 53, dtype=np.uint32)
        q = rng.integers(0, 2 ** 20, 53, dtype=np.uint32)
        h = rng.integers(0, 2 ** 12, 53, dtype=np.uint32)
        dt, dq, dh = to_gpu(t, ctx), to_gpu(q, ctx), to_gpu(h, ctx)
        _lib.check(lib.chroma_allreduce_daq(ctx.handle, dt.ptr, dq.ptr, dh.ptr, 53))
        assert np.array_equal(dt.get(), t) and np.array_equal(dq.get(), q) and np.array_equal(dh.get(), h)
        _lib.check(lib.chroma_comm_destroy(ctx.handle))
        _lib.check(lib.chroma_allreduce_hits(ctx.handle, counts.ptr, earliest.ptr, len(c0)))      # identity again
    finally:
        ctx.pop()
