"""GPUDaq / GPUChannels: per-channel earliest hit time, charge and history
(reference: chroma/gpu/daq.py:8-100 over chroma/cuda/daq.cu).

``ndaq == 1`` runs ``run_daq``; ``ndaq > 1`` runs ``run_daq_many`` (daq.cu:88-150): that many
independent acquisitions of the same photons side by side, each with a unit normal jitter on the hit
time (``GPUChannels.iterate_copies`` walks them).  The random numbers a detected photon consumes
(weight gate, jitter, time smear, charge) come from Philox stream ``1 + acquisition`` of that photon
(copy i from word 8 i on), so propagation draws are not disturbed and two acquisitions of the same
photons differ.
"""
import ctypes

import numpy as np

from chroma_amd import _lib, event
from chroma_amd.gpu.tools import GPUArray, get_context, empty, zeros, to_gpu, RNGStates
from chroma_amd.gpu.photon import _structure


class GPUChannels(object):
    def __init__(self, t, q, flags, ndaq=1, stride=None):
        self.t = t
        self.q = q
        self.flags = flags
        self.ndaq = ndaq
        self.stride = len(t) if stride is None else stride

    def iterate_copies(self):
        for i in range(self.ndaq):
            w = slice(i * self.stride, (i + 1) * self.stride)
            yield GPUChannels(self.t[w], self.q[w], self.flags[w])

    def get(self):
        t = self.t.get()
        q = self.q.get()
        # as in the reference: a channel whose time is still the reset value (1e9) was not hit
        return event.Channels(t < 1e8, t, q, self.flags.get())

    def __len__(self):
        return self.t.size


def _padded_cdf(cdf_x, cdf_y):
    """Device CDF tables of len(cdf_x) points each.  Detector._pdf_to_cdf yields a cdf_y that is
    one entry SHORTER than cdf_x (chroma/detector.py:109-112) and the reference then reads
    cdf_y[len(cdf_x)-1] past the end of its device array (chroma/gpu/detector.py:34-36 passes
    len(cdf_x)); here that entry exists and is 1.0, the value a CDF ends with."""
    x = np.asarray(cdf_x, dtype=np.float32)
    y = np.asarray(cdf_y, dtype=np.float32)
    if len(y) < len(x):
        y = np.concatenate([y, np.ones(len(x) - len(y), dtype=np.float32)])
    return x, y[:len(x)]


class GPUDaq(object):
    def __init__(self, gpu_detector, ndaq=1):
        if ndaq < 1:
            raise ValueError('ndaq must be at least 1')
        self.ctx = gpu_detector.ctx
        self.gpu_detector = gpu_detector
        n = gpu_detector.nchannels * ndaq
        self.earliest_time_gpu = empty(n, np.float32, self.ctx)
        self.earliest_time_int_gpu = empty(n, np.uint32, self.ctx)
        self.channel_history_gpu = zeros(n, np.uint32, self.ctx)
        self.channel_q_int_gpu = zeros(n, np.uint32, self.ctx)
        self.channel_q_gpu = zeros(n, np.float32, self.ctx)
        det = gpu_detector.geometry
        tx, ty = _padded_cdf(*det.time_cdf)
        qx, qy = _padded_cdf(*det.charge_cdf)
        self._tables_host = (tx, ty, qx, qy)
        self._arrays = [to_gpu(a, ctx=self.ctx) for a in (tx, ty, qx, qy)]
        self.charge_unit = float(np.float32(det.charge_cdf[0][-1] / 2 ** 16))
        self.tables = _lib.DaqTables(self._arrays[0].ptr, self._arrays[1].ptr, len(tx),
                                     self._arrays[2].ptr, self._arrays[3].ptr, len(qx), self.charge_unit)
        self.ndaq = ndaq
        self.stride = gpu_detector.nchannels
        self.acquisition = 0

    def begin_acquire(self, nthreads_per_block=64):
        _lib.check(self.ctx._lib.chroma_daq_reset(self.ctx.handle, 1e9, len(self.earliest_time_int_gpu),
                                                  self.earliest_time_int_gpu.ptr, self.channel_q_int_gpu.ptr,
                                                  self.channel_history_gpu.ptr))
        self.channel_q_gpu.fill(0)

    def acquire(self, gpuphotons, rng_states, nthreads_per_block=64, max_blocks=1024, start_photon=None,
                nphotons=None, weight=1.0):
        if start_photon is None:
            start_photon = 0
        if nphotons is None:
            nphotons = len(gpuphotons.pos) - start_photon
        rng = gpuphotons._rng(rng_states)
        s = _structure(gpuphotons)
        if self.ndaq == 1:
            _lib.check(self.ctx._lib.chroma_daq_acquire(self.ctx.handle, self.gpu_detector.handle, ctypes.byref(self.tables),
                                                        int(start_photon), int(nphotons), event.SURFACE_DETECT,
                                                        ctypes.byref(s), rng, self.acquisition, float(weight),
                                                        self.earliest_time_int_gpu.ptr, self.channel_q_int_gpu.ptr,
                                                        self.channel_history_gpu.ptr))
        else:
            _lib.check(self.ctx._lib.chroma_daq_acquire_many(self.ctx.handle, self.gpu_detector.handle, ctypes.byref(self.tables),
                                                             int(start_photon), int(nphotons), event.SURFACE_DETECT,
                                                             ctypes.byref(s), rng, self.acquisition, float(weight),
                                                             int(self.ndaq), int(self.stride),
                                                             self.earliest_time_int_gpu.ptr, self.channel_q_int_gpu.ptr,
                                                             self.channel_history_gpu.ptr))
        self.acquisition += 1
        self.ctx.synchronize()

    def end_acquire(self, nthreads_per_block=64):
        _lib.check(self.ctx._lib.chroma_daq_convert(self.ctx.handle, len(self.earliest_time_int_gpu), self.charge_unit,
                                                    self.earliest_time_int_gpu.ptr, self.channel_q_int_gpu.ptr,
                                                    self.earliest_time_gpu.ptr, self.channel_q_gpu.ptr))
        self.ctx.synchronize()
        return GPUChannels(self.earliest_time_gpu, self.channel_q_gpu, self.channel_history_gpu, self.ndaq, self.stride)
