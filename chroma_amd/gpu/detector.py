"""GPUDetector = GPUGeometry + channel map and DAQ response tables
(reference: chroma/gpu/detector.py:14-40)."""
import numpy as np

from chroma_amd.gpu.geometry import GPUGeometry
from chroma_amd.gpu.tools import to_gpu


class GPUDetector(GPUGeometry):
    def __init__(self, detector, wavelengths=None, print_usage=False, packed=None):
        GPUGeometry.__init__(self, detector, wavelengths=wavelengths, print_usage=False, packed=packed)
        self.solid_id_to_channel_index_gpu = self._device_array('solid_id_to_channel_index', np.int32)
        self.nchannels = detector.num_channels()
        self.time_cdf_x_gpu = to_gpu(detector.time_cdf[0].astype(np.float32), ctx=self.ctx)
        self.time_cdf_y_gpu = to_gpu(detector.time_cdf[1].astype(np.float32), ctx=self.ctx)
        self.charge_cdf_x_gpu = to_gpu(detector.charge_cdf[0].astype(np.float32), ctx=self.ctx)
        self.charge_cdf_y_gpu = to_gpu(detector.charge_cdf[1].astype(np.float32), ctx=self.ctx)
        # q_int = round(q / charge_unit) (chroma/cuda/detector.h:17-21)
        self.charge_unit = np.float32(detector.charge_cdf[0][-1] / 2 ** 16)
        self.detector_gpu = self.handle    # the geometry handle carries the channel map
        if print_usage:
            self.print_device_usage()
