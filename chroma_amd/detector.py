"""Detector = Geometry whose solids can be read-out channels ("PMTs").

Mirrors chroma/detector.py:5-141: channel bookkeeping in ``add_pmt`` and the single
shared time / charge response CDF pair used by the DAQ.
"""
import numpy as np

from chroma_amd.geometry import Geometry


class Detector(Geometry):
    def __init__(self, detector_material=None):
        Geometry.__init__(self, detector_material=detector_material)
        self.solid_id_to_channel_index = []
        self.channel_index_to_solid_id = []
        self.channel_index_to_channel_type = []
        self.channel_index_to_position = []
        # zero time spread and unit charge until set_*_dist is called
        self.time_cdf = (np.array([-0.00000001, 0.00000001]), np.array([0.0, 1.0]))
        self.charge_cdf = (np.array([0.999999999, 1.00000000]), np.array([0.0, 1.0]))

    def add_solid(self, solid, rotation=None, displacement=None):
        solid_id = Geometry.add_solid(self, solid=solid, rotation=rotation, displacement=displacement)
        self.solid_id_to_channel_index.append(-1)   # not a channel unless add_pmt says so
        return solid_id

    def add_pmt(self, pmt, rotation=None, displacement=None, channel_type=None):
        """Add a solid that is a read-out channel.  Returns a dict with ``solid_id``,
        ``channel_index`` (dense, starting at 0) and ``channel_type`` (defaults to the index)."""
        solid_id = self.add_solid(solid=pmt, rotation=rotation, displacement=displacement)
        channel_index = len(self.channel_index_to_solid_id)
        if channel_type is None:
            channel_type = channel_index
        self.solid_id_to_channel_index[solid_id] = channel_index
        self.channel_index_to_solid_id.append(solid_id)
        self.channel_index_to_channel_type.append(channel_type)
        self.channel_index_to_position.append(displacement)
        return {'solid_id': solid_id, 'channel_index': channel_index, 'channel_type': channel_type}

    @staticmethod
    def _pdf_to_cdf(bin_edges, bin_contents):
        """(cdf_x, cdf_y) of a binned PDF; cdf_x are the bin edges.

        As in the reference (chroma/detector.py:109-112, where ``[0.0] + cumsum`` is a
        broadcast add, not a concatenation) cdf_y has one entry per BIN: it starts at the
        first bin's content, not at 0, and is one shorter than cdf_x."""
        cdf_y = 0.0 + np.cumsum(bin_contents)
        cdf_y /= cdf_y[-1]
        return (np.copy(bin_edges), cdf_y)

    def set_time_dist_gaussian(self, rms, lo, hi, nsamples=50):
        edges = np.linspace(lo, hi, nsamples + 1, endpoint=True)
        self.time_cdf = self._pdf_to_cdf(edges, np.exp(-0.5 * (edges[1:] / rms) ** 2))

    def set_time_dist(self, bin_edges, bin_contents):
        self.time_cdf = self._pdf_to_cdf(bin_edges, bin_contents)

    def set_charge_dist_gaussian(self, mean, rms, lo, hi, nsamples=50):
        edges = np.linspace(lo, hi, nsamples + 1, endpoint=True)
        self.charge_cdf = self._pdf_to_cdf(edges, np.exp(-0.5 * ((edges[1:] - mean) / rms) ** 2))

    def num_channels(self):
        return len(self.channel_index_to_channel_type)

    def flatten(self):
        self.solid_id_to_channel_index = np.asarray(self.solid_id_to_channel_index, dtype=np.int32)
        self.channel_index_to_solid_id = np.asarray(self.channel_index_to_solid_id, dtype=np.int32)
        self.channel_index_to_channel_type = np.asarray(self.channel_index_to_channel_type, dtype=np.int32)
        self.channel_index_to_position = np.asarray(self.channel_index_to_position, dtype=np.float32)
        Geometry.flatten(self)
