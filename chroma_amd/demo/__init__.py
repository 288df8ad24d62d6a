"""Demo detectors: a water sphere lined with PMTs on a spherical spiral.

``detector()``, ``tiny()`` follow chroma/demo/__init__.py:19-67; ``detector29k()`` is
the 29k-PMT geometry BASELINE.md names C3/C4 (same builder, radii scaled so the
350 mm pitch is kept).
"""
from math import sin, cos, sqrt

import numpy as np

from chroma_amd.make import sphere
from chroma_amd.geometry import Solid
from chroma_amd.detector import Detector
from chroma_amd.transform import make_rotation_matrix, normalize
from chroma_amd.demo.pmt import build_8inch_pmt_with_lc
from chroma_amd.demo.optics import water, black_surface
from chroma_amd.log import logger


def spherical_spiral(radius, spacing):
    """Points roughly ``spacing`` apart along a spiral wrapping a sphere pole to pole."""
    dl = spacing / radius
    t = 0.0
    a = np.pi / dl
    while t < np.pi:
        yield np.array([sin(t) * sin(a * t), sin(t) * cos(a * t), cos(t)]) * radius
        t += dl / sqrt(1 + a ** 2 * sin(t) ** 2)


def detector(pmt_radius=14000.0, sphere_radius=14500.0, spiral_step=350.0, pmt_profile=None, cone_profile=None):
    """chroma/demo/__init__.py:32-64.  ``pmt_profile`` / ``cone_profile`` (or $CHROMA_PMT_PROFILE / $CHROMA_CONE_PROFILE): the
    reference's digitised outlines (chroma/demo/sno_pmt.txt, sno_cone.txt) where they are at hand, instead of the analytic
    tube of chroma_amd/demo/pmt.py."""
    pmt = build_8inch_pmt_with_lc(pmt_profile=pmt_profile, cone_profile=cone_profile)
    geo = Detector(water)
    geo.add_solid(Solid(sphere(sphere_radius, nsteps=200), water, water,
                        surface=black_surface, color=0xBBFFFFFF))
    y_axis = np.array((0.0, 1.0, 0.0))
    for position in spherical_spiral(pmt_radius, spiral_step):
        direction = -normalize(position)
        # the PMT is built facing +y: turn it to face the centre
        axis = np.cross(direction, y_axis)
        angle = np.arccos(np.dot(y_axis, direction))
        geo.add_pmt(pmt, make_rotation_matrix(angle, axis), position)

    time_rms = 1.5       # ns
    charge_mean = 1.0
    charge_rms = 0.1
    geo.set_time_dist_gaussian(time_rms, -5 * time_rms, 5 * time_rms)
    geo.set_charge_dist_gaussian(charge_mean, charge_rms, 0.0, charge_mean + 5 * charge_rms)
    logger.info('Demo detector: %d PMTs' % geo.num_channels())
    logger.info('               %1.1f ns time RMS' % time_rms)
    logger.info('               %1.1f%% charge RMS' % (100.0 * charge_rms / charge_mean))
    return geo


def tiny(**profiles):
    return detector(2000.0, 2500.0, 700.0, **profiles)


def detector_lite(**profiles):
    """~500 PMTs (BASELINE.md "C2-lite"), for quick turn-around only."""
    return detector(pmt_radius=3120.0, sphere_radius=3620.0, spiral_step=350.0, **profiles)


def detector29k(**profiles):
    """29 007 PMTs at the demo pitch."""
    return detector(pmt_radius=23780.0, sphere_radius=24280.0, spiral_step=350.0, **profiles)


def scintillator_stress():
    """Scintillator + thin film + WLS + dichroic + detecting surface."""
    from chroma_amd.demo.stress import scintillator_stress as build
    return build()
