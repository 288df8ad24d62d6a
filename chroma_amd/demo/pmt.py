"""Demo 8-inch PMT with a light-collecting cone.

Same construction as chroma/demo/pmt.py:7-20 (3 mm glass, R7081HQE photocathode on the
upper half of the vacuum envelope, mirror back surface, reflective cone, nsteps=24),
but the glass outline is an analytic 8-inch tube -- cylindrical neck, ellipsoidal
bulb -- sampled with as many points as the digitised SNO tube drawing the reference
ships (55 per half outline, 16 on the cone), so a PMT has the same triangle count
(5 856 with its collector) and the demo detectors have the sizes quoted in BASELINE.md.
Files in the reference's profile format still load through chroma_amd.pmt.build_pmt.

A site that HAS the reference's digitised drawings (chroma/demo/sno_pmt.txt and sno_cone.txt, which this repository does
not ship) can run the demo detectors -- C1, C2, C3 -- on them: point $CHROMA_PMT_PROFILE / $CHROMA_CONE_PROFILE at the two
files, or pass ``pmt_profile=`` / ``cone_profile=`` to the builders (chroma_amd.demo.tiny / detector / detector29k forward
them).  The files go through build_pmt / build_light_collector_from_file exactly as in chroma/demo/pmt.py:7-20.
"""
import os

import numpy as np

from chroma_amd.pmt import build_pmt, build_light_collector_from_file, build_pmt_from_profile, build_light_collector_from_profile
from chroma_amd.demo.optics import water, glass, vacuum, shiny_surface, r7081hqe_photocathode

NECK_RADIUS = 40.0        # mm
BULB_RADIUS = 101.0       # mm, 8-inch tube
BASE_Y = -184.0           # mm, bottom of the neck
NECK_TOP_Y = -100.0       # mm, where the neck meets the bulb
DOME_HEIGHT = 75.0        # mm, front face above the equator


def pmt_outline():
    """Half outline of the glass as (x, y) rows with x <= 0, base -> face, 55 points."""
    pts = [(-0.85 * NECK_RADIUS, BASE_Y)]                      # moved onto the axis by the builder
    for y in np.linspace(BASE_Y + 0.7, NECK_TOP_Y, 8):         # cylindrical neck
        pts.append((-NECK_RADIUS, y))
    # lower half of the bulb: ellipse through (NECK_RADIUS, NECK_TOP_Y) and (BULB_RADIUS, 0)
    b_low = -NECK_TOP_Y / np.sqrt(1.0 - (NECK_RADIUS / BULB_RADIUS) ** 2)
    phi0 = np.arcsin(NECK_TOP_Y / b_low)
    for phi in np.linspace(phi0, 0.0, 21)[1:]:
        pts.append((-BULB_RADIUS * np.cos(phi), b_low * np.sin(phi)))
    # front dome: ellipse with semi-axes (BULB_RADIUS, DOME_HEIGHT)
    for phi in np.linspace(0.0, np.pi / 2, 27)[1:-1]:
        pts.append((-BULB_RADIUS * np.cos(phi), DOME_HEIGHT * np.sin(phi)))
    pts.append((-0.03 * BULB_RADIUS, DOME_HEIGHT))             # moved onto the axis by the builder
    return np.array(pts)


def cone_outline():
    """Light-collector profile as (x, y) rows, rim -> throat, 16 points."""
    y = np.linspace(130.0, 22.0, 16)
    r = 98.0 + 36.0 * ((y - 22.0) / 108.0) ** 0.8
    return np.column_stack([-r, y])


def build_8inch_pmt(outer_material=water, nsteps=24, pmt_profile=None):
    """chroma/demo/pmt.py:7-13.  ``pmt_profile`` (or $CHROMA_PMT_PROFILE): a glass outline FILE in the reference's format
    (its sno_pmt.txt) instead of the analytic tube."""
    pmt_profile = pmt_profile or os.environ.get('CHROMA_PMT_PROFILE')
    kw = dict(outer_material=outer_material, glass=glass, vacuum=vacuum, photocathode_surface=r7081hqe_photocathode,
              back_surface=shiny_surface, nsteps=nsteps)
    if pmt_profile:
        return build_pmt(pmt_profile, 3.0, **kw)               # 3 mm of glass
    return build_pmt_from_profile(pmt_outline(), 3.0, **kw)


def build_8inch_pmt_with_lc(outer_material=water, nsteps=24, pmt_profile=None, cone_profile=None):
    """chroma/demo/pmt.py:15-20.  ``cone_profile`` (or $CHROMA_CONE_PROFILE): the light collector's profile FILE (the
    reference's sno_cone.txt) instead of the analytic cone."""
    pmt = build_8inch_pmt(outer_material, nsteps, pmt_profile=pmt_profile)
    cone_profile = cone_profile or os.environ.get('CHROMA_CONE_PROFILE')
    if cone_profile:
        lc = build_light_collector_from_file(cone_profile, outer_material=outer_material, surface=shiny_surface, nsteps=nsteps)
    else:
        lc = build_light_collector_from_profile(cone_outline(), outer_material=outer_material, surface=shiny_surface, nsteps=nsteps)
    return pmt + lc
