"""PMT solids from a 2-D glass profile (reference: chroma/pmt.py:6-81).

A PMT is two nested surfaces of revolution: the outer glass envelope and, one glass
thickness inside it, the vacuum envelope whose upper half (triangle centre y > 0)
carries the photocathode surface and whose lower half carries the mirror-like back
surface.  Unlike the reference (chroma/pmt.py:70) the returned Solid
keeps its per-triangle ``outer_material`` array intact, so it can be added to other
solids; the medium outside the tube is stored as ``pmt.outer_medium``.
"""
import numpy as np

from chroma_amd.geometry import Solid
from chroma_amd.make import rotate_extrude
from chroma_amd.tools import read_csv, offset


def _half_profile(profile):
    """Keep the x < 0 half of a full outline, mirror it, order base -> face and close
    both ends on the axis."""
    profile = np.asarray(profile, dtype=float)
    profile = profile[profile[:, 0] < 0].copy()
    profile[:, 0] = -profile[:, 0]
    profile = profile[np.argsort(profile[:, 1], kind='stable')]
    profile[0, 0] = 0.0
    profile[-1, 0] = 0.0
    return profile


def get_lc_profile(radii, a, b, d, rmin, rmax):
    c = -b * np.sqrt(1 - (rmin - d) ** 2 / a ** 2)
    return -c - b * np.sqrt(1 - (radii - d) ** 2 / a ** 2)


def build_light_collector(pmt, a, b, d, rmin, rmax, surface, npoints=10):
    """Elliptical light-collecting cone sitting on the PMT face."""
    if not isinstance(pmt, Solid):
        raise Exception('`pmt` must be an instance of %s' % Solid)
    lc_radii = np.linspace(rmin, rmax, npoints)
    lc_profile = get_lc_profile(lc_radii, a, b, d, rmin, rmax)
    face = pmt.profile[pmt.profile[:, 1] > -1e-3]
    lc_offset = np.interp(lc_radii[0], face[::-1, 0], face[::-1, 1])
    lc_mesh = rotate_extrude(lc_radii, lc_profile + lc_offset, pmt.nsteps)
    return Solid(lc_mesh, pmt.outer_medium, pmt.outer_medium, surface=surface)


def build_pmt_shell_from_profile(profile, outer_material, glass, nsteps=16):
    profile = _half_profile(profile)
    return Solid(rotate_extrude(profile[:, 0], profile[:, 1], nsteps), glass, outer_material, color=0xeeffffff)


def build_pmt_shell(filename, outer_material, glass, nsteps=16):
    return build_pmt_shell_from_profile(read_csv(filename), outer_material, glass, nsteps)


def build_pmt_from_profile(profile, glass_thickness, outer_material, glass, vacuum,
                           photocathode_surface, back_surface, nsteps=16):
    """``profile``: full outline as (x, y) rows (both x signs), as read from a profile file."""
    profile = _half_profile(profile)
    inner_profile = offset(profile, -glass_thickness)
    outer_mesh = rotate_extrude(profile[:, 0], profile[:, 1], nsteps)
    inner_mesh = rotate_extrude(inner_profile[:, 0], inner_profile[:, 1], nsteps)

    outer_envelope = Solid(outer_mesh, glass, outer_material)
    photocathode = np.mean(inner_mesh.assemble(), axis=1)[:, 1] > 0
    surfaces = np.empty(len(photocathode), dtype=object)
    surfaces[photocathode] = photocathode_surface
    surfaces[~photocathode] = back_surface
    inner_envelope = Solid(inner_mesh, vacuum, glass, surface=surfaces,
                           color=np.where(photocathode, 0xff00, 0xff0000))
    pmt = outer_envelope + inner_envelope
    pmt.profile = profile
    pmt.outer_medium = outer_material
    pmt.nsteps = nsteps
    return pmt


def build_pmt(filename, glass_thickness, outer_material, glass, vacuum,
              photocathode_surface, back_surface, nsteps=16):
    return build_pmt_from_profile(read_csv(filename), glass_thickness, outer_material, glass, vacuum,
                                  photocathode_surface, back_surface, nsteps)


def build_light_collector_from_profile(profile, outer_material, surface, nsteps=48):
    profile = np.asarray(profile, dtype=float)
    mesh = rotate_extrude(profile[:, 0], profile[:, 1], nsteps)
    return Solid(mesh, outer_material, outer_material, surface=surface)


def build_light_collector_from_file(filename, outer_material, surface, nsteps=48):
    return build_light_collector_from_profile(read_csv(filename), outer_material, surface, nsteps)
