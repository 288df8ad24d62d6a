import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    return np.load(os.path.join(GOLDEN, 'ref_host_model.npz'))


@pytest.fixture(scope='session')
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


def make_box_geometry(size=100.0, material=None, surface=None):
    """A cube of ``material`` inside vacuum, optional surface on all faces."""
    from chroma_amd.geometry import Geometry, Solid, vacuum
    from chroma_amd.make import box
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.demo.optics import water
    material = water if material is None else material
    g = Geometry(material)
    g.add_solid(Solid(box(size, size, size), material, vacuum, surface=surface))
    return create_geometry_from_obj(g)


@pytest.fixture(scope='session')
def tiny_geometry():
    from chroma_amd import demo
    from chroma_amd.loader import create_geometry_from_obj
    return create_geometry_from_obj(demo.tiny())


@pytest.fixture(scope='session')
def tiny_packed(tiny_geometry):
    from chroma_amd.gpu.geometry import pack_geometry
    return pack_geometry(tiny_geometry)


def make_stress_geometry():
    """chroma_amd.demo.stress.scintillator_stress() flattened with its BVH."""
    from chroma_amd.demo.stress import scintillator_stress
    from chroma_amd.loader import create_geometry_from_obj
    return create_geometry_from_obj(scintillator_stress())


def bomb(n, seed, wavelength=400.0, pos=(0, 0, 0), wavelength_hi=None):
    """Host-side photon bomb with NumPy's generator (used where the oracle's Philox bomb
    is not what is under test)."""
    from chroma_amd.event import Photons
    rng = np.random.default_rng(seed)
    theta = rng.uniform(0, 2 * np.pi, n); u = rng.uniform(-1, 1, n); c = np.sqrt(1 - u * u)
    d = np.column_stack([c * np.cos(theta), c * np.sin(theta), u])
    theta = rng.uniform(0, 2 * np.pi, n); u = rng.uniform(-1, 1, n); c = np.sqrt(1 - u * u)
    a = np.column_stack([c * np.cos(theta), c * np.sin(theta), u])
    pol = np.cross(a, d); pol /= np.linalg.norm(pol, axis=1)[:, None]
    wl = np.full(n, wavelength) if wavelength_hi is None else rng.uniform(wavelength, wavelength_hi, n)
    return Photons(np.tile(np.asarray(pos, dtype=float), (n, 1)), d, pol, wl)
