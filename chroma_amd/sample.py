"""Host-side random direction generators (reference: chroma/sample.py:4-56)."""
import numpy as np
from chroma_amd.transform import rotate


def _sphere_points(theta, u, size, dtype):
    c = np.sqrt(1.0 - u * u)
    if size is None:
        return np.array([c * np.cos(theta), c * np.sin(theta), u])
    pts = np.empty((size, 3), dtype)
    pts[:, 0] = c * np.cos(theta)
    pts[:, 1] = c * np.sin(theta)
    pts[:, 2] = u
    return pts


def uniform_sphere(size=None, dtype=np.double, rng=None):
    """Isotropic unit vectors: theta = U(0, 2pi) first, then u = U(-1, 1)
    (same draw order as chroma/sample.py:16-18 so np.random.seed streams agree)."""
    r = np.random if rng is None else rng
    theta = r.uniform(0.0, 2 * np.pi, size)
    u = r.uniform(-1.0, 1.0, size)
    return _sphere_points(theta, u, size, dtype)


def flashlight(phi=np.pi / 4, direction=(0, 0, 1), size=None, dtype=np.double, rng=None):
    """Directions uniform in the cone of half-angle ``phi`` about ``direction``
    (chroma/sample.py:32-56)."""
    r = np.random if rng is None else rng
    theta = r.uniform(0.0, 2 * np.pi, size)
    u = r.uniform(np.cos(phi), 1, size)
    pts = _sphere_points(theta, u, size, dtype)
    if np.equal(direction, (0, 0, 1)).all():
        axis, angle = (0, 0, 1), 0.0
    else:
        axis = np.cross((0, 0, 1), direction)
        angle = -np.arccos(np.dot(direction, (0, 0, 1)) / np.linalg.norm(direction))
    return rotate(pts, angle, axis)
