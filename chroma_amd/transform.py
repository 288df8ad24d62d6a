"""Rigid transforms on point arrays.

Reference behaviour: chroma/transform.py:4-59 (Rodrigues rotation, counter-clockwise
about ``n`` when looking towards +infinity, i.e. the left-handed convention chroma
uses everywhere).  SciPy is not needed on the propagate path, so ``matrix_to_rotvec``
is written with NumPy only.
"""
import numpy as np


def norm(x):
    """Euclidean norm along the last axis."""
    x = np.asarray(x)
    return np.sqrt(np.sum(x * x, axis=-1))


def normalize(x):
    """Unit vectors along ``x`` (shape (3,) or (n,3))."""
    x = np.atleast_2d(np.asarray(x, dtype=float))
    return (x / norm(x)[:, np.newaxis]).squeeze()


def get_perp(x):
    """An arbitrary vector perpendicular to ``x`` (chroma/transform.py:4-8)."""
    a = np.zeros(3)
    a[np.argmin(np.abs(x))] = 1
    return np.cross(a, x)


def _skew(n):
    return np.array([[0.0, n[2], -n[1]], [-n[2], 0.0, n[0]], [n[1], -n[0], 0.0]])


def make_rotation_matrix(phi, n):
    """Matrix form of :func:`rotate` (chroma/transform.py:10-22)."""
    n = normalize(n)
    c, s = np.cos(phi), np.sin(phi)
    return c * np.identity(3) + (1.0 - c) * np.outer(n, n) + s * _skew(n)


def rotate(x, phi, n):
    """Rotate points ``x`` by ``phi`` about axis ``n`` (chroma/transform.py:34-43)."""
    n = normalize(n)
    x = np.atleast_2d(x)
    phi = np.atleast_1d(phi)
    c = np.cos(phi)[:, np.newaxis]
    s = np.sin(phi)[:, np.newaxis]
    return (x * c + n * np.dot(x, n)[:, np.newaxis] * (1.0 - c) + np.cross(x, n) * s).squeeze()


def rotate_matrix(x, phi, n):
    """Same as :func:`rotate` through the explicit matrix."""
    return np.inner(np.asarray(x), make_rotation_matrix(phi, n))


def matrix_to_rotvec(rot_matrix):
    """(axis, angle) of a rotation matrix in the convention of make_rotation_matrix."""
    r = np.asarray(rot_matrix, dtype=float)
    angle = np.arccos(np.clip((np.trace(r) - 1.0) / 2.0, -1.0, 1.0))
    if angle == 0:
        return np.array([0.0, 0.0, 1.0]), 0
    # for R = cI + (1-c)nn^T + s*skew(n): R[1,2]-R[2,1] = 2 s n_x etc.
    axis = np.array([r[1, 2] - r[2, 1], r[2, 0] - r[0, 2], r[0, 1] - r[1, 0]])
    nrm = np.linalg.norm(axis)
    if nrm < 1e-12:  # angle == pi: take the axis from the symmetric part
        w, v = np.linalg.eigh((r + r.T) / 2.0)
        axis = v[:, np.argmax(w)]
        return axis / np.linalg.norm(axis), angle
    # chroma's matrix is the transpose of the right-handed one
    return -axis / nrm, angle
