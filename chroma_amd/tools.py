"""Small host helpers used on the propagate path.

Reference: chroma/tools.py (read_csv :70-83, offset :85-128, argsort_direction
:175-193, from_film :195-228, filled_array :12-16, profile_if_possible :54-59).
"""
import numpy as np
from chroma_amd.transform import normalize

try:  # kernprof hook, identity otherwise (chroma/tools.py:54-59)
    profile_if_possible = profile  # noqa: F821
except NameError:
    def profile_if_possible(func):
        return func


def count_nonzero(array):
    return int(np.count_nonzero(array))


def filled_array(value, shape, dtype):
    return np.full(shape, value, dtype=dtype)


def read_csv(filename):
    """Rows of comma-separated floats; lines that do not parse are skipped."""
    rows = []
    with open(filename) as f:
        for line in f:
            try:
                rows.append([float(tok) for tok in line.split(',')])
            except ValueError:
                continue
    return np.array(rows)


def _unit_normal_2d(seg, dist):
    """Normal of a 2-D segment rotated 90 degrees clockwise, scaled to ``dist``."""
    v = np.array([seg[1], -seg[0]], dtype=float)
    return v / np.linalg.norm(v) * dist


def offset(points, x):
    """Offset a 2-D open profile by ``x`` (positive = to the right of the path).

    Each vertex moves to the intersection of the two neighbouring edges after both are
    shifted by ``x`` along their normals; the end points use mirrored ghost neighbours.
    Behaviour follows chroma/tools.py:85-128.
    """
    pts = np.asarray(points, dtype=float)
    ext = np.vstack([2 * pts[0] - pts[1], pts, 2 * pts[-1] - pts[-2]])
    out = np.empty_like(pts)
    for i in range(1, len(ext) - 1):
        n1 = _unit_normal_2d(ext[i] - ext[i - 1], x)
        n2 = _unit_normal_2d(ext[i + 1] - ext[i], x)
        a, b = ext[i - 1] + n1, ext[i] + n1
        c, d = ext[i] + n2, ext[i + 1] + n2
        m = np.column_stack([b - a, c - d])
        try:
            t = np.linalg.solve(m, c - a)[0]
            out[i - 1] = a + t * (b - a)
        except np.linalg.LinAlgError:  # collinear neighbours
            out[i - 1] = b
    return out


def interleave(arr, bits):
    """Morton-interleave the columns of an (n,3) integer array."""
    arr = np.asarray(arr)
    if arr.ndim != 2 or arr.shape[1] != 3:
        raise Exception('shape mismatch')
    a = arr.astype(np.uint64)
    z = np.zeros(len(a), dtype=np.uint64)
    for i in range(bits):
        bit = np.uint64(1) << np.uint64(i)
        z |= ((a[:, 2] & bit) << np.uint64(2 * i)) | ((a[:, 1] & bit) << np.uint64(2 * i + 1)) \
            | ((a[:, 0] & bit) << np.uint64(2 * i + 2))
    return z


def argsort_direction(dir):
    """Indices that sort direction vectors by a Morton code of (theta, phi), used to
    make neighbouring photons take neighbouring paths (chroma/tools.py:175-193)."""
    bits = 16
    maxint = 2 ** bits - 1
    dir = np.asarray(dir)
    theta = (np.arccos(np.clip(dir[:, 2], -1, 1)) / np.pi * maxint).astype(np.uint32)
    phi = ((np.arctan2(dir[:, 1], dir[:, 0]) / np.pi / 2.0 + 0.5) * maxint).astype(np.uint32)
    morton = np.zeros(len(dir), dtype=np.uint32)
    for i in range(bits):
        bit = np.uint32(1 << i)
        morton |= ((theta & bit) << np.uint32(i)) | ((phi & bit) << np.uint32(i + 1))
    return np.argsort(morton, kind='stable')


def from_film(position=(0, 0, 0), axis1=(0, 0, 1), axis2=(1, 0, 0), size=(800, 600),
              width=35.0, focal_length=18.0):
    """Ray bundle through a pinhole camera's film (chroma/tools.py:195-228)."""
    height = width * (size[1] / float(size[0]))
    axis1 = normalize(axis1)
    axis2 = normalize(axis2)
    dx0 = width / size[0]
    dx1 = height / size[1]
    yy, xx = np.meshgrid(np.arange(size[1]), np.arange(size[0]))
    n = size[0] * size[1]
    grid = -axis2[np.newaxis, :] * (xx.ravel()[:, np.newaxis] * dx0) \
        + axis1[np.newaxis, :] * (yy.ravel()[:, np.newaxis] * dx1)
    grid += axis2 * width / 2 - axis1 * height / 2
    grid -= np.cross(axis1, axis2) * focal_length
    return np.tile(position, (n, 1)), normalize(-grid)
