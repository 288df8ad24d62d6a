"""Mesh primitives built by sweeping a profile (reference: chroma/make.py:6-163).

``mesh_grid`` turns a (rows, cols) grid of vertex indices into two triangles per
cell, wrapping around in the column direction; the extrusions feed it a grid whose
columns are the copies of the profile.  Winding: profiles traced counter-clockwise
give outward-facing normals.
"""
import numpy as np

from chroma_amd.geometry import Mesh
from chroma_amd.transform import rotate


def mesh_grid(grid):
    """Triangles (a, b, b') and (a, b', a') for every cell of an index grid, where a/b are
    vertically adjacent entries and the primed ones their right-hand (cyclic) neighbours."""
    top, bottom = grid[:-1], grid[1:]
    a, b = top.flatten(), bottom.flatten()
    a_next = np.roll(top, -1, 1).flatten()
    b_next = np.roll(bottom, -1, 1).flatten()
    n = len(a)
    tri = np.empty((2 * n, 3), dtype=a.dtype)
    tri[:n] = np.column_stack([a, b, b_next])
    tri[n:] = np.column_stack([a, b_next, a_next])
    return tri


def linear_extrude(x1, y1, height, x2=None, y2=None, center=None, endcaps=True):
    """Prism of the polygon (x1, y1) extruded along z by ``height`` (optionally tapering to
    (x2, y2)); with ``endcaps`` the ends are closed with fans about the axis."""
    x1, y1 = np.asarray(x1, dtype=float), np.asarray(y1, dtype=float)
    if len(x1) != len(y1):
        raise Exception('`x` and `y` arrays must have the same length.')
    x2 = x1 if x2 is None else np.asarray(x2, dtype=float)
    y2 = y1 if y2 is None else np.asarray(y2, dtype=float)
    if len(x2) != len(y2) or len(x2) != len(x1):
        raise Exception('`x` and `y` arrays must have the same length.')
    n = len(x1)
    lo = np.full(n, -height / 2.0)
    hi = np.full(n, height / 2.0)
    rings = [np.column_stack([x1, y1, lo]), np.column_stack([x2, y2, hi])]
    if endcaps:
        zero = np.zeros(n)
        rings = [np.column_stack([zero, zero, lo])] + rings + [np.column_stack([zero, zero, hi])]
    nring = len(rings)
    # vertex k of polygon point i sits at index i*nring + k
    vertices = np.stack(rings, axis=1).reshape(n * nring, 3)
    if center is not None:
        vertices = vertices + np.asarray(center, dtype=float)
    grid = np.arange(n * nring).reshape(n, nring).transpose()[::-1]
    return Mesh(vertices, mesh_grid(grid), remove_duplicate_vertices=True)


def rotate_extrude(x, y, nsteps=64):
    """Solid of revolution of the profile (x, y) about the y axis in ``nsteps`` steps."""
    if len(x) != len(y):
        raise Exception('`x` and `y` arrays must have the same length.')
    points = np.array([x, y, np.zeros(len(x))]).transpose()
    steps = np.linspace(0, 2 * np.pi, nsteps, endpoint=False)
    vertices = np.vstack([rotate(points, angle, (0, -1, 0)) for angle in steps])
    grid = np.arange(len(vertices)).reshape((len(steps), len(points))).transpose()[::-1]
    return Mesh(vertices, mesh_grid(grid), remove_duplicate_vertices=True)


def box(dx, dy, dz, center=(0, 0, 0)):
    hx, hy = dx / 2.0, dy / 2.0
    return linear_extrude([-hx, hx, hx, -hx], [-hy, -hy, hy, hy], height=dz, center=center)


def cube(size, height=None, center=(0, 0, 0)):
    # as in the reference, ``height`` is accepted but the cube is always size^3
    h = size / 2.0
    return linear_extrude([-h, h, h, -h], [-h, -h, h, h], height=size, center=center)


def cylinder_along_z(radius, height, points=100):
    angles = np.linspace(0, 2 * np.pi, points, endpoint=False)
    return linear_extrude(radius * np.cos(angles), radius * np.sin(angles), height)


def cylinder(radius, height, radius2=None, nsteps=64):
    """Cylinder (or truncated cone when ``radius2`` is given) about the y axis."""
    if radius2 is None:
        radius2 = radius
    return rotate_extrude([0, radius, radius2, 0],
                          [-height / 2.0, -height / 2.0, height / 2.0, height / 2.0], nsteps)


def segmented_cylinder(radius, height, nsteps=64, nsegments=100):
    nr = int((nsegments * radius / (2 * radius + height)) / 2)
    nh = int((nsegments * height / (2 * radius + height)) / 2)
    x = np.concatenate([np.linspace(0, radius, nr, endpoint=False), [radius] * nh,
                        np.linspace(radius, 0, nr, endpoint=False), [0]])
    y = np.concatenate([[-height / 2.0] * nr, np.linspace(-height / 2.0, height / 2.0, nh, endpoint=False),
                        [height / 2.0] * (nr + 1)])
    return rotate_extrude(x, y, nsteps)


def sphere(radius, nsteps=64):
    angles = np.linspace(-np.pi / 2, np.pi / 2, nsteps)
    return rotate_extrude(radius * np.cos(angles), radius * np.sin(angles), nsteps)


def torus(radius, offset, nsteps=64, circle_steps=None):
    if circle_steps is None:
        circle_steps = nsteps
    angles = np.linspace(0, 2 * np.pi, circle_steps)
    return rotate_extrude(radius * np.cos(angles) + offset, radius * np.sin(angles), nsteps)


def convex_polygon(x, y):
    """Fan triangulation of a convex polygon in the x-y plane."""
    vertices = np.column_stack((x, y, np.zeros_like(x)))
    k = np.arange(1, len(vertices) - 1)
    triangles = np.column_stack([np.zeros_like(k), k, k + 1]).astype(np.int32)
    return Mesh(vertices=vertices, triangles=triangles)
