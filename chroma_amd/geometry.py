"""Triangle-mesh geometry model: Mesh, Solid, Material, Surface, DichroicProps, Geometry.

Public names and semantics follow chroma/geometry.py (Mesh :19-110, Solid :115-225,
Material :227-258, DichroicProps :260-264, Surface :266-300, Geometry :302-392) with
two deliberate differences, both recorded in SURVEY.md:

* material / surface numbering is deterministic (order of first appearance) instead of
  the iteration order of a Python ``set`` (chroma/geometry.py:112-113);
* everything is NumPy-2 safe (the reference's ``remove_duplicate_vertices`` relies on
  NumPy-1 ``np.unique`` return shapes, chroma/geometry.py:63-67).
"""
from hashlib import md5

import numpy as np

from chroma_amd.log import logger

# all material/surface properties are resampled on this grid before upload
# (chroma/geometry.py:17): 60..995 nm in 5 nm steps, 188 samples.
standard_wavelengths = np.arange(60, 1000, 5).astype(np.float32)

# meshes with at least this many vertices are de-duplicated by the native (multi-threaded) code
NATIVE_DEDUPE_THRESHOLD = 200000


class Mesh(object):
    """Triangle mesh: float32 vertices (n,3) and int32 vertex indices (m,3)."""

    def __init__(self, vertices, triangles, remove_duplicate_vertices=False, round=True,
                 remove_null_triangles=True):
        vertices = np.asarray(vertices, dtype=np.float32)
        triangles = np.asarray(triangles, dtype=np.int32)
        if vertices.ndim != 2 or vertices.shape[1] != 3:
            raise ValueError('shape mismatch')
        if triangles.ndim != 2 or triangles.shape[1] != 3:
            raise ValueError('shape mismatch')
        if (triangles < 0).any():
            raise ValueError('indices in `triangles` must be positive.')
        if (triangles >= len(vertices)).any():
            raise ValueError('indices in `triangles` must be less than the length of the vertex array.')
        self.vertices = vertices
        self.triangles = triangles
        if len(vertices) == 0:
            logger.warning("Generated mesh has no vertices.")
        if len(triangles) == 0:
            logger.warning("Generated mesh has no triangles.")
        if round:
            self.vertices = self.vertices.round(decimals=12)
        if remove_duplicate_vertices:
            self.remove_duplicate_vertices()
        if remove_null_triangles:
            self.remove_null_triangles()

    def get_triangle_centers(self):
        return np.mean(self.assemble(), axis=1)

    def get_bounds(self):
        return np.min(self.vertices, axis=0), np.max(self.vertices, axis=0)

    def remove_duplicate_vertices(self):
        """Merge identical vertices; the survivors end up in lexicographic (x, y, z) order,
        which is what the reference's structured-array ``np.unique`` produces."""
        if len(self.vertices) >= NATIVE_DEDUPE_THRESHOLD:
            # same result from the multi-threaded native sort (chroma_dedupe_vertices)
            from chroma_amd import _lib
            if not self.triangles.flags['OWNDATA'] or not self.triangles.flags['C_CONTIGUOUS']:
                self.triangles = np.array(self.triangles, order='C')      # remapped in place below
            self.vertices = _lib.dedupe_vertices(self.vertices, self.triangles)
            return
        unique, inverse = np.unique(self.vertices, axis=0, return_inverse=True)
        self.vertices = np.ascontiguousarray(unique, dtype=np.float32)
        self.triangles = np.asarray(inverse).reshape(-1)[self.triangles].astype(self.triangles.dtype)

    def remove_null_triangles(self):
        """Drop triangles that use a vertex twice; returns the mask of kept triangles."""
        if len(self.triangles) == 0:
            return
        t = self.triangles
        mask = (t[:, 0] != t[:, 1]) & (t[:, 1] != t[:, 2]) & (t[:, 0] != t[:, 2])
        self.triangles = t[mask]
        return mask

    def assemble(self, key=slice(None), group=True):
        """Vertex positions of the selected triangles: (N,3,3), or (3N,3) if not grouped."""
        idx = self.triangles[key]
        if not group:
            idx = idx.flatten()
        return self.vertices[idx]

    def __add__(self, other):
        return Mesh(np.concatenate((self.vertices, other.vertices)),
                    np.concatenate((self.triangles, other.triangles + len(self.vertices))))

    def md5(self):
        h = md5(np.ascontiguousarray(self.vertices))
        h.update(np.ascontiguousarray(self.triangles))
        return h.hexdigest()


def ordered_unique(objs):
    """Unique Python objects (by identity) in order of first appearance."""
    seen = {}
    for o in objs:
        seen.setdefault(id(o), o)
    return list(seen.values())


def _per_triangle(value, n, dtype=object):
    """Broadcast a scalar attribute to n triangles, or check an array of length n."""
    if np.iterable(value) and not isinstance(value, (str, bytes)):
        if len(value) != n:
            raise ValueError('shape mismatch')
        out = np.empty(n, dtype=dtype)
        out[:] = list(value) if dtype is object else value
        return out
    out = np.empty(n, dtype=dtype)
    out[:] = value
    return out


def _lookup_indices(objs, table):
    """int32 index of each object of ``objs`` in ``table`` (an id -> index dict)."""
    if len(objs) == 0:
        return np.empty(0, dtype=np.int32)
    # attributes are almost always long runs of the same object: resolve each run once
    ids = np.fromiter((id(o) for o in objs), dtype=np.int64, count=len(objs))
    uniq, inv = np.unique(ids, return_inverse=True)
    return np.array([table[u] for u in uniq], dtype=np.int32)[inv]


class Solid(object):
    """A mesh with per-triangle inner/outer material, surface and colour."""

    def __init__(self, mesh, inner_material=None, outer_material=None, surface=None,
                 color=0x33ffffff, material1=None, material2=None):
        if material1 is not None or material2 is not None:
            logger.warning('material1 and material2 are deprecated.  Use inner_material and outer_material instead.')
            inner_material, outer_material = material1, material2
        n = len(mesh.triangles)
        self.mesh = mesh
        self.inner_material = _per_triangle(inner_material, n)
        self.outer_material = _per_triangle(outer_material, n)
        self.surface = _per_triangle(surface, n)
        self.color = _per_triangle(color, n, dtype=np.uint32)
        self.unique_materials = ordered_unique(np.concatenate([self.inner_material, self.outer_material]))
        self.unique_surfaces = ordered_unique(self.surface)

    def __add__(self, other):
        return Solid(self.mesh + other.mesh,
                     np.concatenate((self.inner_material, other.inner_material)),
                     np.concatenate((self.outer_material, other.outer_material)),
                     np.concatenate((self.surface, other.surface)),
                     np.concatenate((self.color, other.color)))

    def weld(self, other, shared_triangle_surface=None, shared_triangle_color=None):
        """Merge ``other`` into this solid at triangles the two share (not a boolean union).

        Shared triangles keep this solid's surface/colour unless overridden, and get
        ``other``'s inner material as their outer material (chroma/geometry.py:166-208)."""
        def keys(mesh):
            tri = mesh.vertices[mesh.triangles]                      # (n,3,3)
            order = np.lexsort((tri[:, :, 2], tri[:, :, 1], tri[:, :, 0]), axis=1)
            tri = np.take_along_axis(tri, order[:, :, None], axis=1)
            return [t.tobytes() for t in tri]
        mine, theirs = keys(self.mesh), keys(other.mesh)
        where = {}
        for j, k in enumerate(theirs):
            where.setdefault(k, []).append(j)
        mask = np.array([k in where for k in mine], dtype=bool)
        if not mask.any():
            raise Exception('cannot weld solids with no shared triangles')
        duplicates = sorted(j for k in mine if k in where for j in where[k])
        keep = np.ones(len(theirs), dtype=bool)
        keep[duplicates] = False
        self.mesh = self.mesh + Mesh(other.mesh.vertices, other.mesh.triangles[keep])
        self.inner_material = np.concatenate((self.inner_material, other.inner_material[keep]))
        self.outer_material = np.concatenate((self.outer_material, other.outer_material[keep]))
        self.surface = np.concatenate((self.surface, other.surface[keep]))
        self.color = np.concatenate((self.color, other.color[keep]))
        idx = np.nonzero(mask)[0]
        self.outer_material[idx] = other.inner_material[0]
        if shared_triangle_surface is not None:
            self.surface[idx] = shared_triangle_surface
        if shared_triangle_color is not None:
            self.color[idx] = shared_triangle_color
        self.unique_materials = ordered_unique(np.concatenate([self.inner_material, self.outer_material]))
        self.unique_surfaces = ordered_unique(self.surface)

    def inner_material_indices(self, material_lookup):
        return _lookup_indices(self.inner_material, material_lookup)

    def outer_material_indices(self, material_lookup):
        return _lookup_indices(self.outer_material, material_lookup)

    def surface_indices(self, surface_lookup):
        return _lookup_indices(self.surface, surface_lookup)


def _property_table(value, wavelengths):
    if np.iterable(value):
        if len(value) != len(wavelengths):
            raise ValueError('shape mismatch')
        value = np.asarray(value)
    else:
        value = np.full(len(wavelengths), value)
    return np.column_stack([np.asarray(wavelengths), value]).astype(np.float32)


class Material(object):
    """Bulk optical properties; every property is an (n,2) float32 table of (nm, value)."""

    def __init__(self, name='none'):
        self.name = name
        self.refractive_index = None
        self.absorption_length = None
        self.scattering_length = None
        self.scintillation_spectrum = None
        self.scintillation_light_yield = None
        self.scintillation_rise_time = None
        self.scintillation_waveform = None
        self.scintillation_mod = None
        self.comp_reemission_prob = []
        self.comp_reemission_wvl_cdf = []
        self.comp_reemission_times = []
        self.comp_reemission_time_cdf = []
        self.comp_absorption_length = []
        self.density = 0.0       # g/cm^3
        self.composition = {}    # by mass

    def set(self, name, value, wavelengths=standard_wavelengths):
        self.__dict__[name] = _property_table(value, wavelengths)

    def __repr__(self):
        return '<Material %s>' % self.name


# Empty material
vacuum = Material('vacuum')
vacuum.set('refractive_index', 1.0)
vacuum.set('absorption_length', 1e6)
vacuum.set('scattering_length', 1e6)


class DichroicProps(object):
    """Angle-dependent reflect/transmit tables of a dichroic filter surface."""

    def __init__(self, angles, reflect, transmit):
        self.angles = np.asarray(angles)              # [angle]
        self.dichroic_reflect = np.asarray(reflect)   # [angle][(nm, prob)]
        self.dichroic_transmit = np.asarray(transmit)


class Surface(object):
    """Surface optical properties; model 0 default, 1 thin film, 2 WLS, 3 dichroic."""

    def __init__(self, name='none', model=0):
        self.name = name
        self.model = model
        for prop in ('detect', 'absorb', 'reemit', 'reflect_diffuse', 'reflect_specular',
                     'eta', 'k', 'reemission_cdf'):
            self.set(prop, 0)
        self.dichroic_props = None
        self.thickness = 0.0
        self.transmissive = 0

    def set(self, name, value, wavelengths=standard_wavelengths):
        if (np.asarray(value) < 0.0).any():
            raise Exception('all probabilities must be >= 0.0')
        self.__dict__[name] = _property_table(value, wavelengths)

    def __repr__(self):
        return '<Surface %s>' % self.name


class Geometry(object):
    """A list of placed solids; ``flatten()`` produces the single mesh the GPU uses."""

    def __init__(self, detector_material=None):
        self.detector_material = detector_material
        self.solids = []
        self.solid_rotations = []
        self.solid_displacements = []
        self.bvh = None

    def add_solid(self, solid, rotation=None, displacement=None):
        """Place ``solid`` (rotate, then displace); returns its solid id."""
        rotation = np.identity(3) if rotation is None else np.asarray(rotation, dtype=np.float32)
        if rotation.shape != (3, 3):
            raise ValueError('rotation matrix has the wrong shape.')
        displacement = np.zeros(3) if displacement is None else np.asarray(displacement, dtype=np.float32)
        if displacement.shape != (3,):
            raise ValueError('displacement vector has the wrong shape.')
        self.solid_rotations.append(rotation.astype(np.float32))
        self.solid_displacements.append(displacement)
        self.solids.append(solid)
        return len(self.solids) - 1

    def flatten(self):
        """Build ``mesh``, ``colors``, ``solid_id``, ``unique_materials``/``unique_surfaces``
        and the per-triangle material / surface index arrays (surface None -> -1)."""
        if hasattr(self, 'mesh'):
            return
        nv = np.cumsum([0] + [len(s.mesh.vertices) for s in self.solids])
        nt = np.cumsum([0] + [len(s.mesh.triangles) for s in self.solids])
        vertices = np.empty((nv[-1], 3), dtype=np.float32)
        triangles = np.empty((nt[-1], 3), dtype=np.int32)
        logger.info('Flattening detector mesh...')
        logger.info('  triangles: %d' % len(triangles))
        logger.info('  vertices:  %d' % len(vertices))
        # runs of consecutive placements of the same Solid (thousands of identical PMTs) are
        # transformed in one batched product: v' = v . R^T + d, as np.inner(v, R) + d per solid
        runs = []
        i, nsolids = 0, len(self.solids)
        while i < nsolids:
            solid = self.solids[i]
            j = i + 1
            while j < nsolids and self.solids[j] is solid and j - i < 512:
                j += 1
            runs.append((i, j))
            i = j

        def place(run):
            i, j = run
            mv, mt = self.solids[i].mesh.vertices, self.solids[i].mesh.triangles
            if j - i == 1:
                vertices[nv[i]:nv[i + 1]] = np.inner(mv, self.solid_rotations[i]) + self.solid_displacements[i]
                triangles[nt[i]:nt[i + 1]] = mt + nv[i]
            else:
                rot = np.stack(self.solid_rotations[i:j])                                  # (k,3,3)
                disp = np.stack([np.asarray(d, dtype=np.float32) for d in self.solid_displacements[i:j]])
                block = np.matmul(mv[None, :, :], np.transpose(rot, (0, 2, 1))) + disp[:, None, :]
                vertices[nv[i]:nv[j]] = block.reshape(-1, 3)
                triangles[nt[i]:nt[j]] = (mt[None, :, :] + nv[i:j, None, None]).reshape(-1, 3)
        if len(triangles) >= (1 << 22) and len(runs) > 1:
            # (the runs write disjoint slices and NumPy releases the interpreter lock inside its loops: a few threads
            #  place a 29k-PMT detector in a quarter of the time; the arrays do not depend on who wrote which slice)
            import os
            from concurrent.futures import ThreadPoolExecutor
            nthreads = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)))
            with ThreadPoolExecutor(nthreads) as pool:
                list(pool.map(place, runs))
        else:
            for run in runs:
                place(run)
        self.mesh = Mesh(vertices, triangles, remove_duplicate_vertices=True, remove_null_triangles=False)
        self.colors = np.concatenate([s.color for s in self.solids])
        self.solid_id = np.concatenate([np.full(len(s.mesh.triangles), i, dtype=np.uint32)
                                        for i, s in enumerate(self.solids)])

        self.unique_materials = ordered_unique(m for s in self.solids for m in s.unique_materials)
        material_lookup = {id(m): i for i, m in enumerate(self.unique_materials)}
        # the same Solid object is usually placed many times (PMTs): index it once
        def per_solid(method, lookup):
            cache = {}
            for s in self.solids:
                if id(s) not in cache:
                    cache[id(s)] = getattr(s, method)(lookup)
            return np.concatenate([cache[id(s)] for s in self.solids])
        self.inner_material_index = per_solid('inner_material_indices', material_lookup)
        self.outer_material_index = per_solid('outer_material_indices', material_lookup)

        self.unique_surfaces = ordered_unique(x for s in self.solids for x in s.unique_surfaces)
        surface_lookup = {id(x): i for i, x in enumerate(self.unique_surfaces)}
        self.surface_index = per_solid('surface_indices', surface_lookup)
        if id(None) in surface_lookup:
            self.surface_index[self.surface_index == surface_lookup[id(None)]] = -1
