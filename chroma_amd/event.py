"""Event containers: Photons, Event, Vertex, Channels, Steps and the history flags.

Same public names, constructor arguments and array dtypes as chroma/event.py
(flags :5-17, Photons :73-229, Channels :231-262, Event :264-311).  The optional
``particle`` PyPI package is not required: Vertex takes ``pdgcode`` directly and
only looks the name up when that package happens to be importable.
"""
import numpy as np

# Photon history bits (chroma/cuda/photon.h:49-64)
NO_HIT = 0x1 << 0
BULK_ABSORB = 0x1 << 1
SURFACE_DETECT = 0x1 << 2
SURFACE_ABSORB = 0x1 << 3
RAYLEIGH_SCATTER = 0x1 << 4
REFLECT_DIFFUSE = 0x1 << 5
REFLECT_SPECULAR = 0x1 << 6
SURFACE_REEMIT = 0x1 << 7
SURFACE_TRANSMIT = 0x1 << 8
BULK_REEMIT = 0x1 << 9
CHERENKOV = 0x1 << 10
SCINTILLATION = 0x1 << 11
NAN_ABORT = 0x1 << 31

TERMINAL_MASK = NO_HIT | BULK_ABSORB | SURFACE_DETECT | SURFACE_ABSORB | NAN_ABORT

_PHOTON_FIELDS = ('pos', 'dir', 'pol', 'wavelengths', 't', 'last_hit_triangles',
                  'flags', 'weights', 'evidx', 'channel')


class Steps(object):
    """Per-step record of a tracked particle (chroma/event.py:19-30)."""

    def __init__(self, x, y, z, t, dx, dy, dz, ke, edep, qedep):
        self.x, self.y, self.z, self.t = x, y, z, t
        self.dx, self.dy, self.dz = dx, dy, dz
        self.ke, self.edep, self.qedep = ke, edep, qedep


class Vertex(object):
    """A particle vertex (chroma/event.py:33-71)."""

    def __init__(self, particle_name, pos, dir, ke, t0=0.0, pol=None, steps=None,
                 children=None, trackid=-1, pdgcode=None):
        self.particle_name = particle_name
        self.pos = pos
        self.dir = dir
        self.pol = pol
        self.ke = ke
        self.t0 = t0
        self.steps = steps
        self.children = children
        self.trackid = trackid
        if pdgcode is None:
            try:
                from particle import Particle
                pdgcode = Particle.from_evtgen_name(particle_name).pdgid
            except ImportError:
                pdgcode = None
        self.pdgcode = pdgcode

    def __str__(self):
        return 'Vertex(%s,ke=%s,steps=%s)' % (self.particle_name, self.ke, bool(self.steps))

    __repr__ = __str__


def _default(arr, n, dtype, fill):
    if arr is None:
        return np.full(n, fill, dtype=dtype)
    return np.asarray(arr, dtype=dtype)


class Photons(object):
    """A list of n photons as parallel NumPy arrays.

    pos/dir/pol: float32 (n,3); wavelengths (nm), t (ns), weights: float32 (n,);
    last_hit_triangles: int32 (default -1); flags, evidx, channel: uint32.
    """

    def __init__(self, pos=np.empty((0, 3)), dir=np.empty((0, 3)), pol=np.empty((0, 3)),
                 wavelengths=np.empty((0)), t=None, last_hit_triangles=None, flags=None,
                 weights=None, evidx=None, channel=None):
        self.pos = np.asarray(pos, dtype=np.float32)
        self.dir = np.asarray(dir, dtype=np.float32)
        self.pol = np.asarray(pol, dtype=np.float32)
        self.wavelengths = np.asarray(wavelengths, dtype=np.float32)
        n = len(pos)
        self.t = _default(t, n, np.float32, 0)
        self.last_hit_triangles = _default(last_hit_triangles, n, np.int32, -1)
        self.flags = _default(flags, n, np.uint32, 0)
        self.weights = _default(weights, n, np.float32, 1)
        self.evidx = _default(evidx, n, np.uint32, 0)
        self.channel = _default(channel, n, np.uint32, 0)

    def _fields(self):
        return [getattr(self, name) for name in _PHOTON_FIELDS]

    @staticmethod
    def join(photon_list, concatenate=True):
        """Concatenate (or, with concatenate=False, stack scalar-indexed) Photons."""
        combine = np.concatenate if concatenate else np.asarray
        return Photons(*[combine([getattr(p, name) for p in photon_list]) for name in _PHOTON_FIELDS])

    def __add__(self, other):
        return Photons(*[np.concatenate((a, b)) for a, b in zip(self._fields(), other._fields())])

    def __len__(self):
        return len(self.pos)

    def __getitem__(self, key):
        if isinstance(key, (slice, np.ndarray)):
            # (the fields of a Photons object have their types already: a slice or a gather of it needs no conversion --
            #  Simulation hands out tens of thousands of per-channel slices per event)
            out = object.__new__(Photons)
            d = self.__dict__
            out.__dict__ = {name: d[name][key] for name in _PHOTON_FIELDS}
            return out
        return Photons(*[a[key] for a in self._fields()])

    def __str__(self):
        if len(self.pos) == 1:
            return ('Photon(pos=%s,dir=%s,pol=%s,wavelength=%s,t=%s,last_hit_triangle=%s,flag=%s,weight=%s)'
                    % (self.pos[0], self.dir[0], self.pol[0], self.wavelengths[0], self.t[0],
                       self.last_hit_triangles[0], self.flags[0], self.weights[0]))
        return 'Photons[%d]' % len(self.pos)

    __repr__ = __str__

    def reduced(self, reduction_factor=1.0):
        """A random subset of about len(self)*reduction_factor photons."""
        n = len(self)
        return self[np.random.permutation(n)[:int(n * reduction_factor)]]


class Channels(object):
    """Per-channel readout: hit mask, time, charge (chroma/event.py:231-262)."""

    def __init__(self, hit, t, q, flags=None, evidx=None):
        self.hit = hit
        self.t = t
        self.q = q
        self.flags = flags
        self.evidx = evidx

    def hit_channels(self, return_flags=False):
        ids = self.hit.nonzero()[0]
        if return_flags:
            return ids, self.t[self.hit], self.q[self.hit], self.flags[self.hit]
        return ids, self.t[self.hit], self.q[self.hit]


class Event(object):
    """One simulated event (chroma/event.py:264-311)."""

    def __init__(self, id=0, vertices=None, photons_beg=None, photons_end=None,
                 photon_tracks=None, photon_parent_trackids=None, hits=None,
                 flat_hits=None, channels=None):
        self.id = id
        self.nphotons = None
        if vertices is None:
            self.vertices = []
        elif np.iterable(vertices):
            self.vertices = vertices
        else:
            self.vertices = [vertices]
        self.photons_beg = photons_beg
        self.photons_end = photons_end
        self.photon_tracks = photon_tracks
        self.photon_parent_trackids = photon_parent_trackids
        self.hits = hits
        self.flat_hits = flat_hits
        self.channels = channels
