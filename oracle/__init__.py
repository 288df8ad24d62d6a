"""Python access to the CPU oracle (oracle/chroma_oracle.c) -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  It is the checker, never the thing measured or shipped: nothing under
chroma_amd/ imports it.

``variant='contract'`` loads the build whose transcendental functions come from
include/chroma_math.h (bit-exact comparand of the HIP engine); ``variant='libm'`` the
build against the host libm (independent check that the physics does not hinge on the
contract's polynomials).
"""
import ctypes
import os
import subprocess
from ctypes import POINTER, c_float, c_int32, c_uint32, c_uint64, c_void_p

import numpy as np

from chroma_amd import _lib as _abi        # only for the ctypes Structure definitions of the C ABI
from chroma_amd import event

_HERE = os.path.dirname(os.path.abspath(__file__))
_libs = {}


def build(force=False):
    """Compile liboracle.so / liboracle_libm.so (and oracle/_ref when the reference is present)."""
    have = all(os.path.exists(os.path.join(_HERE, n)) for n in ('liboracle.so', 'liboracle_libm.so'))
    if have and not force:
        return
    subprocess.check_call(['make', '-C', _HERE, 'all'], stdout=subprocess.DEVNULL)


def load(variant='contract'):
    if variant in _libs:
        return _libs[variant]
    name = {'contract': 'liboracle.so', 'libm': 'liboracle_libm.so'}[variant]
    path = os.path.join(_HERE, name)
    if variant == 'contract' and os.environ.get('CHROMA_ORACLE_LIBRARY'):       # (a sanitizer build: tools/asan_host.sh)
        path = os.environ['CHROMA_ORACLE_LIBRARY']
    if not os.path.exists(path):
        build()
    lib = ctypes.CDLL(path)
    lib.oracle_variant.restype = ctypes.c_char_p
    assert lib.oracle_variant().decode() == variant
    lib.oracle_propagate.restype = c_int32
    lib.oracle_propagate.argtypes = [POINTER(_abi.GeometryDesc), POINTER(_abi.PhotonArrays), c_uint64, _abi.Rng,
                                     c_int32, c_int32, c_int32, c_int32, POINTER(_abi.PropagateStats)]
    lib.oracle_distance_to_mesh.restype = c_int32
    lib.oracle_distance_to_mesh.argtypes = [POINTER(_abi.GeometryDesc), c_uint64, c_void_p, c_void_p, c_void_p, c_void_p,
                                            POINTER(_abi.PropagateStats)]
    lib.oracle_generate_bomb.restype = c_int32
    lib.oracle_generate_bomb.argtypes = [POINTER(_abi.PhotonArrays), c_uint64, c_uint64, c_uint64, POINTER(c_float),
                                         c_float, c_float]
    lib.oracle_run_daq.restype = c_int32
    lib.oracle_run_daq.argtypes = [POINTER(_abi.GeometryDesc), POINTER(_abi.DaqTables), c_int32, c_int32, c_uint32,
                                   POINTER(_abi.PhotonArrays), _abi.Rng, c_uint32, c_float, c_void_p, c_void_p, c_void_p]
    lib.oracle_run_daq_many.restype = c_int32
    lib.oracle_run_daq_many.argtypes = [POINTER(_abi.GeometryDesc), POINTER(_abi.DaqTables), c_int32, c_int32, c_uint32,
                                        POINTER(_abi.PhotonArrays), _abi.Rng, c_uint32, c_float, c_int32, c_int32,
                                        c_void_p, c_void_p, c_void_p]
    lib.oracle_math.restype = c_int32
    lib.oracle_math.argtypes = [c_int32, c_uint64, c_void_p, c_void_p, c_void_p]
    lib.oracle_philox.restype = None
    lib.oracle_philox.argtypes = [c_void_p, c_void_p, c_void_p]
    lib.oracle_uniform_stream.restype = None
    lib.oracle_uniform_stream.argtypes = [c_uint64, c_uint64, c_uint32, c_uint32, c_void_p]
    lib.oracle_single.restype = c_int32
    lib.oracle_single.argtypes = [POINTER(_abi.GeometryDesc), POINTER(_abi.PhotonArrays), _abi.Rng, c_int32,
                                  POINTER(c_float), c_float, c_float, c_float, c_float, c_int32, c_int32, c_float,
                                  c_int32, c_int32]
    _libs[variant] = lib
    return lib


class HostPhotons(object):
    """Writable host copies of the ten photon arrays + the ctypes struct pointing at them."""
    FIELDS = (('pos', np.float32), ('dir', np.float32), ('pol', np.float32), ('wavelengths', np.float32),
              ('t', np.float32), ('flags', np.uint32), ('last_hit_triangles', np.int32), ('weights', np.float32),
              ('evidx', np.uint32))

    def __init__(self, photons=None, n=None, rng_counters=None):
        if photons is not None:
            n = len(photons)
            for name, dtype in self.FIELDS:
                setattr(self, name, np.array(getattr(photons, name), dtype=dtype, order='C', copy=True))
        else:
            for name, dtype in self.FIELDS:
                shape = (n, 3) if name in ('pos', 'dir', 'pol') else (n,)
                setattr(self, name, np.zeros(shape, dtype=dtype))
        self.rng_counters = np.zeros(n, dtype=np.uint32) if rng_counters is None \
            else np.array(rng_counters, dtype=np.uint32, copy=True)
        self.n = n
        self.struct = _abi.PhotonArrays()
        for name, _ in self.FIELDS:
            setattr(self.struct, name, getattr(self, name).ctypes.data)
        self.struct.rng_counters = self.rng_counters.ctypes.data

    def photons(self):
        return event.Photons(self.pos, self.dir, self.pol, self.wavelengths, self.t, self.last_hit_triangles,
                             self.flags, self.weights, self.evidx)


def propagate(packed, photons, seed, photon_id_base=0, max_steps=10, use_weights=False, scatter_first=0,
              rng_counters=None, nthreads=1, variant='contract'):
    """Run the oracle on ``photons`` (event.Photons).  Returns (Photons, rng_counters, stats dict)."""
    lib = load(variant)
    hp = HostPhotons(photons, rng_counters=rng_counters)
    stats = _abi.PropagateStats()
    rc = lib.oracle_propagate(ctypes.byref(packed.desc), ctypes.byref(hp.struct), hp.n, _abi.Rng(int(seed), int(photon_id_base)),
                              int(max_steps), int(bool(use_weights)), int(scatter_first), int(nthreads), ctypes.byref(stats))
    if rc != 0:
        raise RuntimeError('oracle_propagate failed (%d)' % rc)
    return hp.photons(), hp.rng_counters, stats.as_dict()


def distance_to_mesh(packed, origins, directions, variant='contract', last_hits=None):
    """(distance, triangle) for a ray bundle; misses give distance nan-filled here and triangle -1.
    ``last_hits``: per-ray triangle the ray must not hit (intersect_mesh's last_hit_triangle)."""
    lib = load(variant)
    o = np.ascontiguousarray(origins, dtype=np.float32)
    d = np.ascontiguousarray(directions, dtype=np.float32)
    n = len(o)
    dist = np.full(n, np.nan, dtype=np.float32)
    tri = np.empty(n, dtype=np.int32)
    stats = _abi.PropagateStats()
    lh = None if last_hits is None else np.ascontiguousarray(last_hits, dtype=np.int32)
    lib.oracle_intersect_mesh.restype = c_int32
    lib.oracle_intersect_mesh.argtypes = [POINTER(_abi.GeometryDesc), c_uint64, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p,
                                          POINTER(_abi.PropagateStats)]
    lib.oracle_intersect_mesh(ctypes.byref(packed.desc), n, o.ctypes.data, d.ctypes.data, None if lh is None else lh.ctypes.data,
                              dist.ctypes.data, tri.ctypes.data, ctypes.byref(stats))
    return dist, tri, stats.as_dict()


def reference_test_order(nodes, variant='contract'):
    """Triangles in the order the reference walk tests them when every box is entered (see
    oracle_reference_test_order).  ``nodes``: packed uint4 record array or [n][4] uint32."""
    lib = load(variant)
    raw = np.ascontiguousarray(nodes).view(np.uint32).reshape(-1, 4)
    nleaf = int(((raw[:, 3] >> 28) == 0).sum())
    out = np.empty(max(nleaf, 1), dtype=np.uint32)
    lib.oracle_reference_test_order.restype = ctypes.c_int64
    lib.oracle_reference_test_order.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_uint64]
    n = lib.oracle_reference_test_order(raw.ctypes.data, len(raw), out.ctypes.data, len(out))
    if n < 0:
        raise RuntimeError('reference stack overflow')
    return out[:min(n, len(out))]


def generate_bomb(n, seed, id_base=0, pos=(0, 0, 0), wavelength_lo=400.0, wavelength_hi=0.0, variant='contract'):
    lib = load(variant)
    hp = HostPhotons(n=n)
    p = (c_float * 3)(*[float(x) for x in pos])
    lib.oracle_generate_bomb(ctypes.byref(hp.struct), n, int(seed), int(id_base), p, float(wavelength_lo), float(wavelength_hi))
    return hp.photons()


def run_daq(packed, photons, tables_host, charge_unit, seed, photon_id_base=0, acquisition=0, weight=1.0,
            start_photon=0, nphotons=None, variant='contract'):
    """run_daq on HOST arrays.  ``tables_host`` = (time_cdf_x, time_cdf_y, charge_cdf_x, charge_cdf_y) float32
    arrays of equal pairwise length.  Returns (earliest_time float32, charge float32, histories uint32, hit mask)."""
    lib = load(variant)
    hp = HostPhotons(photons)
    tx, ty, qx, qy = [np.ascontiguousarray(a, dtype=np.float32) for a in tables_host]
    tab = _abi.DaqTables(tx.ctypes.data, ty.ctypes.data, len(tx), qx.ctypes.data, qy.ctypes.data, len(qx), float(charge_unit))
    nch = packed.desc.nchannels
    t_int = np.full(nch, np.float32(1e9).view(np.uint32), dtype=np.uint32)
    q_int = np.zeros(nch, dtype=np.uint32)
    hist = np.zeros(nch, dtype=np.uint32)
    if nphotons is None:
        nphotons = hp.n - start_photon
    lib.oracle_run_daq(ctypes.byref(packed.desc), ctypes.byref(tab), int(start_photon), int(nphotons), event.SURFACE_DETECT,
                       ctypes.byref(hp.struct), _abi.Rng(int(seed), int(photon_id_base)), int(acquisition), float(weight),
                       t_int.ctypes.data, q_int.ctypes.data, hist.ctypes.data)
    t = t_int.view(np.float32)
    return t, (q_int.astype(np.float32) * np.float32(charge_unit)).astype(np.float32), hist, t < 1e8


def run_daq_many(packed, photons, tables_host, charge_unit, seed, ndaq, photon_id_base=0, acquisition=0, weight=1.0,
                 start_photon=0, nphotons=None, variant='contract'):
    """run_daq_many on HOST arrays: ``ndaq`` acquisitions side by side (copy i = entries [i*nch, (i+1)*nch)).
    Returns (earliest_time float32, charge float32, histories uint32, hit mask), each of ndaq * nchannels entries."""
    lib = load(variant)
    hp = HostPhotons(photons)
    tx, ty, qx, qy = [np.ascontiguousarray(a, dtype=np.float32) for a in tables_host]
    tab = _abi.DaqTables(tx.ctypes.data, ty.ctypes.data, len(tx), qx.ctypes.data, qy.ctypes.data, len(qx), float(charge_unit))
    nch = packed.desc.nchannels
    t_int = np.full(nch * ndaq, np.float32(1e9).view(np.uint32), dtype=np.uint32)
    q_int = np.zeros(nch * ndaq, dtype=np.uint32)
    hist = np.zeros(nch * ndaq, dtype=np.uint32)
    if nphotons is None:
        nphotons = hp.n - start_photon
    lib.oracle_run_daq_many(ctypes.byref(packed.desc), ctypes.byref(tab), int(start_photon), int(nphotons), event.SURFACE_DETECT,
                            ctypes.byref(hp.struct), _abi.Rng(int(seed), int(photon_id_base)), int(acquisition), float(weight),
                            int(ndaq), int(nch), t_int.ctypes.data, q_int.ctypes.data, hist.ctypes.data)
    t = t_int.view(np.float32)
    return t, (q_int.astype(np.float32) * np.float32(charge_unit)).astype(np.float32), hist, t < 1e8


def math_fn(fn, x, y=None, variant='contract'):
    names = {'log': 0, 'exp': 1, 'sin': 2, 'cos': 3, 'tan': 4, 'asin': 5, 'acos': 6, 'atan2': 7, 'sqrt': 8, 'uniform': 9}
    lib = load(variant)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = x if y is None else np.ascontiguousarray(y, dtype=np.float32)
    out = np.empty_like(x)
    rc = lib.oracle_math(names[fn], x.size, x.ctypes.data, y.ctypes.data, out.ctypes.data)
    assert rc == 0
    return out


def render(packed, origins, directions, alpha_depth=10, bg_color=0, state=None, variant='contract'):
    """render.cu on host arrays.  ``state`` = (dx, dxlen, color) of a previous call (keep_last_render), else fresh.
    Returns (pixels uint32, (dx [n][alpha_depth], dxlen [n], color [n][alpha_depth][4]))."""
    lib = load(variant)
    lib.oracle_render.restype = c_int32
    lib.oracle_render.argtypes = [POINTER(_abi.GeometryDesc), c_uint64, c_void_p, c_void_p, c_uint32, c_uint32, c_void_p, c_void_p,
                                  c_void_p, c_void_p]
    o = np.ascontiguousarray(origins, dtype=np.float32)
    d = np.ascontiguousarray(directions, dtype=np.float32)
    n = len(o)
    if state is None:
        dx = np.zeros((n, alpha_depth), dtype=np.float32)
        dxlen = np.zeros(n, dtype=np.uint32)
        color = np.zeros((n, alpha_depth, 4), dtype=np.float32)
    else:
        dx, dxlen, color = [np.array(a, copy=True) for a in state]
    pixels = np.zeros(n, dtype=np.uint32)
    rc = lib.oracle_render(ctypes.byref(packed.desc), n, o.ctypes.data, d.ctypes.data, int(alpha_depth), int(bg_color) & 0xFFFFFFFF,
                           pixels.ctypes.data, dx.ctypes.data, dxlen.ctypes.data, color.ctypes.data)
    if rc != 0:
        raise RuntimeError('oracle_render failed (%d)' % rc)
    return pixels, (dx, dxlen, color)


PROBES = {'interp_property': 0, 'interp_idx': 1, 'interp': 2, 'rotate': 3}


def probe(fn, x, tab_x=None, tab_f=None, start=0.0, step=1.0, variant='contract'):
    """One call per element of a single function of the path (oracle_probe).  'rotate': x is [n][7]
    (a, phi, axis) and the result [n][5] (rotated vector, cos phi, sin phi)."""
    lib = load(variant)
    lib.oracle_probe.restype = c_int32
    lib.oracle_probe.argtypes = [c_int32, c_uint64, c_void_p, c_void_p, c_void_p, c_uint32, c_float, c_float, c_void_p]
    x = np.ascontiguousarray(x, dtype=np.float32)
    n = len(x)
    out = np.empty((n, 5) if fn == 'rotate' else n, dtype=np.float32)
    tx = None if tab_x is None else np.ascontiguousarray(tab_x, dtype=np.float32)
    tf = None if tab_f is None else np.ascontiguousarray(tab_f, dtype=np.float32)
    ntab = len(tx) if tx is not None else (len(tf) if tf is not None else 0)
    rc = lib.oracle_probe(PROBES[fn], n, x.ctypes.data, None if tx is None else tx.ctypes.data,
                          None if tf is None else tf.ctypes.data, ntab, float(start), float(step), out.ctypes.data)
    assert rc == 0
    return out


def philox(counter, key):
    lib = load()
    c = np.array(counter, dtype=np.uint32)
    k = np.array(key, dtype=np.uint32)
    out = np.zeros(4, dtype=np.uint32)
    lib.oracle_philox(c.ctypes.data, k.ctypes.data, out.ctypes.data)
    return out


def uniform_stream(seed, photon_id, n, start=0):
    lib = load()
    out = np.empty(n, dtype=np.float32)
    lib.oracle_uniform_stream(int(seed), int(photon_id), int(start), int(n), out.ctypes.data)
    return out
