"""Single functions of the path against the REFERENCE'S OWN headers compiled for gfx950
(oracle/_ref/libchroma_ref_headers.so = oracle/ref_headers_driver.hip + chroma/cuda/rotate.h,
interpolate.h, geometry.h, built by oracle/Makefile where the reference tree is present):

  interp_property  (geometry.h:64-75)     bit for bit: engine == oracle == reference
  interp_idx       (interpolate.h:5-29)   bit for bit (double-precision last line included)
  interp           (interpolate.h:32-57)  bit for bit (the DAQ's CDF sampling, random.h:26-31)
  rotate           (rotate.h:22-28)       engine == oracle bit for bit; against the reference bit for bit
                                           wherever the device library's cosf/sinf (which the reference calls)
                                           equal the contract's, and within 1e-5 of |a| everywhere (the north
                                           star's float tolerance) -- the algebra is pinned, the cosine is
                                           the numeric contract's (include/chroma_math.h)
"""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
REF_LIB = os.path.join(ROOT, 'oracle', '_ref', 'libchroma_ref_headers.so')
needs_ref = pytest.mark.skipif(not os.path.exists(REF_LIB), reason='oracle/_ref not built (needs the reference tree at build time)')
PROBES = {'interp_property': 0, 'interp_idx': 1, 'interp': 2, 'rotate': 3}


def ref_probe(fn, x, tab_x=None, tab_f=None, start=0.0, step=1.0):
    ref = ctypes.CDLL(REF_LIB)
    x = np.ascontiguousarray(x, np.float32)
    n = len(x)
    out = np.empty((n, 5) if fn == 'rotate' else n, np.float32)
    p = lambda a: None if a is None else np.ascontiguousarray(a, np.float32).ctypes.data_as(ctypes.c_void_p)
    tx = None if tab_x is None else np.ascontiguousarray(tab_x, np.float32)
    tf = None if tab_f is None else np.ascontiguousarray(tab_f, np.float32)
    ntab = len(tx) if tx is not None else (len(tf) if tf is not None else 0)
    rc = ref.ref_headers_run(PROBES[fn], n, x.ctypes.data_as(ctypes.c_void_p), p(tx), p(tf), ntab,
                             ctypes.c_float(start), ctypes.c_float(step), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0
    return out


def engine_probe(fn, x, tab_x=None, tab_f=None, start=0.0, step=1.0):
    from chroma_amd import gpu, _lib
    from chroma_amd.gpu.tools import to_gpu, GPUArray
    ctx = gpu.get_context()
    x = np.ascontiguousarray(x, np.float32)
    n = len(x)
    d_x = to_gpu(x.reshape(-1), ctx)
    d_tx = None if tab_x is None else to_gpu(np.ascontiguousarray(tab_x, np.float32), ctx)
    d_tf = None if tab_f is None else to_gpu(np.ascontiguousarray(tab_f, np.float32), ctx)
    ntab = len(tab_x) if tab_x is not None else (len(tab_f) if tab_f is not None else 0)
    out = GPUArray(n * (5 if fn == 'rotate' else 1), np.float32, ctx)
    _lib.check(ctx._lib.chroma_probe(ctx.handle, PROBES[fn], n, d_x.ptr, None if d_tx is None else d_tx.ptr,
                                     None if d_tf is None else d_tf.ptr, ntab, float(start), float(step), out.ptr))
    o = out.get()
    return o.reshape(n, 5) if fn == 'rotate' else o


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@needs_ref
def test_interp_property(oracle_mod):
    rng = np.random.default_rng(1)
    n, start, step = 188, 60.0, 5.0                     # the standard wavelength grid (chroma/geometry.py:17)
    fp = rng.uniform(0.0, 100.0, n).astype(np.float32)
    top = start + (n - 1) * step
    x = np.concatenate([rng.uniform(start - 50, top + 50, 100000), start + step * np.arange(n),       # every grid point
                        [start, top, np.nextafter(np.float32(start), np.float32(0)), np.nextafter(np.float32(top), np.float32(2000)),
                         np.nextafter(np.float32(top), np.float32(0)), -1e30, 1e30]]).astype(np.float32)
    want = ref_probe('interp_property', x, tab_f=fp, start=start, step=step)
    assert np.array_equal(bits(oracle_mod.probe('interp_property', x, tab_f=fp, start=start, step=step)), bits(want))
    assert np.array_equal(bits(engine_probe('interp_property', x, tab_f=fp, start=start, step=step)), bits(want))
    # an irregular grid as well (steps that are not exact in binary)
    start, step = 200.0, 0.7
    x = rng.uniform(start - 5, start + step * n + 5, 100000).astype(np.float32)
    want = ref_probe('interp_property', x, tab_f=fp, start=start, step=step)
    assert np.array_equal(bits(oracle_mod.probe('interp_property', x, tab_f=fp, start=start, step=step)), bits(want))
    assert np.array_equal(bits(engine_probe('interp_property', x, tab_f=fp, start=start, step=step)), bits(want))


@needs_ref
@pytest.mark.parametrize('ntab', [2, 3, 7, 188, 1000])
def test_interp_idx_and_interp(oracle_mod, ntab):
    rng = np.random.default_rng(ntab)
    xp = np.sort(rng.uniform(0.0, 1.5, ntab)).astype(np.float32)
    xp[0] = 0.0
    fp = np.sort(rng.uniform(-6.0, 6.0, ntab)).astype(np.float32)
    x = np.concatenate([rng.uniform(-0.2, 1.7, 100000), xp, [xp[0], xp[-1], -1.0, 5.0]]).astype(np.float32)
    want = ref_probe('interp_idx', x, tab_x=xp)
    assert np.array_equal(bits(oracle_mod.probe('interp_idx', x, tab_x=xp)), bits(want))
    assert np.array_equal(bits(engine_probe('interp_idx', x, tab_x=xp)), bits(want))
    want = ref_probe('interp', x, tab_x=xp, tab_f=fp)
    assert np.array_equal(bits(oracle_mod.probe('interp', x, tab_x=xp, tab_f=fp)), bits(want))
    assert np.array_equal(bits(engine_probe('interp', x, tab_x=xp, tab_f=fp)), bits(want))


@needs_ref
def test_rotate(oracle_mod):
    rng = np.random.default_rng(5)
    n = 100000
    a = rng.normal(size=(n, 3))
    a /= np.linalg.norm(a, axis=1)[:, None]
    axis = rng.normal(size=(n, 3))
    axis /= np.linalg.norm(axis, axis=1)[:, None]
    phi = rng.uniform(-np.pi, np.pi, n)
    phi[:8] = [0.0, np.pi / 2, np.pi, -np.pi / 2, np.pi / 4, 1e-4, -1e-4, 3.0]
    x = np.column_stack([a, phi, axis]).astype(np.float32)
    want = ref_probe('rotate', x)
    orc = oracle_mod.probe('rotate', x)
    eng = engine_probe('rotate', x)
    assert np.array_equal(bits(eng), bits(orc)), 'engine and oracle: the same contract arithmetic'
    same_trig = (bits(orc[:, 3]) == bits(want[:, 3])) & (bits(orc[:, 4]) == bits(want[:, 4]))
    assert same_trig.mean() > 0.3, 'contract and device-library cos/sin agree on %.3f of the angles' % same_trig.mean()
    assert np.array_equal(bits(orc[same_trig, :3]), bits(want[same_trig, :3])), 'rotate algebra differs from rotate.h'
    # everywhere: within the float tolerance of the north star (|a| = 1)
    assert np.abs(orc[:, :3].astype(np.float64) - want[:, :3]).max() < 1e-5
    assert np.abs(orc[:, 3:].astype(np.float64) - want[:, 3:]).max() < 1e-6         # contract cos/sin vs the device library


@needs_ref
def test_point_transform_kernels():
    """chroma_points_translate / _rotate / _rotate_around_point against the reference's own kernels
    (chroma/cuda/transform.cu:9-49, compiled for gfx950 from where they lie): translate bit for bit; the two
    rotations bit for bit wherever the contract's cos/sin of the angle equal the device library's (the test above
    counts how often that is), and within the north star's float tolerance for the others."""
    from chroma_amd import gpu, _lib
    from chroma_amd.gpu.tools import to_gpu
    ctx = gpu.get_context()
    ref = ctypes.CDLL(REF_LIB)
    rng = np.random.default_rng(11)
    n = 50000
    pts = rng.uniform(-2000, 2000, (n, 3)).astype(np.float32)
    f3 = lambda v: (ctypes.c_float * 3)(*[float(c) for c in v])

    def ref_run(fn, params):
        out = np.empty_like(pts)
        t = np.ascontiguousarray(params, np.float32)
        rc = ref.ref_headers_run(fn, n, pts.ctypes.data_as(ctypes.c_void_p), t.ctypes.data_as(ctypes.c_void_p), None, len(t),
                                 ctypes.c_float(0.0), ctypes.c_float(1.0), out.ctypes.data_as(ctypes.c_void_p))
        assert rc == 0
        return out

    v = np.array([12.5, -3.25, 1e-3], np.float32)
    d = to_gpu(pts.reshape(-1).copy(), ctx)
    _lib.check(ctx._lib.chroma_points_translate(ctx.handle, n, d.ptr, f3(v)))
    assert np.array_equal(bits(d.get().reshape(n, 3)), bits(ref_run(4, v)))

    axis = np.array([0.3, -0.5, 0.81], np.float32); axis /= np.linalg.norm(axis)
    point = np.array([100.0, 50.0, -25.0], np.float32)
    exact = 0
    for phi in (0.0, 0.5, np.pi / 2, -2.0, 3.0, 1e-4):
        d = to_gpu(pts.reshape(-1).copy(), ctx)
        _lib.check(ctx._lib.chroma_points_rotate(ctx.handle, n, d.ptr, float(phi), f3(axis)))
        got, want = d.get().reshape(n, 3), ref_run(5, np.concatenate([[phi], axis]))
        assert np.abs(got.astype(np.float64) - want).max() < 1e-5 * 4000
        d = to_gpu(pts.reshape(-1).copy(), ctx)
        _lib.check(ctx._lib.chroma_points_rotate_around_point(ctx.handle, n, d.ptr, float(phi), f3(axis), f3(point)))
        got2, want2 = d.get().reshape(n, 3), ref_run(6, np.concatenate([[phi], axis, point]))
        assert np.abs(got2.astype(np.float64) - want2).max() < 1e-5 * 4000
        exact += int(np.array_equal(bits(got), bits(want)) and np.array_equal(bits(got2), bits(want2)))
    assert exact >= 2, 'bit-exact for %d of 6 angles' % exact
