"""Single functions of the path against the REFERENCE'S OWN headers compiled for gfx950
(oracle/_ref/libchroma_ref_headers.so = oracle/ref_headers_driver.hip + chroma/cuda/rotate.h,
interpolate.h, geometry.h, built by oracle/Makefile where the reference tree is present):

  interp_property  (geometry.h:64-75)     bit for bit: engine == oracle == reference
  interp_idx       (interpolate.h:5-29)   bit for bit (double-precision last line included)
  interp           (interpolate.h:32-57)  bit for bit (the DAQ's CDF sampling, random.h:26-31)
  float3 algebra   (linalg.h)              the reference's OWN unit tests test/linalg_test.py and test/rotate_test.py run here
                                           on their own kernels (test/linalg_test.cu, test/rotate_test.cu compiled for gfx950:
                                           oracle/ref_linalg_driver.hip) with their own assertion (NumPy, allclose), and the
                                           engine's algebra (csrc/device_common.h) held to those kernels bit for bit
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
PROBES = {'interp_property': 0, 'interp_idx': 1, 'interp': 2, 'rotate': 3, 'linalg': 4}
OUT_WIDTH = {'rotate': 5, 'linalg': 32}
REF_LINALG = os.path.join(ROOT, 'oracle', '_ref', 'libchroma_ref_linalg.so')


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
    out = GPUArray(n * OUT_WIDTH.get(fn, 1), np.float32, ctx)
    _lib.check(ctx._lib.chroma_probe(ctx.handle, PROBES[fn], n, d_x.ptr, None if d_tx is None else d_tx.ptr,
                                     None if d_tf is None else d_tf.ptr, ntab, float(start), float(step), out.ptr))
    o = out.get()
    return o.reshape(n, OUT_WIDTH[fn]) if fn in OUT_WIDTH else o


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


# ---- the reference's own unit tests of its float3 algebra and of rotate, on its own kernels ------------------------------
LINALG_OPS = ['float3add', 'float3addequal', 'float3sub', 'float3subequal', 'float3addfloat', 'float3addfloatequal', 'floataddfloat3',
              'float3subfloat', 'float3subfloatequal', 'floatsubfloat3', 'float3mulfloat', 'float3mulfloatequal', 'floatmulfloat3',
              'float3divfloat', 'float3divfloatequal', 'floatdivfloat3', 'dot', 'cross', 'norm', 'minusfloat3']      # test/linalg_test.py:18-37


def ref_linalg(op, a, b, c):
    ref = ctypes.CDLL(REF_LINALG)
    n = len(a)
    out = np.empty(n if op in ('dot', 'norm') else (n, 3), np.float32)
    rc = ref.ref_linalg_run(LINALG_OPS.index(op), n, a.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p),
                            ctypes.c_float(float(c)), out.ctypes.data_as(ctypes.c_void_p))
    assert rc == 0, op
    return out


needs_ref_linalg = pytest.mark.skipif(not os.path.exists(REF_LINALG), reason='oracle/_ref not built (needs the reference tree at build time)')


@needs_ref_linalg
def test_reference_linalg_test_restated_and_the_engine_held_to_its_kernels():
    """test/linalg_test.py:38-214: 256 random vectors a, b in [0, 1)^3 and a random scalar c through every kernel of
    test/linalg_test.cu -- here the reference's kernels themselves, compiled for gfx950 -- each result np.allclose to the NumPy
    expression the reference's test names.  Then the engine's own algebra (chroma_probe fn 4: the operators of
    csrc/device_common.h that every device function of the path is written in) against those kernels BIT FOR BIT, on the test's
    inputs and on 2^16 vectors of mixed sign and magnitude."""
    rng = np.random.default_rng(11)
    for n, draw in ((256, lambda size: rng.random(size)), (1 << 16, lambda size: rng.normal(size=size) * 10.0 ** rng.integers(-3, 4, size))):
        a = np.ascontiguousarray(draw((n, 3)), np.float32)
        b = np.ascontiguousarray(draw((n, 3)), np.float32)
        b[b == 0] = 1.0
        c = np.float32(rng.random() + 0.1)
        want = {'float3add': a + b, 'float3addequal': a + b, 'float3sub': a - b, 'float3subequal': a - b,
                'float3addfloat': a + c, 'float3addfloatequal': a + c, 'floataddfloat3': c + a,
                'float3subfloat': a - c, 'float3subfloatequal': a - c, 'floatsubfloat3': c - a,
                'float3mulfloat': a * c, 'float3mulfloatequal': a * c, 'floatmulfloat3': c * a,
                'float3divfloat': a / c, 'float3divfloatequal': a / c, 'floatdivfloat3': c / a,
                'dot': (a.astype(np.float64) * b).sum(1), 'cross': np.cross(a.astype(np.float64), b),
                'norm': np.linalg.norm(a.astype(np.float64), axis=1), 'minusfloat3': -a}
        got = {op: ref_linalg(op, a, b, c) for op in LINALG_OPS}
        if n == 256:                       # the reference's own assertion (np.allclose, its defaults) on the test's own inputs, on gfx950
            for op in LINALG_OPS:
                assert np.allclose(got[op], want[op]), op
        eng = engine_probe('linalg', np.column_stack([a, b, np.full(n, c, np.float32)]))
        pairs = (('minusfloat3', slice(0, 3)), ('float3add', slice(3, 6)), ('float3sub', slice(6, 9)), ('float3mulfloat', slice(9, 12)),
                 ('floatmulfloat3', slice(12, 15)), ('float3divfloat', slice(15, 18)), ('floatdivfloat3', slice(18, 21)),
                 ('cross', slice(21, 24)), ('dot', 24), ('norm', 25))
        for op, cols in pairs:
            assert np.array_equal(bits(eng[:, cols]), bits(got[op])), 'engine %s differs from the reference kernel' % op
        # normalize = a / norm(a) and the componentwise quotient (linalg.h:171, :17), which linalg_test.cu has no kernel for:
        # against the reference's division and norm kernels composed
        nrm = got['norm']
        assert np.array_equal(bits(eng[:, 26:29]), bits((a / nrm[:, None]).astype(np.float32)))
        assert np.array_equal(bits(eng[:, 29:32]), bits((a / b).astype(np.float32)))


@needs_ref_linalg
def test_reference_rotate_test_restated(oracle_mod):
    """test/rotate_test.py:23-46: random vectors in [0, 1)^3, random angles in [0, 2 pi), ONE random axis, through the kernel of
    test/rotate_test.cu (the reference's, compiled for gfx950) against chroma.transform.rotate on the host within 1e-5 -- the
    reference's own assertion -- and the engine's rotate on the same inputs within the same tolerance (bit for bit where the
    contract's cos / sin equal the device library's: test_rotate above)."""
    from chroma_amd.transform import rotate as host_rotate, normalize
    rng = np.random.default_rng(12)
    n = 1 << 18
    a = rng.random((n, 3)).astype(np.float32)
    t = (rng.random(n) * 2 * np.pi).astype(np.float32)
    w = normalize(rng.random(3))
    ref = ctypes.CDLL(REF_LINALG)
    out = np.empty((n, 3), np.float32)
    axis = (ctypes.c_float * 3)(*[float(x) for x in w])
    assert ref.ref_rotate_test_run(n, a.ctypes.data_as(ctypes.c_void_p), t.ctypes.data_as(ctypes.c_void_p), axis,
                                   out.ctypes.data_as(ctypes.c_void_p)) == 0
    want = host_rotate(a, t, w)
    assert np.allclose(want, out, atol=1e-5)
    w32 = np.array([axis[0], axis[1], axis[2]], np.float32)
    eng = engine_probe('rotate', np.column_stack([a, t, np.tile(w32, (n, 1))]))
    assert np.allclose(want, eng[:, :3], atol=1e-5)
    assert np.abs(eng[:, :3].astype(np.float64) - out).max() < 1e-5
