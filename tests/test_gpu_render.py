"""GPURays.render (chroma/gpu/render.py:7-66, chroma/cuda/render.cu:37-181): an 800 x 600 `from_film` bundle
through demo.tiny() -- pixels, the per-ray alpha-depth lists and their lengths bit for bit against the
oracle's restatement AND against the reference's own render kernel compiled for gfx950 (oracle/_ref), after
the transforms of transform.cu, with a second render that continues the first (keep_last_render) and with a
background colour."""
import ctypes
import os

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
REF_LIB = os.path.join(ROOT, 'oracle', '_ref', 'libchroma_ref_mesh.so')


def _ref_render(geometry, o, d, alpha_depth, bg_color, state=None):
    ref = ctypes.CDLL(REF_LIB)
    mesh, bvh = geometry.mesh, geometry.bvh
    v = np.ascontiguousarray(mesh.vertices, np.float32)
    t = np.ascontiguousarray(mesh.triangles, np.uint32)
    colors = np.ascontiguousarray(geometry.colors, np.uint32)
    nodes = np.ascontiguousarray(bvh.nodes.view(np.uint32).reshape(-1, 4))
    origin = (ctypes.c_float * 3)(*[float(x) for x in bvh.world_coords.world_origin])
    n = len(o)
    if state is None:
        dx, dxlen, color = np.zeros((n, alpha_depth), np.float32), np.zeros(n, np.uint32), np.zeros((n, alpha_depth, 4), np.float32)
    else:
        dx, dxlen, color = [np.array(a, copy=True) for a in state]
    pixels = np.zeros(n, np.uint32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    rc = ref.ref_render_run(p(v), len(v), p(t), len(t), p(colors), p(nodes), len(nodes), origin, ctypes.c_float(float(bvh.world_coords.world_scale)),
                            n, p(np.ascontiguousarray(o, np.float32)), p(np.ascontiguousarray(d, np.float32)), alpha_depth,
                            ctypes.c_uint32(bg_color & 0xFFFFFFFF), p(pixels), p(dx), p(dxlen), p(color))
    assert rc == 0
    return pixels, (dx, dxlen, color)


def _lists_equal(a, b, what):
    (adx, alen, acol), (bdx, blen, bcol) = a, b
    assert np.array_equal(alen, blen), what + ': list lengths'
    k = np.arange(adx.shape[1])[None, :] < alen[:, None]               # entries beyond a ray's length are unspecified
    assert np.array_equal(adx.view(np.uint32)[k], bdx.view(np.uint32)[k]), what + ': distances'
    assert np.array_equal(acol.view(np.uint32)[k], bcol.view(np.uint32)[k]), what + ': colours'


def test_render_matches_oracle_and_compiled_reference(oracle_mod, tiny_geometry, tiny_packed):
    from chroma_amd import gpu
    from chroma_amd.gpu.tools import GPUArray
    from chroma_amd.tools import from_film
    ctx = gpu.create_cuda_context(0)
    try:
        gg = gpu.GPUDetector(tiny_geometry)
        size = (800, 600)
        pos, dirs = from_film(position=(0.0, -6000.0, 0.0), axis1=(0, 0, 1), axis2=(1, 0, 0), size=size, width=35.0, focal_length=18.0)
        rays = gpu.GPURays(pos, dirs, max_alpha_depth=10)
        # camera moves of chroma/camera.py style: transform.cu on the device
        rays.rotate_around_point(0.3, (0.0, 0.0, 1.0), (0.0, 0.0, 0.0))
        rays.translate((0.0, 0.0, 150.0))
        rays.rotate(0.05, (1.0, 0.0, 0.0))
        o = rays.pos.get().view(np.float32).reshape(-1, 3)
        d = rays.dir.get().view(np.float32).reshape(-1, 3)
        assert abs(np.linalg.norm(d, axis=1) - 1).max() < 1e-5 and not np.array_equal(o, pos.astype(np.float32))
        have_ref = os.path.exists(REF_LIB)
        for alpha_depth, bg in ((10, 0x00000000), (3, 0x7F102030)):
            pixels = GPUArray(len(o), np.uint32, ctx)
            rays.render(gg, pixels, alpha_depth=alpha_depth, bg_color=bg)
            got = pixels.get()
            want, wstate = oracle_mod.render(tiny_packed, o, d, alpha_depth=alpha_depth, bg_color=bg)
            assert np.array_equal(got, want), 'pixels differ from the oracle for %d rays (alpha_depth %d)' % ((got != want).sum(), alpha_depth)
            glen = rays.dxlen.get()
            gdx = rays.dx.get()[:len(o) * alpha_depth].reshape(len(o), alpha_depth)
            gcol = rays.color.get().view(np.float32)[:len(o) * alpha_depth * 4].reshape(len(o), alpha_depth, 4)
            _lists_equal((gdx, glen, gcol), wstate, 'engine vs oracle')
            assert (glen > 0).mean() > 0.15 and glen.max() == alpha_depth           # the sphere fills the frame; deep stacks of PMT glass
            if have_ref:
                rpix, rstate = _ref_render(tiny_geometry, o, d, alpha_depth, bg)
                assert np.array_equal(rpix, want), 'oracle vs the compiled reference: %d pixels' % (rpix != want).sum()
                _lists_equal(wstate, rstate, 'oracle vs compiled reference')
        # keep_last_render: a second pass from a shifted camera position merges into the lists of the first
        pixels = GPUArray(len(o), np.uint32, ctx)
        rays.render(gg, pixels, alpha_depth=10)
        first_state = (rays.dx.get().reshape(len(o), 10), rays.dxlen.get(), rays.color.get().view(np.float32).reshape(len(o), 10, 4))
        rays.translate((40.0, 0.0, 0.0))
        o2 = rays.pos.get().view(np.float32).reshape(-1, 3)
        rays.render(gg, pixels, alpha_depth=10, keep_last_render=True)
        want2, wstate2 = oracle_mod.render(tiny_packed, o2, d, alpha_depth=10, state=first_state)
        assert np.array_equal(pixels.get(), want2)
        if have_ref:
            rpix2, _ = _ref_render(tiny_geometry, o2, d, 10, 0, state=first_state)
            assert np.array_equal(rpix2, want2)
        # snapshot() and argument checks as the reference's
        snap = rays.snapshot(gg, alpha_depth=5)
        assert snap.shape == (len(o),) and snap.dtype == np.uint32 and (snap != 0).mean() > 0.15
        with pytest.raises(Exception, match='max_alpha_depth'):
            rays.render(gg, pixels, alpha_depth=11)
        with pytest.raises(ValueError):
            rays.render(gg, GPUArray(5, np.uint32, ctx))
        with pytest.raises(TypeError):
            rays.render(gg, np.zeros(len(o), np.uint32))
    finally:
        ctx.pop()


def test_color_solids_recolours_the_marked_solids_and_render_sees_it(tiny_geometry):
    """GPUGeometry.color_solids (chroma/gpu/geometry.py:283-298, kernel color_solids of chroma/cuda/mesh.h:153-166): the
    triangles of the marked solids take their solid's colour, every other triangle keeps its own -- against NumPy -- through
    the method and through the kernel table (get_cu_module('mesh.h').color_solids, a slice of the triangles); reset_colors
    undoes it; the render kernel reads the recoloured array."""
    from chroma_amd import gpu
    from chroma_amd.gpu.tools import GPUArray
    ctx = gpu.create_cuda_context(0)
    try:
        gg = gpu.GPUGeometry(tiny_geometry)
        solid_id = np.asarray(tiny_geometry.solid_id)
        nsolids = int(solid_id.max()) + 1
        before = gg.colors.get()
        rng = np.random.default_rng(5)
        hit = rng.random(nsolids) < 0.3
        new = rng.integers(0, 1 << 32, nsolids, dtype=np.uint64).astype(np.uint32)
        gg.color_solids(hit, new)
        want = np.where(hit[solid_id], new[solid_id], before)
        assert hit.any() and not hit.all() and not np.array_equal(want, before)
        assert np.array_equal(gg.colors.get(), want)
        gg.reset_colors()
        assert np.array_equal(gg.colors.get(), before)
        # the reference's kernel by name, on a slice of the triangles
        first, count = 1000, len(solid_id) // 2
        fn = gpu.get_cu_module('mesh.h').get_function('color_solids')
        hit_gpu = GPUArray(nsolids, np.uint8, ctx).set(hit.view(np.uint8))
        new_gpu = GPUArray(nsolids, np.uint32, ctx).set(new)
        fn(np.int32(first), np.int32(count), gg.solid_id_map, hit_gpu, new_gpu, gg.gpudata, block=(64, 1, 1), grid=(count // 64 + 1, 1))
        want2 = before.copy()
        want2[first:first + count] = want[first:first + count]
        assert np.array_equal(gg.colors.get(), want2)
        with pytest.raises(Exception):
            fn(np.int32(0), np.int32(len(solid_id) + 1), gg.solid_id_map, hit_gpu, new_gpu, gg.gpudata)
        # a ray that meets a recoloured solid first shows its colour
        gg.reset_colors()
        gg.color_solids(np.ones(nsolids, bool), np.full(nsolids, 0x00FF0000, np.uint32))       # opaque (alpha byte 0) pure red
        d = rng.normal(size=(256, 3)).astype(np.float32)
        rays = gpu.GPURays(np.zeros((256, 3), np.float32), d / np.linalg.norm(d, axis=1)[:, None])
        pixels = GPUArray(256, np.uint32, ctx)
        rays.render(gg, pixels, alpha_depth=2)
        px = pixels.get()
        seen = px != 0                  # (rays that meet nothing keep the background)
        assert seen.sum() > 100 and ((px[seen] & 0x00FFFF) == 0).all() and ((px[seen] >> 16) & 0xFF).max() > 100, hex(int(px[seen][0]))
    finally:
        ctx.pop()
