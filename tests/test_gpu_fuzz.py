"""Random and awkward geometries against the oracle: triangle soups (slivers, zero-area and huge triangles mixed with
tiny ones), exactly overlapping duplicate solids (every hit is a tie between two triangles), nested boxes sharing face
planes, an assembly whose faces lie ON the world box.  For each: `intersect_mesh` of a ray bundle with and without
last hits, and a propagate to completion, engine == oracle bit for bit; the literal reference walk must agree
everywhere, the fast walks on all but (at most) the erratic Moeller-Trumbore class of DESIGN.md section 3.1."""
import numpy as np
import pytest

from chroma_amd import event
from conftest import bomb

pytestmark = pytest.mark.gpu
FIELDS = ('flags', 'last_hit_triangles', 'pos', 'dir', 'pol', 't', 'wavelengths')


@pytest.fixture(scope='module')
def gpu():
    from chroma_amd import gpu as g
    ctx = g.create_cuda_context(0)
    yield g
    ctx.pop()


def _soup(seed, n):
    from chroma_amd.geometry import Mesh
    rng = np.random.default_rng(seed)
    centre = rng.uniform(-500, 500, (n, 3))
    size = 10.0 ** rng.uniform(-2, 2.7, (n, 1, 1))                      # 0.01 mm ... 500 mm
    tri = centre[:, None, :] + rng.normal(size=(n, 3, 3)) * size
    tri[::17, 2] = tri[::17, 1]                                          # zero-area triangles (two equal vertices)
    tri[5::23, :, 2] = np.round(tri[5::23, :1, 2])                       # triangles inside planes z = const
    sl = slice(7, None, 29)
    tri[sl, 2] = tri[sl, 0] + (tri[sl, 1] - tri[sl, 0]) * 0.5 + 1e-4     # slivers
    return Mesh(tri.reshape(-1, 3).astype(np.float32), np.arange(3 * n, dtype=np.int32).reshape(-1, 3), remove_duplicate_vertices=False)


def _geometries():
    from chroma_amd.geometry import Geometry, Solid, Surface, vacuum
    from chroma_amd.demo.optics import water, glass, black_surface
    from chroma_amd.make import box, sphere
    shiny = Surface('shiny'); shiny.set('reflect_specular', 0.5); shiny.set('reflect_diffuse', 0.3)
    for seed in (1, 2, 3):
        g = Geometry(water)
        g.add_solid(Solid(_soup(seed, 4000), glass, water, surface=shiny))
        yield 'soup%d' % seed, g
    g = Geometry(water)                                   # two identical spheres at the same place: every hit is a tie
    for k in range(2):
        g.add_solid(Solid(sphere(300.0, 24), glass, water, surface=None if k else shiny))
    g.add_solid(Solid(box(2000.0, 2000.0, 2000.0), water, vacuum, surface=black_surface))
    yield 'twins', g
    g = Geometry(water)                                   # nested boxes that share face planes, the outer one IS the world box
    g.add_solid(Solid(box(1000.0, 1000.0, 1000.0), water, vacuum, surface=shiny))
    g.add_solid(Solid(box(500.0, 1000.0, 500.0), glass, water))
    g.add_solid(Solid(box(500.0, 500.0, 1000.0), glass, water))
    yield 'nested', g


def _cast(gpu, gg, o, d, last, walk):
    from chroma_amd import _lib
    from chroma_amd.gpu.tools import to_gpu, GPUArray
    ctx = gpu.get_context()
    ctx.set_walk(walk)
    try:
        n = len(o)
        dist = GPUArray(n, np.float32, ctx).fill(np.float32(np.nan)); tri = GPUArray(n, np.int32, ctx)
        d_o, d_d, d_l = to_gpu(o.reshape(-1), ctx), to_gpu(d.reshape(-1), ctx), to_gpu(last, ctx)
        _lib.check(ctx._lib.chroma_intersect_mesh(ctx.handle, gg.handle, n, d_o.ptr, d_d.ptr, d_l.ptr, dist.ptr, tri.ptr))
        return dist.get(), tri.get()
    finally:
        ctx.set_walk('quad')


@pytest.mark.parametrize('which', ['soup1', 'soup2', 'soup3', 'twins', 'nested'])
def test_awkward_geometry(gpu, oracle_mod, which):
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.geometry import pack_geometry
    geometry = create_geometry_from_obj(dict(_geometries())[which])
    packed = pack_geometry(geometry)
    gg = gpu.GPUGeometry(geometry)
    rng = np.random.default_rng(99)
    n = 60000
    o = rng.uniform(-400, 400, (n, 3)).astype(np.float32)
    d = rng.normal(size=(n, 3)).astype(np.float32)
    d[:3000] = np.round(d[:3000])                                        # many exactly axis-parallel / diagonal rays
    d[np.abs(d).sum(axis=1) == 0] = [0, 0, 1]
    o[3000:6000] = np.round(o[3000:6000] / 250.0) * 250.0                # origins ON the shared face planes
    none = np.full(n, -1, np.int32)
    wd, wt, _ = oracle_mod.distance_to_mesh(packed, o, d)
    rd, rt = _cast(gpu, gg, o, d, none, 'reference')
    assert np.array_equal(rt, wt) and np.array_equal(rd.view(np.uint32), wd.view(np.uint32)), 'literal walk vs oracle'
    for walk in ('quad', 'pair', 'coop'):
        gd, gt = _cast(gpu, gg, o, d, none, walk)
        bad = (gt != wt) | (gd.view(np.uint32) != wd.view(np.uint32))
        assert bad.sum() <= 2, '%s walk: %d of %d rays differ from the oracle' % (walk, bad.sum(), n)
    # second-step rays: start on the triangle just hit, that triangle as last hit
    hit = wt >= 0
    o2 = (o[hit].astype(np.float64) + wd[hit, None].astype(np.float64) * (d[hit] / np.linalg.norm(d[hit], axis=1)[:, None])).astype(np.float32)
    d2 = rng.normal(size=o2.shape).astype(np.float32)
    last = wt[hit].astype(np.int32)
    wd2, wt2, _ = oracle_mod.distance_to_mesh(packed, o2, d2, last_hits=last)
    gd2, gt2 = _cast(gpu, gg, o2, d2, last, 'quad')
    bad = (gt2 != wt2) | (gd2.view(np.uint32) != wd2.view(np.uint32))
    assert bad.sum() <= 2, 'second-step rays: %d of %d differ' % (bad.sum(), len(o2))
    assert not np.any(gt2 == last)
    # propagate to completion
    ph = bomb(30000, 21)
    ph.pos[:] = rng.uniform(-300, 300, (len(ph), 3))
    rs = gpu.get_rng_states(64, seed=5)
    gp = gpu.GPUPhotons(ph)
    gp.propagate(gg, rs, max_steps=30)
    got = gp.get()
    want, counters, _ = oracle_mod.propagate(packed, ph, seed=5, max_steps=30, nthreads=8)
    bad = np.zeros(len(ph), dtype=bool)
    for f in FIELDS:
        a, b = getattr(got, f), getattr(want, f)
        same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
        bad |= ~same.reshape(len(a), -1).all(axis=1)
    assert bad.sum() <= 2, 'propagate: %d of %d photons differ' % (bad.sum(), len(ph))


def _random_optics(seed):
    """The stress cube (every surface model + bulk re-emission) with all optical numbers drawn at random: wavelength-
    dependent tables, probabilities that need not sum to one, very short and very long lengths, random film and
    dichroic parameters."""
    from chroma_amd.geometry import Solid, Material, Surface, DichroicProps, vacuum, standard_wavelengths
    from chroma_amd.detector import Detector
    from chroma_amd.make import box
    rng = np.random.default_rng(seed)
    wl = standard_wavelengths.astype(float)

    def table(lo, hi, log=False):
        knots = rng.uniform(np.log(lo), np.log(hi), 6) if log else rng.uniform(lo, hi, 6)
        v = np.interp(wl, np.linspace(wl[0], wl[-1], 6), knots)
        return np.exp(v) if log else v

    def cdf():
        w = rng.uniform(0.0, 1.0, len(wl)); w[rng.integers(0, len(wl), 20)] = 0.0       # flat stretches
        c = np.cumsum(w); return c / c[-1]

    scint = Material('scint')
    scint.set('refractive_index', table(1.2, 1.8))
    scint.set('absorption_length', table(5.0, 5000.0, log=True))
    scint.set('scattering_length', table(20.0, 20000.0, log=True))
    tgrid = np.arange(0, 1000, 0.05)
    for k in range(int(rng.integers(1, 4))):
        tc = 1.0 - np.exp(-tgrid / rng.uniform(1.0, 50.0)); tc /= tc[-1]
        p = Material('tmp'); p.set('x', table(0.0, 0.95)); scint.comp_reemission_prob.append(p.x)
        c = Material('tmp'); c.set('x', cdf()); scint.comp_reemission_wvl_cdf.append(c.x)
        scint.comp_reemission_time_cdf.append(np.column_stack([tgrid, tc]).astype(np.float32))
        a = Material('tmp'); a.set('x', table(10.0, 1000.0, log=True)); scint.comp_absorption_length.append(a.x)

    film = Surface('film', model=1)
    film.set('detect', table(0.0, 0.6)); film.set('eta', table(1.1, 3.0)); film.set('k', table(0.0, 2.5))
    film.set('reflect_diffuse', table(0.0, 0.4))
    film.thickness = float(10.0 ** rng.uniform(-6, -3))
    film.transmissive = int(rng.integers(0, 2))
    wls = Surface('wls', model=2)
    wls.set('absorb', table(0.0, 0.9)); wls.set('reemit', table(0.0, 1.0)); wls.set('reflect_specular', table(0.0, 0.3))
    wls.set('reflect_diffuse', table(0.0, 0.3)); wls.set('reemission_cdf', cdf())
    dich = Surface('dichroic', model=3)
    nang = int(rng.integers(2, 7))
    angles = np.sort(rng.uniform(0.0, np.pi / 2, nang)); angles[0] = 0.0
    refl = [np.column_stack([wl, table(0.0, 0.6)]) for _ in range(nang)]
    tran = [np.column_stack([wl, table(0.0, 0.4)]) for _ in range(nang)]
    dich.dichroic_props = DichroicProps(angles, refl, tran)
    pmt = Surface('pmt')
    pmt.set('detect', table(0.0, 0.5)); pmt.set('absorb', table(0.0, 0.4)); pmt.set('reflect_diffuse', table(0.0, 0.3))
    pmt.set('reflect_specular', table(0.0, 0.3))
    black = Surface('black'); black.set('absorb', 1.0)
    mesh = box(200.0, 150.0, 100.0)
    ntri = len(mesh.triangles)
    surfaces = np.empty(ntri, dtype=object)
    order = rng.permutation(ntri)
    for i, s in enumerate(np.array_split(order, 4)):
        surfaces[s] = [film, wls, dich, pmt][i]
    det = Detector(vacuum)
    det.add_pmt(Solid(mesh, scint, vacuum, surface=surfaces))
    det.add_solid(Solid(box(2000.0, 2000.0, 2000.0), vacuum, vacuum, surface=black))
    return det


@pytest.mark.parametrize('seed', [101, 102, 103, 104])
@pytest.mark.parametrize('mode', ['plain', 'weights'])
def test_random_optics(gpu, oracle_mod, seed, mode):
    """Every surface model and the bulk re-emission with random tables: the all-models physics kernel against the
    oracle, bit for bit, with and without photon weights."""
    from chroma_amd.loader import create_geometry_from_obj
    from chroma_amd.gpu.geometry import pack_geometry
    geometry = create_geometry_from_obj(_random_optics(seed))
    packed = pack_geometry(geometry)
    gg = gpu.GPUDetector(geometry)
    ph = bomb(40000, seed, wavelength=300.0, wavelength_hi=700.0)
    kw = dict(use_weights=True, scatter_first=1) if mode == 'weights' else {}
    rs = gpu.get_rng_states(64, seed=seed)
    gp = gpu.GPUPhotons(ph)
    gp.propagate(gg, rs, max_steps=60, **kw)
    got = gp.get()
    want, counters, _ = oracle_mod.propagate(packed, ph, seed=seed, max_steps=60, nthreads=8, **kw)
    for f in FIELDS + ('weights',):
        a, b = getattr(got, f), getattr(want, f)
        same = (a.view(np.uint32) == b.view(np.uint32)) if a.dtype == np.float32 else (a == b)
        assert same.all(), '%s differs for %d photons' % (f, np.count_nonzero(~same.reshape(len(a), -1).all(axis=1)))
    assert np.array_equal(gp.rng_counters.get(), counters)
    assert int(np.bitwise_or.reduce(got.flags)) & 0x1FE                       # (something happened)
