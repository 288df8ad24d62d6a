"""Find and dissect the photons on which engine and oracle differ in the aimed-ray part of tools/parity_sweep.py
(GPU box).  usage: diag_sweep_diff.py [detector|c3|lite|tiny]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle
from chroma_amd import demo, gpu, event, _lib
from chroma_amd.event import Photons
from chroma_amd.loader import create_geometry_from_obj
from chroma_amd.gpu.geometry import pack_geometry
from chroma_amd.gpu.tools import to_gpu, GPUArray

config = sys.argv[1] if len(sys.argv) > 1 else 'detector'
geo = create_geometry_from_obj({'tiny': demo.tiny, 'lite': demo.detector_lite, 'detector': demo.detector, 'c3': demo.detector29k}[config]())
pk = pack_geometry(geo)
ctx = gpu.create_cuda_context(0)
gg = gpu.GPUDetector(geo, packed=pk)
m = geo.mesh
v = m.vertices.astype(np.float64); t = m.triangles
rng = np.random.default_rng(3)
nthreads = min(64, len(os.sched_getaffinity(0)))
REF = os.path.join(ROOT, 'oracle', '_ref', 'libchroma_ref_mesh.so')


def engine_cast(o, d, last):
    n = len(o)
    dist = GPUArray(n, np.float32, ctx).fill(np.float32(np.nan)); tri = GPUArray(n, np.int32, ctx)
    d_o, d_d, d_l = to_gpu(np.ascontiguousarray(o, np.float32).reshape(-1), ctx), to_gpu(np.ascontiguousarray(d, np.float32).reshape(-1), ctx), to_gpu(np.ascontiguousarray(last, np.int32), ctx)
    _lib.check(ctx._lib.chroma_intersect_mesh(ctx.handle, gg.handle, n, d_o.ptr, d_d.ptr, d_l.ptr, dist.ptr, tri.ptr))
    return dist.get(), tri.get()


def ref_cast(o, d, last):
    ref = ctypes.CDLL(REF)
    vv = np.ascontiguousarray(m.vertices, np.float32); tt = np.ascontiguousarray(m.triangles, np.uint32)
    nodes = np.ascontiguousarray(geo.bvh.nodes.view(np.uint32).reshape(-1, 4))
    origin = (ctypes.c_float * 3)(*[float(x) for x in geo.bvh.world_coords.world_origin])
    n = len(o); rd = np.full(n, np.nan, np.float32); rt = np.full(n, -2, np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    o = np.ascontiguousarray(o, np.float32); d = np.ascontiguousarray(d, np.float32); last = np.ascontiguousarray(last, np.int32)
    assert ref.ref_mesh_run(p(vv), len(vv), p(tt), len(tt), p(nodes), len(nodes), origin, ctypes.c_float(float(geo.bvh.world_coords.world_scale)),
                            n, p(o), p(d), p(last), p(rd), p(rt), 0) == 0
    return rd, rt


for origin in ([0, 0, 0], [300.0, -200.0, 150.0], [0.0, 0.0, 1200.0]):
    pick = rng.choice(len(t), size=min(len(t), 60000), replace=False)
    tri = v[t[pick]]
    targets = np.concatenate([tri.reshape(-1, 3), 0.5 * (tri[:, 0] + tri[:, 1]), 0.5 * (tri[:, 1] + tri[:, 2]), tri.mean(axis=1)])
    d = targets - np.asarray(origin, dtype=np.float64)
    d = d[np.linalg.norm(d, axis=1) > 1e-9]
    d /= np.linalg.norm(d, axis=1)[:, None]
    pol = np.cross(d, np.roll(d, 1, axis=1) + 1e-3); pol /= np.linalg.norm(pol, axis=1)[:, None]
    ph = Photons(np.tile(np.asarray(origin, dtype=float), (len(d), 1)), d, pol, np.full(len(d), 400.0))
    rs = gpu.get_rng_states(64, seed=77)
    gp = gpu.GPUPhotons(ph)
    cur, ctr, ctr_before = ph, None, None
    for step in range(100):
        before = cur
        gp.propagate(gg, rs, max_steps=1)
        cur, ctr, _ = oracle.propagate(pk, cur, seed=77, max_steps=1, rng_counters=ctr, nthreads=nthreads)
        got = gp.get()
        bad = np.flatnonzero((got.last_hit_triangles != cur.last_hit_triangles) | (got.flags != cur.flags) |
                             (got.t.view(np.uint32) != cur.t.view(np.uint32)))
        alive = ((cur.flags & event.TERMINAL_MASK) == 0).sum()
        if len(bad):
            print('origin %s step %d: %d photons differ: %s' % (origin, step, len(bad), bad[:5]), flush=True)
            for i in bad[:3]:
                print('  photon %d BEFORE: pos %r dir %r last %d flags %#x' % (i, before.pos[i].tolist(), before.dir[i].tolist(), before.last_hit_triangles[i], before.flags[i]))
                print('    engine: tri %d flags %#x t %r pos %r' % (got.last_hit_triangles[i], got.flags[i], float(got.t[i]), got.pos[i].tolist()))
                print('    oracle: tri %d flags %#x t %r pos %r' % (cur.last_hit_triangles[i], cur.flags[i], float(cur.t[i]), cur.pos[i].tolist()))
                o1 = before.pos[i:i + 1].astype(np.float32); d1 = before.dir[i:i + 1].astype(np.float32)
                d1 = (d1 / np.sqrt((d1.astype(np.float32) ** 2).sum())).astype(np.float32)
                lh = before.last_hit_triangles[i:i + 1]
                for name, (dd, tt_) in (('engine cast', engine_cast(o1, before.dir[i:i + 1], lh)),
                                        ('oracle cast', oracle.distance_to_mesh(pk, o1, before.dir[i:i + 1], last_hits=lh)[:2]),
                                        ('reference cast', ref_cast(o1, before.dir[i:i + 1], lh) if os.path.exists(REF) else (np.zeros(1), np.zeros(1, int)))):
                    print('    %-15s tri %d distance %r (%#x)' % (name, tt_[0], float(dd[0]), int(np.float32(dd[0]).view(np.uint32))))
                for walk in ('pair', 'coop', 'wide', 'reference'):
                    ctx.set_walk(walk)
                    g2 = gpu.GPUPhotons(before[i:i + 1]); 
                    g2.rng_counters.set(np.asarray(ctr_before[i:i + 1], dtype=np.uint32) if ctr_before is not None else np.zeros(1, np.uint32))
                    g2.propagate(gg, _lib.Rng(77, int(i)), max_steps=1)
                    ctx.set_walk('quad')
                    o = g2.get()
                    print('    single photon, %s walk: tri %d flags %#x' % (walk, o.last_hit_triangles[0], o.flags[0]))
            break
        ctr_before = ctr.copy()
        if alive == 0:
            break
    else:
        pass
    print('origin %s done' % (origin,), flush=True)
