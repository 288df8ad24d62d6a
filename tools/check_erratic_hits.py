"""Are the hits on which the fast walks and the reference differ (tools/diag_sweep_diff.py c3 ->
profiles/r02/diag_erratic_mt_c3.txt) real?  For each of the three rays: Moeller-Trumbore in float64 for the reference's
triangle and for the engine's, the angle between ray and triangle plane, and how far the claimed hit point lies outside
the triangle's vertex bounds.  CPU only (builds the C3 mesh: ~1 min, ~6 GB).  Output: profiles/r02/check_erratic_hits_c3.txt"""
import sys, time, numpy as np
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from chroma_amd import demo
t0 = time.time()
g = demo.detector29k()
g.flatten()
m = g.mesh
print('mesh', len(m.triangles), 'triangles in %.0f s' % (time.time() - t0), flush=True)
cases = [
    ((300.0, -200.0, 150.0), (0.7215774655342102, 0.5825173258781433, 0.3741651177406311), 52782332, 21845.333984375, 52778756, 23551.466796875),
    ((0.0, 0.0, 1200.0), (0.7105749845504761, 0.21008965373039246, 0.6715247631072998), 25879376, 22928.572265625, 25880395, 22934.400390625),
    ((0.0, 0.0, 1200.0), (-0.8305104970932007, 0.011652595363557339, 0.5568810701370239), 34546222, 23063.55078125, 34543534, 23070.66796875),
]
def mt(o, d, v0, v1, v2, dt):
    o, d, v0, v1, v2 = [np.asarray(x, dt) for x in (o, d, v0, v1, v2)]
    e1, e2 = v1 - v0, v2 - v0
    p = np.cross(d, e2); a = e1.dot(p)
    f = 1 / a; s = o - v0; u = f * s.dot(p); q = np.cross(s, e1); v = f * d.dot(q); t = f * e2.dot(q)
    return float(a), float(u), float(v), float(t)
for o, d, tref, dref, teng, deng in cases:
    for name, tri, dist in (('reference', tref, dref), ('engine', teng, deng)):
        v0, v1, v2 = m.vertices[m.triangles[tri]].astype(np.float64)
        n = np.cross(v1 - v0, v2 - v0); n /= np.linalg.norm(n)
        p = np.asarray(o) + dist * np.asarray(d)
        a, u, v, t = mt(o, d, v0, v1, v2, np.float64)
        lo, hi = np.minimum(np.minimum(v0, v1), v2), np.maximum(np.maximum(v0, v1), v2)
        cosang = abs(np.dot(n, d))
        print('%-9s tri %9d: claimed t %.3f | float64 MT: a %.3e u %.4f v %.4f t %.3f | |cos(ray, normal)| %.2e | claimed point is %.1f from the plane, outside the vertex bounds by %s' % (
            name, tri, dist, a, u, v, t, cosang, abs(np.dot(p - v0, n)), np.maximum(np.maximum(lo - p, p - hi), 0).round(1)))
