"""Pure-NumPy restatement of the propagate path -- the "NumPy figure" of BASELINE.md section 5 (config C1).

TEST / BASELINE INFRASTRUCTURE, like the rest of oracle/: nothing under chroma_amd/ imports it.  It is a
vectorised restatement (one NumPy expression per line of the reference's per-photon code, all alive
photons at once) of

    intersect_mesh / intersect_box / intersect_triangle   chroma/cuda/mesh.h:16-118, intersect.h:26-147
    fill_state                                             chroma/cuda/photon.h:83-135
    propagate_to_boundary, rayleigh_scatter                chroma/cuda/photon.h:167-308
    propagate_at_surface (default model only)              chroma/cuda/photon.h:630-733
    propagate_at_boundary, specular / diffuse reflectors   chroma/cuda/photon.h:310-398
    the step loop                                          chroma/cuda/propagate.cu:245-318

for geometries whose surfaces all use the default model and whose materials have no re-emitting
components (demo.tiny / demo.detector: C1-C4).  It is NOT a parity oracle: the random numbers come from
numpy.random (the draws of a photon are not the Philox stream of the engine), libm replaces the numeric
contract, and the BVH is walked breadth-first.  What is checked (tests/test_oracle.py): the ray cast
returns the C oracle's triangles and distances, and the history-flag fractions agree statistically.
"""
import numpy as np

from chroma_amd import event

FLT_EPSILON = np.float32(1.1920929e-07)
EPSILON = 1e-6
SPEED_OF_LIGHT = np.float32(299.792458)      # mm/ns, chroma/cuda/physical_constants.h
F = np.float32


class Tables(object):
    """The host arrays of a packed geometry (chroma_amd.gpu.geometry.pack_geometry)."""

    def __init__(self, packed):
        a, d = packed.arrays, packed.desc
        self.vertices = a['vertices'].reshape(-1, 3)
        self.triangles = a['triangles'].reshape(-1, 3)
        self.codes = a['material_codes']
        nodes = a['nodes'].reshape(-1, 4)
        self.wo = np.array([d.world_origin[0], d.world_origin[1], d.world_origin[2]], dtype=np.float32)
        self.ws = F(d.world_scale)
        self.lower = self.wo + (nodes[:, :3] & 0xFFFF).astype(np.float32) * self.ws         # geometry.h:31-47
        self.upper = self.wo + (nodes[:, :3] >> 16).astype(np.float32) * self.ws
        self.child = (nodes[:, 3] & 0x0FFFFFFF).astype(np.int64)
        self.nchild = (nodes[:, 3] >> 28).astype(np.int64)
        self.wl0, self.wlstep, self.wln = F(d.wavelength_start), F(d.wavelength_step), int(d.wavelength_n)
        n = self.wln
        self.refractive_index = a['mat_refractive_index'].reshape(-1, n)
        self.absorption_length = a['mat_absorption_length'].reshape(-1, n)
        self.scattering_length = a['mat_scattering_length'].reshape(-1, n)
        self.num_comp = a['mat_num_comp']
        if int(d.nsurfaces):
            self.surf = {k: a['surf_' + k].reshape(-1, n) for k in ('detect', 'absorb', 'reflect_diffuse', 'reflect_specular')}
            if np.any(a['surf_model'][:int(d.nsurfaces)] != 0):
                raise ValueError('numpy_propagate handles the default surface model only')
        else:
            self.surf = None
        self.v0 = self.vertices[self.triangles[:, 0]]
        self.v1 = self.vertices[self.triangles[:, 1]]
        self.v2 = self.vertices[self.triangles[:, 2]]

    def interp(self, table, rows, x):
        """interp_property (geometry.h:64-75) for one table row per photon."""
        x = x.astype(np.float32)
        last = self.wl0 + F(self.wln - 1) * self.wlstep
        jl = np.clip(((x - self.wl0) / self.wlstep).astype(np.int64), 0, self.wln - 1)
        ju = np.minimum(jl + 1, self.wln - 1)
        lo, hi = table[rows, jl], table[rows, ju]
        y = lo + (x - (self.wl0 + jl.astype(np.float32) * self.wlstep)) * (hi - lo) / self.wlstep
        y = np.where(x < self.wl0, table[rows, 0], y)
        return np.where(x > last, table[rows, self.wln - 1], y).astype(np.float32)


def _dot(a, b):
    return (a * b).sum(axis=1, dtype=np.float32)


def _normalize(a):
    return (a / np.sqrt(_dot(a, a))[:, None]).astype(np.float32)


def intersect_triangles(origin, direction, v0, v1, v2):
    """intersect_triangle (intersect.h:26-95) for paired rays and triangles: (hit mask, distance)."""
    edge1, edge2 = v1 - v0, v2 - v0
    h = np.cross(direction, edge2).astype(np.float32)
    a = _dot(edge1, h)
    ok = ~((a > -FLT_EPSILON) & (a < FLT_EPSILON))
    f = (1.0 / np.where(ok, a, 1).astype(np.float64)).astype(np.float32)
    s = origin - v0
    u = f * _dot(s, h)
    ok &= ~((u.astype(np.float64) < -EPSILON) | (u.astype(np.float64) > 1.0 + EPSILON))
    q = np.cross(s, edge1).astype(np.float32)
    v = f * _dot(direction, q)
    ok &= ~((v.astype(np.float64) < -EPSILON) | ((u + v).astype(np.float64) > 1.0 + EPSILON))
    t = f * _dot(edge2, q)
    ok &= (t.astype(np.float64) > EPSILON) & np.isfinite(t)
    return ok, t


def intersect_mesh(tab, origin, direction, last_hit):
    """Nearest triangle along every ray: the tree of mesh.h:42-118 walked breadth-first, all rays at
    once.  Returns (triangle index or -1, distance).  Ties go to the lower triangle id (the reference:
    to the first one in ITS test order)."""
    n = len(origin)
    with np.errstate(divide='ignore', invalid='ignore'):
        inv = (F(1.0) / direction).astype(np.float32)
        noid = (-origin / direction).astype(np.float32)
    best_t = np.full(n, np.inf, dtype=np.float32)
    best_tri = np.full(n, -1, dtype=np.int64)
    rays = np.arange(n, dtype=np.int64)
    nodes = np.zeros(n, dtype=np.int64)                      # frontier: (ray, node) pairs, starting at the root
    while len(rays):
        lo, hi = tab.lower[nodes], tab.upper[nodes]
        i, o = inv[rays], noid[rays]
        with np.errstate(invalid='ignore', over='ignore'):
            t0, t1 = lo * i + o, hi * i + o                  # intersect_box (intersect.h:107-147)
        finite = np.isfinite(i)
        tmin = np.where(finite, np.fmin(t0, t1), F(0.0)).max(axis=1)
        tmin = np.maximum(tmin, F(0.0))
        tmax = np.where(finite, np.fmax(t0, t1), np.inf).min(axis=1)
        keep = ~(tmin > tmax) & ~(tmin > best_t[rays])       # intersect_node (mesh.h:16-34)
        rays, nodes = rays[keep], nodes[keep]
        if not len(rays):
            break
        leaf = tab.nchild[nodes] == 0
        lr, lt = rays[leaf], tab.child[nodes[leaf]]
        if len(lr):
            m = lt != last_hit[lr]
            lr, lt = lr[m], lt[m]
            ok, t = intersect_triangles(origin[lr], direction[lr], tab.v0[lt], tab.v1[lt], tab.v2[lt])
            lr, lt, t = lr[ok], lt[ok], t[ok]
            if len(lr):
                order = np.lexsort((lt, t, lr))              # per ray: nearest first, then lowest id
                lr, lt, t = lr[order], lt[order], t[order]
                first = np.concatenate(([True], lr[1:] != lr[:-1]))
                lr, lt, t = lr[first], lt[first], t[first]
                better = (t < best_t[lr]) | ((t == best_t[lr]) & (lt < best_tri[lr]))
                best_t[lr[better]] = t[better]
                best_tri[lr[better]] = lt[better]
        rays, nodes = rays[~leaf], nodes[~leaf]
        if not len(rays):
            break
        k = tab.nchild[nodes]                                 # expand every inner node into its children
        first_child = np.repeat(tab.child[nodes], k)
        offs = np.arange(k.sum(), dtype=np.int64) - np.repeat(np.cumsum(k) - k, k)
        rays, nodes = np.repeat(rays, k), first_child + offs
    return best_tri, np.where(best_tri >= 0, best_t, F(-1.0)).astype(np.float32)


def _uniform_sphere(rng, n):                                 # random.h:15-23
    theta = rng.uniform(0, 2 * np.pi, n).astype(np.float32)
    u = rng.uniform(-1, 1, n).astype(np.float32)
    c = np.sqrt(F(1.0) - u * u)
    return np.column_stack([c * np.cos(theta), c * np.sin(theta), u]).astype(np.float32)


def _u(rng, n):                                              # curand_uniform: (0, 1]
    return (F(1.0) - rng.random(n, dtype=np.float32)).astype(np.float32)


def _rotate(a, phi, nrm):                                    # rotate.h:22-28
    c, s = np.cos(phi)[:, None], np.sin(phi)[:, None]
    return (a * c + nrm * _dot(a, nrm)[:, None] * (F(1.0) - c) + np.cross(a, nrm) * s).astype(np.float32)


def _pick_new_direction(axis, theta, phi):                   # photon.h:137-165
    ct, st, cp, sp = np.cos(theta), np.sin(theta), np.cos(phi), np.sin(phi)
    with np.errstate(invalid='ignore', divide='ignore'):
        sat = np.sqrt(F(1.0) - axis[:, 2] * axis[:, 2])
        small = np.isnan(sat) | (sat < 1e-5)
        cap = np.where(small, F(1.0), axis[:, 0] / sat)
        sap = np.where(small, F(0.0), axis[:, 1] / sat)
    x = ct * axis[:, 0] + st * (axis[:, 2] * cp * cap - sp * sap)
    y = ct * axis[:, 1] + st * (cp * axis[:, 2] * sap + sp * cap)
    z = ct * axis[:, 2] - st * cp * sat
    return np.column_stack([x, y, z]).astype(np.float32)


def propagate(packed, photons, seed=0, max_steps=100, tables=None):
    """All photons to termination or ``max_steps``; returns a new Photons (weights mode and forced
    scattering are not restated)."""
    tab = tables or Tables(packed)
    rng = np.random.default_rng(seed)
    pos = photons.pos.astype(np.float32).copy()
    d = _normalize(photons.dir.astype(np.float32))
    pol = _normalize(photons.pol.astype(np.float32))
    wl = photons.wavelengths.astype(np.float32).copy()
    t = photons.t.astype(np.float32).copy()
    flags = photons.flags.astype(np.uint32).copy()
    last = photons.last_hit_triangles.astype(np.int64).copy()
    alive = np.nonzero((flags & event.TERMINAL_MASK) == 0)[0]
    for _ in range(max_steps):
        if not len(alive):
            break
        a = alive
        bad = np.isnan(d[a].prod(axis=1) * pos[a].prod(axis=1))              # propagate.cu:270-273
        flags[a[bad]] |= event.NO_HIT | event.NAN_ABORT
        a = a[~bad]
        tri, dist = intersect_mesh(tab, pos[a], d[a], last[a])                # fill_state
        last[a] = tri
        miss = tri < 0
        flags[a[miss]] |= event.NO_HIT
        a, tri, dist = a[~miss], tri[~miss], dist[~miss]
        if not len(a):
            alive = a
            break
        code = tab.codes[tri]
        conv = lambda c: np.where(c & 0x80, c.astype(np.int64) - 256, c.astype(np.int64))
        inner, outer, surface = conv((code >> 24) & 0xFF), conv((code >> 16) & 0xFF), conv((code >> 8) & 0xFF)
        normal = _normalize(np.cross(tab.v1[tri] - tab.v0[tri], tab.v2[tri] - tab.v1[tri]).astype(np.float32))
        facing = _dot(normal, -d[a]) > 0
        m1, m2 = np.where(facing, outer, inner), np.where(facing, inner, outer)
        normal = np.where(facing[:, None], normal, -normal)
        n1 = tab.interp(tab.refractive_index, m1, wl[a])
        n2 = tab.interp(tab.refractive_index, m2, wl[a])
        labs = tab.interp(tab.absorption_length, m1, wl[a])
        lscat = tab.interp(tab.scattering_length, m1, wl[a])
        if np.any(tab.num_comp[m1] != 0):
            raise ValueError('numpy_propagate does not restate bulk re-emission')

        # ---- propagate_to_boundary (photon.h:193-308)
        with np.errstate(over='ignore'):
            d_abs = -labs * np.log(_u(rng, len(a)))
            d_scat = -lscat * np.log(_u(rng, len(a)))
        absorb = (d_abs <= d_scat) & (d_abs <= dist)
        scatter = ~(d_abs <= d_scat) & (d_scat <= dist)
        travel = np.where(absorb, d_abs, np.where(scatter, d_scat, dist)).astype(np.float32)
        t[a] += travel / (SPEED_OF_LIGHT / n1)
        pos[a] += travel[:, None] * d[a]
        flags[a[absorb]] |= event.BULK_ABSORB
        last[a[absorb | scatter]] = -1
        s = a[scatter]
        if len(s):                                                           # rayleigh_scatter (photon.h:167-191)
            cos_theta = np.clip(F(2.0) * np.cos((np.arccos(F(1.0) - F(2.0) * _u(rng, len(s))) - F(2 * np.pi)) / F(3.0)), -1, 1).astype(np.float32)
            theta = np.arccos(cos_theta)
            phi = rng.uniform(0, 2 * np.pi, len(s)).astype(np.float32)
            nd = _pick_new_direction(pol[s], theta, phi)
            aligned = (F(1.0) - np.abs(cos_theta)) < 1e-6
            npol = np.where(aligned[:, None], _pick_new_direction(pol[s], np.full(len(s), np.pi / 2, dtype=np.float32), phi),
                            pol[s] - cos_theta[:, None] * nd)
            d[s], pol[s] = _normalize(nd), _normalize(npol.astype(np.float32))
            flags[s] |= event.RAYLEIGH_SCATTER
        at = ~(absorb | scatter)                                             # reached the boundary
        b, normal, n1, n2, surface = a[at], normal[at], n1[at], n2[at], surface[at]

        # ---- propagate_at_surface, default model (photon.h:630-733)
        passed = np.ones(len(b), dtype=bool)
        has = surface >= 0
        if has.any() and tab.surf is not None:
            hs = np.nonzero(has)[0]
            sb, si = b[hs], surface[hs]
            detect = tab.interp(tab.surf['detect'], si, wl[sb])
            absorb_p = tab.interp(tab.surf['absorb'], si, wl[sb])
            diffuse = tab.interp(tab.surf['reflect_diffuse'], si, wl[sb])
            specular = tab.interp(tab.surf['reflect_specular'], si, wl[sb])
            u = _u(rng, len(sb))
            is_abs = u < absorb_p
            is_det = ~is_abs & (u < absorb_p + detect)
            is_dif = ~is_abs & ~is_det & (u < absorb_p + detect + diffuse)
            is_spe = ~is_abs & ~is_det & ~is_dif & (u < absorb_p + detect + diffuse + specular)
            flags[sb[is_abs]] |= event.SURFACE_ABSORB
            flags[sb[is_det]] |= event.SURFACE_DETECT
            k = np.nonzero(is_dif)[0]                                         # propagate_at_diffuse_reflector
            todo = k
            while len(todo):
                nd = _uniform_sphere(rng, len(todo))
                ndotv = _dot(nd, normal[hs[todo]])
                flip = ndotv < 0
                nd[flip], ndotv[flip] = -nd[flip], -ndotv[flip]
                acc = _u(rng, len(todo)) < ndotv
                d[sb[todo[acc]]] = nd[acc]
                todo = todo[~acc]
            if len(k):
                pol[sb[k]] = _normalize(np.cross(_uniform_sphere(rng, len(k)), d[sb[k]]).astype(np.float32))
                flags[sb[k]] |= event.REFLECT_DIFFUSE
            k = np.nonzero(is_spe)[0]                                         # propagate_at_specular_reflector
            if len(k):
                inc = np.arccos(np.clip(_dot(normal[hs[k]], -d[sb[k]]), -1, 1))
                ipn = _normalize(np.cross(d[sb[k]], normal[hs[k]]).astype(np.float32))
                d[sb[k]] = _rotate(normal[hs[k]], inc, ipn)
                flags[sb[k]] |= event.REFLECT_SPECULAR
            passed[hs] = ~(is_abs | is_det | is_dif | is_spe)

        # ---- propagate_at_boundary (photon.h:310-363)
        k = np.nonzero(passed)[0]
        if len(k):
            pb, nrm, r1, r2 = b[k], normal[k], n1[k], n2[k]
            inc = np.arccos(np.clip(_dot(nrm, -d[pb]), -1, 1)).astype(np.float32)
            with np.errstate(invalid='ignore'):
                refr = np.arcsin(np.sin(inc) * r1 / r2).astype(np.float32)
            ipn = np.cross(d[pb], nrm).astype(np.float32)
            ipn_len = np.sqrt(_dot(ipn, ipn))
            with np.errstate(invalid='ignore', divide='ignore'):
                ipn = np.where((ipn_len < 1e-6)[:, None], pol[pb], ipn / ipn_len[:, None]).astype(np.float32)
            ncoef = _dot(pol[pb], ipn)
            s_pol = _u(rng, len(pb)) < ncoef * ncoef
            with np.errstate(invalid='ignore', divide='ignore'):
                rc = np.where(s_pol, -np.sin(inc - refr) / np.sin(inc + refr), np.tan(inc - refr) / np.tan(inc + refr))
            reflect = (_u(rng, len(pb)) < rc * rc) | np.isnan(refr)
            nd = np.where(reflect[:, None], _rotate(nrm, inc, ipn), _rotate(nrm, F(np.pi) - np.nan_to_num(refr), ipn))
            d[pb] = nd.astype(np.float32)
            flags[pb[reflect]] |= event.REFLECT_SPECULAR
            pol[pb] = np.where(s_pol[:, None], ipn, _normalize(np.cross(ipn, nd).astype(np.float32)))
        alive = alive[(flags[alive] & event.TERMINAL_MASK) == 0]
    return event.Photons(pos, d, pol, wl, t, last.astype(np.int32), flags)
