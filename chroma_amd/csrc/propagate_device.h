// propagate_device.h -- ray cast and per-step photon physics, device side.
//
// Behavioural specification: chroma/cuda/{mesh.h,intersect.h,geometry.h,photon.h,random.h,
// interpolate.h,rotate.h,cx.h} of the reference (cited per function).  The code is written for
// one photon per lane of a 64-wide wavefront; the traversal stack and the postponed-leaf FIFO
// live in LDS, laid out [entry][lane] so the 64 lanes of a wave touch 64 different banks
// (lanes l and l+32 are served in different LDS cycles, MI355X_MICROARCH.md "LDS").
// NOTE: intersect_mesh contains wave-wide votes; every lane of a wave must call it together
// (lanes without a ray pass lane_on = false).
#pragma once
#include "device_common.h"
#include <float.h>

#define WEIGHT_LOWER_THRESHOLD 0.0001f
#define CHROMA_EPSILON 1e-6
enum { CMD_BREAK = 0, CMD_CONTINUE = 1, CMD_PASS = 2 };

struct Photon {
    v3 position, direction, polarization;
    float wavelength, time, weight;
    uint32_t history;
    int last_hit_triangle;
    uint32_t evidx;
};

struct State {
    v3 surface_normal;
    float refractive_index1, refractive_index2;
    float absorption_length, scattering_length;
    int material1;
    int surface_index;
    float distance_to_boundary;
};

struct LaneCounters { uint32_t steps, nodes, tris, overflows, spills; };

// ---- geometry.h -------------------------------------------------------------------------
// interp_property (geometry.h:64-75); index clamped where the reference reads one past the
// end with a zero weight.
__device__ inline float interp_property(const GeoView &g, float x, const float *fp)
{
    float start = g.wavelength_start, step = g.wavelength_step;
    uint32_t n = g.wavelength_n;
    if (x < start) return fp[0];
    if (x > (start + (float)(n - 1) * step)) return fp[n - 1];
    int jl = cm_f2i((x - start) / step);
    int ju = (jl + 1 < (int)n) ? jl + 1 : (int)n - 1;
    return fp[jl] + (x - (start + (float)jl * step)) * (fp[ju] - fp[jl]) / step;
}
__device__ inline const float *row(const float *tab, const GeoView &g, int idx) { return tab + (size_t)idx * g.wavelength_n; }

// ---- intersect.h --------------------------------------------------------------------------
// intersect_triangle (intersect.h:26-95): Moeller-Trumbore.  The reference forms f = (float)(1.0 /
// (double)a); that equals the correctly rounded float quotient 1.0f / a for every float a (the double
// quotient 2^k/m of a 24-bit m is at least ~2^-49 relative away from any float rounding boundary, far
// more than the 2^-53 the first rounding can move it, so the second rounding decides alike; checked
// exhaustively over the mantissas of six exponents incl. denormal results and on 1.2e8 random floats),
// and the engine is built with correctly rounded float division -- so it divides in float.
// The reference's epsilon comparisons promote a float to double and compare with
// -1e-6, 1+1e-6 and 1e-6; for a FLOAT x, "x < c" with a double c equals "x < (smallest float >= c)" and
// "x > c" equals "x > (largest float <= c)", so they are done in float against those three constants
// (exactly the same decisions, no conversions; the oracle keeps the reference's form).
#define MT_NEG_EPS  __uint_as_float(0xB58637BDu)    // smallest float >= -1e-6
#define MT_ONE_EPS  __uint_as_float(0x3F800008u)    // largest float <= 1 + 1e-6
#define MT_POS_EPS  __uint_as_float(0x358637BDu)    // largest float <= 1e-6
__device__ inline bool intersect_triangle(v3 origin, v3 direction, v3 v0, v3 v1, v3 v2, float &distance)
{
    v3 edge1 = v1 - v0;
    v3 edge2 = v2 - v0;
    v3 h = cross(direction, edge2);
    float a = dot(edge1, h);
    if (a > -FLT_EPSILON && a < FLT_EPSILON) return false;
    float f = 1.0f / a;
    v3 s = origin - v0;
    float u = f * dot(s, h);
    if (u < MT_NEG_EPS || u > MT_ONE_EPS) return false;
    v3 q = cross(s, edge1);
    float v = f * dot(direction, q);
    if (v < MT_NEG_EPS || (u + v) > MT_ONE_EPS) return false;
    float t = f * dot(edge2, q);
    if (t > MT_POS_EPS && t < cm_inff()) {
        distance = t;
        return true;
    }
    return false;
}

// Slab test of intersect_box (intersect.h:107-147) for one child, in the reference's arithmetic.
// Returns tmin (the distance to the box) or -1 when the ray misses it.  For an axis the ray is exactly
// parallel to (1/d = +-inf) the reference skips the slab altogether (intersect.h:115,124,133), which
// makes such a ray walk every box in its plane; here the parallel axis is a containment test with a
// margin of one quantum `ws` instead: a box whose slab is more than a quantum away from the ray's
// constant coordinate holds no triangle within a quantum of the ray, and Moeller-Trumbore accepts
// nothing farther off than 1e-6 of an edge length (intersect.h:65-72) -- the hits found are the same
// while the walk stays short.
__device__ inline float box_tmin(v3 origin, v3 noid, v3 inv_dir, v3 lower, v3 upper, float ws)
{
    float tmin = 0.0f, tmax = cm_inff();
    float t0, t1;
    if (cm_isfinite(inv_dir.x)) {
        t0 = lower.x * inv_dir.x + noid.x;
        t1 = upper.x * inv_dir.x + noid.x;
        tmin = cm_fmaxf(tmin, cm_fminf(t0, t1));
        tmax = cm_fminf(tmax, cm_fmaxf(t0, t1));
    } else if (origin.x < lower.x - ws || origin.x > upper.x + ws) return -1.0f;
    if (cm_isfinite(inv_dir.y)) {
        t0 = lower.y * inv_dir.y + noid.y;
        t1 = upper.y * inv_dir.y + noid.y;
        tmin = cm_fmaxf(tmin, cm_fminf(t0, t1));
        tmax = cm_fminf(tmax, cm_fmaxf(t0, t1));
    } else if (origin.y < lower.y - ws || origin.y > upper.y + ws) return -1.0f;
    if (cm_isfinite(inv_dir.z)) {
        t0 = lower.z * inv_dir.z + noid.z;
        t1 = upper.z * inv_dir.z + noid.z;
        tmin = cm_fmaxf(tmin, cm_fminf(t0, t1));
        tmax = cm_fminf(tmax, cm_fmaxf(t0, t1));
    } else if (origin.z < lower.z - ws || origin.z > upper.z + ws) return -1.0f;
    if (tmin > tmax) return -1.0f;
    return tmin;
}

// ---- when is a fast walk's answer the reference's? ---------------------------------------------
// The fast walks (fused slab test on boxes grown by a quantum, postponed triangle tests, or another
// tree altogether) test a SUPERSET of the triangles the reference tests.  The reference's own box
// test is not conservative: it evaluates a box as world_origin + q * world_scale in float32
// (geometry.h:31-47), which can round the padding of a leaf box away (an upper face sits only a
// fraction of a quantum above the triangle), so a ray that hits a triangle right at the face of its
// leaf box can miss that box -- and the triangle -- in the reference.  A fast walk finds such a hit.
// So the winner W of a fast walk is accepted only if the reference is sure to test W:
//     the reference's slab test (its arithmetic, literally) passes for W's leaf box, with a box
//     distance not beyond W's hit distance.
// That is enough: the float boxes of W's ancestors contain its leaf box (the conversion and the slab
// arithmetic are monotone), so they pass too; none of them can be pruned before W is tested, because
// pruning needs a hit nearer than the box distance, hence nearer than W, and any such hit the reference
// can find the fast walk has found as well (superset) -- W would not be its winner.  With W tested,
// the reference's result is the (distance, rank)-minimal hit among the triangles it tests, which is W.
// A ray whose winner fails the test is walked again by intersect_mesh_strict, the literal reference
// loop.  Measured: a few per 1e8 random rays; rays aimed at mesh vertices and edges find them readily
// (tests/test_gpu_parity.py::test_exact_ties_follow_the_reference_test_order).
__device__ inline bool reference_tests_leaf(const GeoView &g, uint32_t bx, uint32_t by, uint32_t bz, v3 origin, v3 direction, float t)
{
    const float ws = g.world_scale;
    const v3 noid = (-origin) / direction;
    const v3 inv_dir = 1.0f / direction;
    v3 lower = mk3(g.world_origin[0] + (float)(bx & 0xFFFFu) * ws, g.world_origin[1] + (float)(by & 0xFFFFu) * ws,
                   g.world_origin[2] + (float)(bz & 0xFFFFu) * ws);
    v3 upper = mk3(g.world_origin[0] + (float)(bx >> 16) * ws, g.world_origin[1] + (float)(by >> 16) * ws,
                   g.world_origin[2] + (float)(bz >> 16) * ws);
    // intersect_box (intersect.h:107-147): a slab is skipped when 1/d is not finite
    float tmin = 0.0f, tmax = cm_inff();
    float t0, t1;
    if (cm_isfinite(inv_dir.x)) {
        t0 = lower.x * inv_dir.x + noid.x; t1 = upper.x * inv_dir.x + noid.x;
        tmin = cm_fmaxf(tmin, cm_fminf(t0, t1)); tmax = cm_fminf(tmax, cm_fmaxf(t0, t1));
    }
    if (cm_isfinite(inv_dir.y)) {
        t0 = lower.y * inv_dir.y + noid.y; t1 = upper.y * inv_dir.y + noid.y;
        tmin = cm_fmaxf(tmin, cm_fminf(t0, t1)); tmax = cm_fminf(tmax, cm_fmaxf(t0, t1));
    }
    if (cm_isfinite(inv_dir.z)) {
        t0 = lower.z * inv_dir.z + noid.z; t1 = upper.z * inv_dir.z + noid.z;
        tmin = cm_fmaxf(tmin, cm_fminf(t0, t1)); tmax = cm_fminf(tmax, cm_fmaxf(t0, t1));
    }
    return !(tmin > tmax) && !(tmin > t);
}
// The same question asked with the triangle's vertices (the per-step kernels do not carry the box of a
// postponed triangle).  Cheap part first: the leaf box reaches at least a quantum below the vertices'
// minimum and, in exact arithmetic, above their maximum, so a hit point between (minimum - ws/2) and
// (maximum - margin) is inside the float box whatever the rounding did (margin ~ 32 ulp of the largest
// world coordinate): the slab test passes.  Only a hit within the margin of a maximum face (~1e-3 of
// the hits) takes the exact route: the leaf box by the reference's rule, then the test above.
__device__ inline void leaf_words(const GeoView &g, v3 v0, v3 v1, v3 v2, uint32_t &bx, uint32_t &by, uint32_t &bz);
// the cheap sufficient part: true = the reference is sure to test this triangle
__device__ inline bool record_hit_is_plainly_regular(const GeoView &g, float4 a, float4 b, float4 c, v3 origin, v3 direction, float t)
{
    const float m = 2.0f * g.suspect_margin, half = 0.5f * g.world_scale;
    float px = origin.x + t * direction.x, py = origin.y + t * direction.y, pz = origin.z + t * direction.z;
    float lx = fminf(fminf(a.x, b.x), c.x), hx = fmaxf(fmaxf(a.x, b.x), c.x);
    float ly = fminf(fminf(a.y, b.y), c.y), hy = fmaxf(fmaxf(a.y, b.y), c.y);
    float lz = fminf(fminf(a.z, b.z), c.z), hz = fmaxf(fmaxf(a.z, b.z), c.z);
    // (a leaf whose lower bound quantises to 0 is not padded downwards, bvh.cu:181: exact route)
    bool low_ok = px >= lx - half && py >= ly - half && pz >= lz - half &&
                  lx >= g.world_origin[0] + g.world_scale && ly >= g.world_origin[1] + g.world_scale &&
                  lz >= g.world_origin[2] + g.world_scale;
    return low_ok && px <= hx - m && py <= hy - m && pz <= hz - m;
}
// the exact route: the leaf box by the reference's rule, then the reference's slab test
__device__ inline bool record_hit_is_exactly_regular(const GeoView &g, float4 a, float4 b, float4 c, v3 origin, v3 direction, float t)
{
    uint32_t bx, by, bz;
    leaf_words(g, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), bx, by, bz);
    return reference_tests_leaf(g, bx, by, bz, origin, direction, t);
}
__device__ inline bool record_hit_is_regular(const GeoView &g, float4 a, float4 b, float4 c, v3 origin, v3 direction, float t)
{
    return record_hit_is_plainly_regular(g, a, b, c, origin, direction, t) || record_hit_is_exactly_regular(g, a, b, c, origin, direction, t);
}

// the leaf box of a triangle by the reference's rule (cuda/bvh.cu:149-203: truncate, one quantum down,
// one up), for the lane-per-ray kernels, which do not carry the box of a postponed triangle
__device__ inline uint32_t leaf_word(float lo, float hi, float org, float ws)
{
    uint32_t ql = (uint32_t)((lo - org) / ws);
    if (ql > 0) ql--;
    uint32_t qu = (uint32_t)((hi - org) / ws) + 1u;
    return ql | qu << 16;
}
__device__ inline void leaf_words(const GeoView &g, v3 v0, v3 v1, v3 v2, uint32_t &bx, uint32_t &by, uint32_t &bz)
{
    const float ws = g.world_scale;
    bx = leaf_word(fminf(fminf(v0.x, v1.x), v2.x), fmaxf(fmaxf(v0.x, v1.x), v2.x), g.world_origin[0], ws);
    by = leaf_word(fminf(fminf(v0.y, v1.y), v2.y), fmaxf(fmaxf(v0.y, v1.y), v2.y), g.world_origin[1], ws);
    bz = leaf_word(fminf(fminf(v0.z, v1.z), v2.z), fmaxf(fmaxf(v0.z, v1.z), v2.z), g.world_origin[2], ws);
}

// pruning rule of intersect_node (mesh.h:16-34) given the box distance
__device__ inline bool node_passes(float tmin, float min_distance)
{
    if (tmin < 0.0f) return false;
    if (min_distance < 0.0f) return true;
    return !(tmin > min_distance);
}

// ---- mesh.h ---------------------------------------------------------------------------------
// Traversal stack: an entry is the packed `w` word of a node (nchild<<28 | first_child), what the
// reference keeps in its two 1000-entry local arrays (mesh.h:58-59).  The first LDS_N entries
// live in LDS ([entry][lane]: one bank per lane), deeper ones -- rare: the observed depth is
// ~20 -- in a per-lane scratch array, so LDS use stays at LDS_N*256 B per wave.
#define STACK_SCRATCH 104
#ifndef TRAV_PENDING
#define TRAV_PENDING 8      // postponed leaf (triangle) tests per lane, kept in LDS
#endif
template <int LDS_N, int BLOCK>
struct TravStack {
    uint32_t *lds;                       // this lane's column
    uint32_t spill[STACK_SCRATCH];
    __device__ inline void put(int i, uint32_t w) { if (i < LDS_N) lds[i * BLOCK] = w; else spill[i - LDS_N] = w; }
    __device__ inline uint32_t get(int i) const { return (i < LDS_N) ? lds[i * BLOCK] : spill[i - LDS_N]; }
};
// LDS words a block of BLOCK lanes needs for intersect_mesh
#define TRAV_LDS_WORDS(LDS_N, BLOCK) (((LDS_N) + TRAV_PENDING) * (BLOCK))

// Fast slab test used when every component of 1/d is finite and moderate (|1/d| < 1e30, i.e.
// all rays but the exactly/nearly axis-parallel ones).  The box is the node's box grown by G quanta
// on every side (G <= 1: ray_growth below) and the dequantisation is folded into two per-ray constants, so a bound
// costs one convert and one fma:
//     t = (origin_w + (q -+ G) * scale - o) / d  =  fma(q, a, b -+ G a),  a = scale/d, b = (origin_w - o)/d
// Growing the box makes this test strictly more permissive than the reference's slab test
// (intersect.h:107-147, whose rounding differs by far less than a quantum), so it never culls a
// box the reference would enter: the set of triangles tested still contains the reference's.
// HOW MUCH the box is grown (round 3).  The growth only has to cover the difference between the two evaluations of
// the same slab distance -- the reference's fl(fl(fl(wo + fl(q ws)) I) + N) and the fused fl(fma(q, fl(ws I), fl(fl(fma(wo,
// I, N)) -+ G |a|))) with the same I = fl(1/d) and N = fl(-o/d).  With u = 2^-24, E the extent of the 16-bit grid, Lmax the
// largest world coordinate of a grid face and the ray's origin within 1.5 E of the grid's centre, the reference's value is
// within u |I| (3 E + 2 Lmax) of the exact one and the fused value within u |I| 7 E, so the two differ by at most
// u (10 E + 2 Lmax) |I| = u (10 * 65534 + 2 Lmax / ws) quanta * |a| -- 0.04 of a quantum for a world centred on the origin.
// GeoView::slab_grow is four times that bound, at least a quarter of a quantum and at most one (chroma_geometry_create); rounds
// 1 and 2 grew every box by a whole quantum, i.e. a triangle of ~13 quanta tested as if it were ~17 wide instead of ~15.5.
// A ray whose origin lies farther out than 1.5 E from the centre keeps the whole quantum.
__device__ inline float ray_growth(const GeoView &g, v3 origin)
{
    const float half = 32767.0f * g.world_scale, reach = 3.0f * half;         // 1.5 E
    const bool near = cm_fabsf(origin.x - (g.world_origin[0] + half)) <= reach && cm_fabsf(origin.y - (g.world_origin[1] + half)) <= reach &&
                      cm_fabsf(origin.z - (g.world_origin[2] + half)) <= reach;
    return near ? g.slab_grow : 1.0f;
}
struct RayFast { v3 a, blo, bhi; };
__device__ inline RayFast ray_fast(const GeoView &g, v3 noid, v3 inv_dir, float grow)
{
    RayFast r;
    float ws = g.world_scale;
    r.a = mk3(ws * inv_dir.x, ws * inv_dir.y, ws * inv_dir.z);
    v3 b = mk3(cm_fmaf(g.world_origin[0], inv_dir.x, noid.x), cm_fmaf(g.world_origin[1], inv_dir.y, noid.y),
               cm_fmaf(g.world_origin[2], inv_dir.z, noid.z));
    r.blo = b - grow * r.a;
    r.bhi = b + grow * r.a;
    return r;
}
__device__ inline float box_tmin_fast(const RayFast &r, uint4 nd)
{
    float t0x = cm_fmaf((float)(nd.x & 0xFFFFu), r.a.x, r.blo.x), t1x = cm_fmaf((float)(nd.x >> 16), r.a.x, r.bhi.x);
    float t0y = cm_fmaf((float)(nd.y & 0xFFFFu), r.a.y, r.blo.y), t1y = cm_fmaf((float)(nd.y >> 16), r.a.y, r.bhi.y);
    float t0z = cm_fmaf((float)(nd.z & 0xFFFFu), r.a.z, r.blo.z), t1z = cm_fmaf((float)(nd.z >> 16), r.a.z, r.bhi.z);
    // v_min_f32 / v_max_f32: a NaN operand yields the other one, like CUDA's min/max
    float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                                 __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
    float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                                 __builtin_fmaxf(t0z, t1z));
    return (tmin > tmax) ? -1.0f : tmin;
}

// the same slab test returning the interval (the caller compares; no -1 encoding to undo)
__device__ inline void box_interval_fast(const RayFast &r, uint4 nd, float &tmin, float &tmax)
{
    float t0x = cm_fmaf((float)(nd.x & 0xFFFFu), r.a.x, r.blo.x), t1x = cm_fmaf((float)(nd.x >> 16), r.a.x, r.bhi.x);
    float t0y = cm_fmaf((float)(nd.y & 0xFFFFu), r.a.y, r.blo.y), t1y = cm_fmaf((float)(nd.y >> 16), r.a.y, r.bhi.y);
    float t0z = cm_fmaf((float)(nd.z & 0xFFFFu), r.a.z, r.blo.z), t1z = cm_fmaf((float)(nd.z >> 16), r.a.z, r.bhi.z);
    tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(t0x, t1x), __builtin_fminf(t0y, t1y)),
                           __builtin_fmaxf(__builtin_fminf(t0z, t1z), 0.0f));
    tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(t0x, t1x), __builtin_fmaxf(t0y, t1y)),
                           __builtin_fmaxf(t0z, t1z));
}

// intersect_mesh (mesh.h:42-118), literally: the reference's order, its box arithmetic, every triangle
// tested the moment its leaf box is entered.  Lanes are independent (no wave votes), so it is slow and
// used where exactness matters more than speed: rays handed over by the fast walks, the fused
// lane-per-photon kernel, distance_to_mesh.  Works on triangle RECORD indices in and out.
template <int LDS_N, int BLOCK, bool COUNT>
__device__ inline int intersect_mesh_strict(const GeoView &g, v3 origin, v3 direction, float &min_distance,
                                            int last_hit_record, uint32_t *lds, LaneCounters &cnt, bool lane_on)
{
    int triangle_index = -1;
    min_distance = -1.0f;
    if (!lane_on) return -1;
    const v3 noid = (-origin) / direction;
    const v3 inv_dir = 1.0f / direction;
    const v3 wo = mk3(g.world_origin[0], g.world_origin[1], g.world_origin[2]);
    const float ws = g.world_scale;
#define S_LO(nd) mk3(wo.x + (float)((nd).x & 0xFFFFu) * ws, wo.y + (float)((nd).y & 0xFFFFu) * ws, wo.z + (float)((nd).z & 0xFFFFu) * ws)
#define S_HI(nd) mk3(wo.x + (float)((nd).x >> 16) * ws, wo.y + (float)((nd).y >> 16) * ws, wo.z + (float)((nd).z >> 16) * ws)
    TravStack<LDS_N, BLOCK> stack;
    stack.lds = lds;
    uint4 root = g.nodes[0];
    if (!node_passes(box_tmin(origin, noid, inv_dir, S_LO(root), S_HI(root), ws), min_distance)) return -1;
    int sp = 0;
    stack.put(sp++, root.w);
    while (sp > 0) {
        uint32_t w = stack.get(--sp);
        uint32_t first = w & ~CHROMA_NCHILD_MASK, n = w >> CHROMA_CHILD_BITS;
        // the children of a range are contiguous: fetched four at a time (one latency per four),
        // tested strictly in order
        for (uint32_t i0 = first; i0 < first + n; i0 += 4) {
            const uint32_t last = first + n - 1;
            uint4 quad[4];
#pragma unroll
            for (int k = 0; k < 4; k++) quad[k] = g.nodes[min(i0 + (uint32_t)k, last)];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (i0 + (uint32_t)k > last) break;
                const uint4 nd = quad[k];
                if (COUNT) cnt.nodes++;
                if (!node_passes(box_tmin(origin, noid, inv_dir, S_LO(nd), S_HI(nd), ws), min_distance)) continue;
                uint32_t child = nd.w & ~CHROMA_NCHILD_MASK;
                if ((nd.w >> CHROMA_CHILD_BITS) == 0) {
                    if ((int)child == last_hit_record) continue;
                    if (COUNT) cnt.tris++;
                    const float4 *t = g.tri + TRI_STRIDE * (size_t)child;
                    float4 a = t[0], b = t[1], c = t[2];
                    float distance;
                    if (intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance)) {
                        if (triangle_index == -1 || distance < min_distance) {
                            triangle_index = (int)child;
                            min_distance = distance;
                        }
                    }
                } else if (sp >= LDS_N + STACK_SCRATCH) {      // cannot happen when the host check passed
                    cnt.overflows++;
                    return triangle_index;
                } else {
                    stack.put(sp++, nd.w);
                }
            }
        }
    }
#undef S_LO
#undef S_HI
    return triangle_index;
}

// by triangle id in and out (fused kernel, distance_to_mesh)
template <int LDS_N, int BLOCK, bool COUNT>
__device__ inline int intersect_mesh(const GeoView &g, v3 origin, v3 direction, float &min_distance,
                                     int last_hit_triangle, uint32_t *lds, LaneCounters &cnt, bool lane_on = true)
{
    int last_hit_dev = (lane_on && last_hit_triangle >= 0) ? (int)g.tri_to_dev[last_hit_triangle] : -1;
    int found = intersect_mesh_strict<LDS_N, BLOCK, COUNT>(g, origin, direction, min_distance, last_hit_dev, lds, cnt, lane_on);
    return (found >= 0) ? (int)g.dev_to_tri[found] : found;
}
// by record index in and out (retry pass, tail kernel)
template <int LDS_N, int BLOCK, bool COUNT>
__device__ inline int intersect_mesh_dev(const GeoView &g, v3 origin, v3 direction, float &min_distance,
                                         int last_hit_dev, uint32_t *lds, LaneCounters &cnt, bool lane_on = true)
{
    return intersect_mesh_strict<LDS_N, BLOCK, COUNT>(g, origin, direction, min_distance, last_hit_dev, lds, cnt, lane_on);
}

// ---- random.h / interpolate.h -----------------------------------------------------------------
__device__ inline float rng_u(cm_rng &r) { return cm_rng_uniform(&r); }
__device__ inline float uniform(cm_rng &r, float low, float high) { return low + rng_u(r) * (high - low); }   // random.h:9-13
__device__ inline v3 uniform_sphere(cm_rng &r)                                                               // random.h:15-23
{
    float theta = uniform(r, 0.0f, 2 * CM_PI_F);
    float u = uniform(r, -1.0f, 1.0f);
    float c = cm_sqrtf(1.0f - u * u);
    float st, ct;
    cm_sincosf(theta, &st, &ct);
    return mk3(c * ct, c * st, u);
}
// sample_cdf on a uniform grid (random.h:35-55)
__device__ inline float sample_cdf_uniform(cm_rng &r, int ncdf, float x0, float delta, const float *cdf_y)
{
    float u = rng_u(r);
    int lower = 0;
    int upper = ncdf - 1;
    while (lower < upper - 1) {
        int half = (lower + upper) / 2;
        if (u < cdf_y[half]) upper = half; else lower = half;
    }
    float delta_cdf_y = cdf_y[upper] - cdf_y[lower];
    return x0 + delta * (float)lower + delta * (u - cdf_y[lower]) / delta_cdf_y;
}
// interp_idx (interpolate.h:5-29); the last line is evaluated in double as in the reference
__device__ inline float interp_idx(float x, int n, const float *xp)
{
    int lower = 0;
    int upper = n - 1;
    if (x <= xp[lower]) return (float)lower;
    if (x >= xp[upper]) return (float)upper;
    while (lower < upper - 1) {
        int half = (lower + upper) / 2;
        if (x < xp[half]) upper = half; else lower = half;
    }
    float dx = xp[upper] - xp[lower];
    return (float)((double)lower + 1.0 * (double)(x - xp[lower]) / (double)dx);
}

// ---- rotate.h:22-28 ----------------------------------------------------------------------------
__device__ inline v3 rotate(v3 a, float phi, v3 n)
{
    float cos_phi = cm_cosf(phi);
    float sin_phi = cm_sinf(phi);
    return (a * cos_phi + (n * dot(a, n)) * (1.0f - cos_phi)) + cross(a, n) * sin_phi;
}

// ---- photon.h -------------------------------------------------------------------------------------
__device__ inline int convert(int c) { return (c & 0x80) ? (int)(0xFFFFFF00u | (unsigned)c) : c; }   // photon.h:68-75
__device__ inline float get_theta(v3 a, v3 b)                                                         // photon.h:77-81
{ return cm_acosf(cm_fmaxf(-1.0f, cm_fminf(1.0f, dot(a, b)))); }

// fill_state (photon.h:83-135), the part after the ray cast.  The triangle record already holds
// the three vertices and the material code, so the second triangle fetch of the reference is one
// 48-B read (an L2 hit right after the cast).  `record` is the index of the triangle's record.
__device__ inline void apply_hit_record(State &s, Photon &p, const GeoView &g, size_t record, float distance,
                                        float4 a, float4 b, float4 c)
{
    s.distance_to_boundary = distance;
    v3 v0 = mk3(a.x, a.y, a.z), v1 = mk3(b.x, b.y, b.z), v2 = mk3(c.x, c.y, c.z);
    uint32_t material_code = __float_as_uint(a.w);

    int inner_material_index = convert(0xFF & (material_code >> 24));
    int outer_material_index = convert(0xFF & (material_code >> 16));
    s.surface_index = convert(0xFF & (material_code >> 8));

    v3 v01 = v1 - v0;
    v3 v12 = v2 - v1;
    s.surface_normal = normalize(cross(v01, v12));

    int material1, material2;
    if (dot(s.surface_normal, -p.direction) > 0.0f) {
        material1 = outer_material_index;
        material2 = inner_material_index;
    } else {
        material1 = inner_material_index;
        material2 = outer_material_index;
        s.surface_normal = -s.surface_normal;
    }
    s.refractive_index1 = interp_property(g, p.wavelength, row(g.mat_refractive_index, g, material1));
    s.refractive_index2 = interp_property(g, p.wavelength, row(g.mat_refractive_index, g, material2));
    s.absorption_length = interp_property(g, p.wavelength, row(g.mat_absorption_length, g, material1));
    s.scattering_length = interp_property(g, p.wavelength, row(g.mat_scattering_length, g, material1));
    s.material1 = material1;
}
// by triangle id (fused kernel)
__device__ inline void apply_hit(State &s, Photon &p, const GeoView &g, int triangle, float distance)
{
    p.last_hit_triangle = triangle;
    s.distance_to_boundary = distance;
    if (triangle == -1) {
        p.history |= CHROMA_NO_HIT;
        return;
    }
    size_t record = g.tri_to_dev[triangle];
    const float4 *t = g.tri + TRI_STRIDE * record;
    apply_hit_record(s, p, g, record, distance, t[0], t[1], t[2]);
}
// by record index (what the per-step ray cast hands over); the record names its triangle
__device__ inline void apply_hit_dev(State &s, Photon &p, const GeoView &g, int record, float distance)
{
    s.distance_to_boundary = distance;
    if (record == -1) {
        p.last_hit_triangle = -1;
        p.history |= CHROMA_NO_HIT;
        return;
    }
    const float4 *t = g.tri + TRI_STRIDE * (size_t)record;
    float4 a = t[0], b = t[1], c = t[2];
    p.last_hit_triangle = (int)__float_as_uint(b.w);
    apply_hit_record(s, p, g, (size_t)record, distance, a, b, c);
}


// pick_new_direction (photon.h:137-165)
__device__ inline v3 pick_new_direction(v3 axis, float theta, float phi)
{
    float cos_theta, sin_theta;
    cm_sincosf(theta, &sin_theta, &cos_theta);
    float cos_phi, sin_phi;
    cm_sincosf(phi, &sin_phi, &cos_phi);

    float sin_axis_theta = cm_sqrtf(1.0f - axis.z * axis.z);
    float cos_axis_phi, sin_axis_phi;
    if (cm_isnan(sin_axis_theta) || sin_axis_theta < 0.00001f) {
        cos_axis_phi = 1.0f;
        sin_axis_phi = 0.0f;
    } else {
        cos_axis_phi = axis.x / sin_axis_theta;
        sin_axis_phi = axis.y / sin_axis_theta;
    }
    float dirx = cos_theta * axis.x + sin_theta * (axis.z * cos_phi * cos_axis_phi - sin_phi * sin_axis_phi);
    float diry = cos_theta * axis.y + sin_theta * (cos_phi * axis.z * sin_axis_phi + sin_phi * cos_axis_phi);
    float dirz = cos_theta * axis.z - sin_theta * cos_phi * sin_axis_theta;
    return mk3(dirx, diry, dirz);
}

// rayleigh_scatter (photon.h:167-191)
__device__ inline void rayleigh_scatter(Photon &p, cm_rng &rng)
{
    float cos_theta = 2.0f * cm_cosf((cm_acosf(1.0f - 2.0f * rng_u(rng)) - 2 * CM_PI_F) / 3.0f);
    if (cos_theta > 1.0f) cos_theta = 1.0f;
    else if (cos_theta < -1.0f) cos_theta = -1.0f;

    float theta = cm_acosf(cos_theta);
    float phi = uniform(rng, 0.0f, 2.0f * CM_PI_F);

    p.direction = pick_new_direction(p.polarization, theta, phi);

    if (1.0f - cm_fabsf(cos_theta) < 1e-6f)
        p.polarization = pick_new_direction(p.polarization, CM_PI_F / 2.0f, phi);
    else
        p.polarization = p.polarization - cos_theta * p.direction;

    p.direction = p.direction / norm(p.direction);
    p.polarization = p.polarization / norm(p.polarization);
}

// propagate_to_boundary (photon.h:193-308)
// FULL = false: the build for geometries whose materials have no re-emitting components and whose surfaces
// all use the default model (GeoView::plain_optics, decided once at chroma_geometry_create): the bulk
// re-emission, thin-film, wavelength-shifter and dichroic code is not compiled in, which halves the
// registers of the per-step physics kernel for the detectors of configs C1-C4.
// DEFER_SCATTER (k_physics_deal): when the photon scatters, stop after moving it to the scattering point and return
// CMD_SCATTER -- the caller runs rayleigh_scatter (and sets the flag, forgets the triangle) later, among photons that all do.
#define CMD_SCATTER 3
template <bool FULL, bool DEFER_SCATTER = false>
__device__ inline int propagate_to_boundary(Photon &p, State &s, cm_rng &rng, const GeoView &g,
                                            bool use_weights, int scatter_first)
{
    float absorption_distance = -s.absorption_length * cm_logf(rng_u(rng));
    float scattering_distance = -s.scattering_length * cm_logf(rng_u(rng));

    if (use_weights && p.weight > WEIGHT_LOWER_THRESHOLD)
        absorption_distance = 1e30f;
    else
        use_weights = false;

    if (scatter_first == 1) {
        float scatter_prob = 1.0f - cm_expf(-s.distance_to_boundary / s.scattering_length);
        if (scatter_prob > WEIGHT_LOWER_THRESHOLD) {
            int i = 0;
            const int max_i = 1000;
            while (i < max_i && scattering_distance > s.distance_to_boundary) {
                scattering_distance = -s.scattering_length * cm_logf(rng_u(rng));
                i++;
            }
            p.weight *= scatter_prob;
        }
    } else if (scatter_first == -1) {
        float no_scatter_prob = cm_expf(-s.distance_to_boundary / s.scattering_length);
        if (no_scatter_prob > WEIGHT_LOWER_THRESHOLD) {
            int i = 0;
            const int max_i = 1000;
            while (i < max_i && scattering_distance <= s.distance_to_boundary) {
                scattering_distance = -s.scattering_length * cm_logf(rng_u(rng));
                i++;
            }
            p.weight *= no_scatter_prob;
        }
    }

    if (absorption_distance <= scattering_distance) {
        if (absorption_distance <= s.distance_to_boundary) {
            p.time += absorption_distance / (CM_SPEED_OF_LIGHT / s.refractive_index1);
            p.position = p.position + absorption_distance * p.direction;

            uint32_t num_comp = FULL ? g.mat_num_comp[s.material1] : 0u;
            if (num_comp == 0) {
                p.last_hit_triangle = -1;
                p.history |= CHROMA_BULK_ABSORB;
                return CMD_BREAK;
            }
            uint32_t comp_base = g.mat_comp_offset[s.material1];
            float uniform_sample_comp = rng_u(rng);
            float prob = 0.0f;
            uint32_t comp;
            for (comp = 0;; comp++) {
                float comp_abs = interp_property(g, p.wavelength, row(g.comp_absorption_length, g, (int)(comp_base + comp)));
                prob += s.absorption_length / comp_abs;
                if (uniform_sample_comp < prob || comp + 1 == num_comp) break;
            }
            float uniform_sample_reemit = rng_u(rng);
            float comp_reemit_prob = interp_property(g, p.wavelength, row(g.comp_reemission_prob, g, (int)(comp_base + comp)));
            if (uniform_sample_reemit < comp_reemit_prob) {
                p.wavelength = sample_cdf_uniform(rng, (int)g.wavelength_n, g.wavelength_start, g.wavelength_step,
                                                  row(g.comp_reemission_wvl_cdf, g, (int)(comp_base + comp)));
                p.time += sample_cdf_uniform(rng, (int)g.time_n, g.time_start, g.time_step,
                                             g.comp_reemission_time_cdf + (size_t)(comp_base + comp) * g.time_n);
                p.direction = uniform_sphere(rng);
                p.polarization = cross(uniform_sphere(rng), p.direction);
                p.polarization = p.polarization / norm(p.polarization);
                p.last_hit_triangle = -1;
                p.history |= CHROMA_BULK_REEMIT;
                return CMD_CONTINUE;
            } else {
                p.last_hit_triangle = -1;
                p.history |= CHROMA_BULK_ABSORB;
                return CMD_BREAK;
            }
        }
    } else {
        if (scattering_distance <= s.distance_to_boundary) {
            if (use_weights)
                p.weight *= cm_expf(-scattering_distance / s.absorption_length);
            p.time += scattering_distance / (CM_SPEED_OF_LIGHT / s.refractive_index1);
            p.position = p.position + scattering_distance * p.direction;
            if (DEFER_SCATTER) return CMD_SCATTER;
            rayleigh_scatter(p, rng);
            p.history |= CHROMA_RAYLEIGH_SCATTER;
            p.last_hit_triangle = -1;
            return CMD_CONTINUE;
        }
    }

    if (use_weights)
        p.weight *= cm_expf(-s.distance_to_boundary / s.absorption_length);

    p.position = p.position + s.distance_to_boundary * p.direction;
    p.time += s.distance_to_boundary / (CM_SPEED_OF_LIGHT / s.refractive_index1);
    return CMD_PASS;
}

// propagate_at_boundary (photon.h:310-363): Fresnel reflection / refraction
__device__ inline void propagate_at_boundary(Photon &p, State &s, cm_rng &rng)
{
    float incident_angle = get_theta(s.surface_normal, -p.direction);
    float refracted_angle = cm_asinf(cm_sinf(incident_angle) * s.refractive_index1 / s.refractive_index2);

    v3 incident_plane_normal = cross(p.direction, s.surface_normal);
    float incident_plane_normal_length = norm(incident_plane_normal);
    if (incident_plane_normal_length < 1e-6f)
        incident_plane_normal = p.polarization;
    else
        incident_plane_normal = incident_plane_normal / incident_plane_normal_length;

    float normal_coefficient = dot(p.polarization, incident_plane_normal);
    float normal_probability = normal_coefficient * normal_coefficient;

    float reflection_coefficient;
    bool s_pol = rng_u(rng) < normal_probability;
    if (s_pol)
        reflection_coefficient = -cm_sinf(incident_angle - refracted_angle) / cm_sinf(incident_angle + refracted_angle);
    else
        reflection_coefficient = cm_tanf(incident_angle - refracted_angle) / cm_tanf(incident_angle + refracted_angle);

    float u2 = rng_u(rng);
    if ((u2 < reflection_coefficient * reflection_coefficient) || cm_isnan(refracted_angle)) {
        p.direction = rotate(s.surface_normal, incident_angle, incident_plane_normal);
        p.history |= CHROMA_REFLECT_SPECULAR;
    } else {
        p.direction = rotate(s.surface_normal, CM_PI_F - refracted_angle, incident_plane_normal);
    }
    if (s_pol) {
        p.polarization = incident_plane_normal;
    } else {
        p.polarization = cross(incident_plane_normal, p.direction);
        p.polarization = p.polarization / norm(p.polarization);
    }
}

// propagate_at_specular_reflector (photon.h:365-377)
__device__ inline int propagate_at_specular_reflector(Photon &p, State &s)
{
    float incident_angle = get_theta(s.surface_normal, -p.direction);
    v3 incident_plane_normal = cross(p.direction, s.surface_normal);
    incident_plane_normal = incident_plane_normal / norm(incident_plane_normal);
    p.direction = rotate(s.surface_normal, incident_angle, incident_plane_normal);
    p.history |= CHROMA_REFLECT_SPECULAR;
    return CMD_CONTINUE;
}

// propagate_at_diffuse_reflector (photon.h:379-398)
__device__ inline int propagate_at_diffuse_reflector(Photon &p, State &s, cm_rng &rng)
{
    float ndotv;
    do {
        p.direction = uniform_sphere(rng);
        ndotv = dot(p.direction, s.surface_normal);
        if (ndotv < 0.0f) {
            p.direction = -p.direction;
            ndotv = -ndotv;
        }
    } while (!(rng_u(rng) < ndotv));

    p.polarization = cross(uniform_sphere(rng), p.direction);
    p.polarization = p.polarization / norm(p.polarization);
    p.history |= CHROMA_REFLECT_DIFFUSE;
    return CMD_CONTINUE;
}

// ---- complex helpers: cuComplex.h semantics (CUDA toolkit) + chroma/cuda/cx.h:27-37 -----------------
struct cxf { float x, y; };
__device__ inline cxf cx(float r, float i) { return cxf{r, i}; }
__device__ inline cxf cx_add(cxf a, cxf b) { return cx(a.x + b.x, a.y + b.y); }
__device__ inline cxf cx_sub(cxf a, cxf b) { return cx(a.x - b.x, a.y - b.y); }
__device__ inline cxf cx_mul(cxf a, cxf b) { return cx((a.x * b.x) - (a.y * b.y), (a.x * b.y) + (a.y * b.x)); }
__device__ inline cxf cx_div(cxf a, cxf b)
{
    float s = cm_fabsf(b.x) + cm_fabsf(b.y);
    float oos = 1.0f / s;
    float ars = a.x * oos, ais = a.y * oos;
    float brs = b.x * oos, bis = b.y * oos;
    s = (brs * brs) + (bis * bis);
    oos = 1.0f / s;
    return cx(((ars * brs) + (ais * bis)) * oos, ((ais * brs) - (ars * bis)) * oos);
}
__device__ inline float cx_abs(cxf a)
{
    float p = cm_fabsf(a.x), q = cm_fabsf(a.y);
    float v, w, t;
    if (p > q) { v = p; w = q; } else { v = q; w = p; }
    t = w / v;
    t = 1.0f + t * t;
    t = v * cm_sqrtf(t);
    if ((v == 0.0f) || (v > 3.402823466e38f) || (w > 3.402823466e38f)) t = v + w;
    return t;
}
__device__ inline float cx_arg(cxf a) { return cm_atan2f(a.y, a.x); }
__device__ inline cxf cx_sqrt(cxf a)
{
    float r = cm_sqrtf(cx_abs(a));
    float t = cx_arg(a) / 2.0f;
    return cx(r * cm_cosf(t), r * cm_sinf(t));
}

struct RT { float r, t; };
// reflectance / transmittance of the film for one polarisation (photon.h:440-458 and twins)
__device__ inline RT film_rt(cxf r12, cxf r23, cxf t12, cxf t23, cxf gg, float u, float v, float e)
{
    float abs_r12 = cx_abs(r12), abs_r23 = cx_abs(r23);
    float abs_t12 = cx_abs(t12), abs_t23 = cx_abs(t23);
    float arg_r12 = cx_arg(r12), arg_r23 = cx_arg(r23);
    float exp1 = cm_expf(2.0f * v * e);
    float exp2 = 1.0f / exp1;
    float denom = exp1 + abs_r12 * abs_r12 * abs_r23 * abs_r23 * exp2 +
                  2.0f * abs_r12 * abs_r23 * cm_cosf(arg_r23 + arg_r12 + 2.0f * u * e);
    float r = abs_r12 * abs_r12 * exp1 + abs_r23 * abs_r23 * exp2 +
              2.0f * abs_r12 * abs_r23 * cm_cosf(arg_r23 - arg_r12 + 2.0f * u * e);
    r /= denom;
    float t = gg.x * abs_t12 * abs_t12 * abs_t23 * abs_t23;
    t /= denom;
    return RT{r, t};
}

// propagate_complex (photon.h:400-590): thin-film surface
__device__ inline int propagate_complex(Photon &p, State &s, cm_rng &rng, const GeoView &g, int si, bool use_weights)
{
    float detect = interp_property(g, p.wavelength, row(g.surf_detect, g, si));
    float reflect_diffuse = interp_property(g, p.wavelength, row(g.surf_reflect_diffuse, g, si));
    float n2_eta = interp_property(g, p.wavelength, row(g.surf_eta, g, si));
    float n2_k = interp_property(g, p.wavelength, row(g.surf_k, g, si));
    SurfaceInfo info = g.surf_info[si];
    bool transmissive = info.transmissive != 0;

    cxf n1 = cx(s.refractive_index1, 0.0f);
    cxf n2 = cx(n2_eta, n2_k);
    cxf n3 = cx(s.refractive_index2, 0.0f);

    float cos_t1 = dot(p.direction, s.surface_normal);
    if (cos_t1 < 0.0f) cos_t1 = -cos_t1;
    float theta = cm_acosf(cos_t1);

    cxf cos1 = cx(cm_cosf(theta), 0.0f);
    cxf sin1 = cx(cm_sinf(theta), 0.0f);

    float e = (float)((double)(2.0f * CM_PI_F * info.thickness) * 1.0e6 / (double)p.wavelength);   // photon.h:422
    cxf one = cx(1.0f, 0.0f), two = cx(2.0f, 0.0f);
    cxf ratio13sin = cx_mul(cx_mul(cx_div(n1, n3), cx_div(n1, n3)), cx_mul(sin1, sin1));
    cxf cos3 = cx_sqrt(cx_sub(one, ratio13sin));
    cxf ratio12sin = cx_mul(cx_mul(cx_div(n1, n2), cx_div(n1, n2)), cx_mul(sin1, sin1));
    cxf cos2 = cx_sqrt(cx_sub(one, ratio12sin));
    float u = cx_mul(n2, cos2).x;
    float v = cx_mul(n2, cos2).y;

    cxf s_n1c1 = cx_mul(n1, cos1), s_n2c2 = cx_mul(n2, cos2), s_n3c3 = cx_mul(n3, cos3);
    RT srt = film_rt(cx_div(cx_sub(s_n1c1, s_n2c2), cx_add(s_n1c1, s_n2c2)),
                     cx_div(cx_sub(s_n2c2, s_n3c3), cx_add(s_n2c2, s_n3c3)),
                     cx_div(cx_mul(two, s_n1c1), cx_add(s_n1c1, s_n2c2)),
                     cx_div(cx_mul(two, s_n2c2), cx_add(s_n2c2, s_n3c3)),
                     cx_div(s_n3c3, s_n1c1), u, v, e);
    cxf p_n2c1 = cx_mul(n2, cos1), p_n3c2 = cx_mul(n3, cos2), p_n2c3 = cx_mul(n2, cos3), p_n1c2 = cx_mul(n1, cos2);
    RT prt = film_rt(cx_div(cx_sub(p_n2c1, p_n1c2), cx_add(p_n2c1, p_n1c2)),
                     cx_div(cx_sub(p_n3c2, p_n2c3), cx_add(p_n3c2, p_n2c3)),
                     cx_div(cx_mul(cx_mul(two, n1), cos1), cx_add(p_n2c1, p_n1c2)),
                     cx_div(cx_mul(cx_mul(two, n2), cos2), cx_add(p_n3c2, p_n2c3)),
                     cx_div(cx_mul(n3, cos3), cx_mul(n1, cos1)), u, v, e);
    RT nrt = film_rt(cx_div(cx_sub(n1, n2), cx_add(n1, n2)),
                     cx_div(cx_sub(n2, n3), cx_add(n2, n3)),
                     cx_div(cx_mul(two, n1), cx_add(n1, n2)),
                     cx_div(cx_mul(two, n2), cx_add(n2, n3)),
                     cx_div(n3, n1), n2_eta, n2_k, e);

    float incident_angle = get_theta(s.surface_normal, -p.direction);
    float refracted_angle = cm_asinf(cm_sinf(incident_angle) * s.refractive_index1 / s.refractive_index2);
    v3 incident_plane_normal = cross(p.direction, s.surface_normal);
    float incident_plane_normal_length = norm(incident_plane_normal);
    if (incident_plane_normal_length < 1e-6f)
        incident_plane_normal = p.polarization;
    else
        incident_plane_normal = incident_plane_normal / incident_plane_normal_length;
    float normal_coefficient = dot(p.polarization, incident_plane_normal);
    float normal_probability = normal_coefficient * normal_coefficient;

    float transmit = normal_probability * srt.t + (1.0f - normal_probability) * prt.t;
    float transmit_normal_incidence = nrt.t;
    if (!transmissive) {
        transmit = 0.0f;
        transmit_normal_incidence = 0.0f;
    }
    float reflect = normal_probability * srt.r + (1.0f - normal_probability) * prt.r;
    float reflect_normal_incidence = nrt.r;
    float absorb = 1.0f - transmit - reflect;
    float absorb_normal_incidence = 1.0f - transmit_normal_incidence - reflect_normal_incidence;

    detect /= absorb_normal_incidence;
    if (use_weights && p.weight > WEIGHT_LOWER_THRESHOLD && absorb < (1.0f - WEIGHT_LOWER_THRESHOLD)) {
        float survive = 1.0f - absorb;
        absorb = 0.0f;
        p.weight *= survive;
        detect /= survive;
        reflect /= survive;
        transmit /= survive;
    }
    if (use_weights && detect > 0.0f) {
        p.history |= CHROMA_SURFACE_DETECT;
        p.weight *= detect;
        return CMD_BREAK;
    }

    float uniform_sample = rng_u(rng);
    if (uniform_sample < absorb) {
        float uniform_sample_detect = rng_u(rng);
        if (uniform_sample_detect < detect) p.history |= CHROMA_SURFACE_DETECT;
        else p.history |= CHROMA_SURFACE_ABSORB;
        return CMD_BREAK;
    } else if (uniform_sample < absorb + reflect || !transmissive) {
        float uniform_sample_reflect = rng_u(rng);
        if (uniform_sample_reflect < reflect_diffuse)
            return propagate_at_diffuse_reflector(p, s, rng);
        else
            return propagate_at_specular_reflector(p, s);
    } else {
        p.direction = rotate(s.surface_normal, CM_PI_F - refracted_angle, incident_plane_normal);
        p.polarization = cross(incident_plane_normal, p.direction);
        p.polarization = p.polarization / norm(p.polarization);
        p.history |= CHROMA_SURFACE_TRANSMIT;
        return CMD_CONTINUE;
    }
}

// propagate_at_wls (photon.h:592-637)
__device__ inline int propagate_at_wls(Photon &p, State &s, cm_rng &rng, const GeoView &g, int si, bool use_weights)
{
    float absorb = interp_property(g, p.wavelength, row(g.surf_absorb, g, si));
    float reflect_specular = interp_property(g, p.wavelength, row(g.surf_reflect_specular, g, si));
    float reflect_diffuse = interp_property(g, p.wavelength, row(g.surf_reflect_diffuse, g, si));
    float reemit = interp_property(g, p.wavelength, row(g.surf_reemit, g, si));

    float uniform_sample = rng_u(rng);

    if (use_weights && p.weight > WEIGHT_LOWER_THRESHOLD && absorb < (1.0f - WEIGHT_LOWER_THRESHOLD)) {
        float survive = 1.0f - absorb;
        absorb = 0.0f;
        p.weight *= survive;
        reflect_diffuse /= survive;
        reflect_specular /= survive;
    }

    if (uniform_sample < absorb) {
        float uniform_sample_reemit = rng_u(rng);
        if (uniform_sample_reemit < reemit) {
            p.history |= CHROMA_SURFACE_REEMIT;
            p.wavelength = sample_cdf_uniform(rng, (int)g.wavelength_n, g.wavelength_start, g.wavelength_step,
                                              row(g.surf_reemission_cdf, g, si));
            p.direction = uniform_sphere(rng);
            p.polarization = cross(uniform_sphere(rng), p.direction);
            p.polarization = p.polarization / norm(p.polarization);
            return CMD_CONTINUE;
        } else {
            p.history |= CHROMA_SURFACE_ABSORB;
            return CMD_BREAK;
        }
    } else if (uniform_sample < absorb + reflect_specular + reflect_diffuse) {
        float uniform_sample_reflect = rng_u(rng) * (reflect_specular + reflect_diffuse);
        if (uniform_sample_reflect < reflect_specular)
            return propagate_at_specular_reflector(p, s);
        else
            return propagate_at_diffuse_reflector(p, s, rng);
    } else {
        p.history |= CHROMA_SURFACE_TRANSMIT;
        return CMD_PASS;
    }
}

// propagate_at_dichroic (photon.h:640-670)
__device__ inline int propagate_at_dichroic(Photon &p, State &s, cm_rng &rng, const GeoView &g, int si)
{
    float incident_angle = get_theta(s.surface_normal, -p.direction);
    int di = g.surf_info[si].dichroic_index;
    uint32_t nangles = g.dichroic_nangles[di];
    uint32_t base = g.dichroic_offset[di];
    float idx = interp_idx(incident_angle, (int)nangles, g.dichroic_angles + base);
    uint32_t iidx = (uint32_t)cm_f2i(idx);
    uint32_t iidx_hi = iidx < nangles - 2 ? iidx + 1 : iidx;
    float reflect_prob_low = interp_property(g, p.wavelength, row(g.dichroic_reflect, g, (int)(base + iidx)));
    float reflect_prob_high = interp_property(g, p.wavelength, row(g.dichroic_reflect, g, (int)(base + iidx_hi)));
    float transmit_prob_low = interp_property(g, p.wavelength, row(g.dichroic_transmit, g, (int)(base + iidx)));
    float transmit_prob_high = interp_property(g, p.wavelength, row(g.dichroic_transmit, g, (int)(base + iidx_hi)));

    float frac = idx - (float)iidx;
    float reflect_prob = reflect_prob_low + (reflect_prob_high - reflect_prob_low) * frac;
    float transmit_prob = transmit_prob_low + (transmit_prob_high - transmit_prob_low) * frac;

    float uniform_sample = rng_u(rng);
    if (uniform_sample < reflect_prob) {
        return propagate_at_specular_reflector(p, s);
    } else if (uniform_sample < transmit_prob + reflect_prob) {
        p.history |= CHROMA_SURFACE_TRANSMIT;
        return CMD_PASS;
    } else {
        p.history |= CHROMA_SURFACE_ABSORB;
        return CMD_BREAK;
    }
}

// propagate_at_surface (photon.h:672-733)
template <bool FULL>
__device__ inline int propagate_at_surface(Photon &p, State &s, cm_rng &rng, const GeoView &g, bool use_weights)
{
    int si = s.surface_index;
    if (FULL) {
        uint32_t model = g.surf_info[si].model;
        if (model == CHROMA_SURFACE_COMPLEX)
            return propagate_complex(p, s, rng, g, si, use_weights);
        else if (model == CHROMA_SURFACE_WLS)
            return propagate_at_wls(p, s, rng, g, si, use_weights);
        else if (model == CHROMA_SURFACE_DICHROIC)
            return propagate_at_dichroic(p, s, rng, g, si);
    }

    float detect = interp_property(g, p.wavelength, row(g.surf_detect, g, si));
    float absorb = interp_property(g, p.wavelength, row(g.surf_absorb, g, si));
    float reflect_diffuse = interp_property(g, p.wavelength, row(g.surf_reflect_diffuse, g, si));
    float reflect_specular = interp_property(g, p.wavelength, row(g.surf_reflect_specular, g, si));

    float uniform_sample = rng_u(rng);

    if (use_weights && p.weight > WEIGHT_LOWER_THRESHOLD && absorb < (1.0f - WEIGHT_LOWER_THRESHOLD)) {
        float survive = 1.0f - absorb;
        absorb = 0.0f;
        p.weight *= survive;
        detect /= survive;
        reflect_diffuse /= survive;
        reflect_specular /= survive;
    }
    if (use_weights && detect > 0.0f) {
        p.history |= CHROMA_SURFACE_DETECT;
        p.weight *= detect;
        return CMD_BREAK;
    }
    if (uniform_sample < absorb) {
        p.history |= CHROMA_SURFACE_ABSORB;
        return CMD_BREAK;
    } else if (uniform_sample < absorb + detect) {
        p.history |= CHROMA_SURFACE_DETECT;
        return CMD_BREAK;
    } else if (uniform_sample < absorb + detect + reflect_diffuse)
        return propagate_at_diffuse_reflector(p, s, rng);
    else if (uniform_sample < absorb + detect + reflect_diffuse + reflect_specular)
        return propagate_at_specular_reflector(p, s);
    else
        return CMD_PASS;
}

// The part of one loop iteration of propagate.cu:264-301 that follows fill_state.
// Returns false when the photon's loop ends (BREAK), true when it goes on.
template <bool FULL = true>
__device__ inline bool step_after_hit(Photon &p, State &s, cm_rng &rng, const GeoView &g, bool use_weights, int scatter_first)
{
    int command = propagate_to_boundary<FULL>(p, s, rng, g, use_weights, scatter_first);
    if (command == CMD_BREAK) return false;
    if (command == CMD_CONTINUE) return true;
    if (s.surface_index != -1) {
        command = propagate_at_surface<FULL>(p, s, rng, g, use_weights);
        if (command == CMD_BREAK) return false;
        if (command == CMD_CONTINUE) return true;
    }
    propagate_at_boundary(p, s, rng);
    return true;
}
