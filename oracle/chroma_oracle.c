/* chroma_oracle.c -- CPU restatement of chroma's photon propagation path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product (chroma_amd/) never does.
 *
 * Each function restates one reference function in plain scalar C and cites the
 * reference file:line it follows (paths relative to the reference root,
 * pennneutrinos/chroma).  Arithmetic is IEEE-754 single precision with the
 * reference's double-precision islands kept (intersect.h:76-87, interpolate.h:28,
 * photon.h:422); transcendental functions come from include/chroma_math.h (the
 * numeric contract shared with the HIP engine) or, when built with -DORACLE_LIBM,
 * from the host libm.  The random stream is per-photon Philox4x32-10 mapped to
 * (0,1] the way curand_uniform does (include/chroma_math.h); draws are taken in
 * exactly the order the reference takes them.
 *
 * PARITY PIN STATUS (the reference's own sources compiled for gfx950 by oracle/Makefile into oracle/_ref,
 * from where they lie, nothing copied; the tests run on the GPU box)
 *   - ray cast (intersect_mesh / intersect_box / intersect_triangle / get_node, with last_hit_triangle and
 *     with rays through vertices and edges, where the reference's test order decides): PINNED on
 *     chroma/cuda/mesh.h (tests/test_gpu_ref_mesh.py).
 *   - interp_property, interp_idx, interp (the DAQ's CDF sampling), rotate: PINNED on chroma/cuda/geometry.h,
 *     interpolate.h, rotate.h (tests/test_gpu_ref_headers.py; rotate: the algebra -- its cosine is the numeric
 *     contract's, the reference calls the device library's).
 *   - render: PINNED on chroma/cuda/render.cu (tests/test_gpu_render.py).
 *   - host tables, meshes, spiral, flatten: PINNED on vectors made by importing the reference's NumPy modules
 *     (tests/golden/ref_host_model.npz, tools/gen_golden.py).
 *   - physics (photon.h, random.h, cx.h) and daq.cu: the reference's device code needs curand_kernel.h and
 *     cuComplex.h, bvh.cu needs cuda.h; this image has none of them, so they are unbuildable here (no stand-in
 *     headers were written); the reference holds no golden vectors for them, only statistical tests
 *     (test/test_rayleigh.py, test/test_propagation.py, test/test_detector.py), which tests/ restate.  Bit-level
 *     parity of the physics with the CUDA reference is therefore UNPINNED ("parity unpinned").
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>
#include <float.h>
#include <pthread.h>

#include "../include/chroma_hip.h"
#include "../include/chroma_math.h"

#ifdef ORACLE_LIBM
#define M_LOGF   logf
#define M_EXPF   expf
#define M_SINF   sinf
#define M_COSF   cosf
#define M_TANF   tanf
#define M_ASINF  asinf
#define M_ACOSF  acosf
#define M_ATAN2F atan2f
static inline void M_SINCOSF(float x, float *s, float *c) { *s = sinf(x); *c = cosf(x); }
#else
#define M_LOGF   cm_logf
#define M_EXPF   cm_expf
#define M_SINF   cm_sinf
#define M_COSF   cm_cosf
#define M_TANF   cm_tanf
#define M_ASINF  cm_asinf
#define M_ACOSF  cm_acosf
#define M_ATAN2F cm_atan2f
#define M_SINCOSF cm_sincosf
#endif

#define PI CM_PI_F
#define SPEED_OF_LIGHT CM_SPEED_OF_LIGHT
#define WEIGHT_LOWER_THRESHOLD 0.0001f     /* photon.h:13 */
#define STACK_SIZE 1000                    /* mesh.h:9 */
#define CHROMA_EPSILON 1e-6                /* intersect.h:14 (a double literal) */

enum { BREAK, CONTINUE, PASS };            /* photon.h:66 */

/* ---- linalg.h ---------------------------------------------------------- */
typedef struct { float x, y, z; } f3;
static inline f3 mk3(float x, float y, float z) { f3 r = {x, y, z}; return r; }
static inline f3 neg3(f3 a) { return mk3(-a.x, -a.y, -a.z); }
static inline f3 add3(f3 a, f3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline f3 sub3(f3 a, f3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline f3 mul3s(f3 a, float c) { return mk3(a.x * c, a.y * c, a.z * c); }
static inline f3 smul3(float c, f3 a) { return mk3(c * a.x, c * a.y, c * a.z); }
static inline f3 div3s(f3 a, float c) { return mk3(a.x / c, a.y / c, a.z / c); }
static inline f3 div33(f3 a, f3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline f3 sdiv3(float c, f3 a) { return mk3(c / a.x, c / a.y, c / a.z); }
static inline float dot3(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       /* linalg.h:147 */
static inline f3 cross3(f3 a, f3 b)                                                         /* linalg.h:153 */
{ return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
static inline float norm3(f3 a) { return cm_sqrtf(dot3(a, a)); }                           /* linalg.h:165 */
static inline f3 normalize3(f3 a) { return div3s(a, norm3(a)); }                           /* linalg.h:171 */

/* rotate.h:22-28 */
static inline f3 rotate3(f3 a, float phi, f3 n)
{
    float cos_phi = M_COSF(phi);
    float sin_phi = M_SINF(phi);
    f3 t1 = mul3s(a, cos_phi);
    f3 t2 = mul3s(mul3s(n, dot3(a, n)), 1.0f - cos_phi);
    f3 t3 = mul3s(cross3(a, n), sin_phi);
    return add3(add3(t1, t2), t3);
}

/* ---- photon / state (photon.h:15-47) ----------------------------------- */
typedef struct {
    f3 position, direction, polarization;
    float wavelength, time, weight;
    uint32_t history;
    int last_hit_triangle;
    uint32_t evidx;
} Photon;

typedef struct {
    int inside_to_outside;
    f3 surface_normal;
    float refractive_index1, refractive_index2;
    float absorption_length, scattering_length;
    int material1;
    int surface_index;
    float distance_to_boundary;
} State;

typedef struct {
    uint64_t photon_steps, nodes_visited, triangles_tested, stack_overflows;
} Counters;

typedef const chroma_geometry_desc Geo;

/* ---- geometry.h --------------------------------------------------------- */
typedef struct { f3 lower, upper; uint32_t child, nchild; } Node;

/* geometry.h:31-47 */
static inline Node get_node(Geo *g, uint32_t i)
{
    const uint32_t *n = g->nodes + 4 * (size_t)i;
    Node r;
    f3 o = mk3(g->world_origin[0], g->world_origin[1], g->world_origin[2]);
    float s = g->world_scale;
    r.lower = mk3(o.x + (float)(n[0] & 0xFFFFu) * s, o.y + (float)(n[1] & 0xFFFFu) * s, o.z + (float)(n[2] & 0xFFFFu) * s);
    r.upper = mk3(o.x + (float)(n[0] >> 16) * s, o.y + (float)(n[1] >> 16) * s, o.z + (float)(n[2] >> 16) * s);
    r.child = n[3] & ~CHROMA_NCHILD_MASK;
    r.nchild = n[3] >> CHROMA_CHILD_BITS;
    return r;
}

/* geometry.h:50-60 */
static inline void get_triangle(Geo *g, uint32_t i, f3 *v0, f3 *v1, f3 *v2)
{
    const uint32_t *t = g->triangles + 3 * (size_t)i;
    const float *a = g->vertices + 3 * (size_t)t[0];
    const float *b = g->vertices + 3 * (size_t)t[1];
    const float *c = g->vertices + 3 * (size_t)t[2];
    *v0 = mk3(a[0], a[1], a[2]);
    *v1 = mk3(b[0], b[1], b[2]);
    *v2 = mk3(c[0], c[1], c[2]);
}

/* geometry.h:64-75.  The reference reads fp[n] when x sits exactly on the last grid
 * point (jl = n-1); that term is multiplied by 0, so clamping the index keeps the value. */
static inline float interp_property(Geo *g, float x, const float *fp)
{
    float start = g->wavelength_start, step = g->wavelength_step;
    uint32_t n = g->wavelength_n;
    if (x < start) return fp[0];
    if (x > (start + (float)(n - 1) * step)) return fp[n - 1];
    int jl = cm_f2i((x - start) / step);
    int ju = (jl + 1 < (int)n) ? jl + 1 : (int)n - 1;
    return fp[jl] + (x - (start + (float)jl * step)) * (fp[ju] - fp[jl]) / step;
}

/* ---- intersect.h ---------------------------------------------------------- */
/* intersect.h:26-95 (Moeller-Trumbore with double-precision islands) */
static inline int intersect_triangle(f3 origin, f3 direction, f3 v0, f3 v1, f3 v2, float *distance)
{
    f3 edge1 = sub3(v1, v0);
    f3 edge2 = sub3(v2, v0);
    f3 h = cross3(direction, edge2);
    float a = dot3(edge1, h);
    if (a > -FLT_EPSILON && a < FLT_EPSILON) return 0;
    float f = (float)(1.0 / (double)a);
    f3 s = sub3(origin, v0);
    float u = f * dot3(s, h);
    if ((double)u < -CHROMA_EPSILON || (double)u > 1.0 + CHROMA_EPSILON) return 0;
    f3 q = cross3(s, edge1);
    float v = f * dot3(direction, q);
    if ((double)v < -CHROMA_EPSILON || (double)(u + v) > 1.0 + CHROMA_EPSILON) return 0;
    float t = f * dot3(edge2, q);
    if ((double)t > CHROMA_EPSILON && t < cm_inff()) {
        *distance = t;
        return 1;
    }
    return 0;
}

/* intersect.h:107-147 */
static inline int intersect_box(f3 noid, f3 inv_dir, f3 lower, f3 upper, float *distance_to_box)
{
    float tmin = 0.0f, tmax = cm_inff();
    float t0, t1;
    if (cm_isfinite(inv_dir.x)) {
        t0 = lower.x * inv_dir.x + noid.x;
        t1 = upper.x * inv_dir.x + noid.x;
        tmin = cm_fmaxf(tmin, cm_fminf(t0, t1));
        tmax = cm_fminf(tmax, cm_fmaxf(t0, t1));
    }
    if (cm_isfinite(inv_dir.y)) {
        t0 = lower.y * inv_dir.y + noid.y;
        t1 = upper.y * inv_dir.y + noid.y;
        tmin = cm_fmaxf(tmin, cm_fminf(t0, t1));
        tmax = cm_fminf(tmax, cm_fmaxf(t0, t1));
    }
    if (cm_isfinite(inv_dir.z)) {
        t0 = lower.z * inv_dir.z + noid.z;
        t1 = upper.z * inv_dir.z + noid.z;
        tmin = cm_fmaxf(tmin, cm_fminf(t0, t1));
        tmax = cm_fminf(tmax, cm_fmaxf(t0, t1));
    }
    if (tmin > tmax) return 0;
    *distance_to_box = tmin;
    return 1;
}

/* ---- mesh.h ---------------------------------------------------------------- */
/* mesh.h:16-34 */
static inline int intersect_node(f3 noid, f3 inv_dir, const Node *node, float min_distance)
{
    float distance_to_box;
    if (intersect_box(noid, inv_dir, node->lower, node->upper, &distance_to_box)) {
        if (min_distance < 0.0f) return 1;
        if (distance_to_box > min_distance) return 0;
        return 1;
    }
    return 0;
}

/* mesh.h:42-118 */
static int intersect_mesh(f3 origin, f3 direction, Geo *g, float *min_distance_out,
                          int last_hit_triangle, Counters *cnt)
{
    int triangle_index = -1;
    float distance;
    float min_distance = -1.0f;

    Node root = get_node(g, 0);
    f3 noid = div33(neg3(origin), direction);
    f3 inv_dir = sdiv3(1.0f, direction);

    if (!intersect_node(noid, inv_dir, &root, min_distance)) {
        *min_distance_out = min_distance;
        return -1;
    }

    uint32_t child_ptr_stack[STACK_SIZE];
    uint32_t nchild_ptr_stack[STACK_SIZE];
    child_ptr_stack[0] = root.child;
    nchild_ptr_stack[0] = root.nchild;
    int curr = 0;

    while (curr >= 0) {
        uint32_t first_child = child_ptr_stack[curr];
        uint32_t nchild = nchild_ptr_stack[curr];
        curr--;
        for (uint32_t i = first_child; i < first_child + nchild; i++) {
            Node node = get_node(g, i);
            cnt->nodes_visited++;
            if (intersect_node(noid, inv_dir, &node, min_distance)) {
                if (node.nchild == 0) {
                    if ((int)node.child != last_hit_triangle) {
                        cnt->triangles_tested++;
                        f3 v0, v1, v2;
                        get_triangle(g, node.child, &v0, &v1, &v2);
                        if (intersect_triangle(origin, direction, v0, v1, v2, &distance)) {
                            if (triangle_index == -1 || distance < min_distance) {
                                triangle_index = (int)node.child;
                                min_distance = distance;
                            }
                        }
                    }
                } else {
                    if (curr + 1 >= STACK_SIZE) {   /* mesh.h:104-107: warn and stop */
                        cnt->stack_overflows++;
                        curr = -1;
                        break;
                    }
                    curr++;
                    child_ptr_stack[curr] = node.child;
                    nchild_ptr_stack[curr] = node.nchild;
                }
            }
        }
    }
    *min_distance_out = min_distance;
    return triangle_index;
}

/* ---- random.h ---------------------------------------------------------------- */
static inline float rng_u(cm_rng *r) { return cm_rng_uniform(r); }                  /* curand_uniform */
static inline float uniform(cm_rng *r, float low, float high)                       /* random.h:9-13 */
{ return low + rng_u(r) * (high - low); }

static inline f3 uniform_sphere(cm_rng *r)                                          /* random.h:15-23 */
{
    float theta = uniform(r, 0.0f, 2 * PI);
    float u = uniform(r, -1.0f, 1.0f);
    float c = cm_sqrtf(1.0f - u * u);
    return mk3(c * M_COSF(theta), c * M_SINF(theta), u);
}

/* random.h:35-55: sample from a uniformly-sampled CDF */
static inline float sample_cdf_uniform(cm_rng *r, int ncdf, float x0, float delta, const float *cdf_y)
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

/* interpolate.h:5-29 */
static inline float interp_idx(float x, int n, const float *xp)
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

/* ---- photon.h ------------------------------------------------------------------ */
static inline int convert(int c) { return (c & 0x80) ? (int)(0xFFFFFF00u | (unsigned)c) : c; }   /* photon.h:68-75 */

static inline float get_theta(f3 a, f3 b)                                                          /* photon.h:77-81 */
{ return M_ACOSF(cm_fmaxf(-1.0f, cm_fminf(1.0f, dot3(a, b)))); }

static inline const float *mat_row(const float *tab, Geo *g, int idx) { return tab + (size_t)idx * g->wavelength_n; }

/* photon.h:83-135 */
static void fill_state(State *s, Photon *p, Geo *g, Counters *cnt)
{
    p->last_hit_triangle = intersect_mesh(p->position, p->direction, g, &s->distance_to_boundary,
                                          p->last_hit_triangle, cnt);
    if (p->last_hit_triangle == -1) {
        p->history |= CHROMA_NO_HIT;
        return;
    }
    f3 v0, v1, v2;
    get_triangle(g, (uint32_t)p->last_hit_triangle, &v0, &v1, &v2);
    uint32_t material_code = g->material_codes[p->last_hit_triangle];
    int inner_material_index = convert(0xFF & (material_code >> 24));
    int outer_material_index = convert(0xFF & (material_code >> 16));
    s->surface_index = convert(0xFF & (material_code >> 8));

    f3 v01 = sub3(v1, v0);
    f3 v12 = sub3(v2, v1);
    s->surface_normal = normalize3(cross3(v01, v12));

    int material1, material2;
    if (dot3(s->surface_normal, neg3(p->direction)) > 0.0f) {
        material1 = outer_material_index;
        material2 = inner_material_index;
        s->inside_to_outside = 0;
    } else {
        material1 = inner_material_index;
        material2 = outer_material_index;
        s->surface_normal = neg3(s->surface_normal);
        s->inside_to_outside = 1;
    }
    s->refractive_index1 = interp_property(g, p->wavelength, mat_row(g->mat_refractive_index, g, material1));
    s->refractive_index2 = interp_property(g, p->wavelength, mat_row(g->mat_refractive_index, g, material2));
    s->absorption_length = interp_property(g, p->wavelength, mat_row(g->mat_absorption_length, g, material1));
    s->scattering_length = interp_property(g, p->wavelength, mat_row(g->mat_scattering_length, g, material1));
    s->material1 = material1;
}

/* photon.h:137-165 */
static f3 pick_new_direction(f3 axis, float theta, float phi)
{
    float cos_theta, sin_theta;
    M_SINCOSF(theta, &sin_theta, &cos_theta);
    float cos_phi, sin_phi;
    M_SINCOSF(phi, &sin_phi, &cos_phi);

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

/* photon.h:167-191 */
static void rayleigh_scatter(Photon *p, cm_rng *rng)
{
    float cos_theta = 2.0f * M_COSF((M_ACOSF(1.0f - 2.0f * rng_u(rng)) - 2 * PI) / 3.0f);
    if (cos_theta > 1.0f) cos_theta = 1.0f;
    else if (cos_theta < -1.0f) cos_theta = -1.0f;

    float theta = M_ACOSF(cos_theta);
    float phi = uniform(rng, 0.0f, 2.0f * PI);

    p->direction = pick_new_direction(p->polarization, theta, phi);

    if (1.0f - cm_fabsf(cos_theta) < 1e-6f) {
        p->polarization = pick_new_direction(p->polarization, PI / 2.0f, phi);
    } else {
        p->polarization = sub3(p->polarization, smul3(cos_theta, p->direction));
    }
    p->direction = div3s(p->direction, norm3(p->direction));
    p->polarization = div3s(p->polarization, norm3(p->polarization));
}

/* photon.h:193-308 */
static int propagate_to_boundary(Photon *p, State *s, cm_rng *rng, Geo *g, int use_weights, int scatter_first)
{
    float absorption_distance = -s->absorption_length * M_LOGF(rng_u(rng));
    float scattering_distance = -s->scattering_length * M_LOGF(rng_u(rng));

    if (use_weights && p->weight > WEIGHT_LOWER_THRESHOLD)
        absorption_distance = 1e30f;
    else
        use_weights = 0;

    if (scatter_first == 1) {
        float scatter_prob = 1.0f - M_EXPF(-s->distance_to_boundary / s->scattering_length);
        if (scatter_prob > WEIGHT_LOWER_THRESHOLD) {
            int i = 0;
            const int max_i = 1000;
            while (i < max_i && scattering_distance > s->distance_to_boundary) {
                scattering_distance = -s->scattering_length * M_LOGF(rng_u(rng));
                i++;
            }
            p->weight *= scatter_prob;
        }
    } else if (scatter_first == -1) {
        float no_scatter_prob = M_EXPF(-s->distance_to_boundary / s->scattering_length);
        if (no_scatter_prob > WEIGHT_LOWER_THRESHOLD) {
            int i = 0;
            const int max_i = 1000;
            while (i < max_i && scattering_distance <= s->distance_to_boundary) {
                scattering_distance = -s->scattering_length * M_LOGF(rng_u(rng));
                i++;
            }
            p->weight *= no_scatter_prob;
        }
    }

    if (absorption_distance <= scattering_distance) {
        if (absorption_distance <= s->distance_to_boundary) {
            p->time += absorption_distance / (SPEED_OF_LIGHT / s->refractive_index1);
            p->position = add3(p->position, smul3(absorption_distance, p->direction));

            uint32_t num_comp = g->mat_num_comp[s->material1];
            if (num_comp == 0) {
                p->last_hit_triangle = -1;
                p->history |= CHROMA_BULK_ABSORB;
                return BREAK;
            }
            uint32_t comp_base = g->mat_comp_offset[s->material1];
            float uniform_sample_comp = rng_u(rng);
            float prob = 0.0f;
            uint32_t comp;
            for (comp = 0; ; comp++) {
                float comp_abs = interp_property(g, p->wavelength, mat_row(g->comp_absorption_length, g, (int)(comp_base + comp)));
                prob += s->absorption_length / comp_abs;
                if (uniform_sample_comp < prob || comp + 1 == num_comp) break;
            }
            float uniform_sample_reemit = rng_u(rng);
            float comp_reemit_prob = interp_property(g, p->wavelength, mat_row(g->comp_reemission_prob, g, (int)(comp_base + comp)));
            if (uniform_sample_reemit < comp_reemit_prob) {
                p->wavelength = sample_cdf_uniform(rng, (int)g->wavelength_n, g->wavelength_start, g->wavelength_step,
                                                   mat_row(g->comp_reemission_wvl_cdf, g, (int)(comp_base + comp)));
                p->time += sample_cdf_uniform(rng, (int)g->time_n, g->time_start, g->time_step,
                                              g->comp_reemission_time_cdf + (size_t)(comp_base + comp) * g->time_n);
                p->direction = uniform_sphere(rng);
                p->polarization = cross3(uniform_sphere(rng), p->direction);
                p->polarization = div3s(p->polarization, norm3(p->polarization));
                p->last_hit_triangle = -1;
                p->history |= CHROMA_BULK_REEMIT;
                return CONTINUE;
            } else {
                p->last_hit_triangle = -1;
                p->history |= CHROMA_BULK_ABSORB;
                return BREAK;
            }
        }
    } else {
        if (scattering_distance <= s->distance_to_boundary) {
            if (use_weights)
                p->weight *= M_EXPF(-scattering_distance / s->absorption_length);
            p->time += scattering_distance / (SPEED_OF_LIGHT / s->refractive_index1);
            p->position = add3(p->position, smul3(scattering_distance, p->direction));
            rayleigh_scatter(p, rng);
            p->history |= CHROMA_RAYLEIGH_SCATTER;
            p->last_hit_triangle = -1;
            return CONTINUE;
        }
    }

    if (use_weights)
        p->weight *= M_EXPF(-s->distance_to_boundary / s->absorption_length);

    p->position = add3(p->position, smul3(s->distance_to_boundary, p->direction));
    p->time += s->distance_to_boundary / (SPEED_OF_LIGHT / s->refractive_index1);
    return PASS;
}

/* photon.h:310-363 */
static void propagate_at_boundary(Photon *p, State *s, cm_rng *rng)
{
    float incident_angle = get_theta(s->surface_normal, neg3(p->direction));
    float refracted_angle = M_ASINF(M_SINF(incident_angle) * s->refractive_index1 / s->refractive_index2);

    f3 incident_plane_normal = cross3(p->direction, s->surface_normal);
    float incident_plane_normal_length = norm3(incident_plane_normal);

    if (incident_plane_normal_length < 1e-6f)
        incident_plane_normal = p->polarization;
    else
        incident_plane_normal = div3s(incident_plane_normal, incident_plane_normal_length);

    float normal_coefficient = dot3(p->polarization, incident_plane_normal);
    float normal_probability = normal_coefficient * normal_coefficient;

    float reflection_coefficient;
    if (rng_u(rng) < normal_probability) {
        reflection_coefficient = -M_SINF(incident_angle - refracted_angle) / M_SINF(incident_angle + refracted_angle);
        float u2 = rng_u(rng);
        if ((u2 < reflection_coefficient * reflection_coefficient) || cm_isnan(refracted_angle)) {
            p->direction = rotate3(s->surface_normal, incident_angle, incident_plane_normal);
            p->history |= CHROMA_REFLECT_SPECULAR;
        } else {
            p->direction = rotate3(s->surface_normal, PI - refracted_angle, incident_plane_normal);
        }
        p->polarization = incident_plane_normal;
    } else {
        reflection_coefficient = M_TANF(incident_angle - refracted_angle) / M_TANF(incident_angle + refracted_angle);
        float u2 = rng_u(rng);
        if ((u2 < reflection_coefficient * reflection_coefficient) || cm_isnan(refracted_angle)) {
            p->direction = rotate3(s->surface_normal, incident_angle, incident_plane_normal);
            p->history |= CHROMA_REFLECT_SPECULAR;
        } else {
            p->direction = rotate3(s->surface_normal, PI - refracted_angle, incident_plane_normal);
        }
        p->polarization = cross3(incident_plane_normal, p->direction);
        p->polarization = div3s(p->polarization, norm3(p->polarization));
    }
}

/* photon.h:365-377 */
static int propagate_at_specular_reflector(Photon *p, State *s)
{
    float incident_angle = get_theta(s->surface_normal, neg3(p->direction));
    f3 incident_plane_normal = cross3(p->direction, s->surface_normal);
    incident_plane_normal = div3s(incident_plane_normal, norm3(incident_plane_normal));
    p->direction = rotate3(s->surface_normal, incident_angle, incident_plane_normal);
    p->history |= CHROMA_REFLECT_SPECULAR;
    return CONTINUE;
}

/* photon.h:379-398 */
static int propagate_at_diffuse_reflector(Photon *p, State *s, cm_rng *rng)
{
    float ndotv;
    do {
        p->direction = uniform_sphere(rng);
        ndotv = dot3(p->direction, s->surface_normal);
        if (ndotv < 0.0f) {
            p->direction = neg3(p->direction);
            ndotv = -ndotv;
        }
    } while (!(rng_u(rng) < ndotv));

    p->polarization = cross3(uniform_sphere(rng), p->direction);
    p->polarization = div3s(p->polarization, norm3(p->polarization));
    p->history |= CHROMA_REFLECT_DIFFUSE;
    return CONTINUE;
}

/* ---- complex arithmetic: cuComplex.h (CUDA toolkit, absent from the reference tree;
 * pinned only by the Docker base image nvidia/cudagl:11.4.2) + chroma/cuda/cx.h -------- */
typedef struct { float x, y; } cxf;
static inline cxf cx(float r, float i) { cxf c = {r, i}; return c; }
static inline cxf cx_add(cxf a, cxf b) { return cx(a.x + b.x, a.y + b.y); }
static inline cxf cx_sub(cxf a, cxf b) { return cx(a.x - b.x, a.y - b.y); }
static inline cxf cx_mul(cxf a, cxf b)                                   /* cuCmulf */
{ return cx((a.x * b.x) - (a.y * b.y), (a.x * b.y) + (a.y * b.x)); }
static inline cxf cx_div(cxf a, cxf b)                                   /* cuCdivf: scaled division */
{
    float s = cm_fabsf(b.x) + cm_fabsf(b.y);
    float oos = 1.0f / s;
    float ars = a.x * oos, ais = a.y * oos;
    float brs = b.x * oos, bis = b.y * oos;
    s = (brs * brs) + (bis * bis);
    oos = 1.0f / s;
    return cx(((ars * brs) + (ais * bis)) * oos, ((ais * brs) - (ars * bis)) * oos);
}
static inline float cx_abs(cxf a)                                        /* cuCabsf: scaled hypot */
{
    float p = cm_fabsf(a.x), q = cm_fabsf(a.y);
    float v, w, t;                         /* v = larger, w = smaller */
    if (p > q) { v = p; w = q; } else { v = q; w = p; }
    t = w / v;
    t = 1.0f + t * t;
    t = v * cm_sqrtf(t);
    if ((v == 0.0f) || (v > 3.402823466e38f) || (w > 3.402823466e38f)) t = v + w;
    return t;
}
static inline float cx_arg(cxf a) { return M_ATAN2F(a.y, a.x); }         /* cx.h:27-30 */
static inline cxf cx_sqrt(cxf a)                                          /* cx.h:32-37 */
{
    float r = cm_sqrtf(cx_abs(a));
    float t = cx_arg(a) / 2.0f;
    return cx(r * M_COSF(t), r * M_SINF(t));
}

typedef struct { float r, t; } RT;
/* the s/p/normal-incidence blocks of photon.h:430-514 share this algebra */
static inline RT film_rt(cxf r12, cxf r23, cxf t12, cxf t23, cxf gg, float u, float v, float e)
{
    float abs_r12 = cx_abs(r12), abs_r23 = cx_abs(r23);
    float abs_t12 = cx_abs(t12), abs_t23 = cx_abs(t23);
    float arg_r12 = cx_arg(r12), arg_r23 = cx_arg(r23);
    float exp1 = M_EXPF(2.0f * v * e);
    float exp2 = 1.0f / exp1;
    float denom = exp1 + abs_r12 * abs_r12 * abs_r23 * abs_r23 * exp2 +
                  2.0f * abs_r12 * abs_r23 * M_COSF(arg_r23 + arg_r12 + 2.0f * u * e);
    float r = abs_r12 * abs_r12 * exp1 + abs_r23 * abs_r23 * exp2 +
              2.0f * abs_r12 * abs_r23 * M_COSF(arg_r23 - arg_r12 + 2.0f * u * e);
    r /= denom;
    float t = gg.x * abs_t12 * abs_t12 * abs_t23 * abs_t23;
    t /= denom;
    RT out = {r, t};
    return out;
}

/* photon.h:400-590 */
static int propagate_complex(Photon *p, State *s, cm_rng *rng, Geo *g, int si, int use_weights)
{
    float detect = interp_property(g, p->wavelength, mat_row(g->surf_detect, g, si));
    float reflect_specular = interp_property(g, p->wavelength, mat_row(g->surf_reflect_specular, g, si));
    float reflect_diffuse = interp_property(g, p->wavelength, mat_row(g->surf_reflect_diffuse, g, si));
    float n2_eta = interp_property(g, p->wavelength, mat_row(g->surf_eta, g, si));
    float n2_k = interp_property(g, p->wavelength, mat_row(g->surf_k, g, si));
    (void)reflect_specular;
    int transmissive = (int)g->surf_transmissive[si];

    cxf n1 = cx(s->refractive_index1, 0.0f);
    cxf n2 = cx(n2_eta, n2_k);
    cxf n3 = cx(s->refractive_index2, 0.0f);

    float cos_t1 = dot3(p->direction, s->surface_normal);
    if (cos_t1 < 0.0f) cos_t1 = -cos_t1;
    float theta = M_ACOSF(cos_t1);

    cxf cos1 = cx(M_COSF(theta), 0.0f);
    cxf sin1 = cx(M_SINF(theta), 0.0f);

    /* photon.h:422: double-precision island, then narrowed */
    float e = (float)((double)(2.0f * PI * g->surf_thickness[si]) * 1.0e6 / (double)p->wavelength);
    cxf one = cx(1.0f, 0.0f), two = cx(2.0f, 0.0f);
    cxf ratio13sin = cx_mul(cx_mul(cx_div(n1, n3), cx_div(n1, n3)), cx_mul(sin1, sin1));
    cxf cos3 = cx_sqrt(cx_sub(one, ratio13sin));
    cxf ratio12sin = cx_mul(cx_mul(cx_div(n1, n2), cx_div(n1, n2)), cx_mul(sin1, sin1));
    cxf cos2 = cx_sqrt(cx_sub(one, ratio12sin));
    float u = cx_mul(n2, cos2).x;
    float v = cx_mul(n2, cos2).y;

    /* s polarization (photon.h:430-458) */
    cxf s_n1c1 = cx_mul(n1, cos1), s_n2c2 = cx_mul(n2, cos2), s_n3c3 = cx_mul(n3, cos3);
    RT srt = film_rt(cx_div(cx_sub(s_n1c1, s_n2c2), cx_add(s_n1c1, s_n2c2)),
                     cx_div(cx_sub(s_n2c2, s_n3c3), cx_add(s_n2c2, s_n3c3)),
                     cx_div(cx_mul(two, s_n1c1), cx_add(s_n1c1, s_n2c2)),
                     cx_div(cx_mul(two, s_n2c2), cx_add(s_n2c2, s_n3c3)),
                     cx_div(s_n3c3, s_n1c1), u, v, e);
    /* p polarization (photon.h:460-489) */
    cxf p_n2c1 = cx_mul(n2, cos1), p_n3c2 = cx_mul(n3, cos2), p_n2c3 = cx_mul(n2, cos3), p_n1c2 = cx_mul(n1, cos2);
    RT prt = film_rt(cx_div(cx_sub(p_n2c1, p_n1c2), cx_add(p_n2c1, p_n1c2)),
                     cx_div(cx_sub(p_n3c2, p_n2c3), cx_add(p_n3c2, p_n2c3)),
                     cx_div(cx_mul(cx_mul(two, n1), cos1), cx_add(p_n2c1, p_n1c2)),
                     cx_div(cx_mul(cx_mul(two, n2), cos2), cx_add(p_n3c2, p_n2c3)),
                     cx_div(cx_mul(n3, cos3), cx_mul(n1, cos1)), u, v, e);
    /* normal incidence (photon.h:490-514) */
    RT nrt = film_rt(cx_div(cx_sub(n1, n2), cx_add(n1, n2)),
                     cx_div(cx_sub(n2, n3), cx_add(n2, n3)),
                     cx_div(cx_mul(two, n1), cx_add(n1, n2)),
                     cx_div(cx_mul(two, n2), cx_add(n2, n3)),
                     cx_div(n3, n1), n2_eta, n2_k, e);

    /* photon.h:516-529 */
    float incident_angle = get_theta(s->surface_normal, neg3(p->direction));
    float refracted_angle = M_ASINF(M_SINF(incident_angle) * s->refractive_index1 / s->refractive_index2);
    f3 incident_plane_normal = cross3(p->direction, s->surface_normal);
    float incident_plane_normal_length = norm3(incident_plane_normal);
    if (incident_plane_normal_length < 1e-6f)
        incident_plane_normal = p->polarization;
    else
        incident_plane_normal = div3s(incident_plane_normal, incident_plane_normal_length);
    float normal_coefficient = dot3(p->polarization, incident_plane_normal);
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
    if (use_weights && p->weight > WEIGHT_LOWER_THRESHOLD && absorb < (1.0f - WEIGHT_LOWER_THRESHOLD)) {
        float survive = 1.0f - absorb;
        absorb = 0.0f;
        p->weight *= survive;
        detect /= survive;
        reflect /= survive;
        transmit /= survive;
    }
    if (use_weights && detect > 0.0f) {
        p->history |= CHROMA_SURFACE_DETECT;
        p->weight *= detect;
        return BREAK;
    }

    float uniform_sample = rng_u(rng);
    if (uniform_sample < absorb) {
        float uniform_sample_detect = rng_u(rng);
        if (uniform_sample_detect < detect) p->history |= CHROMA_SURFACE_DETECT;
        else p->history |= CHROMA_SURFACE_ABSORB;
        return BREAK;
    } else if (uniform_sample < absorb + reflect || !transmissive) {
        float uniform_sample_reflect = rng_u(rng);
        if (uniform_sample_reflect < reflect_diffuse)
            return propagate_at_diffuse_reflector(p, s, rng);
        else
            return propagate_at_specular_reflector(p, s);
    } else {
        p->direction = rotate3(s->surface_normal, PI - refracted_angle, incident_plane_normal);
        p->polarization = cross3(incident_plane_normal, p->direction);
        p->polarization = div3s(p->polarization, norm3(p->polarization));
        p->history |= CHROMA_SURFACE_TRANSMIT;
        return CONTINUE;
    }
}

/* photon.h:592-637 */
static int propagate_at_wls(Photon *p, State *s, cm_rng *rng, Geo *g, int si, int use_weights)
{
    float absorb = interp_property(g, p->wavelength, mat_row(g->surf_absorb, g, si));
    float reflect_specular = interp_property(g, p->wavelength, mat_row(g->surf_reflect_specular, g, si));
    float reflect_diffuse = interp_property(g, p->wavelength, mat_row(g->surf_reflect_diffuse, g, si));
    float reemit = interp_property(g, p->wavelength, mat_row(g->surf_reemit, g, si));

    float uniform_sample = rng_u(rng);

    if (use_weights && p->weight > WEIGHT_LOWER_THRESHOLD && absorb < (1.0f - WEIGHT_LOWER_THRESHOLD)) {
        float survive = 1.0f - absorb;
        absorb = 0.0f;
        p->weight *= survive;
        reflect_diffuse /= survive;
        reflect_specular /= survive;
    }

    if (uniform_sample < absorb) {
        float uniform_sample_reemit = rng_u(rng);
        if (uniform_sample_reemit < reemit) {
            p->history |= CHROMA_SURFACE_REEMIT;
            p->wavelength = sample_cdf_uniform(rng, (int)g->wavelength_n, g->wavelength_start, g->wavelength_step,
                                               mat_row(g->surf_reemission_cdf, g, si));
            p->direction = uniform_sphere(rng);
            p->polarization = cross3(uniform_sphere(rng), p->direction);
            p->polarization = div3s(p->polarization, norm3(p->polarization));
            return CONTINUE;
        } else {
            p->history |= CHROMA_SURFACE_ABSORB;
            return BREAK;
        }
    } else if (uniform_sample < absorb + reflect_specular + reflect_diffuse) {
        float uniform_sample_reflect = rng_u(rng) * (reflect_specular + reflect_diffuse);
        if (uniform_sample_reflect < reflect_specular)
            return propagate_at_specular_reflector(p, s);
        else
            return propagate_at_diffuse_reflector(p, s, rng);
    } else {
        p->history |= CHROMA_SURFACE_TRANSMIT;
        return PASS;
    }
}

/* photon.h:640-670 */
static int propagate_at_dichroic(Photon *p, State *s, cm_rng *rng, Geo *g, int si)
{
    float incident_angle = get_theta(s->surface_normal, neg3(p->direction));
    int di = g->surf_dichroic_index[si];
    uint32_t nangles = g->dichroic_nangles[di];
    uint32_t base = g->dichroic_offset[di];
    float idx = interp_idx(incident_angle, (int)nangles, g->dichroic_angles + base);
    uint32_t iidx = (uint32_t)cm_f2i(idx);
    uint32_t iidx_hi = iidx < nangles - 2 ? iidx + 1 : iidx;
    float reflect_prob_low = interp_property(g, p->wavelength, mat_row(g->dichroic_reflect, g, (int)(base + iidx)));
    float reflect_prob_high = interp_property(g, p->wavelength, mat_row(g->dichroic_reflect, g, (int)(base + iidx_hi)));
    float transmit_prob_low = interp_property(g, p->wavelength, mat_row(g->dichroic_transmit, g, (int)(base + iidx)));
    float transmit_prob_high = interp_property(g, p->wavelength, mat_row(g->dichroic_transmit, g, (int)(base + iidx_hi)));

    float frac = idx - (float)iidx;
    float reflect_prob = reflect_prob_low + (reflect_prob_high - reflect_prob_low) * frac;
    float transmit_prob = transmit_prob_low + (transmit_prob_high - transmit_prob_low) * frac;

    float uniform_sample = rng_u(rng);
    if (uniform_sample < reflect_prob) {
        return propagate_at_specular_reflector(p, s);
    } else if (uniform_sample < transmit_prob + reflect_prob) {
        p->history |= CHROMA_SURFACE_TRANSMIT;
        return PASS;
    } else {
        p->history |= CHROMA_SURFACE_ABSORB;
        return BREAK;
    }
}

/* photon.h:672-733 */
static int propagate_at_surface(Photon *p, State *s, cm_rng *rng, Geo *g, int use_weights)
{
    int si = s->surface_index;
    uint32_t model = g->surf_model[si];
    if (model == CHROMA_SURFACE_COMPLEX)
        return propagate_complex(p, s, rng, g, si, use_weights);
    else if (model == CHROMA_SURFACE_WLS)
        return propagate_at_wls(p, s, rng, g, si, use_weights);
    else if (model == CHROMA_SURFACE_DICHROIC)
        return propagate_at_dichroic(p, s, rng, g, si);
    else {
        float detect = interp_property(g, p->wavelength, mat_row(g->surf_detect, g, si));
        float absorb = interp_property(g, p->wavelength, mat_row(g->surf_absorb, g, si));
        float reflect_diffuse = interp_property(g, p->wavelength, mat_row(g->surf_reflect_diffuse, g, si));
        float reflect_specular = interp_property(g, p->wavelength, mat_row(g->surf_reflect_specular, g, si));

        float uniform_sample = rng_u(rng);

        if (use_weights && p->weight > WEIGHT_LOWER_THRESHOLD && absorb < (1.0f - WEIGHT_LOWER_THRESHOLD)) {
            float survive = 1.0f - absorb;
            absorb = 0.0f;
            p->weight *= survive;
            detect /= survive;
            reflect_diffuse /= survive;
            reflect_specular /= survive;
        }
        if (use_weights && detect > 0.0f) {
            p->history |= CHROMA_SURFACE_DETECT;
            p->weight *= detect;
            return BREAK;
        }
        if (uniform_sample < absorb) {
            p->history |= CHROMA_SURFACE_ABSORB;
            return BREAK;
        } else if (uniform_sample < absorb + detect) {
            p->history |= CHROMA_SURFACE_DETECT;
            return BREAK;
        } else if (uniform_sample < absorb + detect + reflect_diffuse)
            return propagate_at_diffuse_reflector(p, s, rng);
        else if (uniform_sample < absorb + detect + reflect_diffuse + reflect_specular)
            return propagate_at_specular_reflector(p, s);
        else
            return PASS;
    }
}

/* ---- propagate.cu:217-319, one photon ------------------------------------------ */
static void propagate_one(Geo *g, const chroma_photon_arrays *a, size_t photon_id, chroma_rng rng_desc,
                          int max_steps, int use_weights, int scatter_first, Counters *cnt)
{
    Photon p;
    p.position = mk3(a->pos[3 * photon_id], a->pos[3 * photon_id + 1], a->pos[3 * photon_id + 2]);
    p.direction = mk3(a->dir[3 * photon_id], a->dir[3 * photon_id + 1], a->dir[3 * photon_id + 2]);
    p.direction = div3s(p.direction, norm3(p.direction));
    p.polarization = mk3(a->pol[3 * photon_id], a->pol[3 * photon_id + 1], a->pol[3 * photon_id + 2]);
    p.polarization = div3s(p.polarization, norm3(p.polarization));
    p.wavelength = a->wavelengths[photon_id];
    p.time = a->t[photon_id];
    p.last_hit_triangle = a->last_hit_triangles[photon_id];
    p.history = a->flags[photon_id];
    p.weight = a->weights[photon_id];
    p.evidx = a->evidx[photon_id];

    if (p.history & CHROMA_TERMINAL_MASK) return;          /* propagate.cu:258-259 */

    cm_rng rng;
    cm_rng_init(&rng, rng_desc.seed, rng_desc.photon_id_base + photon_id, a->rng_counters[photon_id]);

    State s;
    memset(&s, 0, sizeof s);
    int steps = 0;
    while (steps < max_steps) {
        steps++;
        int command;
        if (cm_isnan(p.direction.x * p.direction.y * p.direction.z * p.position.x * p.position.y * p.position.z)) {
            p.history |= CHROMA_NO_HIT | CHROMA_NAN_ABORT;
            break;
        }
        cnt->photon_steps++;
        fill_state(&s, &p, g, cnt);
        if (p.last_hit_triangle == -1) break;

        command = propagate_to_boundary(&p, &s, &rng, g, use_weights, scatter_first);
        scatter_first = 0;
        if (command == BREAK) break;
        if (command == CONTINUE) continue;

        if (s.surface_index != -1) {
            command = propagate_at_surface(&p, &s, &rng, g, use_weights);
            if (command == BREAK) break;
            if (command == CONTINUE) continue;
        }
        propagate_at_boundary(&p, &s, &rng);
    }

    a->rng_counters[photon_id] = rng.counter;
    a->pos[3 * photon_id] = p.position.x; a->pos[3 * photon_id + 1] = p.position.y; a->pos[3 * photon_id + 2] = p.position.z;
    a->dir[3 * photon_id] = p.direction.x; a->dir[3 * photon_id + 1] = p.direction.y; a->dir[3 * photon_id + 2] = p.direction.z;
    a->pol[3 * photon_id] = p.polarization.x; a->pol[3 * photon_id + 1] = p.polarization.y; a->pol[3 * photon_id + 2] = p.polarization.z;
    a->wavelengths[photon_id] = p.wavelength;
    a->t[photon_id] = p.time;
    a->flags[photon_id] = p.history;
    a->last_hit_triangles[photon_id] = p.last_hit_triangle;
    a->weights[photon_id] = p.weight;
    a->evidx[photon_id] = p.evidx;
}

/* A launch is shared by the host threads in blocks of JOB_BLOCK queue entries taken from one atomic
 * cursor (photons differ a lot in length, so equal static shares leave threads idle); every thread counts
 * in a Counters of its OWN on its stack -- the Job structs are adjacent in memory, and counters bumped per
 * node visit inside them made 256 threads fight over cache lines -- and adds it to its Job once, at the end. */
#define JOB_BLOCK 1024
typedef struct {
    Geo *g; const chroma_photon_arrays *a; const uint32_t *ids; size_t n; size_t *cursor; chroma_rng rng;
    int max_steps, use_weights, scatter_first; Counters cnt;
} Job;

static void *job_main(void *arg)
{
    Job *j = (Job *)arg;
    Counters cnt; memset(&cnt, 0, sizeof cnt);
    for (;;) {
        size_t begin = __atomic_fetch_add(j->cursor, (size_t)JOB_BLOCK, __ATOMIC_RELAXED);
        if (begin >= j->n) break;
        size_t end = begin + JOB_BLOCK < j->n ? begin + JOB_BLOCK : j->n;
        for (size_t i = begin; i < end; i++)
            propagate_one(j->g, j->a, j->ids[i], j->rng, j->max_steps, j->use_weights, j->scatter_first, &cnt);
    }
    j->cnt = cnt;
    return NULL;
}

/* GPUPhotons.propagate (chroma/gpu/photon.py:193-259) for HOST photon arrays: the step loop with
 * the reference's launch policy -- one step per launch while at least `finish_threshold`
 * (nthreads_per_block*16*8 = 8192, photon.py:227) photons are alive and weights are off,
 * otherwise all remaining steps in one launch.  The policy matters at the bit level because
 * every launch re-normalises direction and polarisation on load (propagate.cu:248,250).
 * Survivors are re-queued as propagate.cu:315-318 does; `nthreads` host threads share a launch. */
int oracle_propagate(const chroma_geometry_desc *g, const chroma_photon_arrays *a, uint64_t nphotons,
                     chroma_rng rng, int32_t max_steps, int32_t use_weights, int32_t scatter_first,
                     int32_t nthreads, chroma_propagate_stats *stats)
{
    const uint64_t finish_threshold = 64 * 16 * 8;
    if (nthreads < 1) nthreads = 1;
    if (nphotons == 0 || max_steps <= 0) return 0;
    Job *jobs = (Job *)calloc((size_t)nthreads, sizeof(Job));
    pthread_t *th = (pthread_t *)calloc((size_t)nthreads, sizeof(pthread_t));
    uint32_t *alive = (uint32_t *)malloc(sizeof(uint32_t) * nphotons);
    if (!jobs || !th || !alive) { free(jobs); free(th); free(alive); return -1; }
    for (uint64_t i = 0; i < nphotons; i++) alive[i] = (uint32_t)i;
    uint64_t n = nphotons, launches = 0;
    Counters total; memset(&total, 0, sizeof total);
    int step = 0;
    while (step < max_steps) {
        int nsteps = (n < finish_threshold || use_weights) ? (max_steps - step) : 1;
        int nt = nthreads;
        if ((uint64_t)nt > (n + JOB_BLOCK - 1) / JOB_BLOCK) nt = (int)((n + JOB_BLOCK - 1) / JOB_BLOCK);
        size_t cursor = 0;
        for (int i = 0; i < nt; i++) {
            memset(&jobs[i], 0, sizeof(Job));
            jobs[i].g = g; jobs[i].a = a; jobs[i].ids = alive; jobs[i].rng = rng;
            jobs[i].n = (size_t)n; jobs[i].cursor = &cursor;
            jobs[i].max_steps = nsteps; jobs[i].use_weights = use_weights; jobs[i].scatter_first = scatter_first;
        }
        if (nt == 1) {
            job_main(&jobs[0]);
        } else {
            for (int i = 0; i < nt; i++) pthread_create(&th[i], NULL, job_main, &jobs[i]);
            for (int i = 0; i < nt; i++) pthread_join(th[i], NULL);
        }
        launches++;
        for (int i = 0; i < nt; i++) {
            total.photon_steps += jobs[i].cnt.photon_steps;
            total.nodes_visited += jobs[i].cnt.nodes_visited;
            total.triangles_tested += jobs[i].cnt.triangles_tested;
            total.stack_overflows += jobs[i].cnt.stack_overflows;
        }
        step += nsteps;
        scatter_first = 0;
        if (step < max_steps) {
            uint64_t m = 0;
            for (uint64_t i = 0; i < n; i++)
                if ((a->flags[alive[i]] & CHROMA_TERMINAL_MASK) == 0) alive[m++] = alive[i];
            n = m;
            if (n == 0) break;
        }
    }
    if (stats) {
        stats->photon_steps += total.photon_steps;
        stats->nodes_visited += total.nodes_visited;
        stats->triangles_tested += total.triangles_tested;
        stats->stack_overflows += total.stack_overflows;
        stats->launches += launches;
    }
    free(jobs); free(th); free(alive);
    return 0;
}

/* distance_to_mesh kernel (mesh.h:124-151) + the triangle id; misses leave distance untouched */
/* mesh.h:124-151 (last_hits == NULL), and intersect_mesh (mesh.h:42-118) with its last_hit_triangle argument */
int oracle_intersect_mesh(const chroma_geometry_desc *g, uint64_t n, const float *origin, const float *direction,
                          const int32_t *last_hits, float *distance, int32_t *triangle, chroma_propagate_stats *stats)
{
    Counters cnt; memset(&cnt, 0, sizeof cnt);
    for (uint64_t i = 0; i < n; i++) {
        f3 o = mk3(origin[3 * i], origin[3 * i + 1], origin[3 * i + 2]);
        f3 d = mk3(direction[3 * i], direction[3 * i + 1], direction[3 * i + 2]);
        d = div3s(d, norm3(d));
        float dist;
        int tri = intersect_mesh(o, d, g, &dist, last_hits ? last_hits[i] : -1, &cnt);
        if (tri != -1) distance[i] = dist;
        if (triangle) triangle[i] = tri;
    }
    if (stats) {
        stats->nodes_visited += cnt.nodes_visited;
        stats->triangles_tested += cnt.triangles_tested;
        stats->stack_overflows += cnt.stack_overflows;
    }
    return 0;
}

int oracle_distance_to_mesh(const chroma_geometry_desc *g, uint64_t n, const float *origin, const float *direction,
                            float *distance, int32_t *triangle, chroma_propagate_stats *stats)
{
    return oracle_intersect_mesh(g, n, origin, direction, NULL, distance, triangle, stats);
}

/* ---- render (chroma/cuda/render.cu) ------------------------------------------------------------------ */
/* sorting.h:64-87 */
static uint32_t render_searchsorted(uint32_t n, const float *arr, float x)
{
    uint32_t jl = 0, ju = n;
    int ascnd = (arr[n - 1] >= arr[0]);
    while (ju - jl > 1) {
        uint32_t jm = (ju + jl) >> 1;
        if ((x > arr[jm]) == ascnd) jl = jm; else ju = jm;
    }
    if ((x <= arr[0]) == ascnd) return 0;
    return ju;
}

/* intersect_node without a distance (mesh.h:16-34, min_distance = -1) */
static int box_hit(f3 noid, f3 inv_dir, const Node *node)
{
    float t;
    return intersect_box(noid, inv_dir, node->lower, node->upper, &t);
}

/* render.cu:37-181 for host arrays: dx [n][alpha_depth], dxlen [n], color [n][alpha_depth][4] in and out */
int oracle_render(const chroma_geometry_desc *g, uint64_t n, const float *origins, const float *directions, uint32_t alpha_depth,
                  uint32_t bg_color, uint32_t *pixels, float *dx_all, uint32_t *dxlen, float *color_all)
{
    if (!g->colors || alpha_depth < 1) return -1;
    for (uint64_t id = 0; id < n; id++) {
        f3 origin = mk3(origins[3 * id], origins[3 * id + 1], origins[3 * id + 2]);
        f3 direction = mk3(directions[3 * id], directions[3 * id + 1], directions[3 * id + 2]);
        uint32_t cnt = dxlen[id];
        Node root = get_node(g, 0);
        f3 noid = div33(neg3(origin), direction);
        f3 inv_dir = sdiv3(1.0f, direction);
        if (cnt < 1 && !box_hit(noid, inv_dir, &root)) { pixels[id] = bg_color; continue; }
        uint32_t child_ptr_stack[STACK_SIZE], nchild_ptr_stack[STACK_SIZE];
        child_ptr_stack[0] = root.child;
        nchild_ptr_stack[0] = root.nchild;
        int curr = 0;
        float *dx = dx_all + id * alpha_depth;
        float *color_a = color_all + 4 * id * alpha_depth;
        while (curr >= 0) {
            uint32_t first_child = child_ptr_stack[curr], nchild = nchild_ptr_stack[curr];
            curr--;
            for (uint32_t i = first_child; i < first_child + nchild; i++) {
                Node node = get_node(g, i);
                if (!box_hit(noid, inv_dir, &node)) continue;
                if (node.nchild != 0) {
                    if (curr + 1 >= STACK_SIZE) return -2;
                    curr++;
                    child_ptr_stack[curr] = node.child;
                    nchild_ptr_stack[curr] = node.nchild;
                    continue;
                }
                f3 v0, v1, v2;
                float distance;
                get_triangle(g, node.child, &v0, &v1, &v2);
                if (!intersect_triangle(origin, direction, v0, v1, v2, &distance)) continue;
                /* get_color (render.cu:11-32) */
                f3 normal = normalize3(cross3(sub3(v1, v0), sub3(v2, v1)));
                float cos_theta = dot3(normal, neg3(direction));
                if (cos_theta < 0.0f) cos_theta = -cos_theta;
                uint32_t rgba = g->colors[node.child];
                float col[4] = {(float)(0xffu & (rgba >> 16)) * cos_theta, (float)(0xffu & (rgba >> 8)) * cos_theta,
                                (float)(0xffu & rgba) * cos_theta, (float)(255u - (0xffu & (rgba >> 24))) / 255.0f};
                if (cnt < 1) {
                    dx[0] = distance;
                    memcpy(color_a, col, sizeof col);
                } else {
                    uint32_t j = render_searchsorted(cnt, dx, distance);
                    if (j <= alpha_depth - 1u) {
                        for (uint32_t k = alpha_depth - 1u; k > j; k--) { dx[k] = dx[k - 1]; memcpy(color_a + 4 * k, color_a + 4 * (k - 1), 16); }
                        dx[j] = distance;
                        memcpy(color_a + 4 * j, col, sizeof col);
                    }
                }
                if (cnt < alpha_depth) cnt++;
            }
        }
        if (cnt < 1) { pixels[id] = bg_color; continue; }
        dxlen[id] = cnt;
        float scale = 1.0f, fr = 0.0f, fg = 0.0f, fb = 0.0f;
        for (uint32_t i = 0; i < cnt; i++) {
            float alpha = color_a[4 * i + 3];
            fr += scale * color_a[4 * i] * alpha;
            fg += scale * color_a[4 * i + 1] * alpha;
            fb += scale * color_a[4 * i + 2] * alpha;
            scale *= (1.0f - alpha);
        }
        float alpha = (float)((double)(float)((bg_color & 0xFF000000u) >> 24) / 255.0);
        fr += scale * (float)((bg_color & 0xFF0000u) >> 16) * alpha;
        fg += scale * (float)((bg_color & 0xFF00u) >> 8) * alpha;
        fb += scale * (float)(bg_color & 0xFFu) * alpha;
        scale *= (1.0f - alpha);
        uint32_t av = (cnt < alpha_depth) ? cm_f2u32(cm_floorf(255.0f * (1.0f - scale))) : 255u;
        uint32_t red = cm_f2u32(cm_floorf(fr / (1.0f - scale)));
        uint32_t green = cm_f2u32(cm_floorf(fg / (1.0f - scale)));
        uint32_t blue = cm_f2u32(cm_floorf(fb / (1.0f - scale)));
        pixels[id] = av << 24 | red << 16 | green << 8 | blue;
    }
    return 0;
}

/* Test order of the reference walk: the loop of intersect_mesh (mesh.h:58-110) with EVERY box test
 * succeeding and no triangle hit (so nothing is pruned).  order_out[k] = k-th triangle tested;
 * returns the number of tests, or -1 when the explicit stack (mesh.h: 1000 entries) would overflow.
 * Pins the `rank` array the engine's wide walk uses to break exact distance ties. */
int64_t oracle_reference_test_order(const uint32_t *nodes, uint64_t nnodes, uint32_t *order_out, uint64_t capacity)
{
    enum { STACK = 1000 };
    uint32_t child_ptr_stack[STACK], nchild_ptr_stack[STACK];
    int64_t ntests = 0;
    if (nnodes == 0) return 0;
    uint32_t root_w = nodes[3];
    child_ptr_stack[0] = root_w & ~CHROMA_NCHILD_MASK;
    nchild_ptr_stack[0] = root_w >> CHROMA_CHILD_BITS;
    int curr = 0;
    while (curr >= 0) {
        uint32_t first_child = child_ptr_stack[curr];
        uint32_t nchild = nchild_ptr_stack[curr];
        curr--;
        for (uint32_t i = first_child; i < first_child + nchild; i++) {
            uint32_t w = nodes[4 * (uint64_t)i + 3];
            uint32_t k = w >> CHROMA_CHILD_BITS, c = w & ~CHROMA_NCHILD_MASK;
            if (k == 0) {
                if ((uint64_t)ntests < capacity) order_out[ntests] = c;
                ntests++;
            } else {
                curr++;
                if (curr >= STACK) return -1;
                child_ptr_stack[curr] = c;
                nchild_ptr_stack[curr] = k;
            }
        }
    }
    return ntests;
}

/* ---- DAQ: run_daq (chroma/cuda/daq.cu:35-86) with interp (interpolate.h:32-57) -------------- */
static float interp_table(float x, int n, const float *xp, const float *fp)
{
    int lower = 0;
    int upper = n - 1;
    if (x <= xp[lower]) return fp[lower];
    if (x >= xp[upper]) return fp[upper];
    while (lower < upper - 1) {
        int half = (lower + upper) / 2;
        if (x < xp[half]) upper = half; else lower = half;
    }
    float df = fp[upper] - fp[lower];
    float dx = xp[upper] - xp[lower];
    return fp[lower] + df * (x - xp[lower]) / dx;
}

/* Channel arrays are HOST arrays of nchannels entries, accumulated into (min / add / or). */
int oracle_run_daq(const chroma_geometry_desc *g, const chroma_daq_tables *tab, int32_t first_photon, int32_t nphotons,
                   uint32_t detection_state, const chroma_photon_arrays *a, chroma_rng rng_desc, uint32_t acquisition,
                   float global_weight, uint32_t *earliest_time_int, uint32_t *channel_q_int, uint32_t *channel_histories)
{
    for (int id = 0; id < nphotons; id++) {
        int photon_id = id + first_photon;
        int triangle_id = a->last_hit_triangles[photon_id];
        if (triangle_id <= -1) continue;
        uint32_t history = a->flags[photon_id];
        int channel_index = g->solid_id_to_channel_index[g->solid_id_map[triangle_id]];
        if (channel_index < 0 || !(history & detection_state)) continue;
        cm_rng rng;
        cm_rng_init(&rng, rng_desc.seed, rng_desc.photon_id_base + (uint64_t)photon_id, 0);
        rng.stream = 1u + acquisition;
        float weight = a->weights[photon_id] * global_weight;
        if (rng_u(&rng) < weight) {
            float time = a->t[photon_id] + interp_table(rng_u(&rng), tab->time_cdf_len, tab->d_time_cdf_y, tab->d_time_cdf_x);
            float charge = interp_table(rng_u(&rng), tab->charge_cdf_len, tab->d_charge_cdf_y, tab->d_charge_cdf_x);
            uint32_t charge_int = (uint32_t)cm_roundf(charge / tab->charge_unit);
            uint32_t time_int = cm_f2u(time);
            if (time_int < earliest_time_int[channel_index]) earliest_time_int[channel_index] = time_int;
            channel_q_int[channel_index] += charge_int;
            channel_histories[channel_index] |= history;
        }
    }
    return 0;
}

/* run_daq_many (chroma/cuda/daq.cu:88-150): ndaq acquisitions side by side, copy i in channels
 * [i * stride, (i + 1) * stride), each with a unit normal jitter on the hit time.  Copy i of a photon
 * draws from words 8 i ... of the photon's DAQ stream (mirrors chroma_daq_acquire_many). */
int oracle_run_daq_many(const chroma_geometry_desc *g, const chroma_daq_tables *tab, int32_t first_photon, int32_t nphotons,
                        uint32_t detection_state, const chroma_photon_arrays *a, chroma_rng rng_desc, uint32_t acquisition,
                        float global_weight, int32_t ndaq, int32_t channel_stride,
                        uint32_t *earliest_time_int, uint32_t *channel_q_int, uint32_t *channel_histories)
{
    for (int id = 0; id < nphotons; id++) {
        int photon_id = id + first_photon;
        int triangle_id = a->last_hit_triangles[photon_id];
        if (triangle_id <= -1) continue;
        uint32_t history = a->flags[photon_id];
        int channel_index = g->solid_id_to_channel_index[g->solid_id_map[triangle_id]];
        if (channel_index < 0 || !(history & detection_state)) continue;
        float weight = a->weights[photon_id] * global_weight;
        for (int copy = 0; copy < ndaq; copy++) {
            cm_rng rng;
            cm_rng_init(&rng, rng_desc.seed, rng_desc.photon_id_base + (uint64_t)photon_id, 8u * (uint32_t)copy);
            rng.stream = 1u + acquisition;
            int channel_offset = channel_index + copy * channel_stride;
            if (rng_u(&rng) < weight) {
                float jitter = cm_rng_normal(&rng);
                float time = a->t[photon_id] + jitter + interp_table(rng_u(&rng), tab->time_cdf_len, tab->d_time_cdf_y, tab->d_time_cdf_x);
                float charge = interp_table(rng_u(&rng), tab->charge_cdf_len, tab->d_charge_cdf_y, tab->d_charge_cdf_x);
                uint32_t charge_int = (uint32_t)cm_roundf(charge / tab->charge_unit);
                uint32_t time_int = cm_f2u(time);
                if (time_int < earliest_time_int[channel_offset]) earliest_time_int[channel_offset] = time_int;
                channel_q_int[channel_offset] += charge_int;
                channel_histories[channel_offset] |= history;
            }
        }
    }
    return 0;
}

/* Isotropic photon bomb, the benchmark source of chroma/benchmark.py:77-83 with the
 * formulas of chroma/sample.py:16-30 in single precision; photon i is drawn from the Philox
 * stream (seed, 0xB0B0000000000000 + id_base + i): dir = uniform_sphere (2 draws), an auxiliary
 * isotropic vector (2 draws), pol = normalize(cross(aux, dir)), then the wavelength if a range
 * is given.  Mirrors chroma_generate_bomb of the engine. */
int oracle_generate_bomb(const chroma_photon_arrays *a, uint64_t n, uint64_t seed, uint64_t id_base,
                         const float pos[3], float wl_lo, float wl_hi)
{
    for (uint64_t i = 0; i < n; i++) {
        cm_rng rng;
        cm_rng_init(&rng, seed, 0xB0B0000000000000ull + id_base + i, 0);
        f3 dir = uniform_sphere(&rng);
        f3 aux = uniform_sphere(&rng);
        f3 pol = cross3(aux, dir);
        pol = div3s(pol, norm3(pol));
        float wl = (wl_hi > wl_lo) ? uniform(&rng, wl_lo, wl_hi) : wl_lo;
        a->pos[3 * i] = pos[0]; a->pos[3 * i + 1] = pos[1]; a->pos[3 * i + 2] = pos[2];
        a->dir[3 * i] = dir.x; a->dir[3 * i + 1] = dir.y; a->dir[3 * i + 2] = dir.z;
        a->pol[3 * i] = pol.x; a->pol[3 * i + 1] = pol.y; a->pol[3 * i + 2] = pol.z;
        a->wavelengths[i] = wl; a->t[i] = 0.0f; a->flags[i] = 0u; a->last_hit_triangles[i] = -1;
        a->weights[i] = 1.0f; a->evidx[i] = 0u; a->rng_counters[i] = 0u;
    }
    return 0;
}

/* ---- probes used by tests/ to pin the numeric contract ---------------------------- */
/* fn: 0 log 1 exp 2 sin 3 cos 4 tan 5 asin 6 acos 7 atan2(x,y) 8 sqrt 9 uniform(u32 bits) */
int oracle_math(int fn, uint64_t n, const float *x, const float *y, float *out)
{
    for (uint64_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = M_LOGF(x[i]); break;
        case 1: out[i] = M_EXPF(x[i]); break;
        case 2: out[i] = M_SINF(x[i]); break;
        case 3: out[i] = M_COSF(x[i]); break;
        case 4: out[i] = M_TANF(x[i]); break;
        case 5: out[i] = M_ASINF(x[i]); break;
        case 6: out[i] = M_ACOSF(x[i]); break;
        case 7: out[i] = M_ATAN2F(x[i], y[i]); break;
        case 8: out[i] = cm_sqrtf(x[i]); break;
        case 9: out[i] = cm_u32_to_uniform(cm_f2u(x[i])); break;
        default: return -1;
        }
    }
    return 0;
}

/* Single functions of the path, one call per element, for the pins on the reference's own headers
 * (oracle/ref_headers_driver.hip) and on the engine's device code (chroma_probe):
 * fn 0 interp_property (geometry.h:64-75)   x[n], tab_f[ntab], grid (start, step)
 * fn 1 interp_idx (interpolate.h:5-29)      x[n], tab_x[ntab]
 * fn 2 interp (interpolate.h:32-57)         x[n], tab_x[ntab], tab_f[ntab]   (interp_table of the DAQ)
 * fn 3 rotate (rotate.h:22-28)              x[7 n] = a.xyz, phi, axis.xyz -> out[5 n] = r.xyz, cos(phi), sin(phi) */
static float interp_table(float x, int n, const float *xp, const float *fp);
int oracle_probe(int fn, uint64_t n, const float *x, const float *tab_x, const float *tab_f, uint32_t ntab,
                 float start, float step, float *out)
{
    chroma_geometry_desc g;
    memset(&g, 0, sizeof g);
    g.wavelength_n = ntab; g.wavelength_start = start; g.wavelength_step = step;
    for (uint64_t i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = interp_property(&g, x[i], tab_f); break;
        case 1: out[i] = interp_idx(x[i], (int)ntab, tab_x); break;
        case 2: out[i] = interp_table(x[i], (int)ntab, tab_x, tab_f); break;
        case 3: {
            const float *p = x + 7 * i;
            f3 r = rotate3(mk3(p[0], p[1], p[2]), p[3], mk3(p[4], p[5], p[6]));
            float *o = out + 5 * i;
            o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = M_COSF(p[3]); o[4] = M_SINF(p[3]);
            break;
        }
        default: return -1;
        }
    }
    return 0;
}

void oracle_philox(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    cm_philox4x32_10(ctr[0], ctr[1], ctr[2], ctr[3], key[0], key[1], out);
}

/* the first `n` uniforms of one photon's stream */
void oracle_uniform_stream(uint64_t seed, uint64_t photon_id, uint32_t start, uint32_t n, float *out)
{
    cm_rng r;
    cm_rng_init(&r, seed, photon_id, start);
    for (uint32_t i = 0; i < n; i++) out[i] = cm_rng_uniform(&r);
}

/* single physics routines exposed for branch-level tests: runs ONE call of the named
 * routine on photon 0 of `a` with an explicit State.  which: 0 rayleigh_scatter,
 * 1 propagate_at_boundary, 2 specular, 3 diffuse, 4 propagate_at_surface, 5 propagate_to_boundary */
int oracle_single(const chroma_geometry_desc *g, const chroma_photon_arrays *a, chroma_rng rng_desc, int which,
                  const float normal[3], float n1, float n2, float absorption_length, float scattering_length,
                  int material1, int surface_index, float distance_to_boundary, int use_weights, int scatter_first)
{
    Photon p;
    p.position = mk3(a->pos[0], a->pos[1], a->pos[2]);
    p.direction = mk3(a->dir[0], a->dir[1], a->dir[2]);
    p.polarization = mk3(a->pol[0], a->pol[1], a->pol[2]);
    p.wavelength = a->wavelengths[0]; p.time = a->t[0]; p.last_hit_triangle = a->last_hit_triangles[0];
    p.history = a->flags[0]; p.weight = a->weights[0]; p.evidx = a->evidx[0];
    State s; memset(&s, 0, sizeof s);
    s.surface_normal = mk3(normal[0], normal[1], normal[2]);
    s.refractive_index1 = n1; s.refractive_index2 = n2;
    s.absorption_length = absorption_length; s.scattering_length = scattering_length;
    s.material1 = material1; s.surface_index = surface_index; s.distance_to_boundary = distance_to_boundary;
    cm_rng rng;
    cm_rng_init(&rng, rng_desc.seed, rng_desc.photon_id_base, a->rng_counters[0]);
    int command = -1;
    switch (which) {
    case 0: rayleigh_scatter(&p, &rng); break;
    case 1: propagate_at_boundary(&p, &s, &rng); break;
    case 2: command = propagate_at_specular_reflector(&p, &s); break;
    case 3: command = propagate_at_diffuse_reflector(&p, &s, &rng); break;
    case 4: command = propagate_at_surface(&p, &s, &rng, g, use_weights); break;
    case 5: command = propagate_to_boundary(&p, &s, &rng, g, use_weights, scatter_first); break;
    default: return -100;
    }
    a->rng_counters[0] = rng.counter;
    a->pos[0] = p.position.x; a->pos[1] = p.position.y; a->pos[2] = p.position.z;
    a->dir[0] = p.direction.x; a->dir[1] = p.direction.y; a->dir[2] = p.direction.z;
    a->pol[0] = p.polarization.x; a->pol[1] = p.polarization.y; a->pol[2] = p.polarization.z;
    a->wavelengths[0] = p.wavelength; a->t[0] = p.time; a->flags[0] = p.history;
    a->last_hit_triangles[0] = p.last_hit_triangle; a->weights[0] = p.weight;
    return command;
}

const char *oracle_variant(void)
{
#ifdef ORACLE_LIBM
    return "libm";
#else
    return "contract";
#endif
}
