// kernels_daq_render.h -- DAQ (chroma/cuda/daq.cu), distance_to_mesh, render (chroma/cuda/render.cu), point transforms, the bomb generator, the probe kernel.
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

// ---- DAQ (chroma/cuda/daq.cu) ------------------------------------------------------------------
// interp (interpolate.h:32-57) as used by sample_cdf(rng, n, cdf_x, cdf_y) (random.h:26-31)
__device__ inline float interp_table(float x, int n, const float *xp, const float *fp)
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

__global__ void k_daq_reset(float maxtime, uint32_t n, uint32_t *time_ints, uint32_t *q_ints, uint32_t *histories)
{
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) {
        time_ints[id] = __float_as_uint(maxtime);
        q_ints[id] = 0u;
        histories[id] = 0u;
    }
}

// run_daq (daq.cu:35-86)
__global__ void k_run_daq(GeoView g, chroma_daq_tables tab, int first_photon, int nphotons, uint32_t detection_state,
                          const float *photon_times, const uint32_t *photon_histories, const int32_t *last_hit_triangles,
                          const float *weights, uint64_t seed, uint64_t id_base, uint32_t acquisition, float global_weight,
                          uint32_t *earliest_time_int, uint32_t *channel_q_int, uint32_t *channel_histories)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nphotons) return;
    int photon_id = id + first_photon;
    int triangle_id = last_hit_triangles[photon_id];
    if (triangle_id <= -1) return;
    uint32_t history = photon_histories[photon_id];
    int channel_index = g.solid_id_to_channel_index[g.solid_id_map[triangle_id]];
    if (channel_index < 0 || !(history & detection_state)) return;
    cm_rng rng;
    cm_rng_init(&rng, seed, id_base + (uint64_t)photon_id, 0);
    rng.stream = 1u + acquisition;
    float weight = weights[photon_id] * global_weight;
    if (cm_rng_uniform(&rng) < weight) {
        float time = photon_times[photon_id] + interp_table(cm_rng_uniform(&rng), tab.time_cdf_len, tab.d_time_cdf_y, tab.d_time_cdf_x);
        float charge = interp_table(cm_rng_uniform(&rng), tab.charge_cdf_len, tab.d_charge_cdf_y, tab.d_charge_cdf_x);
        uint32_t charge_int = (uint32_t)cm_roundf(charge / tab.charge_unit);
        atomicMin(earliest_time_int + channel_index, __float_as_uint(time));
        atomicAdd(channel_q_int + channel_index, charge_int);
        atomicOr(channel_histories + channel_index, history);
    }
}

// run_daq_many (daq.cu:88-150): ndaq independent acquisitions of the same photons side by side, copy i
// in channels [i * stride, (i + 1) * stride); a copy adds a unit normal jitter to the hit time.  The
// reference gives a photon a block and its copies the block's threads; here a thread is one (photon,
// copy) pair and copy i draws from words 8 i ... of the photon's DAQ stream, so copies are independent
// and the result does not depend on the launch shape.
__global__ void k_run_daq_many(GeoView g, chroma_daq_tables tab, int first_photon, int nphotons, uint32_t detection_state,
                               const float *photon_times, const uint32_t *photon_histories, const int32_t *last_hit_triangles,
                               const float *weights, uint64_t seed, uint64_t id_base, uint32_t acquisition, float global_weight,
                               int ndaq, int channel_stride,
                               uint32_t *earliest_time_int, uint32_t *channel_q_int, uint32_t *channel_histories)
{
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)nphotons * ndaq) return;
    int photon_id = (int)(id / ndaq) + first_photon, copy = (int)(id % ndaq);
    int triangle_id = last_hit_triangles[photon_id];
    if (triangle_id <= -1) return;
    uint32_t history = photon_histories[photon_id];
    int channel_index = g.solid_id_to_channel_index[g.solid_id_map[triangle_id]];
    if (channel_index < 0 || !(history & detection_state)) return;
    cm_rng rng;
    cm_rng_init(&rng, seed, id_base + (uint64_t)photon_id, 8u * (uint32_t)copy);
    rng.stream = 1u + acquisition;
    float weight = weights[photon_id] * global_weight;
    int channel_offset = channel_index + copy * channel_stride;
    if (cm_rng_uniform(&rng) < weight) {
        float jitter = cm_rng_normal(&rng);
        float time = photon_times[photon_id] + jitter +
                     interp_table(cm_rng_uniform(&rng), tab.time_cdf_len, tab.d_time_cdf_y, tab.d_time_cdf_x);
        float charge = interp_table(cm_rng_uniform(&rng), tab.charge_cdf_len, tab.d_charge_cdf_y, tab.d_charge_cdf_x);
        uint32_t charge_int = (uint32_t)cm_roundf(charge / tab.charge_unit);
        atomicMin(earliest_time_int + channel_offset, __float_as_uint(time));
        atomicAdd(channel_q_int + channel_offset, charge_int);
        atomicOr(channel_histories + channel_offset, history);
    }
}
__global__ void k_daq_convert(uint32_t n, float charge_unit, const uint32_t *time_ints, const uint32_t *q_ints, float *t_out, float *q_out)
{
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) {
        t_out[id] = __uint_as_float(time_ints[id]);
        q_out[id] = (float)q_ints[id] * charge_unit;
    }
}

// distance_to_mesh (chroma/cuda/mesh.h:124-151)
template <int LDS_N, bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) void
k_distance_to_mesh(GeoView g, int nthreads, const float *origin, const float *direction, const int32_t *last_hit_in,
                   float *distance_out, int32_t *triangle_out, DeviceCounters *counters)
{
    __shared__ uint32_t s_lds[TRAV_LDS_WORDS(LDS_N, PROP_BLOCK)];
    int id = blockIdx.x * PROP_BLOCK + threadIdx.x;
    LaneCounters cnt = {0, 0, 0, 0};
    bool on = id < nthreads;
    v3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
    if (on) {
        o = load3(origin, id);
        d = load3(direction, id);
        d = d / norm(d);
    }
    float dist;
    const int last_hit = (on && last_hit_in) ? last_hit_in[id] : -1;
    int tri = intersect_mesh<LDS_N, PROP_BLOCK, COUNT>(g, o, d, dist, last_hit, s_lds + threadIdx.x, cnt, on);
    if (on) {
        if (tri != -1) distance_out[id] = dist;
        if (triangle_out) triangle_out[id] = tri;
    }
    unsigned long long ov = wave_sum_u64(cnt.overflows);
    if (COUNT) {
        unsigned long long nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane_id() == 0) { atomicAdd(&counters->nodes_visited, nd); atomicAdd(&counters->triangles_tested, tr); }
    }
    if (lane_id() == 0 && ov) atomicAdd(&counters->stack_overflows, ov);
}

// ---- render (chroma/cuda/render.cu:37-181) ---------------------------------------------------------------
// One lane per ray: EVERY triangle along the ray (no nearest-hit pruning: intersect_node without a distance,
// render.cu:107), the `alpha_depth` nearest kept as a list sorted by distance -- an equal distance goes in FRONT of
// the ones already there (searchsorted/insert, sorting.h:64-98), so the order of discovery is part of the
// result -- then composited front to back over the background colour.  The walk is therefore the reference's
// own: its tree, its child order, its box arithmetic; the lists live in the caller's arrays (GPURays.dx /
// .color / .dxlen), which is what lets a second render continue the first (keep_last_render).
__device__ inline uint32_t render_searchsorted(uint32_t n, const float *arr, float x)       // sorting.h:64-87
{
    uint32_t jl = 0, ju = n;
    const bool ascnd = arr[n - 1] >= arr[0];
    while (ju - jl > 1) {
        const uint32_t jm = (ju + jl) >> 1;
        if ((x > arr[jm]) == ascnd) jl = jm; else ju = jm;
    }
    return ((x <= arr[0]) == ascnd) ? 0u : ju;
}

template <int LDS_N>
__global__ __launch_bounds__(PROP_BLOCK) void
k_render(GeoView g, const uint32_t *colors, int nthreads, const float *origin_in, const float *direction_in, uint32_t alpha_depth,
         uint32_t *pixels, float *dx_all, uint32_t *dxlen, float4 *color_all, uint32_t bg_color, DeviceCounters *counters)
{
    __shared__ uint32_t s_lds[LDS_N * PROP_BLOCK];
    const int id = blockIdx.x * PROP_BLOCK + threadIdx.x;
    if (id >= nthreads) return;                    // (lanes are independent: no wave-wide votes below)
    const v3 origin = load3(origin_in, id), direction = load3(direction_in, id);      // as given: NOT normalised (render.cu:57-58)
    uint32_t n = dxlen[id];
    const v3 noid = (-origin) / direction;
    const v3 inv_dir = 1.0f / direction;
    const v3 wo = mk3(g.world_origin[0], g.world_origin[1], g.world_origin[2]);
    const float ws = g.world_scale;
#define R_LO(nd) mk3(wo.x + (float)((nd).x & 0xFFFFu) * ws, wo.y + (float)((nd).y & 0xFFFFu) * ws, wo.z + (float)((nd).z & 0xFFFFu) * ws)
#define R_HI(nd) mk3(wo.x + (float)((nd).x >> 16) * ws, wo.y + (float)((nd).y >> 16) * ws, wo.z + (float)((nd).z >> 16) * ws)
    const uint4 root = g.nodes[0];
    if (n < 1 && box_tmin(origin, noid, inv_dir, R_LO(root), R_HI(root), ws) < 0.0f) {
        pixels[id] = bg_color;
        return;
    }
    TravStack<LDS_N, PROP_BLOCK> stack;
    stack.lds = s_lds + threadIdx.x;
    int sp = 0;
    bool overflow = false;
    stack.put(sp++, root.w);
    float *dx = dx_all + (size_t)id * alpha_depth;
    float4 *color_a = color_all + (size_t)id * alpha_depth;
    while (sp > 0 && !overflow) {
        const uint32_t w = stack.get(--sp);
        const uint32_t first = w & ~CHROMA_NCHILD_MASK, nchild = w >> CHROMA_CHILD_BITS;
        for (uint32_t i = first; i < first + nchild; i++) {
            const uint4 nd = g.nodes[i];
            if (box_tmin(origin, noid, inv_dir, R_LO(nd), R_HI(nd), ws) < 0.0f) continue;
            const uint32_t child = nd.w & ~CHROMA_NCHILD_MASK;
            if ((nd.w >> CHROMA_CHILD_BITS) != 0) {
                if (sp >= LDS_N + STACK_SCRATCH) { overflow = true; break; }      // cannot happen when the host check passed
                stack.put(sp++, nd.w);
                continue;
            }
            const float4 *t = g.tri + TRI_STRIDE * (size_t)child;                  // leaf: the triangle record (device order)
            const float4 a = t[0], b = t[1], c = t[2];
            const v3 v0 = mk3(a.x, a.y, a.z), v1 = mk3(b.x, b.y, b.z), v2 = mk3(c.x, c.y, c.z);
            float distance;
            if (!intersect_triangle(origin, direction, v0, v1, v2, distance)) continue;
            // get_color (render.cu:11-32)
            const v3 normal = normalize(cross(v1 - v0, v2 - v1));
            float cos_theta = dot(normal, -direction);
            if (cos_theta < 0.0f) cos_theta = -cos_theta;
            const uint32_t rgba = colors[__float_as_uint(b.w)];
            const float4 color = make_float4((float)(0xffu & (rgba >> 16)) * cos_theta, (float)(0xffu & (rgba >> 8)) * cos_theta,
                                             (float)(0xffu & rgba) * cos_theta, (float)(255u - (0xffu & (rgba >> 24))) / 255.0f);
            if (n < 1) {
                dx[0] = distance;
                color_a[0] = color;
            } else {
                const uint32_t j = render_searchsorted(n, dx, distance);
                if (j <= alpha_depth - 1u) {
                    for (uint32_t k = alpha_depth - 1u; k > j; k--) { dx[k] = dx[k - 1]; color_a[k] = color_a[k - 1]; }     // sorting.h:89-98
                    dx[j] = distance;
                    color_a[j] = color;
                }
            }
            if (n < alpha_depth) n++;
        }
    }
#undef R_LO
#undef R_HI
    if (overflow) atomicAdd(&counters->stack_overflows, 1ull);
    if (n < 1) {
        pixels[id] = bg_color;
        return;
    }
    dxlen[id] = n;
    float scale = 1.0f, fr = 0.0f, fg = 0.0f, fb = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        const float4 ci = color_a[i];
        const float alpha = ci.w;
        fr += scale * ci.x * alpha;
        fg += scale * ci.y * alpha;
        fb += scale * ci.z * alpha;
        scale *= (1.0f - alpha);
    }
    // (the reference divides by the double literal 255.0 here, render.cu:163)
    const float alpha = (float)((double)(float)((bg_color & 0xFF000000u) >> 24) / 255.0);
    fr += scale * (float)((bg_color & 0xFF0000u) >> 16) * alpha;
    fg += scale * (float)((bg_color & 0xFF00u) >> 8) * alpha;
    fb += scale * (float)(bg_color & 0xFFu) * alpha;
    scale *= (1.0f - alpha);
    const uint32_t av = (n < alpha_depth) ? cm_f2u32(cm_floorf(255.0f * (1.0f - scale))) : 255u;
    const uint32_t red = cm_f2u32(cm_floorf(fr / (1.0f - scale)));
    const uint32_t green = cm_f2u32(cm_floorf(fg / (1.0f - scale)));
    const uint32_t blue = cm_f2u32(cm_floorf(fb / (1.0f - scale)));
    pixels[id] = av << 24 | red << 16 | green << 8 | blue;
}

// chroma/cuda/mesh.h:153-166 color_solids: the triangles of the solids marked in solid_hit take their solid's colour
// (an id beyond the caller's arrays is left alone instead of read)
__global__ void k_color_solids(int first_triangle, int nthreads, const uint32_t *solid_id_map, const uint8_t *solid_hit,
                               const uint32_t *solid_colors, uint32_t nsolids, uint32_t *colors)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nthreads) return;
    const int triangle_id = first_triangle + id;
    const uint32_t solid_id = solid_id_map[triangle_id];
    if (solid_id < nsolids && solid_hit[solid_id]) colors[triangle_id] = solid_colors[solid_id];
}

// chroma/cuda/transform.cu: translate / rotate / rotate_around_point of a point array
__global__ void k_rays_transform(int n, float *a, int mode, float phi, float ax, float ay, float az, float px, float py, float pz)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    v3 p = load3(a, id);
    const v3 axis = mk3(ax, ay, az), point = mk3(px, py, pz);
    if (mode == 0) p = p + point;                                        // translate by `point`
    else if (mode == 1) p = rotate(p, phi, axis);
    else { p = p - point; p = rotate(p, phi, axis); p = p + point; }
    store3(a, id, p);
}

// isotropic photon bomb (chroma/benchmark.py:77-83 with chroma/sample.py:16-30's formulas)
__global__ void k_generate_bomb(PhotonView pv, uint64_t n, uint64_t seed, uint64_t id_base, float px, float py, float pz,
                                float wl_lo, float wl_hi)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cm_rng rng;
    cm_rng_init(&rng, seed, 0xB0B0000000000000ull + id_base + i, 0);
    v3 dir = uniform_sphere(rng);
    v3 aux = uniform_sphere(rng);
    v3 pol = cross(aux, dir);
    pol = pol / norm(pol);
    float wl = (wl_hi > wl_lo) ? uniform(rng, wl_lo, wl_hi) : wl_lo;
    store3(pv.pos, i, mk3(px, py, pz));
    store3(pv.dir, i, dir);
    store3(pv.pol, i, pol);
    pv.wavelengths[i] = wl;
    pv.t[i] = 0.0f;
    pv.flags[i] = 0u;
    pv.last_hit_triangles[i] = -1;
    pv.weights[i] = 1.0f;
    pv.evidx[i] = 0u;
    pv.rng_counters[i] = 0u;
}

// chroma_probe: single device functions of the path, one call per element (tests pin them on the oracle
// and on the reference's own headers compiled for gfx950 by the test infrastructure)
__global__ void k_probe(int fn, uint64_t n, const float *x, const float *tab_x, const float *tab_f, uint32_t ntab,
                        float start, float step, float *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn == 0) {
        GeoView g;
        g.wavelength_n = ntab; g.wavelength_start = start; g.wavelength_step = step;
        out[i] = interp_property(g, x[i], tab_f);
    } else if (fn == 1) {
        out[i] = interp_idx(x[i], (int)ntab, tab_x);
    } else if (fn == 2) {
        out[i] = interp_table(x[i], (int)ntab, tab_x, tab_f);
    } else if (fn == 3) {
        const float *p = x + 7 * i;
        v3 r = rotate(mk3(p[0], p[1], p[2]), p[3], mk3(p[4], p[5], p[6]));
        float *o = out + 5 * i;
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = cm_cosf(p[3]); o[4] = cm_sinf(p[3]);
    } else {
        // the float3 algebra every device function of the path is written in (device_common.h; chroma/cuda/linalg.h)
        const float *p = x + 7 * i;
        const v3 a = mk3(p[0], p[1], p[2]), b = mk3(p[3], p[4], p[5]);
        const float c = p[6];
        float *o = out + 32 * i;
        const v3 r[8] = {-a, a + b, a - b, a * c, c * a, a / c, c / a, cross(a, b)};
        for (int k = 0; k < 8; k++) { o[3 * k] = r[k].x; o[3 * k + 1] = r[k].y; o[3 * k + 2] = r[k].z; }
        o[24] = dot(a, b);
        o[25] = norm(a);
        const v3 u = normalize(a), q = a / b;
        o[26] = u.x; o[27] = u.y; o[28] = u.z;
        o[29] = q.x; o[30] = q.y; o[31] = q.z;
    }
}
