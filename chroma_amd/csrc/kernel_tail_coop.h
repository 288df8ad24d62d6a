// kernel_tail_coop.h -- k_tail_coop: all remaining steps of the last (< 8192) photons in one launch, eight lanes per photon.
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

// ---- fused tail: all remaining steps of the last few photons, eight lanes per photon ---------------
// Once fewer than 64*16*8 photons are alive the reference finishes them in ONE launch
// (chroma/gpu/photon.py:227-230).  Per-step launches are a poor fit for that tail -- a few thousand
// rays, each a chain of ~30 dependent fetches, ~70 steps deep -- so it gets its own kernel: a group of
// 8 lanes owns one photon for all its remaining steps, casts its rays cooperatively (coop_cast: the
// walk of k_raycast_coop without the refill) and runs the physics redundantly in its 8 lanes (same
// inputs, same arithmetic, so the lanes stay identical; lane 0 of the group stores).  No launch or
// queue round trip between steps: the tail takes as long as its longest photon, not 70 launches.
// Rays the wide walk cannot take go through the general walk on the group's first lane.
template <bool COUNT>
__device__ inline int coop_cast(const GeoView &g, v3 origin, v3 direction, int last_hit, bool on, float &min_distance,
                                uint32_t *stack_n, float *stack_t, uint32_t *pending, uint2 *spill,
                                unsigned j, unsigned gshift, uint32_t below, LaneCounters &cnt)
{
    const float inf = cm_inff();
    int triangle_index = -1;
    uint32_t best_rank = 0;
    min_distance = -1.0f;
    uint32_t cur = WIDE_NONE;
    int sp = 0, npend = 0;
    bool active = false;
    RayFast rf;
    rf.a = rf.blo = rf.bhi = mk3(0.f, 0.f, 0.f);
    if (on) {
        v3 noid = (-origin) / direction;
        v3 inv_dir = 1.0f / direction;
        bool moderate = cm_fabsf(inv_dir.x) < 1e30f && cm_fabsf(inv_dir.y) < 1e30f && cm_fabsf(inv_dir.z) < 1e30f &&
                        cm_fabsf(noid.x) < 1e30f && cm_fabsf(noid.y) < 1e30f && cm_fabsf(noid.z) < 1e30f;
        if (!moderate) {
            triangle_index = HIT_RETRY;
        } else {
            rf = ray_fast(g, noid, inv_dir, ray_growth(g, origin));
            cur = 0;
            active = true;
        }
    }
    while (__any(active || npend > 0)) {
        // node phase
        while (__any(active) && !__any(npend >= 8)) {
            if (active && cur == WIDE_NONE) {
                while (sp > 0) {
                    sp--;
                    uint32_t n; float t;
                    if (sp < COOP_STACK) { n = stack_n[sp]; t = stack_t[sp]; }
                    else { uint2 se = spill[sp - COOP_STACK]; n = se.x; t = __uint_as_float(se.y); }
                    if (min_distance < 0.0f || !(t > min_distance)) { cur = n; break; }
                }
                if (cur == WIDE_NONE) active = false;
            }
            if (active) {
                const uint4 e = g.wnodes[8 * (size_t)cur + j];
                if (COUNT && j == 0) cnt.nodes += 8;
                const float t = box_tmin_fast(rf, e);
                const uint32_t w = e.w;
                const bool pass = (w != WIDE_NONE) && node_passes(t, min_distance);
                const bool isleaf = (w & 0x80000000u) != 0u;
                const bool leaf = pass && isleaf && (int)(w & 0x7FFFFFFFu) != last_hit;
                const bool inner = pass && !isleaf;
                const uint32_t gl = (uint32_t)(__ballot(leaf) >> gshift) & 0xFFu;
                const uint32_t gi = (uint32_t)(__ballot(inner) >> gshift) & 0xFFu;
                if (leaf) pending[npend + __popc(gl & below)] = w & 0x7FFFFFFFu;
                npend += __popc(gl);
                cur = WIDE_NONE;
                if (gi) {
                    const float tm = group8_min(inner ? t : inf);
                    const uint32_t gn = (uint32_t)(__ballot(inner && t == tm) >> gshift) & 0xFFu;
                    const uint32_t nj = (uint32_t)__ffs((int)gn) - 1u;
                    const uint32_t others = gi & ~(1u << nj);
                    if (inner && j != nj) {
                        int pos = sp + __popc(others & below);
                        if (pos < COOP_STACK) { stack_n[pos] = w; stack_t[pos] = t; }
                        else if (pos < COOP_STACK + COOP_SPILL) { spill[pos - COOP_STACK] = make_uint2(w, __float_as_uint(t)); if (COUNT) cnt.spills++; }
                    }
                    sp += __popc(others);
                    cur = (uint32_t)__shfl((int)w, (int)(gshift + nj));
                    if (sp > COOP_STACK + COOP_SPILL) {          // cannot happen: the host checked the tree's need
                        triangle_index = HIT_RETRY;
                        active = false; npend = 0; cur = WIDE_NONE; sp = 0;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // leaf phase
        while (__any(npend > 0)) {
            if (npend > 0) {
                const int take = min(npend, 8);
                bool hit = false;
                float distance = inf;
                uint32_t tri = 0, rank = 0xFFFFFFFFu;
                if ((int)j < take) {
                    tri = pending[j];
                    if (COUNT) cnt.tris++;
                    const float4 *tp = g.tri + TRI_STRIDE * (size_t)tri;
                    float4 a = tp[0], b = tp[1], c = tp[2];
                    hit = intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance);
                    rank = __float_as_uint(c.w);
                }
                const float dm = group8_min(hit ? distance : inf);
                if (dm < inf) {
                    const bool cand = hit && distance == dm;
                    const uint32_t rm = group8_min_u32(cand ? rank : 0xFFFFFFFFu);
                    const uint32_t gw = (uint32_t)(__ballot(cand && rank == rm) >> gshift) & 0xFFu;
                    const uint32_t wj = (uint32_t)__ffs((int)gw) - 1u;
                    const int wtri = __shfl((int)tri, (int)(gshift + wj));
                    if (triangle_index == -1 || dm < min_distance || (dm == min_distance && rm < best_rank)) {
                        triangle_index = wtri;
                        min_distance = dm;
                        best_rank = rm;
                    }
                }
                if (npend > 8) {
                    uint32_t mv = pending[j + 8];
                    if ((int)j + 8 < npend) pending[j] = mv;
                }
                npend -= take;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return triangle_index;
}

template <bool COUNT, bool LITERAL = false>
__global__ __launch_bounds__(PROP_BLOCK) void
k_tail_coop(GeoView g, PhotonView pv, const StepState *st, const float4 *work_in,
            uint64_t seed, uint64_t id_base, int max_steps, int use_weights, int scatter_first, uint2 *spill_base,
            DeviceCounters *counters, HitsOut h, uint32_t *words)
{
    // (`words` != NULL: a chroma_propagate_hits call whose k_finalize_hits runs BESIDE this kernel on another stream and leaves
    //  this kernel's photons alone (k_mark_tail): their abort bits and hits are reported from here)
    __shared__ uint32_t s_coop[8 * COOP_STRIDE];
    __shared__ uint32_t s_walk[TRAV_LDS_WORDS(STACK_LDS, PROP_BLOCK)];
    const int nthreads = (int)st->n, renorm = (int)st->renorm;
    if ((long long)blockIdx.x * 8 >= nthreads) return;
    const unsigned lane = lane_id();
    const unsigned j = lane & 7u, gshift = lane & ~7u, grp = lane >> 3;
    const uint32_t below = (1u << j) - 1u;
    uint32_t *stack_n = s_coop + grp * COOP_STRIDE;
    float *stack_t = (float *)(stack_n + COOP_STACK);
    uint32_t *pending = stack_n + 2 * COOP_STACK;
    uint2 *spill = spill_base + ((size_t)blockIdx.x * 8 + grp) * COOP_SPILL;
    LaneCounters cnt = {0, 0, 0, 0};

    const int id = (int)blockIdx.x * 8 + (int)grp;          // one photon per group
    bool loaded = false;
    uint32_t photon_id = 0;
    int last_hit_dev = -1;
    Photon p;
    cm_rng rng;
    State s;
    if (id < nthreads) {
        const float4 *w = work_in + 4 * (size_t)id;
        const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        photon_id = __float_as_uint(w3.w);
        p.position = mk3(w0.x, w0.y, w0.z);
        p.direction = mk3(w1.x, w1.y, w1.z);
        p.polarization = mk3(w2.x, w2.y, w2.z);
        if (renorm) {
            p.direction = p.direction / norm(p.direction);
            p.polarization = p.polarization / norm(p.polarization);
        }
        p.wavelength = w0.w;
        p.time = w1.w;
        p.weight = w2.w;
        p.history = __float_as_uint(w3.x);
        last_hit_dev = __float_as_int(w3.z);
        p.last_hit_triangle = last_hit_dev >= 0 ? (int)g.dev_to_tri[last_hit_dev] : -1;
        p.evidx = 0;
        loaded = true;                                      // (the working set holds live photons only)
        cm_rng_init(&rng, seed, id_base + photon_id, __float_as_uint(w3.y));
    }

    bool live = loaded;
    int steps = 0;
    while (__any(live && steps < max_steps)) {
        bool stepping = live && steps < max_steps;
        if (stepping) {
            steps++;
            if (cm_isnan(p.direction.x * p.direction.y * p.direction.z * p.position.x * p.position.y * p.position.z)) {
                p.history |= CHROMA_NO_HIT | CHROMA_NAN_ABORT;
                live = false;
                stepping = false;
            } else if (COUNT && j == 0) cnt.steps++;
        }
        float distance;
        // (LITERAL: the exact walk's cast -- chroma/cuda/mesh.h:42-118 itself, kernel_raycast_literal.h; its stack words live
        //  where the wide walk keeps its (node, distance) entries)
        int record = LITERAL ? literal_cast_group8<COUNT>(g, p.position, p.direction, last_hit_dev, stepping, distance, stack_n,
                                                          (uint32_t *)spill, j, cnt)
                             : coop_cast<COUNT>(g, p.position, p.direction, last_hit_dev, stepping, distance, stack_n, stack_t, pending,
                                                spill, j, gshift, below, cnt);
        // the reference's own walk for the rays the wide walk cannot take, and for winners that are not
        // regular (record_hit_is_regular): first lane of the group, then shared
        bool general = stepping && record == HIT_RETRY;
        if (!LITERAL && stepping && record >= 0) {
            const float4 *t = g.tri + TRI_STRIDE * (size_t)record;
            general = !record_hit_is_regular(g, t[0], t[1], t[2], p.position, p.direction, distance);
        }
        if (__any(general)) {
            float d2 = 0.0f;
            int r2 = intersect_mesh_dev<STACK_LDS, PROP_BLOCK, COUNT>(g, p.position, p.direction, d2, last_hit_dev,
                                                                       s_walk + threadIdx.x, cnt, general && j == 0);
            r2 = __shfl(r2, (int)gshift);
            d2 = __shfl(d2, (int)gshift);
            if (general) { record = r2; distance = d2; }
        }
        if (stepping) {
            apply_hit_dev(s, p, g, record, distance);
            if (record == -1) {
                live = false;
                last_hit_dev = -1;
            } else {
                live = step_after_hit(p, s, rng, g, use_weights != 0, scatter_first);
                scatter_first = 0;
                last_hit_dev = (p.last_hit_triangle < 0) ? -1 : record;
            }
        }
    }

    if (loaded && j == 0) {                                 // the call ends with this kernel: everything goes back
        pv.rng_counters[photon_id] = rng.counter;
        store3(pv.pos, photon_id, p.position);
        store3(pv.dir, photon_id, p.direction);
        store3(pv.pol, photon_id, p.polarization);
        pv.wavelengths[photon_id] = p.wavelength;
        pv.t[photon_id] = p.time;
        pv.flags[photon_id] = p.history;
        pv.last_hit_triangles[photon_id] = p.last_hit_triangle;
        pv.weights[photon_id] = p.weight;
        if (words) {
            if (p.history & CHROMA_NAN_ABORT) atomicOr(words + 2, CHROMA_NAN_ABORT);
            const int ch = h.want ? hit_channel(g, p.history, p.last_hit_triangle, h.detection_state) : -1;
            if (ch >= 0) {
                const uint32_t off = atomicAdd(words, 1u);
                if (h.channels && off < h.capacity) {
                    store3(h.dst.pos, off, p.position);
                    store3(h.dst.dir, off, p.direction);
                    store3(h.dst.pol, off, p.polarization);
                    h.dst.wavelengths[off] = p.wavelength;
                    h.dst.t[off] = p.time;
                    h.dst.flags[off] = p.history;
                    h.dst.last_hit_triangles[off] = p.last_hit_triangle;
                    h.dst.weights[off] = p.weight;
                    h.dst.evidx[off] = pv.evidx[photon_id];
                    h.channels[off] = ch;
                }
                if (h.hit_count) {
                    atomicAdd(&h.hit_count[ch], 1u);
                    if (h.earliest) atomicMin(&h.earliest[ch], __float_as_uint(p.time));
                }
            }
        }
    }

    if (COUNT) {
        unsigned long long sts = wave_sum_u64(cnt.steps), nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        unsigned long long sx = wave_sum_u64(cnt.spills);
        if (lane == 0) {
            atomicAdd(&counters->photon_steps, sts);
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
            if (sx) atomicAdd(&counters->stack_spills, sx);
        }
    }
}

// the photons the tail kernel is about to finish: their final-record slots are stamped, so that the k_finalize_hits that runs
// beside the tail kernel leaves them alone (slot 0 of the queue = tail index, as everywhere)
__global__ void k_mark_tail(const uint32_t *queue, float4 *final_rec, uint32_t tail_mark)
{
    const uint32_t n = queue[0] - 1u;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        final_rec[4 * (size_t)queue[1 + i] + 3] = make_float4(0.f, 0.f, 0.f, __uint_as_float(tail_mark));
}
