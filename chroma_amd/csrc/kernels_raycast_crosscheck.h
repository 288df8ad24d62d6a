// kernels_raycast_crosscheck.h -- the cross-check walks: the reference tree with postponed tests (k_raycast_persistent), the wide tree with one lane (k_raycast_wide) and eight lanes (k_raycast_coop) per ray; the eight-lane helpers also serve the tail kernel.
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

// ---- persistent ray cast with lane refill ---------------------------------------------------------
// One ray per lane, but a lane that finishes its ray takes the next one from the queue (one atomic
// per wave per refill), so the 64 lanes of a wave stay busy although their rays need very
// different numbers of node visits (measured: 26 % of the lanes active without refill).
// Traversal is the walk of intersect_mesh (same visit order, postponed triangle tests); the stack
// lives in LDS only.  The rare rays this kernel cannot take -- a component of 1/d that is not
// "moderate" (exactly or nearly axis-parallel) or a stack deeper than RAY_LDS_STACK -- are marked
// HIT_RETRY and done by k_raycast_retry with the general code.
#ifndef RAY_LDS_STACK
#define RAY_LDS_STACK 24
#endif
#ifndef RAY_REFILL_MIN
#define RAY_REFILL_MIN 12     // refill once this many lanes are idle
#endif

template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK, RAY_WAVES) void
k_raycast_persistent(GeoView g, const float4 *rays, int first_photon, StepState *st,
                     int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, DeviceCounters *counters)
{
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * PROP_BLOCK >= nthreads) return;
    uint32_t *work_counter = &st->work, *retry_counter = &st->retry;
    __shared__ uint32_t s_lds[(RAY_LDS_STACK + TRAV_PENDING) * PROP_BLOCK];
    uint32_t *stack = s_lds + threadIdx.x;
    uint32_t *pending = stack + RAY_LDS_STACK * PROP_BLOCK;
    const unsigned lane = lane_id();
    LaneCounters cnt = {0, 0, 0, 0};

    // per-lane ray state
    bool has_ray = false, active = false;
    int slot = 0;
    v3 origin = mk3(0.f, 0.f, 0.f), direction = mk3(0.f, 0.f, 1.f);
    RayFast rf;
    rf.a = rf.blo = rf.bhi = mk3(0.f, 0.f, 0.f);
    int last_hit = -1, triangle_index = -1;
    float min_distance = -1.0f;
    uint32_t cur = 1, end = 0;
    int sp = 0, npend = 0;
    bool exhausted = false;     // wave-uniform: the queue has no more rays

    for (;;) {
        // ---- refill idle lanes
        unsigned long long idle_mask = __ballot(!has_ray);
        int n_idle = __popcll(idle_mask);
        if (!exhausted && (n_idle >= RAY_REFILL_MIN || n_idle == WAVE)) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(work_counter, (uint32_t)n_idle);
            base = __shfl(base, 0);
            if (base + (uint32_t)n_idle >= (uint32_t)nthreads) exhausted = true;
            if (!has_ray) {
                uint32_t idx = base + (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                if (idx < (uint32_t)nthreads) {
                    slot = first_photon + (int)idx;
                    const float4 *r = rays + 4 * (size_t)slot;
                    const float4 r0 = r[0], r1 = r[1];
                    if (__float_as_int(r1.w) == 0) {             // (other slots were settled by k_ray_setup)
                        const float4 r2 = r[2], r3 = r[3];
                        origin = mk3(r0.x, r0.y, r0.z);
                        direction = mk3(r1.x, r1.y, r1.z);
                        last_hit = __float_as_int(r0.w);
                        rf.a = mk3(r2.x, r2.y, r2.z);
                        const v3 bb = mk3(r3.x, r3.y, r3.z);
                        rf.blo = bb - r2.w * rf.a;
                        rf.bhi = bb + r2.w * rf.a;
                        triangle_index = -1;
                        min_distance = -1.0f;
                        sp = 0;
                        npend = 0;
                        uint4 root = g.nodes[0];
                        has_ray = true;
                        if (node_passes(box_tmin_fast(rf, root), min_distance)) {
                            active = true;
                            cur = root.w & ~CHROMA_NCHILD_MASK;
                            end = cur + (root.w >> CHROMA_CHILD_BITS) - 1;
                        } else {
                            active = false;      // misses the world box: result -1 written below
                        }
                    }
                }
            }
        }
        if (!__any(has_ray)) {
            if (exhausted) break;
            continue;
        }

        // ---- node phase: one node per active lane per iteration; it ends when a lane's FIFO of
        // postponed leaves is full, or enough lanes have finished to make a refill worthwhile
        const int stop_at = exhausted ? 0 : max(0, __popcll(__ballot(active)) - RAY_REFILL_MIN);
        do {
            if (active) {
                if (cur > end) {
                    if (sp == 0) {
                        active = false;
                    } else {
                        sp--;
                        uint32_t w = stack[sp * PROP_BLOCK];
                        cur = w & ~CHROMA_NCHILD_MASK;
                        end = cur + (w >> CHROMA_CHILD_BITS) - 1;
                    }
                }
                if (active) {
                    uint4 nd = g.nodes[cur];
                    cur++;
                    if (COUNT) cnt.nodes++;
                    float tmin = box_tmin_fast(rf, nd);
                    if (node_passes(tmin, min_distance)) {
                        uint32_t nd_child = nd.w & ~CHROMA_NCHILD_MASK;
                        if ((nd.w >> CHROMA_CHILD_BITS) == 0) {
                            if ((int)nd_child != last_hit) {
                                pending[npend * PROP_BLOCK] = nd_child;
                                npend++;
                            }
                        } else if (sp >= RAY_LDS_STACK) {
                            // deeper than the LDS stack: hand the whole ray to the retry kernel
                            active = false;
                            npend = 0;
                            triangle_index = HIT_RETRY;
                        } else {
                            stack[sp * PROP_BLOCK] = nd.w;
                            sp++;
                        }
                    }
                }
            }
        } while (!__any(npend >= TRAV_PENDING) && __popcll(__ballot(active)) > stop_at);

        // ---- leaf phase: postponed triangle tests, oldest first
        for (int j = 0; __any(j < npend); j++) {
            if (j < npend) {
                uint32_t tri = pending[j * PROP_BLOCK];
                if (COUNT) cnt.tris++;
                const float4 *t = g.tri + TRI_STRIDE * (size_t)tri;
                float4 a = t[0], b = t[1], c = t[2];
                float distance;
                if (intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance)) {
                    if (triangle_index == -1 || distance < min_distance) {
                        triangle_index = (int)tri;
                        min_distance = distance;
                    }
                }
            }
        }
        npend = 0;

        // ---- retire finished rays
        if (has_ray && !active) {
            hit_triangle[slot] = triangle_index;                 // record index, or a HIT_* code
            hit_distance[slot] = min_distance;
            if (triangle_index == HIT_RETRY) retry_list[atomicAdd(retry_counter, 1u)] = (uint32_t)slot;
            has_ray = false;
        }
    }

    if (COUNT) {
        unsigned long long st = wave_sum_u64(cnt.steps), nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane == 0) {
            atomicAdd(&counters->photon_steps, st);
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
        }
    }
}

// ---- persistent ray cast over the derived 8-wide tree ---------------------------------------------
// Same frame as k_raycast_persistent (one ray per lane, lanes refilled from the queue), but a node
// visit is one 128-byte line: eight child boxes tested with the fast slab test, triangle children
// noted for the leaf phase, the nearest inner child walked next and the others pushed with their
// box distance so that a popped entry farther than the best hit costs nothing.  The visiting order
// is NOT the reference's; the result is, because the walk is conservative and exact ties between
// triangles are broken by the reference's test order (`rank`, see csrc/wide_build.cpp).
// Rays this kernel cannot take (1/d not moderate, more than WIDE_STACK entries) go to
// k_raycast_retry as before.
#ifndef WIDE_STACK
#define WIDE_STACK 16        // (node, distance) entries per lane in LDS
#endif
#ifndef WIDE_PENDING
#define WIDE_PENDING 12      // postponed triangle tests per lane in LDS
#endif
#ifndef WIDE_FLUSH
#define WIDE_FLUSH 5         // run the leaf phase once a lane holds this many (a visit adds up to 8)
#endif
#ifndef WIDE_SPILL
#define WIDE_SPILL 112       // further entries per lane in global memory (rarely touched)
#endif
#define WIDE_NONE 0xFFFFFFFFu

template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK, RAY_WAVES) void
k_raycast_wide(GeoView g, const float4 *rays, int first_photon, StepState *st,
               int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, uint2 *spill_base, DeviceCounters *counters,
               int big_chunk)
{
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * PROP_BLOCK >= nthreads) return;
    uint32_t *work_counter = &st->work, *retry_counter = &st->retry;
    // rays taken from the queue per atomic: many for big batches (a hot word serves only ~88 atomics/us),
    // one wave-load when every wave gets only a few rounds anyway
    const int chunk = ((long long)nthreads > 4ll * big_chunk * (long long)gridDim.x) ? big_chunk : PROP_BLOCK;
    static_assert(WIDE_FLUSH - 1 + 8 <= WIDE_PENDING, "a node visit must fit the FIFO");
    static_assert(PROP_BLOCK == WAVE, "one wave per workgroup: blockIdx.x names the wave's spill area");
    // stack entries beyond the LDS part live in this wave's slice of a global buffer, [entry][lane]
    uint2 *spill = spill_base + (size_t)blockIdx.x * WIDE_SPILL * PROP_BLOCK + threadIdx.x;
    __shared__ uint32_t s_lds[(2 * WIDE_STACK + WIDE_PENDING) * PROP_BLOCK];
    uint32_t *stack_n = s_lds + threadIdx.x;
    float *stack_t = (float *)(stack_n + WIDE_STACK * PROP_BLOCK);
    uint32_t *pending = stack_n + 2 * WIDE_STACK * PROP_BLOCK;
    const unsigned lane = lane_id();
    LaneCounters cnt = {0, 0, 0, 0};

    bool has_ray = false, active = false;
    int slot = 0;
    v3 origin = mk3(0.f, 0.f, 0.f), direction = mk3(0.f, 0.f, 1.f);
    RayFast rf;
    rf.a = rf.blo = rf.bhi = mk3(0.f, 0.f, 0.f);
    int last_hit = -1, triangle_index = -1;
    uint32_t best_rank = 0;
    float min_distance = -1.0f;
    uint32_t cur = WIDE_NONE;
    int sp = 0, npend = 0;
    // the wave's share of the queue, [loc_next, loc_end), taken `chunk` rays per atomic: a hot word
    // serves only ~88 atomics/us, far fewer than the refills 1e8 rays need
    uint32_t loc_next = 0, loc_end = 0;
    bool exhausted = false;

    for (;;) {
        // ---- refill idle lanes
        unsigned long long idle_mask = __ballot(!has_ray);
        int n_idle = __popcll(idle_mask);
        bool more = !exhausted || loc_next < loc_end;
        if (more && (n_idle >= RAY_REFILL_MIN || n_idle == WAVE)) {
            if (loc_next >= loc_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (uint32_t)chunk);
                base = __shfl(base, 0);
                if (base + (uint32_t)chunk >= (uint32_t)nthreads) exhausted = true;
                loc_next = min(base, (uint32_t)nthreads);
                loc_end = min(base + (uint32_t)chunk, (uint32_t)nthreads);
            }
            uint32_t idx = loc_next + (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
            loc_next = min(loc_end, loc_next + (uint32_t)n_idle);
            if (!has_ray) {
                if (idx < loc_end) {
                    slot = first_photon + (int)idx;
                    const float4 *r = rays + 4 * (size_t)slot;
                    const float4 r0 = r[0], r1 = r[1];
                    if (__float_as_int(r1.w) == 0) {             // (other slots were settled by k_ray_setup)
                        const float4 r2 = r[2], r3 = r[3];
                        origin = mk3(r0.x, r0.y, r0.z);
                        direction = mk3(r1.x, r1.y, r1.z);
                        last_hit = __float_as_int(r0.w);
                        rf.a = mk3(r2.x, r2.y, r2.z);
                        const v3 bb = mk3(r3.x, r3.y, r3.z);
                        rf.blo = bb - r2.w * rf.a;
                        rf.bhi = bb + r2.w * rf.a;
                        triangle_index = -1;
                        min_distance = -1.0f;
                        sp = 0;
                        npend = 0;
                        cur = 0;                 // the wide root holds the children of the reference root
                        has_ray = true;
                        active = true;
                    }
                }
            }
        }
        if (!__any(has_ray)) {
            if (exhausted && loc_next >= loc_end) break;
            continue;
        }

        // ---- node phase: one wide node per active lane per iteration
        more = !exhausted || loc_next < loc_end;
        const int stop_at = more ? max(0, __popcll(__ballot(active)) - RAY_REFILL_MIN) : 0;
        do {
            if (active && cur == WIDE_NONE) {
                // next entry that can still hold a nearer hit
                while (sp > 0) {
                    sp--;
                    uint32_t n; float t;
                    if (sp < WIDE_STACK) { n = stack_n[sp * PROP_BLOCK]; t = stack_t[sp * PROP_BLOCK]; }
                    else { uint2 e = spill[(size_t)(sp - WIDE_STACK) * PROP_BLOCK]; n = e.x; t = __uint_as_float(e.y); }
                    if (min_distance < 0.0f || !(t > min_distance)) { cur = n; break; }
                }
                if (cur == WIDE_NONE) active = false;
            }
            if (active) {
                const uint4 *wn = g.wnodes + 8 * (size_t)cur;
                uint4 c[8];
#pragma unroll
                for (int j = 0; j < 8; j++) c[j] = wn[j];
                if (COUNT) cnt.nodes += 8;
                uint32_t nxt = WIDE_NONE;
                float nxt_t = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float t = box_tmin_fast(rf, c[j]);
                    uint32_t w = c[j].w;
                    if (w != WIDE_NONE && node_passes(t, min_distance)) {
                        if (w & 0x80000000u) {
                            uint32_t tri = w & 0x7FFFFFFFu;
                            if ((int)tri != last_hit) {
                                pending[npend * PROP_BLOCK] = tri;
                                npend++;
                            }
                        } else if (nxt == WIDE_NONE) {
                            nxt = w; nxt_t = t;
                        } else {
                            uint32_t pw = w; float pt = t;
                            if (t < nxt_t) { pw = nxt; pt = nxt_t; nxt = w; nxt_t = t; }
                            if (sp < WIDE_STACK) {
                                stack_n[sp * PROP_BLOCK] = pw;
                                stack_t[sp * PROP_BLOCK] = pt;
                                sp++;
                            } else if (sp < WIDE_STACK + WIDE_SPILL) {
                                spill[(size_t)(sp - WIDE_STACK) * PROP_BLOCK] = make_uint2(pw, __float_as_uint(pt));
                                sp++;
                            } else {                                 // cannot happen: the host checked the tree's need
                                triangle_index = HIT_RETRY;
                            }
                        }
                    }
                }
                cur = nxt;
                if (triangle_index == HIT_RETRY) { active = false; npend = 0; cur = WIDE_NONE; sp = 0; }
            }
        } while (!__any(npend >= WIDE_FLUSH) && __popcll(__ballot(active)) > stop_at);

        // ---- leaf phase: postponed triangle tests
        for (int j = 0; __any(j < npend); j++) {
            if (j < npend) {
                uint32_t tri = pending[j * PROP_BLOCK];
                if (COUNT) cnt.tris++;
                const float4 *t = g.tri + TRI_STRIDE * (size_t)tri;
                float4 a = t[0], b = t[1], cc = t[2];
                float distance;
                if (intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(cc.x, cc.y, cc.z), distance)) {
                    uint32_t rank = __float_as_uint(cc.w);
                    if (triangle_index == -1 || distance < min_distance || (distance == min_distance && rank < best_rank)) {
                        triangle_index = (int)tri;
                        min_distance = distance;
                        best_rank = rank;
                    }
                }
            }
        }
        npend = 0;

        // ---- retire finished rays
        if (has_ray && !active) {
            hit_triangle[slot] = triangle_index;                 // record index, or a HIT_* code
            hit_distance[slot] = min_distance;
            if (triangle_index == HIT_RETRY) retry_list[atomicAdd(retry_counter, 1u)] = (uint32_t)slot;
            has_ray = false;
        }
    }

    if (COUNT) {
        unsigned long long st = wave_sum_u64(cnt.steps), nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane == 0) {
            atomicAdd(&counters->photon_steps, st);
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
        }
    }
}

// ---- cooperative ray cast over the 8-wide tree: eight lanes per ray ---------------------------------
// A wavefront carries 8 rays; the 8 lanes of a group each own ONE of the eight child entries of the
// node their ray is visiting.  A node visit is therefore one coalesced 128-byte read per group (one
// dwordx4 per lane, 8 lines per wave instruction instead of 64), one slab test per lane, and a few
// group-wide operations: ballots give the set of children hit, DPP min-reductions pick the nearest
// inner child, and every other hit lane writes its own (node, distance) entry at its own stack slot,
// so nothing in the visit is serial.  Triangle tests are shared the same way: up to 8 postponed
// triangles of a ray are tested at once, one per lane, and reduced by (distance, rank).
// The per-ray state (origin, direction, slab constants, best hit, stack pointer) is replicated in
// the 8 lanes of the group and stays identical because every lane computes it from the same
// ballots and broadcasts.  LDS: 8 groups x (24 stack entries + 16 postponed triangles) = 2 KB per
// wave, so residency is limited by wave slots only.  Results are those of k_raycast_wide (and of
// the reference): same conservative tree, same tie-break.
#ifndef COOP_STACK
#define COOP_STACK 24
#endif
#define COOP_PENDING 16
#define COOP_STRIDE (2 * COOP_STACK + COOP_PENDING + 1)     // words per group, +1 staggers the banks
#ifndef COOP_SPILL
#define COOP_SPILL 104       // stack entries per ray beyond the LDS part (global memory)
#endif
#ifndef COOP_REFILL_MIN
#define COOP_REFILL_MIN 2    // refill once this many of the 8 groups are idle
#endif

// group-wide (8 lanes) minimum with DPP: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror;
// every lane of the group ends with the result
__device__ inline float group8_min(float v)
{
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));
    return v;
}
__device__ inline uint32_t group8_min_u32(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));
    return v;
}

#ifndef COOP_WAVES_PER_EU
#define COOP_WAVES_PER_EU 7
#endif
template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) __attribute__((amdgpu_waves_per_eu(COOP_WAVES_PER_EU, COOP_WAVES_PER_EU))) void
k_raycast_coop(GeoView g, const float4 *rays, int first_photon, StepState *st,
               int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, uint2 *spill_base, DeviceCounters *counters,
               int big_chunk)
{
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * 8 >= nthreads) return;
    uint32_t *work_counter = &st->work, *retry_counter = &st->retry;
    const int chunk = ((long long)nthreads > 4ll * big_chunk * (long long)gridDim.x) ? big_chunk : 8;
    static_assert(PROP_BLOCK == WAVE, "one wave per workgroup");
    __shared__ uint32_t s_lds[8 * COOP_STRIDE];
    const unsigned lane = lane_id();
    const unsigned j = lane & 7u, gshift = lane & ~7u, grp = lane >> 3;
    const uint32_t below = (1u << j) - 1u;
    uint32_t *stack_n = s_lds + grp * COOP_STRIDE;
    float *stack_t = (float *)(stack_n + COOP_STACK);
    uint32_t *pending = stack_n + 2 * COOP_STACK;
    uint2 *spill = spill_base + ((size_t)blockIdx.x * 8 + grp) * COOP_SPILL;
    LaneCounters cnt = {0, 0, 0, 0};
    const float inf = cm_inff();

    // per-ray state, identical in the 8 lanes of a group
    bool has_ray = false, active = false;
    int slot = 0;
    v3 origin = mk3(0.f, 0.f, 0.f), direction = mk3(0.f, 0.f, 1.f);
    RayFast rf;
    rf.a = rf.blo = rf.bhi = mk3(0.f, 0.f, 0.f);
    int last_hit = -1, triangle_index = -1;
    uint32_t best_rank = 0;
    float min_distance = -1.0f;
    uint32_t cur = WIDE_NONE;
    int sp = 0, npend = 0;
    // the wave's share of the queue: [loc_next, loc_end) taken `chunk` rays at a time
    uint32_t loc_next = 0, loc_end = 0;
    bool exhausted = false;

    for (;;) {
        // ---- refill idle groups
        unsigned long long idle_mask = __ballot(!has_ray && j == 0);
        int n_idle = __popcll(idle_mask);
        bool more = !exhausted || loc_next < loc_end;
        if (more && (n_idle >= COOP_REFILL_MIN || n_idle == 8)) {
            if (loc_next >= loc_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (uint32_t)chunk);
                base = __shfl(base, 0);
                if (base + (uint32_t)chunk >= (uint32_t)nthreads) exhausted = true;
                loc_next = min(base, (uint32_t)nthreads);
                loc_end = min(base + (uint32_t)chunk, (uint32_t)nthreads);
            }
            uint32_t idx = loc_next + (uint32_t)__popcll(idle_mask & ((1ull << gshift) - 1ull));
            loc_next = min(loc_end, loc_next + (uint32_t)n_idle);
            if (!has_ray && idx < loc_end) {
                slot = first_photon + (int)idx;
                const float4 *r = rays + 4 * (size_t)slot;
                const float4 r0 = r[0], r1 = r[1];
                if (__float_as_int(r1.w) == 0) {                 // (other slots were settled by k_ray_setup)
                    const float4 r2 = r[2], r3 = r[3];
                    origin = mk3(r0.x, r0.y, r0.z);
                    direction = mk3(r1.x, r1.y, r1.z);
                    last_hit = __float_as_int(r0.w);
                    rf.a = mk3(r2.x, r2.y, r2.z);
                    const v3 bb = mk3(r3.x, r3.y, r3.z);
                    rf.blo = bb - r2.w * rf.a;
                    rf.bhi = bb + r2.w * rf.a;
                    triangle_index = -1;
                    min_distance = -1.0f;
                    sp = 0;
                    npend = 0;
                    cur = 0;
                    has_ray = true;
                    active = true;
                }
            }
        }
        if (!__any(has_ray)) {
            if (exhausted && loc_next >= loc_end) break;
            continue;
        }

        // ---- node phase: every active group visits one node per iteration
        more = !exhausted || loc_next < loc_end;
        const int stop_at = more ? max(0, (int)__popcll(__ballot(active && j == 0)) - (int)COOP_REFILL_MIN) : 0;
        do {
            if (active && cur == WIDE_NONE) {
                // next entry that can still hold a nearer hit
                while (sp > 0) {
                    sp--;
                    uint32_t n; float t;
                    if (sp < COOP_STACK) { n = stack_n[sp]; t = stack_t[sp]; }
                    else { uint2 e = spill[sp - COOP_STACK]; n = e.x; t = __uint_as_float(e.y); }
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
                    const uint32_t nj = (uint32_t)__ffs((int)gn) - 1u;          // lane of the nearest inner child
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
        } while (!__any(npend >= 8) && __popcll(__ballot(active && j == 0)) > stop_at);
        __builtin_amdgcn_wave_barrier();      // (scheduling fence: the lanes of a group exchange data through LDS)

        // ---- leaf phase: up to 8 postponed triangles of a ray at once, one per lane
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
                if (npend > 8) {                      // keep the rest: move entries 8.. down
                    uint32_t mv = pending[j + 8];
                    if ((int)j + 8 < npend) pending[j] = mv;
                }
                npend -= take;
            }
        }

        // ---- retire finished rays
        if (has_ray && !active) {
            if (j == 0) {
                hit_triangle[slot] = triangle_index;                 // record index, or a HIT_* code
                hit_distance[slot] = min_distance;
                if (triangle_index == HIT_RETRY) retry_list[atomicAdd(retry_counter, 1u)] = (uint32_t)slot;
            }
            has_ray = false;
        }
    }

    if (COUNT) {
        unsigned long long st = wave_sum_u64(cnt.steps), nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        unsigned long long sx = wave_sum_u64(cnt.spills);
        if (lane == 0) {
            atomicAdd(&counters->photon_steps, st);
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
            if (sx) atomicAdd(&counters->stack_spills, sx);
        }
    }
}
