// kernel_raycast_literal.h -- k_raycast_literal, the exact walk: chroma/cuda/mesh.h:42-118 for every ray, four lanes per ray.
//
// (Included by chroma_hip.hip; needs its StepState, HIT_* codes and the quad DPP helpers.)
//
// What "literal" means.  The reference's intersect_mesh is a depth-first walk over child RANGES: pop a range,
// test its children IN ORDER against the ray -- intersect_box in the reference's float arithmetic on boxes
// dequantised as world_origin + q * world_scale (geometry.h:31-47, intersect.h:107-147), pruned by the best hit
// SO FAR (mesh.h:16-34) -- a leaf that passes has its triangle tested AT ONCE (mesh.h:82-101), an inner node that
// passes is pushed; the range pushed last is walked next.  The answer depends on that order whenever a
// Moeller-Trumbore result is numerically erratic, so an exact walk has to keep it.
//
// How it runs here.  A wavefront carries 16 rays; the 4 lanes of a quad take FOUR CONSECUTIVE CHILDREN of the
// range their ray is scanning: one 16-byte node each, 64 contiguous bytes per quad.
//   * The four box tests are independent of each other and run in parallel: the slab arithmetic is the
//     reference's own, operation for operation (two multiply-add pairs per bound, not fused; the two faces of an
//     axis as one packed operation).
//   * What the reference does with the results is sequential -- a triangle hit in child k changes the pruning
//     distance for children k+1.. -- and is replayed in order from the four (box distance, hit distance) pairs:
//     a handful of DPP broadcasts.  A chunk without a triangle to test (most: inner ranges) cannot change the
//     pruning distance, so its passing children are pushed at once at prefix-count positions.
//   * Triangle tests are the divergent part: a quad that has one to do WAITS (its lanes sit out the node
//     iterations) until LIT_TRI_MIN quads are waiting or nothing else can run, then all waiting quads test
//     their up-to-four triangles in one pass and replay.  Nothing is postponed past anything it could
//     influence: a ray never looks at another node before its triangle has been tested.
//   * Triangles are tested speculatively within a chunk (child k+1's triangle before child k's result is
//     known); the replay discards a test the reference would not have made.  Discarded tests change nothing.
//   * Persistent waves with refill, an LDS stack of LIT_STACK words per ray (a stack entry is the packed `w`
//     word of a node, as in the reference's two arrays) with the rest in the per-ray slice of the global spill
//     area the other walks use.
// Rays whose 1/d is not "moderate" (|1/d| >= 1e30 or not finite: one per ~3e7 bomb photons) do not come here:
// k_ray_setup lists them for k_raycast_retry (intersect_mesh_strict, whose intersect_box handles them).
#pragma once

#ifndef LIT_STACK
#define LIT_STACK 32          // stack words per ray in LDS
#endif
#define LIT_STRIDE (LIT_STACK + 1)       // odd: staggers the banks
#ifndef LIT_TRI_MIN
#define LIT_TRI_MIN 8         // run the triangle pass once this many of the 16 quads wait for it
#endif
#ifndef LIT_TRI_END_HALF
#define LIT_TRI_END_HALF 1    // with no ray left to take, the triangle pass runs once half of the wave's remaining rays wait for it
#endif
#ifndef LIT_REFILL_MIN
#define LIT_REFILL_MIN 4      // refill once this many of the 16 rays are done
#endif
#ifndef LIT_WAVES_PER_EU
#define LIT_WAVES_PER_EU 8
#endif
#ifndef LIT_DIAG
#define LIT_DIAG 0            // diagnostic build: the counting kernel also reports wave iterations, quads running in them, triangle passes
#endif                        // (through the packet_* fields of the statistics, which the literal walk does not otherwise use)
#define LIT_SPILL (2 * COOP_SPILL)       // words of a ray's slice of the global spill area (uint2 entries there)

// lane k of the quad's value in every lane of the quad (quad_perm [k,k,k,k])
template <int K>
__device__ inline uint32_t quad_bcast_u32(uint32_t v)
{
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, K * 0x55, 0xF, 0xF, false);
}
template <int K>
__device__ inline float quad_bcast_f32(float v) { return __uint_as_float(quad_bcast_u32<K>(__float_as_uint(v))); }

// intersect_box (intersect.h:107-147) on a packed node (geometry.h:31-47), for a ray whose 1/d is finite on every
// axis: bit for bit box_tmin() of propagate_device.h.  lower/upper = wo + q * ws, t = bound * (1/d) + (-o/d); every
// multiplication and addition rounded on its own (-ffp-contract=off), the two faces of an axis side by side.
__device__ inline float box_tmin_exact(uint4 nd, float ws, float wox, float woy, float woz, v3 inv_dir, v3 noid)
{
    const f32x2 qx = {(float)(nd.x & 0xFFFFu), (float)(nd.x >> 16)};
    const f32x2 qy = {(float)(nd.y & 0xFFFFu), (float)(nd.y >> 16)};
    const f32x2 qz = {(float)(nd.z & 0xFFFFu), (float)(nd.z >> 16)};
    const f32x2 s = {ws, ws};
    const f32x2 bx = (f32x2){wox, wox} + qx * s, by = (f32x2){woy, woy} + qy * s, bz = (f32x2){woz, woz} + qz * s;
    const f32x2 tx = bx * (f32x2){inv_dir.x, inv_dir.x} + (f32x2){noid.x, noid.x};
    const f32x2 ty = by * (f32x2){inv_dir.y, inv_dir.y} + (f32x2){noid.y, noid.y};
    const f32x2 tz = bz * (f32x2){inv_dir.z, inv_dir.z} + (f32x2){noid.z, noid.z};
    const float tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)),
                                       __builtin_fmaxf(__builtin_fminf(tz.x, tz.y), 0.0f));
    const float tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)),
                                       __builtin_fmaxf(tz.x, tz.y));
    return (tmin > tmax) ? -1.0f : tmin;
}

template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) __attribute__((amdgpu_waves_per_eu(LIT_WAVES_PER_EU, LIT_WAVES_PER_EU))) void
k_raycast_literal(GeoView g, const float4 *rays, StepState *st, int32_t *hit_triangle, float *hit_distance,
                  uint2 *spill_base, DeviceCounters *counters, int big_chunk, int settle, uint32_t *retry_list, int static_eighths = 0)
{
    // (`settle`: the records came from k_load_working / k_physics, not from k_ray_setup: nobody has written the hit entries
    //  of the slots whose record says "not to be cast" yet -- NaN, or 1/d not moderate: the strict loop's -- as in k_raycast_quad)
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * 16 >= nthreads) return;
    uint32_t *work_counter = &st->work;
    const int chunk = ((long long)nthreads > 4ll * big_chunk * (long long)gridDim.x) ? big_chunk : 16;
    WorkClaim wc(nthreads, chunk, static_eighths);          // (kernel_step_control.h: most chunks without the counter)
    static_assert(PROP_BLOCK == WAVE, "one wave per workgroup");
    __shared__ uint32_t s_lds[16 * LIT_STRIDE];
    const unsigned lane = lane_id();
    const unsigned j = lane & 3u, gshift = lane & ~3u, grp = lane >> 2;
    const uint32_t jbit = 1u << j, below = jbit - 1u;
    uint32_t *stack = s_lds + grp * LIT_STRIDE;
    uint32_t *spill = (uint32_t *)(spill_base + ((size_t)blockIdx.x * 16 + grp) * COOP_SPILL);
    const float ws = g.world_scale, wox = g.world_origin[0], woy = g.world_origin[1], woz = g.world_origin[2];
    LaneCounters cnt = {0, 0, 0, 0};

    // per-ray state, identical in the 4 lanes of a quad
    bool has_ray = false, active = false, waiting = false;
    int slot = 0;
    v3 origin = mk3(0.f, 0.f, 0.f), direction = mk3(0.f, 0.f, 1.f), inv_dir = mk3(0.f, 0.f, 1.f), noid = mk3(0.f, 0.f, 0.f);
    int last_hit = -1, triangle_index = -1;
    float min_distance = -1.0f;
    uint32_t cur = 0, end = 0;           // the children of the range being scanned that are still to do: [cur, end)
    int sp = 0;
    // what a waiting quad holds of its chunk: this lane's node word and box distance, the quad's decision word
    uint32_t w_node = 0, w_qm = 0;
    float w_tmin = 0.f;
    uint32_t loc_next = 0, loc_end = 0;
    bool exhausted = false;
#if LIT_DIAG
    unsigned diag_iters = 0, diag_running = 0, diag_passes = 0, diag_waiting = 0;
#endif

    for (;;) {
        // ---- refill idle quads
        const unsigned long long idle_mask = __ballot(!has_ray && j == 0);
        const int n_idle = __popcll(idle_mask);
        bool more = !exhausted || loc_next < loc_end;
        if (more && (n_idle >= LIT_REFILL_MIN || n_idle == 16)) {
            if (loc_next >= loc_end) {
                const uint32_t base = wc.next(work_counter, lane, exhausted);
                loc_next = min(base, (uint32_t)nthreads);
                loc_end = min(base + (uint32_t)chunk, (uint32_t)nthreads);
            }
            const uint32_t idx = loc_next + (uint32_t)__popcll(idle_mask & ((1ull << gshift) - 1ull));
            loc_next = min(loc_end, loc_next + (uint32_t)n_idle);
            if (!has_ray && idx < loc_end) {
                slot = (int)idx;
                const float4 *r = rays + 4 * (size_t)slot;
                const float4 r0 = r[0], r1 = r[1];
                if (__float_as_int(r1.w) == 0) {                 // (the other slots were settled by k_ray_setup)
                    const float4 r2 = r[2], r3 = r[3];           // 1/d and -o/d (k_ray_setup, literal records)
                    origin = mk3(r0.x, r0.y, r0.z);
                    direction = mk3(r1.x, r1.y, r1.z);
                    inv_dir = mk3(r2.x, r2.y, r2.z);
                    noid = mk3(r3.x, r3.y, r3.z);
                    last_hit = __float_as_int(r0.w);
                    triangle_index = -1;
                    min_distance = -1.0f;
                    sp = 0;
                    // the root is tested like any other node (mesh.h:55): a one-node range
                    cur = 0; end = 1;
                    has_ray = true;
                    active = true;
                    waiting = false;
                } else if (settle && j == 0) {
                    const int status = __float_as_int(r1.w);             // HIT_NAN, or HIT_RETRY: 1/d not moderate
                    hit_triangle[slot] = status;
                    hit_distance[slot] = 0.0f;
                    if (status == HIT_RETRY) retry_list[atomicAdd(&st->retry, 1u)] = (uint32_t)slot;
                }
            }
        }
        if (!__any(has_ray)) {
            if (exhausted && loc_next >= loc_end) break;
            continue;
        }

        // ---- node phase: every quad that is not waiting for its triangles scans four children of its range
        more = !exhausted || loc_next < loc_end;
        const int stop_at = more ? max(0, (int)__popcll(__ballot(active && !waiting) & 0x1111111111111111ull) - (int)LIT_REFILL_MIN) : 0;
        // (no ray left to take: the wave's rays are fewer every pass, and a quad that waits for LIT_TRI_MIN of them waits for
        //  all the others -- the triangle pass then runs once half of the rays that are left wait for it, the share LIT_TRI_MIN
        //  is of a full wave; what a small launch takes is its longest ray)
        const int tri_min = (more || !LIT_TRI_END_HALF) ? (int)LIT_TRI_MIN : max(1, min((int)LIT_TRI_MIN, ((int)__popcll(__ballot(has_ray) & 0x1111111111111111ull) + 1) / 2));
        do {
#if LIT_DIAG
            if (COUNT) { diag_iters++; diag_running += (unsigned)__popcll(__ballot(active && !waiting) & 0x1111111111111111ull); }
#endif
            __builtin_amdgcn_s_setprio(3);
            // the range is done: the one pushed last is next (mesh.h:68-72).  (A stack that reaches into the global spill
            // area -- rare -- pops from there in a branch of its own: one pointer for both would be a flat load.)
            const bool pop = active && !waiting && cur >= end;
            if (__any(pop && sp > LIT_STACK)) {
                if (pop && sp > LIT_STACK) {
                    sp--;
                    const uint32_t w = spill[sp - LIT_STACK];
                    cur = w & ~CHROMA_NCHILD_MASK;
                    end = cur + (w >> CHROMA_CHILD_BITS);
                }
            } 
            if (pop && cur >= end) {
                if (sp == 0) {
                    active = false;
                } else {
                    sp--;
                    const uint32_t w = stack[sp];
                    cur = w & ~CHROMA_NCHILD_MASK;
                    end = cur + (w >> CHROMA_CHILD_BITS);
                }
            }
            if (active && !waiting) {
                const uint32_t idx = cur + j;
                const bool valid = idx < end;
                const uint4 nd = g.nodes[valid ? idx : end - 1u];
                __builtin_amdgcn_s_setprio(0);
                if (COUNT && j == 0 && end != 1u) cnt.nodes += min(4u, end - cur);      // (children only: mesh.h counts inside its loop, the root is tested before it)
                cur += 4u;
                const float tmin = box_tmin_exact(nd, ws, wox, woy, woz, inv_dir, noid);
                // intersect_node (mesh.h:16-34) against the best hit known when the chunk is entered: a superset of what
                // the replay lets through, the pruning distance only ever shrinks
                const bool pass = valid && node_passes(tmin, min_distance);
                const bool leaf = (nd.w >> CHROMA_CHILD_BITS) == 0u;
                const bool is_tri = pass && leaf && (int)(nd.w & ~CHROMA_NCHILD_MASK) != last_hit;      // mesh.h:82
                const bool is_inner = pass && !leaf;
                const uint32_t qm = quad_or_u32((is_inner ? (jbit << 4) : 0u) | (is_tri ? jbit : 0u));
                if ((qm & 0xFu) == 0u) {
                    // no triangle in this chunk: nothing can change the pruning distance, the inner children that
                    // pass go on the stack in order
                    const uint32_t mi = qm >> 4;
                    const int pos = sp + (int)__popc(mi & below);
                    if (is_inner) {
                        if (pos < LIT_STACK) stack[pos] = nd.w;
                        else if (pos < LIT_STACK + LIT_SPILL) { spill[pos - LIT_STACK] = nd.w; if (COUNT) cnt.spills++; }
                    }
                    sp += (int)__popc(mi);
                    // (cannot happen: chroma_geometry_create works the tree's need out and the launch code checks it)
                    if (sp > LIT_STACK + LIT_SPILL) { if (j == 0) cnt.overflows++; sp = LIT_STACK + LIT_SPILL; }
                } else {
                    waiting = true;
                    w_node = nd.w;
                    w_tmin = tmin;
                    w_qm = qm;
                }
            }
        } while ((int)__popcll(__ballot(waiting) & 0x1111111111111111ull) < tri_min &&
                 (int)__popcll(__ballot(active && !waiting) & 0x1111111111111111ull) > stop_at);
        __builtin_amdgcn_s_setprio(0);

        // ---- triangle pass: every waiting quad tests the triangles of its chunk (one per lane), then replays the
        // reference's loop over the four children in order
        if (__any(waiting)) {
#if LIT_DIAG
            if (COUNT) { diag_passes++; diag_waiting += (unsigned)__popcll(__ballot(waiting) & 0x1111111111111111ull); }
#endif
            bool hit = false;
            float distance = 0.0f;
            if (waiting && (w_qm & jbit)) {
                const float4 *tp = g.tri + TRI_STRIDE * (size_t)(w_node & ~CHROMA_NCHILD_MASK);
                const float4 a = tp[0], b = tp[1], c = tp[2];
                hit = intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance);
            }
            const uint32_t hm = quad_or_u32(hit ? jbit : 0u);
            // Replay of mesh.h:74-108 over the chunk's four children in order.  Every lane of the quad fetches the four box
            // distances and the four hit distances once (DPP broadcasts) and runs the same short chain on its own registers:
            // `m` is the pruning distance as the reference would hold it after each child (negative: nothing hit yet, which
            // is also what triangle_index == -1 says, a hit distance being > 1e-6), `win` the child whose triangle became
            // the best hit, `acc` the inner children that pass and are pushed.
            if (waiting) {
                const float t0 = quad_bcast_f32<0>(w_tmin), t1 = quad_bcast_f32<1>(w_tmin), t2 = quad_bcast_f32<2>(w_tmin), t3 = quad_bcast_f32<3>(w_tmin);
                const float d0 = quad_bcast_f32<0>(distance), d1 = quad_bcast_f32<1>(distance), d2 = quad_bcast_f32<2>(distance), d3 = quad_bcast_f32<3>(distance);
                float m = min_distance;
                uint32_t win = 4u, acc = 0u, ntested = 0u;
#define LIT_REPLAY(K, TK, DK)                                                                                          \
                if ((w_qm & (0x11u << K)) && node_passes(TK, m)) {                                                     \
                    if (w_qm & (1u << K)) {                                                                            \
                        ntested++;                                                                                     \
                        if ((hm & (1u << K)) && (m < 0.0f || DK < m)) { m = DK; win = K; }          /* mesh.h:88 */     \
                    } else {                                                                                           \
                        acc |= 1u << K;                                                                                \
                    }                                                                                                  \
                }
                LIT_REPLAY(0, t0, d0) LIT_REPLAY(1, t1, d1) LIT_REPLAY(2, t2, d2) LIT_REPLAY(3, t3, d3)
#undef LIT_REPLAY
                if (COUNT && j == 0) cnt.tris += ntested;
                const uint32_t wchild = quad_max_u32(j == win ? (w_node & ~CHROMA_NCHILD_MASK) + 1u : 0u);
                if (win != 4u) { triangle_index = (int)wchild - 1; min_distance = m; }
                if (acc) {                       // (a chunk that holds triangles AND inner nodes: rare)
                    const int pos = sp + (int)__popc(acc & below);
                    if (acc & jbit) {
                        if (pos < LIT_STACK) stack[pos] = w_node;
                        else if (pos < LIT_STACK + LIT_SPILL) { spill[pos - LIT_STACK] = w_node; if (COUNT) cnt.spills++; }
                    }
                    sp += (int)__popc(acc);
                    if (sp > LIT_STACK + LIT_SPILL) { if (j == 0) cnt.overflows++; sp = LIT_STACK + LIT_SPILL; }
                }
            }
            waiting = false;
            __builtin_amdgcn_wave_barrier();      // (scheduling fence: lane 0 of a quad wrote stack words its other lanes will read)
        }

        // ---- retire finished rays
        if (has_ray && !active) {
            if (j == 0) {
                hit_triangle[slot] = triangle_index;                 // record index or -1
                hit_distance[slot] = min_distance;
            }
            has_ray = false;
        }
    }

    const unsigned long long ov = wave_sum_u64(cnt.overflows);
    if (COUNT) {
        const unsigned long long nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris), sx = wave_sum_u64(cnt.spills);
        if (lane == 0) {
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
            if (sx) atomicAdd(&counters->stack_spills, sx);
        }
    }
    if (lane == 0 && ov) atomicAdd(&counters->stack_overflows, ov);
#if LIT_DIAG
    if (COUNT && lane == 0) {
        atomicAdd(&counters->packet_nodes, (unsigned long long)diag_iters);
        atomicAdd(&counters->packet_rays, (unsigned long long)diag_running);
        atomicAdd(&counters->packet_tris, ((unsigned long long)diag_waiting << 32) | diag_passes);
    }
#endif
}

// ---- the same walk for ONE ray per group of 8 lanes, on its own: the cast of the tail kernel under the exact walk ------------
// k_tail_coop<.., LITERAL> finishes the last (< 8192) photons of a call in one launch, a group of 8 lanes per photon; its ray cast
// is this function: the chunk loop of k_raycast_literal without the scheduling against other rays (a group with a triangle to
// test tests it at once).  The two quads of a group do the same work on the same values (lane j and lane j + 4 take the same
// child): the tail is a chain of dependent fetches, not a throughput problem.  Returns the record index, -1, or HIT_RETRY for
// a ray whose 1/d is not moderate (the caller's strict loop takes it).
template <bool COUNT>
__device__ inline int literal_cast_group8(const GeoView &g, v3 origin, v3 direction, int last_hit, bool on, float &min_distance,
                                          uint32_t *stack, uint32_t *spill, unsigned j8, LaneCounters &cnt)
{
    const unsigned j = j8 & 3u;
    const uint32_t jbit = 1u << j, below = jbit - 1u;
    const float ws = g.world_scale, wox = g.world_origin[0], woy = g.world_origin[1], woz = g.world_origin[2];
    int triangle_index = -1;
    min_distance = -1.0f;
    v3 noid = mk3(0.f, 0.f, 0.f), inv_dir = mk3(0.f, 0.f, 1.f);
    bool active = false;
    if (on) {
        noid = (-origin) / direction;
        inv_dir = 1.0f / direction;
        const bool moderate = cm_fabsf(inv_dir.x) < 1e30f && cm_fabsf(inv_dir.y) < 1e30f && cm_fabsf(inv_dir.z) < 1e30f &&
                              cm_fabsf(noid.x) < 1e30f && cm_fabsf(noid.y) < 1e30f && cm_fabsf(noid.z) < 1e30f;
        if (!moderate) triangle_index = HIT_RETRY; else active = true;
    }
    uint32_t cur = 0, end = 1;           // the root, tested like any other node (mesh.h:55)
    int sp = 0;
    while (__any(active)) {
        if (active && cur >= end) {
            if (sp == 0) {
                active = false;
            } else {
                sp--;
                const uint32_t w = (sp < LIT_STACK) ? stack[sp] : spill[sp - LIT_STACK];
                cur = w & ~CHROMA_NCHILD_MASK;
                end = cur + (w >> CHROMA_CHILD_BITS);
            }
        }
        if (active) {
            const uint32_t idx = cur + j;
            const bool valid = idx < end;
            const uint4 nd = g.nodes[valid ? idx : end - 1u];
            if (COUNT && j8 == 0 && end != 1u) cnt.nodes += min(4u, end - cur);
            cur += 4u;
            const float tmin = box_tmin_exact(nd, ws, wox, woy, woz, inv_dir, noid);
            const bool pass = valid && node_passes(tmin, min_distance);
            const bool leaf = (nd.w >> CHROMA_CHILD_BITS) == 0u;
            const bool is_tri = pass && leaf && (int)(nd.w & ~CHROMA_NCHILD_MASK) != last_hit;
            const bool is_inner = pass && !leaf;
            const uint32_t qm = quad_or_u32((is_inner ? (jbit << 4) : 0u) | (is_tri ? jbit : 0u));
            uint32_t acc = qm >> 4;                      // the inner children to push: all that pass, unless a triangle intervenes
            if (qm & 0xFu) {
                bool hit = false;
                float distance = 0.0f;
                if (is_tri) {
                    const float4 *tp = g.tri + TRI_STRIDE * (size_t)(nd.w & ~CHROMA_NCHILD_MASK);
                    const float4 a = tp[0], b = tp[1], c = tp[2];
                    hit = intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance);
                }
                const uint32_t hm = quad_or_u32(hit ? jbit : 0u);
                const float t0 = quad_bcast_f32<0>(tmin), t1 = quad_bcast_f32<1>(tmin), t2 = quad_bcast_f32<2>(tmin), t3 = quad_bcast_f32<3>(tmin);
                const float d0 = quad_bcast_f32<0>(distance), d1 = quad_bcast_f32<1>(distance), d2 = quad_bcast_f32<2>(distance), d3 = quad_bcast_f32<3>(distance);
                float m = min_distance;
                uint32_t win = 4u, ntested = 0u;
                acc = 0u;
#define LIT_REPLAY(K, TK, DK)                                                                                          \
                if ((qm & (0x11u << K)) && node_passes(TK, m)) {                                                       \
                    if (qm & (1u << K)) {                                                                              \
                        ntested++;                                                                                     \
                        if ((hm & (1u << K)) && (m < 0.0f || DK < m)) { m = DK; win = K; }          /* mesh.h:88 */     \
                    } else {                                                                                           \
                        acc |= 1u << K;                                                                                \
                    }                                                                                                  \
                }
                LIT_REPLAY(0, t0, d0) LIT_REPLAY(1, t1, d1) LIT_REPLAY(2, t2, d2) LIT_REPLAY(3, t3, d3)
#undef LIT_REPLAY
                if (COUNT && j8 == 0) cnt.tris += ntested;
                const uint32_t wchild = quad_max_u32(j == win ? (nd.w & ~CHROMA_NCHILD_MASK) + 1u : 0u);
                if (win != 4u) { triangle_index = (int)wchild - 1; min_distance = m; }
            }
            if (acc) {
                const int pos = sp + (int)__popc(acc & below);
                if ((acc & jbit) && j8 < 4u) {
                    if (pos < LIT_STACK) stack[pos] = nd.w;
                    else if (pos < LIT_STACK + LIT_SPILL) { spill[pos - LIT_STACK] = nd.w; if (COUNT) cnt.spills++; }
                }
                sp += (int)__popc(acc);
                if (sp > LIT_STACK + LIT_SPILL) { if (j8 == 0) cnt.overflows++; sp = LIT_STACK + LIT_SPILL; }
            }
            __builtin_amdgcn_wave_barrier();      // (scheduling fence: lanes 0..3 wrote stack words every lane of the group will read)
        }
    }
    return triangle_index;
}
