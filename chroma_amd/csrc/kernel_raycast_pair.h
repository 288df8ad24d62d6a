// kernel_raycast_pair.h -- k_raycast_pair: the wide tree with two lanes per ray (a cross-check walk).
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

// ---- the same ray cast with TWO lanes per ray, four child entries per lane -----------------------------
// 32 rays per wavefront.  The ray cast is bound by instruction issue plus the latency of a wave's chain
// per node visit, and the four-lane kernel pays its per-visit bookkeeping (the quad-wide word, prefix
// counts, the reduction that picks the nearest child, pop and loop control: two thirds of a visit's
// instructions) once per 16 rays.  Here the same bookkeeping serves 32 rays: a lane tests four entries (one
// 64-byte read, the pair of lanes reading one 128-byte line), what the two lanes decide travels as one word
// exchanged by a single DPP swap, and a stack entry is one 8-byte LDS word pair written without branches
// (an entry that is not pushed goes to a scratch slot of the ray's LDS area).  Same tree, same
// (distance, rank) tie-break, same results as k_raycast_quad.
#ifndef PAIR_STACK
#define PAIR_STACK 18        // (node, distance) entries per ray in LDS; deeper ones go through the global spill area
#endif
#define PAIR_PENDING 16      // ring of postponed triangles per ray
#define PAIR_STRIDE (2 * PAIR_STACK + PAIR_PENDING + 2)     // words per ray: stack pairs, ring, one scratch pair (even: 8-byte aligned)
#ifndef PAIR_REFILL_MIN
#define PAIR_REFILL_MIN 8    // refill once this many of the 32 rays are done
#endif
#ifndef PAIR_WAVES_PER_EU
#define PAIR_WAVES_PER_EU 5
#endif
#ifndef PAIR_FLUSH
#define PAIR_FLUSH 8
#endif

__device__ inline uint32_t pair_swap(uint32_t v) { return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false); }
__device__ inline uint32_t pair_min_u32(uint32_t v) { return min(v, pair_swap(v)); }
__device__ inline uint32_t pair_max_u32(uint32_t v) { return max(v, pair_swap(v)); }

template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) __attribute__((amdgpu_waves_per_eu(PAIR_WAVES_PER_EU, PAIR_WAVES_PER_EU))) void
k_raycast_pair(GeoView g, const float4 *rays, int first_photon, StepState *st,
               int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, uint2 *spill_base, DeviceCounters *counters,
               int big_chunk)
{
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * 32 >= nthreads) return;
    uint32_t *work_counter = &st->work, *retry_counter = &st->retry;
    const int chunk = ((long long)nthreads > 4ll * big_chunk * (long long)gridDim.x) ? big_chunk : 32;
    static_assert(PROP_BLOCK == WAVE, "one wave per workgroup");
    static_assert((PAIR_PENDING & (PAIR_PENDING - 1)) == 0 && PAIR_FLUSH - 1 + 8 <= PAIR_PENDING && (PAIR_STRIDE & 1) == 0, "LDS layout");
    __shared__ __attribute__((aligned(8))) uint32_t s_lds[32 * PAIR_STRIDE];
    const unsigned lane = lane_id();
    const unsigned j = lane & 1u, pshift = lane & ~1u, grp = lane >> 1;
    const uint32_t low4 = j ? 0xFu : 0u;               // the partner's entries, when they come before this lane's
    uint2 *stack = (uint2 *)(s_lds + grp * PAIR_STRIDE);                         // [PAIR_STACK] (node, distance bits)
    uint32_t *pending = s_lds + grp * PAIR_STRIDE + 2 * PAIR_STACK;               // [PAIR_PENDING]
    uint2 *const scratch_pair = (uint2 *)(pending + PAIR_PENDING);               // where an entry that is not pushed goes
    uint32_t *const scratch_word = pending + PAIR_PENDING;
    uint2 *spill = spill_base + ((size_t)blockIdx.x * 32 + grp) * COOP_SPILL;
    LaneCounters cnt = {0, 0, 0, 0};

    // per-ray state, identical in the 2 lanes of a pair
    bool has_ray = false, active = false;
    int slot = 0;
    v3 origin = mk3(0.f, 0.f, 0.f), direction = mk3(0.f, 0.f, 1.f);
    float rax = 0.f, ray_ = 0.f, raz = 0.f;
    f32x2 rbx = {0.f, 0.f}, rby = {0.f, 0.f}, rbz = {0.f, 0.f};
    uint32_t last_hit_w = WIDE_NONE;
    int triangle_index = -1;
    uint32_t best_rank = 0;
    float min_distance = -1.0f;
    float prune_t = cm_inff();
    uint32_t cur = WIDE_NONE;
    int sp = 0, npend = 0;
    uint32_t phead = 0;
    uint32_t loc_next = 0, loc_end = 0;
    bool exhausted = false;

    for (;;) {
        // ---- refill idle pairs
        unsigned long long idle_mask = __ballot(!has_ray && j == 0);
        int n_idle = __popcll(idle_mask);
        bool more = !exhausted || loc_next < loc_end;
        if (more && (n_idle >= PAIR_REFILL_MIN || n_idle == 32)) {
            if (loc_next >= loc_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (uint32_t)chunk);
                base = __shfl(base, 0);
                if (base + (uint32_t)chunk >= (uint32_t)nthreads) exhausted = true;
                loc_next = min(base, (uint32_t)nthreads);
                loc_end = min(base + (uint32_t)chunk, (uint32_t)nthreads);
            }
            uint32_t idx = loc_next + (uint32_t)__popcll(idle_mask & ((1ull << pshift) - 1ull));
            loc_next = min(loc_end, loc_next + (uint32_t)n_idle);
            if (!has_ray && idx < loc_end) {
                slot = first_photon + (int)idx;
                const float4 *r = rays + 4 * (size_t)slot;
                const float4 r0 = r[0], r1 = r[1];
                if (__float_as_int(r1.w) == 0) {                 // (other slots were settled by k_ray_setup)
                    const float4 r2 = r[2], r3 = r[3];
                    origin = mk3(r0.x, r0.y, r0.z);
                    direction = mk3(r1.x, r1.y, r1.z);
                    { const int lh = __float_as_int(r0.w); last_hit_w = lh >= 0 ? (0x80000000u | (uint32_t)lh) : WIDE_NONE; }
                    rax = r2.x; ray_ = r2.y; raz = r2.z;
                    rbx = (f32x2){r3.x - r2.w * rax, r3.x + r2.w * rax};
                    rby = (f32x2){r3.y - r2.w * ray_, r3.y + r2.w * ray_};
                    rbz = (f32x2){r3.z - r2.w * raz, r3.z + r2.w * raz};
                    triangle_index = -1;
                    min_distance = -1.0f;
                    prune_t = cm_inff();
                    sp = 0;
                    npend = 0;
                    phead = 0;
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

        // ---- node phase: every active pair visits one node per iteration
        more = !exhausted || loc_next < loc_end;
        const int stop_at = more ? max(0, (int)__popcll(__ballot(active && j == 0)) - (int)PAIR_REFILL_MIN) : 0;
        do {
            __builtin_amdgcn_s_setprio(3);       // a wave about to fetch its next node goes before waves that compute
            if (!__any(sp > PAIR_STACK)) {
                if (active && cur == WIDE_NONE) {
                    while (sp > 0) {
                        sp--;
                        const uint2 se = stack[sp];
                        if (!(__uint_as_float(se.y) > prune_t)) { cur = se.x; break; }
                    }
                    if (cur == WIDE_NONE) active = false;
                }
            } else if (active && cur == WIDE_NONE) {
                while (sp > 0) {
                    sp--;
                    const uint2 se = sp < PAIR_STACK ? stack[sp] : spill[sp - PAIR_STACK];
                    if (!(__uint_as_float(se.y) > prune_t)) { cur = se.x; break; }
                }
                if (cur == WIDE_NONE) active = false;
            }
            if (active) {
                const uint4 *np = g.wnodes + 8 * (size_t)cur + 4 * j;       // this lane's four entries: 64 bytes
                const uint4 e0 = np[0], e1 = np[1], e2 = np[2], e3 = np[3];
                __builtin_amdgcn_s_setprio(0);
                if (COUNT && j == 0) cnt.nodes += 8;
                float t0, t1, t2, t3, f0, f1, f2, f3;
                box_interval_pk(rax, ray_, raz, rbx, rby, rbz, e0, t0, f0);
                box_interval_pk(rax, ray_, raz, rbx, rby, rbz, e1, t1, f1);
                box_interval_pk(rax, ray_, raz, rbx, rby, rbz, e2, t2, f2);
                box_interval_pk(rax, ray_, raz, rbx, rby, rbz, e3, t3, f3);
                // intersect_node (mesh.h:16-34) with prune_t = +inf until something is hit
                const bool p0 = (e0.w != WIDE_NONE) & !(t0 > f0) & !(t0 > prune_t);
                const bool p1 = (e1.w != WIDE_NONE) & !(t1 > f1) & !(t1 > prune_t);
                const bool p2 = (e2.w != WIDE_NONE) & !(t2 > f2) & !(t2 > prune_t);
                const bool p3 = (e3.w != WIDE_NONE) & !(t3 > f3) & !(t3 > prune_t);
                const bool l0 = p0 & ((int)e0.w < 0) & (e0.w != last_hit_w), i0 = p0 & ((int)e0.w >= 0);
                const bool l1 = p1 & ((int)e1.w < 0) & (e1.w != last_hit_w), i1 = p1 & ((int)e1.w >= 0);
                const bool l2 = p2 & ((int)e2.w < 0) & (e2.w != last_hit_w), i2 = p2 & ((int)e2.w >= 0);
                const bool l3 = p3 & ((int)e3.w < 0) & (e3.w != last_hit_w), i3 = p3 & ((int)e3.w >= 0);
                // the pair's word: bits 0-7 = entry k is a leaf to test, bits 8-15 = an inner node to visit
                // (entry number = 4 * lane-in-pair + k)
                const uint32_t own = ((l0 ? 0x001u : 0u) | (l1 ? 0x002u : 0u) | (l2 ? 0x004u : 0u) | (l3 ? 0x008u : 0u) |
                                      (i0 ? 0x100u : 0u) | (i1 ? 0x200u : 0u) | (i2 ? 0x400u : 0u) | (i3 ? 0x800u : 0u)) << (4u * j);
                const uint32_t pm = own | pair_swap(own);
                // postponed triangles: ring slots after the ones already there, lower entries first; an entry
                // that is no leaf writes to the scratch word instead (no branches)
                {
                    uint32_t off = phead + (uint32_t)npend + __popc(pm & low4);
                    uint32_t *a0 = l0 ? pending + (off & (PAIR_PENDING - 1u)) : scratch_word; off += l0 ? 1u : 0u;
                    uint32_t *a1 = l1 ? pending + (off & (PAIR_PENDING - 1u)) : scratch_word; off += l1 ? 1u : 0u;
                    uint32_t *a2 = l2 ? pending + (off & (PAIR_PENDING - 1u)) : scratch_word; off += l2 ? 1u : 0u;
                    uint32_t *a3 = l3 ? pending + (off & (PAIR_PENDING - 1u)) : scratch_word;
                    *a0 = e0.w & 0x7FFFFFFFu; *a1 = e1.w & 0x7FFFFFFFu; *a2 = e2.w & 0x7FFFFFFFu; *a3 = e3.w & 0x7FFFFFFFu;
                    npend += __popc(pm & 0xFFu);
                }
                cur = WIDE_NONE;
                const uint32_t mi = pm >> 8;                 // inner entries by entry number
                if (mi) {
                    // nearest inner child: smallest (distance, entry) key -- the entry number replaces the
                    // low 3 mantissa bits, which only matters for the ORDER of the visits
                    const uint32_t eb = 4u * j;
                    const uint32_t k0 = i0 ? ((__float_as_uint(t0) & ~7u) | eb) : 0xFFFFFFFFu;
                    const uint32_t k1 = i1 ? ((__float_as_uint(t1) & ~7u) | (eb + 1u)) : 0xFFFFFFFFu;
                    const uint32_t k2 = i2 ? ((__float_as_uint(t2) & ~7u) | (eb + 2u)) : 0xFFFFFFFFu;
                    const uint32_t k3 = i3 ? ((__float_as_uint(t3) & ~7u) | (eb + 3u)) : 0xFFFFFFFFu;
                    const uint32_t ne = pair_min_u32(min(min(k0, k1), min(k2, k3))) & 7u;        // entry number of the nearest
                    const uint32_t nk = ne - eb;                                                  // 0..3 when it is this lane's
                    const uint32_t mine = nk == 0u ? e0.w : nk == 1u ? e1.w : nk == 2u ? e2.w : nk == 3u ? e3.w : 0u;
                    cur = pair_max_u32(mine);
                    // every other inner child goes on the stack at its own slot
                    const uint32_t mo = mi & ~(1u << ne);
                    int pos = sp + __popc(mo & low4);
                    sp += __popc(mo);
                    const bool q0 = i0 & (nk != 0u), q1 = i1 & (nk != 1u), q2 = i2 & (nk != 2u), q3 = i3 & (nk != 3u);
                    if (!__any(sp > PAIR_STACK)) {
                        // every ray of the wave stays inside its LDS stack (almost always): four unconditional stores
                        uint2 *s0 = q0 ? stack + pos : scratch_pair; pos += q0 ? 1 : 0;
                        uint2 *s1 = q1 ? stack + pos : scratch_pair; pos += q1 ? 1 : 0;
                        uint2 *s2 = q2 ? stack + pos : scratch_pair; pos += q2 ? 1 : 0;
                        uint2 *s3 = q3 ? stack + pos : scratch_pair;
                        *s0 = make_uint2(e0.w, __float_as_uint(t0)); *s1 = make_uint2(e1.w, __float_as_uint(t1));
                        *s2 = make_uint2(e2.w, __float_as_uint(t2)); *s3 = make_uint2(e3.w, __float_as_uint(t3));
                    } else {
#define PAIR_PUSH(q, e, t)                                                                                              \
                        if (q) {                                                                                        \
                            if (pos < PAIR_STACK) stack[pos] = make_uint2(e.w, __float_as_uint(t));                     \
                            else if (pos < PAIR_STACK + COOP_SPILL) { spill[pos - PAIR_STACK] = make_uint2(e.w, __float_as_uint(t)); \
                                                                      if (COUNT) atomicAdd(&counters->stack_spills, 1ull); } \
                            pos++;                                                                                      \
                        }
                        PAIR_PUSH(q0, e0, t0) PAIR_PUSH(q1, e1, t1) PAIR_PUSH(q2, e2, t2) PAIR_PUSH(q3, e3, t3)
#undef PAIR_PUSH
                        if (sp > PAIR_STACK + COOP_SPILL) {          // cannot happen: the host checked the tree's need
                            triangle_index = HIT_RETRY;
                            active = false; npend = 0; cur = WIDE_NONE; sp = 0;
                        }
                    }
                }
            }
        } while (!__any(npend >= PAIR_FLUSH) && __popcll(__ballot(active && j == 0)) > stop_at);
        __builtin_amdgcn_wave_barrier();      // (scheduling fence: the lanes of a pair exchange data through LDS)

        // ---- leaf phase: up to 2 postponed triangles of a ray at once, one per lane
        while (__any(npend > 0)) {
            if (npend > 0) {
                const int take = min(npend, 2);
                bool hit = false;
                float distance = 0.0f;
                uint32_t tri = 0, rank = 0xFFFFFFFFu;
                if ((int)j < take) {
                    tri = pending[(phead + j) & (PAIR_PENDING - 1u)];
                    if (COUNT) cnt.tris++;
                    const float4 *tp = g.tri + TRI_STRIDE * (size_t)tri;
                    float4 a = tp[0], b = tp[1], c = tp[2];
                    hit = intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance);
                    rank = __float_as_uint(c.w);
                }
                // distances are positive: their bit patterns order like the floats
                const uint32_t dkey = hit ? __float_as_uint(distance) : 0x7F800000u;
                const uint32_t dmin = pair_min_u32(dkey);
                if (dmin != 0x7F800000u) {
                    const float dm = __uint_as_float(dmin);
                    const bool cand = hit && dkey == dmin;
                    const uint32_t rm = pair_min_u32(cand ? rank : 0xFFFFFFFFu);
                    const uint32_t wtri = pair_max_u32((cand && rank == rm) ? tri + 1u : 0u) - 1u;
                    if (triangle_index == -1 || dm < min_distance || (dm == min_distance && rm < best_rank)) {
                        triangle_index = (int)wtri;
                        min_distance = dm;
                        prune_t = dm;
                        best_rank = rm;
                    }
                }
                phead = (phead + (uint32_t)take) & (PAIR_PENDING - 1u);
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
        unsigned long long nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane == 0) {
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
        }
    }
}
