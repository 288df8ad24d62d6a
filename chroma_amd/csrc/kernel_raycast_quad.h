// kernel_raycast_quad.h -- k_raycast_quad: the DEFAULT ray cast -- the derived 8-wide tree, four lanes per ray, persistent waves.
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

// ---- the same ray cast with FOUR lanes per ray, two child entries per lane ---------------------------
// 16 rays per wavefront.  The per-visit bookkeeping of k_raycast_coop (ballots, prefix counts, the
// reduction that picks the nearest child, the loop control) costs as much as the eight slab tests it
// serves; here one pass of that bookkeeping serves 16 rays instead of 8, the reductions run inside a
// quad (two DPP steps), and a lane's two entries are one 32-byte read.  Postponed triangles live in a
// ring per ray and are tested four at a time.  Same tree, same tie-break, same results.
#ifndef QUAD_PENDING
#define QUAD_PENDING 16      // ring of postponed triangles per ray (a power of two; 32 costs residency, measured slower)
#endif
#ifndef QUAD_STACK
#define QUAD_STACK COOP_STACK    // (node, distance) entries per ray in LDS
#endif
#define QUAD_OD_WORDS 6       // origin and direction of a ray wait in LDS between its triangle rounds
#define QUAD_STRIDE (2 * QUAD_STACK + QUAD_PENDING + QUAD_OD_WORDS + 1)     // words per ray, odd: staggers the banks
#ifndef QUAD_REFILL_MIN
#define QUAD_REFILL_MIN 4    // refill once this many of the 16 rays are done
#endif
#ifndef QUAD_WAVES_PER_EU
#define QUAD_WAVES_PER_EU 8  // 63 VGPRs; the one value that does not fit (the base of the global spill area) is reloaded from
#endif                       // scratch in the rare deep-stack push only.  -4 % against 7 (68 VGPRs), profiles/r02/ab_quad_8waves.txt
#ifndef QUAD_TIMING
#define QUAD_TIMING 0        // diagnostic build: s_memtime stamps around the phases of a wave, printed by a few waves
#endif
#ifndef QUAD_FLUSH
#define QUAD_FLUSH 8         // run the triangle tests once a ray has this many postponed (a visit adds up to 8)
#endif
#ifndef QUAD_KEEP
#define QUAD_KEEP 7          // a triangle phase runs rounds until no ray with node work left holds more than this many (7: one round unless a ray holds 12+; -5 % against 0)
#endif

__device__ inline uint32_t quad_min_u32(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    return v;
}
__device__ inline uint32_t quad_max_u32(uint32_t v)
{
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
    v = max(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    return v;
}

// box_interval_fast with the two faces of an axis as one packed operation (v_pk_fma_f32: same fused
// multiply-add per half, half the issue slots)
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ inline void box_interval_pk(float ax, float ay, float az, f32x2 bx, f32x2 by, f32x2 bz, uint4 nd, float &tmin, float &tmax)
{
    f32x2 qx = {(float)(nd.x & 0xFFFFu), (float)(nd.x >> 16)};
    f32x2 qy = {(float)(nd.y & 0xFFFFu), (float)(nd.y >> 16)};
    f32x2 qz = {(float)(nd.z & 0xFFFFu), (float)(nd.z >> 16)};
    const f32x2 tx = __builtin_elementwise_fma(qx, (f32x2){ax, ax}, bx);
    const f32x2 ty = __builtin_elementwise_fma(qy, (f32x2){ay, ay}, by);
    const f32x2 tz = __builtin_elementwise_fma(qz, (f32x2){az, az}, bz);
    tmin = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(tx.x, tx.y), __builtin_fminf(ty.x, ty.y)),
                           __builtin_fmaxf(__builtin_fminf(tz.x, tz.y), 0.0f));
    tmax = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(tx.x, tx.y), __builtin_fmaxf(ty.x, ty.y)),
                           __builtin_fmaxf(tz.x, tz.y));
}
// The same test with the faces picked by the SIGN of the direction instead of by min/max: the packed bounds of an
// axis the ray runs down are rotated by 16 bits (one v_alignbit with a per-ray shift), so that the low half always
// is the face the ray meets first.  near = fma(q_near, a, b - |a|), far = fma(q_far, a, b + |a|) are the very values
// min and max picked (fma and the offsets are monotone), so the result is bit-identical for a real box -- and an EMPTY
// entry (lo = 0xFFFF, hi = 0 on every axis) now fails by itself, because nothing swaps its faces back.
__device__ inline void box_interval_signed(float ax, float ay, float az, uint32_t sx, uint32_t sy, uint32_t sz,
                                           f32x2 bx, f32x2 by, f32x2 bz, uint4 nd, float &tmin, float &tmax)
{
    const uint32_t x = __builtin_amdgcn_alignbit(nd.x, nd.x, sx), y = __builtin_amdgcn_alignbit(nd.y, nd.y, sy),
                   z = __builtin_amdgcn_alignbit(nd.z, nd.z, sz);
    f32x2 qx = {(float)(x & 0xFFFFu), (float)(x >> 16)};
    f32x2 qy = {(float)(y & 0xFFFFu), (float)(y >> 16)};
    f32x2 qz = {(float)(z & 0xFFFFu), (float)(z >> 16)};
    const f32x2 tx = __builtin_elementwise_fma(qx, (f32x2){ax, ax}, bx);
    const f32x2 ty = __builtin_elementwise_fma(qy, (f32x2){ay, ay}, by);
    const f32x2 tz = __builtin_elementwise_fma(qz, (f32x2){az, az}, bz);
    tmin = __builtin_fmaxf(__builtin_fmaxf(tx.x, ty.x), __builtin_fmaxf(tz.x, 0.0f));
    tmax = __builtin_fminf(__builtin_fminf(tx.y, ty.y), tz.y);
}
__device__ inline uint32_t quad_or_u32(uint32_t v)
{
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false);
    return v;
}

template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) __attribute__((amdgpu_waves_per_eu(QUAD_WAVES_PER_EU, QUAD_WAVES_PER_EU))) void
k_raycast_quad(GeoView g, const float4 *rays, int first_photon, StepState *st,
               int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, uint2 *spill_base, DeviceCounters *counters,
               int big_chunk, int settle, const uint32_t *skip = nullptr, int static_eighths = 0)
{
    // (`settle`: nobody has written the hit entries of the slots whose ray record says "not to be cast" yet)
    // (`skip`: the step has been given to k_raycast_packet, launched before this kernel)
    if (skip && *skip != 0u) return;
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * 16 >= nthreads) return;
    uint32_t *work_counter = &st->work, *retry_counter = &st->retry;
    const int chunk = ((long long)nthreads > 4ll * big_chunk * (long long)gridDim.x) ? big_chunk : 16;
    WorkClaim wc(nthreads, chunk, static_eighths);
    static_assert(PROP_BLOCK == WAVE, "one wave per workgroup");
    static_assert((QUAD_PENDING & (QUAD_PENDING - 1)) == 0 && QUAD_FLUSH - 1 + 8 <= QUAD_PENDING && QUAD_KEEP < QUAD_FLUSH, "ring of postponed triangles");
    __shared__ uint32_t s_lds[16 * QUAD_STRIDE];
    const unsigned lane = lane_id();
    const unsigned j = lane & 3u, gshift = lane & ~3u, grp = lane >> 2;
    // what the 4 lanes of a quad decide about their 8 entries travels as ONE word, OR-ed across the quad
    // by two DPP steps: bit 2j / 2j+1 of byte 0 = lane j's first / second entry is a leaf to test, of byte 1 =
    // it is an inner node to visit.  (Wave ballots cost two VALU operations each plus the extract.)
    // (entry e of lane j is bit 2j+e: the entries of a node in memory order)
    const uint32_t jbit = 1u << (2u * j), below2 = jbit - 1u;       // (below2: the entries of lower lanes, within a byte)
    uint32_t *stack_n = s_lds + grp * QUAD_STRIDE;
    float *stack_t = (float *)(stack_n + QUAD_STACK);
    uint32_t *pending = stack_n + 2 * QUAD_STACK;
    uint2 *spill = spill_base + ((size_t)blockIdx.x * 16 + grp) * COOP_SPILL;
    LaneCounters cnt = {0, 0, 0, 0};

    // per-ray state, identical in the 4 lanes of a quad
    bool has_ray = false, active = false;
    int slot = 0;
    float *ray_od = (float *)(stack_n + 2 * QUAD_STACK + QUAD_PENDING);      // origin, direction of this quad's ray
    float rax = 0.f, ray_ = 0.f, raz = 0.f; // RayFast::a (three scalars: as a struct it ended up in LDS), and {blo, bhi} per axis
    f32x2 rbx = {0.f, 0.f}, rby = {0.f, 0.f}, rbz = {0.f, 0.f};
    uint32_t rsx = 0, rsy = 0, rsz = 0;     // 16 for an axis the ray runs down (box_interval_signed)
    uint32_t last_hit_w = WIDE_NONE;        // the leaf word of the photon's last hit (never entered)
    int triangle_index = -1;
    uint32_t best_rank = 0;
    float prune_t = cm_inff();              // distance of the best hit, +inf while nothing was hit
    uint32_t cur = WIDE_NONE;
    int sp = 0, npend = 0;
    uint32_t phead = 0;                     // first postponed triangle in the ring
    uint32_t loc_next = 0, loc_end = 0;
    bool exhausted = false;
#if QUAD_TIMING
    // where a wave's cycles go (diagnostic build, tools/quad_timing.sh): s_memtime stamps around the phases
    unsigned long long tq_refill = 0, tq_pop = 0, tq_wait = 0, tq_node = 0, tq_leaf = 0, tq_retire = 0, tq_a, tq_b;
    unsigned tq_iters = 0, tq_rounds = 0, tq_outer = 0, tq_active = 0, tq_tests = 0;
#define TQ_STAMP(v) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v) :: "memory")
    const unsigned long long tq_start = __builtin_readcyclecounter();
#endif

    for (;;) {
#if QUAD_TIMING
        TQ_STAMP(tq_a); tq_outer++;
#endif
        // ---- refill idle quads
        unsigned long long idle_mask = __ballot(!has_ray && j == 0);
        int n_idle = __popcll(idle_mask);
        bool more = !exhausted || loc_next < loc_end;
        if (more && (n_idle >= QUAD_REFILL_MIN || n_idle == 16)) {
            if (loc_next >= loc_end) {
                const uint32_t base = wc.next(work_counter, lane, exhausted);        // (wave-uniform: scalar registers)
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
                    if (j == 0) { ray_od[0] = r0.x; ray_od[1] = r0.y; ray_od[2] = r0.z; ray_od[3] = r1.x; ray_od[4] = r1.y; ray_od[5] = r1.z; }
                    { const int lh = __float_as_int(r0.w); last_hit_w = lh >= 0 ? (0x80000000u | (uint32_t)lh) : WIDE_NONE; }
                    rax = r2.x; ray_ = r2.y; raz = r2.z;
                    { const float mx = r2.w * cm_fabsf(rax), my = r2.w * cm_fabsf(ray_), mz = r2.w * cm_fabsf(raz);       // (growth of the boxes: ray_growth)
                      rbx = (f32x2){r3.x - mx, r3.x + mx}; rby = (f32x2){r3.y - my, r3.y + my}; rbz = (f32x2){r3.z - mz, r3.z + mz}; }
                    rsx = rax < 0.f ? 16u : 0u; rsy = ray_ < 0.f ? 16u : 0u; rsz = raz < 0.f ? 16u : 0u;
                    triangle_index = -1;
                    prune_t = cm_inff();
                    sp = 0;
                    npend = 0;
                    phead = 0;
                    cur = 0;
                    has_ray = true;
                    active = true;
                } else if (settle && j == 0) {
                    const int status = __float_as_int(r1.w);             // HIT_NAN, or HIT_RETRY: 1/d not moderate
                    hit_triangle[slot] = status;
                    hit_distance[slot] = 0.0f;
                    if (status == HIT_RETRY) retry_list[atomicAdd(retry_counter, 1u)] = (uint32_t)slot;
                }
            }
        }
        if (!__any(has_ray)) {
            if (exhausted && loc_next >= loc_end) break;
            continue;
        }

#if QUAD_TIMING
        TQ_STAMP(tq_b); tq_refill += tq_b - tq_a;
#endif
        // ---- node phase: every active quad visits one node per iteration
        more = !exhausted || loc_next < loc_end;
        // (one lane per quad counts: masks and counts stay in scalar registers)
        const int stop_at = more ? max(0, (int)__popcll(__ballot(active) & 0x1111111111111111ull) - (int)QUAD_REFILL_MIN) : 0;
        do {
#if QUAD_TIMING
            TQ_STAMP(tq_a); tq_iters++; tq_active += (unsigned)__popcll(__ballot(active && j == 0));
#endif
            __builtin_amdgcn_s_setprio(3);       // a wave about to fetch its next node goes before waves that compute
            // a ray whose stack reaches into the global spill area (a few in 1e8) first pops from there -- a prefix that
            // changes its state in place -- and every ray then runs the ONE pop loop over the LDS part: no second arm whose
            // state has to be merged with the first at every node visit
            if (__any(sp > QUAD_STACK)) {
                while (active && cur == WIDE_NONE && sp > QUAD_STACK) {
                    sp--;
                    const uint2 se = spill[sp - QUAD_STACK];
                    cur = (__uint_as_float(se.y) > prune_t) ? WIDE_NONE : se.x;
                }
            }
            // (node, distance) read together, the entry kept or dropped by a select: no branch inside the loop
            while (active && cur == WIDE_NONE) {
                if (sp == 0) { active = false; break; }
                sp--;
                const uint32_t n = stack_n[sp];
                const float t = stack_t[sp];
                cur = (t > prune_t) ? WIDE_NONE : n;
            }
#if QUAD_TIMING
            TQ_STAMP(tq_b); tq_pop += tq_b - tq_a;
#endif
            if (active) {
                const uint4 *np = g.wnodes + 8 * (size_t)cur + 2 * j;       // this lane's two entries: 32 bytes
                const uint4 ea = np[0], eb = np[1];
#if QUAD_TIMING
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                TQ_STAMP(tq_a); tq_wait += tq_a - tq_b;
#endif
                __builtin_amdgcn_s_setprio(0);
                if (COUNT && j == 0) cnt.nodes += 8;
                float ta, tb, fa, fb;
                box_interval_signed(rax, ray_, raz, rsx, rsy, rsz, rbx, rby, rbz, ea, ta, fa);
                box_interval_signed(rax, ray_, raz, rsx, rsy, rsz, rbx, rby, rbz, eb, tb, fb);
                const bool pa = !(ta > fa) & !(ta > prune_t);        // (an empty entry fails the first test by itself)
                const bool pb = !(tb > fb) & !(tb > prune_t);
                // (the photon's last hit is left out when its turn to be tested comes: one compare per triangle
                //  round instead of two per visit)
                const bool fa_leaf = (int)ea.w < 0, fb_leaf = (int)eb.w < 0;
                const bool la = pa & fa_leaf, lb = pb & fb_leaf;
                const bool ia = pa & !fa_leaf, ib = pb & !fb_leaf;
                const uint32_t qm = quad_or_u32((((ib ? 2u * jbit : 0u) | (ia ? jbit : 0u)) << 8) | (lb ? 2u * jbit : 0u) | (la ? jbit : 0u));
                // postponed triangles: ring slots after the ones already there, lower lanes first
                {
                    uint32_t off = phead + (uint32_t)npend + __popc(qm & below2);
                    if (la) pending[off & (QUAD_PENDING - 1u)] = ea.w & 0x7FFFFFFFu;
                    if (lb) pending[(off + (la ? 1u : 0u)) & (QUAD_PENDING - 1u)] = eb.w & 0x7FFFFFFFu;
                    npend += __popc(qm & 0xFFu);
                }
                cur = WIDE_NONE;
                const uint32_t mi = qm >> 8;                 // inner entries, bit = entry number
                if (mi) {
                    // nearest inner child: smallest (distance, entry) key -- the entry number replaces
                    // the low 3 mantissa bits, which only matters for the ORDER of the visits
                    const uint32_t ka = ia ? ((__float_as_uint(ta) & ~7u) | (2u * j)) : 0xFFFFFFFFu;
                    const uint32_t kb = ib ? ((__float_as_uint(tb) & ~7u) | (2u * j + 1u)) : 0xFFFFFFFFu;
                    const uint32_t ne = quad_min_u32(min(ka, kb)) & 7u;          // entry number of the nearest
                    const bool na = ia && ne == 2u * j, nb = ib && ne == 2u * j + 1u;
                    cur = quad_max_u32(na ? ea.w : (nb ? eb.w : 0u));
                    // every other inner child goes on the stack at its own slot
                    const bool qa = ia && !na, qb = ib && !nb;
                    const uint32_t mo = mi & ~(1u << ne);
                    int pos = sp + __popc(mo & below2);
                    sp += __popc(mo);
                    if (!__any(sp > QUAD_STACK)) {
                        // every ray of the wave stays inside its LDS stack (almost always): two plain stores
                        if (qa) { stack_n[pos] = ea.w; stack_t[pos] = ta; pos++; }
                        if (qb) { stack_n[pos] = eb.w; stack_t[pos] = tb; }
                    } else
                    {
                        if (qa) {
                            if (pos < QUAD_STACK) { stack_n[pos] = ea.w; stack_t[pos] = ta; }
                            else if (pos < QUAD_STACK + COOP_SPILL) { spill[pos - QUAD_STACK] = make_uint2(ea.w, __float_as_uint(ta)); if (COUNT) atomicAdd(&counters->stack_spills, 1ull); }
                            pos++;
                        }
                        if (qb) {
                            if (pos < QUAD_STACK) { stack_n[pos] = eb.w; stack_t[pos] = tb; }
                            else if (pos < QUAD_STACK + COOP_SPILL) { spill[pos - QUAD_STACK] = make_uint2(eb.w, __float_as_uint(tb)); if (COUNT) atomicAdd(&counters->stack_spills, 1ull); }
                        }
                        // A stack deeper than LDS part + spill area cannot happen: chroma_geometry_create works the tree's need out
                        // and the launch code only picks this walk when it fits.  Rounds 1-3 nevertheless reset the ray's whole
                        // state here -- a merge of five loop-carried values with an arm that never runs, which cost the arm that
                        // always runs eight register copies per node visit.  The guards above already keep every write inside
                        // the two areas; clamping the depth keeps every later read inside them too, and the overflow is counted
                        // (stats.stack_overflows, which the tests hold at zero).
                        if (sp > QUAD_STACK + COOP_SPILL) { atomicAdd(&counters->stack_overflows, 1ull); sp = QUAD_STACK + COOP_SPILL; }
                    }
                }
            }
#if QUAD_TIMING
            TQ_STAMP(tq_b); tq_node += tq_b - tq_a;       // (a_ = after the wait when the wave fetched, else the pop stamp)
#endif
        } while (!__any(npend >= QUAD_FLUSH) && (int)__popcll(__ballot(active) & 0x1111111111111111ull) > stop_at);
        __builtin_amdgcn_wave_barrier();      // (scheduling fence: the lanes of a quad exchange data through LDS)
#if QUAD_TIMING
        TQ_STAMP(tq_a);
#endif

        // ---- leaf phase: up to 4 postponed triangles of a ray at once, one per lane
        while (__any(npend > (active ? QUAD_KEEP : 0))) {
#if QUAD_TIMING
            tq_rounds++; tq_tests += (unsigned)__popcll(__ballot(npend > 0 && (int)j < min(npend, 4)));
#endif
            {
                // every lane runs the round; a ray without postponed triangles takes none and keeps its state
                // through selects (the reductions are a few DPP operations: cheaper than the copies that
                // branches around them cost)
                const int take = min(npend, 4);
                bool hit = false;
                float distance = 0.0f;
                uint32_t tri = 0, rank = 0xFFFFFFFFu;
                if ((int)j < take) tri = pending[(phead + j) & (QUAD_PENDING - 1u)];
                if ((int)j < take && (0x80000000u | tri) != last_hit_w) {
                    if (COUNT) cnt.tris++;
                    const float4 *tp = g.tri + TRI_STRIDE * (size_t)tri;
                    float4 a = tp[0], b = tp[1], c = tp[2];
                    const v3 origin = mk3(ray_od[0], ray_od[1], ray_od[2]), direction = mk3(ray_od[3], ray_od[4], ray_od[5]);
                    hit = intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance);
                    rank = __float_as_uint(c.w);
                }
                const uint32_t dkey = hit ? __float_as_uint(distance) : 0x7F800000u;
                const uint32_t dmin = quad_min_u32(dkey);
                const float dm = __uint_as_float(dmin);
                const bool cand = hit && dkey == dmin;
                const uint32_t rm = quad_min_u32(cand ? rank : 0xFFFFFFFFu);
                const uint32_t wtri = quad_max_u32((cand && rank == rm) ? tri + 1u : 0u) - 1u;
                // (prune_t is the best distance, +inf before the first hit: no separate "nothing yet" test)
                const bool better = dmin != 0x7F800000u && (dm < prune_t || (dm == prune_t && rm < best_rank));
                triangle_index = better ? (int)wtri : triangle_index;
                prune_t = better ? dm : prune_t;
                best_rank = better ? rm : best_rank;
                phead = (phead + (uint32_t)take) & (QUAD_PENDING - 1u);
                npend -= take;
            }
        }


#if QUAD_TIMING
        TQ_STAMP(tq_b); tq_leaf += tq_b - tq_a;
#endif
        // ---- retire finished rays
        if (has_ray && !active) {
            if (j == 0) {
                hit_triangle[slot] = triangle_index;                 // record index, or a HIT_* code
                hit_distance[slot] = triangle_index == -1 ? -1.0f : prune_t;
                if (triangle_index == HIT_RETRY) retry_list[atomicAdd(retry_counter, 1u)] = (uint32_t)slot;
            }
            has_ray = false;
        }
    }

#if QUAD_TIMING
    if (lane == 0 && (blockIdx.x & 1023u) == 0u && nthreads > 1000000) {
        const unsigned long long total = __builtin_readcyclecounter() - tq_start;
        printf("QT rays %d wave %u total %llu refill %llu pop %llu wait %llu node %llu leaf %llu outer %u iters %u active %u rounds %u tests %u\n",
               nthreads, blockIdx.x, total, tq_refill, tq_pop, tq_wait, tq_node, tq_leaf, tq_outer, tq_iters, tq_active, tq_rounds, tq_tests);
    }
#endif
    if (COUNT) {
        unsigned long long nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane == 0) {
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
        }
    }
}
