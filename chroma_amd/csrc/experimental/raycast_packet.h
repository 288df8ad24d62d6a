// experimental/raycast_packet.h -- k_raycast_packet: the first step of a call as packets of 64 rays (profiles/r03/ab_packet_first_step.txt, pmc_packet.txt)
// Built only into build_variants/libchroma_hip_experimental.so (-DCHROMA_EXPERIMENTAL=1, `make variants`): measured, parity-green,
// and NOT faster than the product path (the A/B records are named below).  The product library does not contain this code.
#pragma once

// ---- the ray cast for COHERENT rays: one packet of 64 rays per wavefront ---------------------------------
// The first step of a batch whose photons come in direction order from a common origin (tools.argsort_direction, as
// chroma/benchmark.py:80-82 prepares them; a photon bomb; the Cherenkov cone of a track) is a third of the ray-cast
// time of the whole batch, and its rays are as coherent as rays get: the 64 rays of consecutive slots cross the same
// nodes down to the last levels of the tree.  k_raycast_quad cannot use that -- every ray keeps its own stack and pays
// the per-visit bookkeeping alone.  Here a wavefront IS a packet: ONE traversal stack (LDS), ONE node fetch per visit
// for all 64 rays (the node's eight entries are wave-uniform: scalar loads, SGPRs), every lane tests the eight boxes
// against its own ray, leaf triangles are tested at once by all lanes whose ray enters the leaf box (64 of 64 lanes on a
// uniform triangle record instead of 7-12 of 64), and the bookkeeping of a visit -- order of the children, push, pop --
// is wave-uniform scalar work done once for 64 rays.
// Same tree, same slab test, same (distance, rank) rule: a lane tests exactly the triangles whose leaf entry its OWN
// ray passes in nodes its own ray entered (a stack entry carries the mask of the lanes that passed the node's box; the
// others sit the visit out), so the argument of DESIGN.md section 3.1 applies lane by lane and the result is the
// quad walk's bit for bit (tests/test_gpu_packet.py) -- whatever the rays look like.  Only the SPEED depends on their
// coherence: a packet of unrelated rays visits the union of 64 traversals with a few lanes active each time, so the
// kernel can be switched in where the photons say they are coherent (k_load_working counts the waves whose rays share
// an origin and lie within a narrow cone; chroma_propagate's first step only).
// MEASURED (profiles/r03/ab_packet_first_step.txt, pmc_packet.txt): 31.3 ms for the 1e8 direction-sorted rays of a C3
// batch's first step against 29.5 ms for k_raycast_quad -- the slab work per (ray, entry) pair is the same in both, and
// what a packet saves in bookkeeping it pays for the UNION of its rays' paths (~40 nodes, ~35 triangles per packet where
// one ray needs 19 and 9.4).  So it is an opt-in (CHROMA_PACKET=on|auto, chroma_set_packet), off by default.
#ifndef PACKET_STACK
#define PACKET_STACK 96      // entries of the packet's stack in LDS (node, box distance, lane mask): deeper trees keep the quad walk
#endif
// (wave-uniform reads through the constant address space: the compiler emits scalar loads, the data lands in SGPRs)
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) u32x4_t *const_u32x4_p;
typedef const __attribute__((address_space(4))) f32x4_t *const_f32x4_p;

// minimum over the 64 lanes (every lane active), for non-negative floats and +inf
__device__ inline float wave_min_f32(float v)
{
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));     // quad_perm [1,0,3,2]
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));     // quad_perm [2,3,0,1]
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));    // row_half_mirror
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false)));    // row_mirror
    const float a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), d = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return __builtin_fminf(__builtin_fminf(a, b), __builtin_fminf(c, d));
}

template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) __attribute__((amdgpu_waves_per_eu(8, 8))) void
k_raycast_packet(GeoView g, const float4 *rays, StepState *st, int32_t *hit_triangle, float *hit_distance,
                 uint32_t *retry_list, DeviceCounters *counters, const uint32_t *use_packet)
{
    // (launched beside k_raycast_quad: the step's photons decide on the device which of the two has work to do)
    if (*use_packet == 0u) return;
    const uint32_t nthreads = st->n;
    static_assert(PROP_BLOCK == WAVE, "one wave per workgroup");
    __shared__ uint32_t s_node[PACKET_STACK];
    __shared__ float s_t[PACKET_STACK];
    __shared__ unsigned long long s_mask[PACKET_STACK];
    const unsigned lane = lane_id();
    const unsigned long long lane_bit = 1ull << lane;
    const float inf = cm_inff();
    LaneCounters cnt = {0, 0, 0, 0};
    const const_u32x4_p wnodes = (const_u32x4_p)(uintptr_t)g.wnodes;
    const const_f32x4_p tris = (const_f32x4_p)(uintptr_t)g.tri;

    for (;;) {
        uint32_t base = 0;
        if (lane == 0) base = atomicAdd(&st->work, (uint32_t)WAVE);
        base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
        if (base >= nthreads) break;
        const uint32_t slot = base + lane;
        // ---- this lane's ray
        bool on = false;
        float ox = 0.f, oy = 0.f, oz = 0.f, dx = 0.f, dy = 0.f, dz = 1.f;
        float rax = 0.f, ray_ = 0.f, raz = 0.f;
        f32x2 rbx = {0.f, 0.f}, rby = {0.f, 0.f}, rbz = {0.f, 0.f};
        uint32_t rsx = 0, rsy = 0, rsz = 0;
        int last_hit = -1;
        if (slot < nthreads) {
            const float4 *r = rays + 4 * (size_t)slot;
            const float4 r0 = r[0], r1 = r[1];
            const int status = __float_as_int(r1.w);
            if (status == 0) {
                const float4 r2 = r[2], r3 = r[3];
                ox = r0.x; oy = r0.y; oz = r0.z; dx = r1.x; dy = r1.y; dz = r1.z;
                last_hit = __float_as_int(r0.w);
                rax = r2.x; ray_ = r2.y; raz = r2.z;
                const float mx = r2.w * cm_fabsf(rax), my = r2.w * cm_fabsf(ray_), mz = r2.w * cm_fabsf(raz);
                rbx = (f32x2){r3.x - mx, r3.x + mx}; rby = (f32x2){r3.y - my, r3.y + my}; rbz = (f32x2){r3.z - mz, r3.z + mz};
                rsx = rax < 0.f ? 16u : 0u; rsy = ray_ < 0.f ? 16u : 0u; rsz = raz < 0.f ? 16u : 0u;
                on = true;
            } else {                                         // HIT_NAN, or HIT_RETRY: 1/d not moderate (as k_raycast_quad settles them)
                hit_triangle[slot] = status;
                hit_distance[slot] = 0.0f;
                if (status == HIT_RETRY) retry_list[atomicAdd(&st->retry, 1u)] = slot;
            }
        }
        int triangle_index = -1;
        uint32_t best_rank = 0;
        float prune_t = inf;
        // ---- the packet's traversal: wave-uniform control flow from here to the end of the packet
        int sp = 0;
        uint32_t cur = 0u;
        unsigned long long cur_mask = __ballot(on);
        bool have = cur_mask != 0ull;
        while (have) {
            const bool here = (cur_mask & lane_bit) != 0ull;        // this lane's ray entered the node
            uint4 e[8];
#pragma unroll
            for (int j = 0; j < 8; j++) { const u32x4_t v = wnodes[8 * (size_t)cur + j]; e[j] = make_uint4(v.x, v.y, v.z, v.w); }
            if (COUNT && here) cnt.nodes += 8;
            uint32_t nxt = WIDE_NONE;
            float nxt_t = inf;
            unsigned long long nxt_mask = 0ull;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const uint32_t w = e[j].w;
                if (w == WIDE_NONE) continue;                        // (uniform)
                float tn, tf;
                box_interval_signed(rax, ray_, raz, rsx, rsy, rsz, rbx, rby, rbz, e[j], tn, tf);
                const bool pass = here & !(tn > tf) & !(tn > prune_t);
                if ((int)w < 0) {                                    // a triangle (uniform)
                    const uint32_t rec = w & 0x7FFFFFFFu;
                    const bool test = pass & ((int)rec != last_hit);
                    if (__any(test)) {
                        const f32x4_t a = tris[TRI_STRIDE * (size_t)rec], b = tris[TRI_STRIDE * (size_t)rec + 1], c = tris[TRI_STRIDE * (size_t)rec + 2];
                        if (test) {
                            if (COUNT) cnt.tris++;
                            float distance;
                            if (intersect_triangle(mk3(ox, oy, oz), mk3(dx, dy, dz), mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance)) {
                                const uint32_t rank = __float_as_uint(c.w);
                                if (distance < prune_t || (distance == prune_t && rank < best_rank)) {
                                    triangle_index = (int)rec;
                                    prune_t = distance;
                                    best_rank = rank;
                                }
                            }
                        }
                    }
                } else {
                    const unsigned long long m = __ballot(pass);
                    if (m != 0ull) {
                        const float t = wave_min_f32(pass ? tn : inf);      // the box distance of the nearest of the rays that enter
                        uint32_t pn = w; float pt = t; unsigned long long pm = m;
                        if (t < nxt_t) { pn = nxt; pt = nxt_t; pm = nxt_mask; nxt = w; nxt_t = t; nxt_mask = m; }
                        if (pn != WIDE_NONE) {
                            if (sp < PACKET_STACK) { s_node[sp] = pn; s_t[sp] = pt; s_mask[sp] = pm; }
                            sp++;
                        }
                    }
                }
            }
            cur = nxt;
            cur_mask = nxt_mask;
            have = cur != WIDE_NONE;
            // next entry that can still hold a nearer hit for one of the rays that entered its box
            while (!have && sp > 0) {
                sp--;
                if (sp >= PACKET_STACK) continue;                    // (cannot happen: the host checked the tree's need)
                const float t = s_t[sp];
                const unsigned long long m = s_mask[sp] & __ballot(!(t > prune_t));
                if (m != 0ull) { cur = s_node[sp]; cur_mask = m; have = true; }
            }
        }
        if (on) {
            hit_triangle[slot] = triangle_index;
            hit_distance[slot] = triangle_index == -1 ? -1.0f : prune_t;
            if (COUNT) cnt.steps++;
        }
    }
    if (COUNT) {
        unsigned long long nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris), ry = wave_sum_u64(cnt.steps);
        if (lane == 0) {
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
            atomicAdd(&counters->packet_nodes, nd);
            atomicAdd(&counters->packet_tris, tr);
            atomicAdd(&counters->packet_rays, ry);
        }
    }
}

// which ray cast takes the first step: the packet kernel when three quarters of the waves are coherent (and the batch
// is large enough for its persistent grid); `mode` 1 = always, 0 = never (CHROMA_PACKET=on|off)
__global__ void k_packet_decide(const uint32_t *coherence, uint32_t *use_packet, uint64_t n, int mode)
{
    uint32_t use = 0u;
    if (mode == 1) use = 1u;
    else if (mode == 2) use = (n >= (1u << 18) && coherence[1] > 0u && 4ull * coherence[0] >= 3ull * coherence[1]) ? 1u : 0u;
    *use_packet = use;
}
