// experimental/physics_deal.h -- k_physics_deal: the photons of a block dealt to its threads by outcome (profiles/r03/ab_physics_deal.txt)
// Built only into build_variants/libchroma_hip_experimental.so (-DCHROMA_EXPERIMENTAL=1, `make variants`): measured, parity-green,
// and NOT faster than the product path (the A/B records are named below).  The product library does not contain this code.
#pragma once

// ---- k_physics for plain optics, with the photons of a block DEALT BY WHAT HAPPENS TO THEM -----------------------------
// k_physics<false> issues VALU work in 78 % of its SIMD cycles at 42 % lane utilisation (profiles/pmc_traffic.json): what a
// photon does in a step -- absorbed in the bulk, Rayleigh-scattered, met by a surface (detected, absorbed, reflected), or
// refracted / reflected at a material boundary -- is decided by its own random draws, and a wave that holds all kinds runs
// every branch with a fraction of its lanes.  Here a round has two halves.  First half, every thread on its own slot: hit
// entry, photon record, the check that the reference tests the winner, the triangle, the optical constants, the two
// distance draws and the DECISION (propagate_to_boundary with the scattering itself deferred).  Then the photons of the
// block change threads through LDS -- 23 words each: the photon, the draw counter, the surface normal and the two indices of
// refraction -- so that threads t = 0, 1, 2, ... hold first all photons that scatter, then all that meet a material
// boundary, then those at a surface, then those that ended, and the second half (rayleigh_scatter / propagate_at_boundary /
// propagate_at_surface, then the survivor's next ray record or the ended photon's stores) runs on waves that mostly hold one
// kind.  Same functions, same arguments, same draws in the same order (the generator is re-seeded from the photon's draw
// counter after the move: one Philox block): the results are k_physics<false>'s bit for bit; only the ORDER in which a
// block appends its survivors changes, which nothing depends on.
// RESULT (profiles/r03/ab_physics_deal.txt, C3): 0.072-0.073 s per 3 steps outside the ray cast with the deal (512-thread
// blocks; 256: 0.070; 1024: 0.091) against 0.070-0.071 s for k_physics<false> -- the exchange (23 LDS words each way, three
// more barriers, one more Philox block) costs what the purer waves save.  Off by default; kept as a build option.
#ifndef PHYS_DEAL
#define PHYS_DEAL 0       // MEASURED (profiles/r03/ab_physics_deal.txt): parity-green, and no faster -- see below
#endif
#ifndef PHYS_DEAL_BLOCK
#define PHYS_DEAL_BLOCK 512
#endif
#ifndef PHYS_DEAL_WAVES_PER_EU
#define PHYS_DEAL_WAVES_PER_EU 4
#endif
#define DEAL_WORDS 23
#define DEAL_CLASSES 5       // 0 scatter, 1 material boundary, 2 surface, 3 ended (stores only), 4 nothing to do
__global__ __launch_bounds__(PHYS_DEAL_BLOCK) __attribute__((amdgpu_waves_per_eu(PHYS_DEAL_WAVES_PER_EU))) void
k_physics_deal(GeoView g, PhotonView pv, StepState *st, const float4 *work_in, uint32_t *output_queue, float4 *work_out,
               const int32_t *hit_triangle, const float *hit_distance, uint64_t seed, uint64_t id_base,
               int use_weights, int scatter_first, uint32_t *retry_list, int fixup, DeviceCounters *counters, float4 *rays_next)
{
    constexpr int BLOCK = PHYS_DEAL_BLOCK, NW = BLOCK / WAVE;
    __shared__ uint32_t s_counts[NW + 1];
    __shared__ uint32_t s_x[DEAL_WORDS][BLOCK];
    __shared__ uint32_t s_class[NW][DEAL_CLASSES];          // per wave: photons of each class, then where they start
    __shared__ uint32_t s_start[DEAL_CLASSES + 1];
    const int nthreads = fixup == 1 ? (int)st->retry : (int)st->n, renorm = (int)st->renorm, renorm_next = st->in_tail ? 0 : 1;
    const unsigned lane = lane_id(), wave = threadIdx.x / WAVE;
    unsigned long long nsteps = 0;
    for (int block_base = blockIdx.x * BLOCK; block_base < nthreads; block_base += gridDim.x * BLOCK) {
        // ---- first half: this thread's own slot, up to the decision
        const int id = block_base + (int)threadIdx.x;
        uint32_t cls = 4u;
        Photon p;
        State s;
        uint32_t photon_id = 0, counter = 0;
        int tri = HIT_RETRY;
        p.position = p.direction = p.polarization = mk3(0.f, 0.f, 0.f);
        p.wavelength = p.time = p.weight = 0.f; p.history = 0u; p.last_hit_triangle = -1; p.evidx = 0u;
        s.surface_normal = mk3(0.f, 0.f, 0.f); s.refractive_index1 = s.refractive_index2 = 0.f; s.surface_index = -1;
        if (id < nthreads) {
            const int slot = fixup == 1 ? (int)retry_list[id] : id;
            const float4 *w = work_in + 4 * (size_t)slot;
            const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
            tri = hit_triangle[slot];
            const float hit_dist = hit_distance[slot];
            if (tri != HIT_RETRY) {
                photon_id = __float_as_uint(w3.w);
                p.position = mk3(w0.x, w0.y, w0.z);
                p.direction = mk3(w1.x, w1.y, w1.z);
                p.polarization = mk3(w2.x, w2.y, w2.z);
                if (renorm) {
                    p.direction = p.direction / norm(p.direction);
                    p.polarization = p.polarization / norm(p.polarization);
                }
                if (!fixup && tri >= 0) {
                    const float4 *t = g.tri + TRI_STRIDE * (size_t)tri;
                    if (!record_hit_is_plainly_regular(g, t[0], t[1], t[2], p.position, p.direction, hit_dist)) {
                        retry_list[atomicAdd(&st->retry, 1u)] = (uint32_t)slot;
                        tri = HIT_RETRY;
                    }
                }
            }
            if (tri != HIT_RETRY) {
                if (tri != HIT_NAN) nsteps++;
                p.wavelength = w0.w;
                p.time = w1.w;
                p.weight = w2.w;
                p.history = __float_as_uint(w3.x);
                cm_rng rng;
                cm_rng_init(&rng, seed, id_base + photon_id, __float_as_uint(w3.y));
                if (tri == HIT_NAN) {
                    const int lhr = __float_as_int(w3.z);                       // the last hit stays what it was (propagate.cu:270-273)
                    p.last_hit_triangle = lhr >= 0 ? (int)g.dev_to_tri[lhr] : -1;
                    p.history |= CHROMA_NO_HIT | CHROMA_NAN_ABORT;
                    cls = 3u;
                } else {
                    apply_hit_dev(s, p, g, tri, hit_dist);
                    if (tri == -1) cls = 3u;                                     // NO_HIT: ended
                    else {
                        const int cmd = propagate_to_boundary<false, true>(p, s, rng, g, use_weights != 0, scatter_first);
                        cls = cmd == CMD_BREAK ? 3u : cmd == CMD_SCATTER ? 0u : (s.surface_index != -1 ? 2u : 1u);
                    }
                }
                counter = rng.counter;
            }
        }
        // ---- the deal: position of this photon among the block's, classes in order, waves in order within a class
        uint32_t my_rank = 0;
#pragma unroll
        for (uint32_t c = 0; c < DEAL_CLASSES; c++) {
            const unsigned long long m = __ballot(cls == c);
            if (cls == c) my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_class[wave][c] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            uint32_t run = 0;
            for (uint32_t c = 0; c < DEAL_CLASSES; c++) {
                s_start[c] = run;
                for (uint32_t w = 0; w < (uint32_t)NW; w++) { const uint32_t k = s_class[w][c]; s_class[w][c] = run; run += k; }
            }
            s_start[DEAL_CLASSES] = run;
        }
        __syncthreads();
        {
            const uint32_t dst = s_class[wave][cls] + my_rank;
            s_x[0][dst] = __float_as_uint(p.position.x); s_x[1][dst] = __float_as_uint(p.position.y); s_x[2][dst] = __float_as_uint(p.position.z);
            s_x[3][dst] = __float_as_uint(p.direction.x); s_x[4][dst] = __float_as_uint(p.direction.y); s_x[5][dst] = __float_as_uint(p.direction.z);
            s_x[6][dst] = __float_as_uint(p.polarization.x); s_x[7][dst] = __float_as_uint(p.polarization.y); s_x[8][dst] = __float_as_uint(p.polarization.z);
            s_x[9][dst] = __float_as_uint(p.wavelength); s_x[10][dst] = __float_as_uint(p.time); s_x[11][dst] = __float_as_uint(p.weight);
            s_x[12][dst] = p.history; s_x[13][dst] = (uint32_t)p.last_hit_triangle; s_x[14][dst] = photon_id; s_x[15][dst] = counter;
            s_x[16][dst] = (uint32_t)tri;
            s_x[17][dst] = __float_as_uint(s.surface_normal.x); s_x[18][dst] = __float_as_uint(s.surface_normal.y); s_x[19][dst] = __float_as_uint(s.surface_normal.z);
            s_x[20][dst] = __float_as_uint(s.refractive_index1); s_x[21][dst] = __float_as_uint(s.refractive_index2); s_x[22][dst] = (uint32_t)s.surface_index;
        }
        __syncthreads();
        // ---- second half: thread t takes the photon dealt to position t
        const uint32_t t = threadIdx.x;
        const uint32_t mine = t < s_start[1] ? 0u : t < s_start[2] ? 1u : t < s_start[3] ? 2u : t < s_start[4] ? 3u : 4u;
        bool alive = false;
        int last_hit_record = -1;
        if (mine < 4u) {
            p.position = mk3(__uint_as_float(s_x[0][t]), __uint_as_float(s_x[1][t]), __uint_as_float(s_x[2][t]));
            p.direction = mk3(__uint_as_float(s_x[3][t]), __uint_as_float(s_x[4][t]), __uint_as_float(s_x[5][t]));
            p.polarization = mk3(__uint_as_float(s_x[6][t]), __uint_as_float(s_x[7][t]), __uint_as_float(s_x[8][t]));
            p.wavelength = __uint_as_float(s_x[9][t]); p.time = __uint_as_float(s_x[10][t]); p.weight = __uint_as_float(s_x[11][t]);
            p.history = s_x[12][t]; p.last_hit_triangle = (int)s_x[13][t]; photon_id = s_x[14][t]; counter = s_x[15][t];
            tri = (int)s_x[16][t];
            if (mine < 3u) {
                cm_rng rng;
                cm_rng_init(&rng, seed, id_base + photon_id, counter);
                if (mine == 0u) {
                    rayleigh_scatter(p, rng);
                    p.history |= CHROMA_RAYLEIGH_SCATTER;
                    p.last_hit_triangle = -1;
                } else {
                    s.surface_normal = mk3(__uint_as_float(s_x[17][t]), __uint_as_float(s_x[18][t]), __uint_as_float(s_x[19][t]));
                    s.refractive_index1 = __uint_as_float(s_x[20][t]); s.refractive_index2 = __uint_as_float(s_x[21][t]);
                    s.surface_index = (int)s_x[22][t];
                    int cmd = CMD_PASS;
                    if (mine == 2u) cmd = propagate_at_surface<false>(p, s, rng, g, use_weights != 0);
                    if (cmd == CMD_PASS) propagate_at_boundary(p, s, rng);
                }
                counter = rng.counter;
            }
            // (a photon scattered or absorbed in the bulk forgets the triangle, photon.h:232,262,283)
            last_hit_record = (p.last_hit_triangle < 0) ? -1 : tri;
            alive = (p.history & CHROMA_TERMINAL_MASK) == 0;
            if (!alive) {
                pv.rng_counters[photon_id] = counter;
                store3(pv.pos, photon_id, p.position);
                store3(pv.dir, photon_id, p.direction);
                store3(pv.pol, photon_id, p.polarization);
                pv.wavelengths[photon_id] = p.wavelength;
                pv.t[photon_id] = p.time;
                pv.flags[photon_id] = p.history;
                pv.last_hit_triangles[photon_id] = p.last_hit_triangle;
                pv.weights[photon_id] = p.weight;
            }
        }
        const uint32_t at = block_queue_append<NW>(output_queue, alive, photon_id, s_counts);
        if (alive) {
            float4 *w = work_out + 4 * (size_t)(at - 1u);
            w[0] = make_float4(p.position.x, p.position.y, p.position.z, p.wavelength);
            w[1] = make_float4(p.direction.x, p.direction.y, p.direction.z, p.time);
            w[2] = make_float4(p.polarization.x, p.polarization.y, p.polarization.z, p.weight);
            w[3] = make_float4(__uint_as_float(p.history), __uint_as_float(counter), __int_as_float(last_hit_record), __uint_as_float(photon_id));
            if (rays_next) make_ray_record(g, rays_next + 4 * (size_t)(at - 1u), p.position, p.direction, renorm_next, last_hit_record);
        }
        __syncthreads();        // s_counts, s_x and s_class are reused by the next round
    }
    if (counters) {
        nsteps = wave_sum_u64(nsteps);
        if (lane_id() == 0 && nsteps) atomicAdd(&counters->photon_steps, nsteps);
    }
}

