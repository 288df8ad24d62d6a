// kernel_physics.h -- k_physics: one step of photon physics for every queued photon.
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

#ifndef PHYS_BLOCK
#define PHYS_BLOCK 512
#endif
#ifndef PHYS_WAVES_PER_EU
#define PHYS_WAVES_PER_EU 4
#endif
#ifndef PHYS_SORT
#define PHYS_SORT 1          // all-models build: deal the slots of a block to its threads by the kind of surface hit
#endif
#define PHYS_CLASSES 8
#ifndef PHYS_PLAIN_WAVES_PER_EU
#define PHYS_PLAIN_WAVES_PER_EU 5   // 96 VGPRs, no scratch (round 3: the photon's record is asked for together with its hit entry);
#endif                              // five waves per SIMD need blocks of FOUR waves -- 20 waves per CU are five such blocks, but only two of eight
#ifndef PHYS_PLAIN_BLOCK
#define PHYS_PLAIN_BLOCK 256        // -2 ms per C3 step against 512 threads at 4 waves (profiles/r03/ab_physics_occupancy.txt)
#endif
#define PHYS_BLOCK_OF(FULL) ((FULL) ? PHYS_BLOCK : PHYS_PLAIN_BLOCK)
template <bool FULL>
__global__ __launch_bounds__(PHYS_BLOCK_OF(FULL)) __attribute__((amdgpu_waves_per_eu(FULL ? PHYS_WAVES_PER_EU : PHYS_PLAIN_WAVES_PER_EU))) void
k_physics(GeoView g, PhotonView pv, StepState *st, const float4 *work_in, uint32_t *output_queue, float4 *work_out,
          const int32_t *hit_triangle, const float *hit_distance, uint64_t seed, uint64_t id_base,
          int use_weights, int scatter_first, uint32_t *retry_list, int fixup, DeviceCounters *counters, float4 *rays_next,
          float4 *final_rec = nullptr, uint32_t epoch = 0u, int literal_rays = 0)
{
    // (`literal_rays`: the survivors' next ray records carry 1/d and -o/d, what k_raycast_literal reads: the exact walk)
    // Two passes per step.  Main pass (fixup = 0): every slot of the working set; a slot the ray cast
    // handed to the strict walk (HIT_RETRY) is left alone, and so is a hit that is not REGULAR
    // (record_hit_is_regular, propagate_device.h): its slot joins retry_list.  Fix-up pass (fixup = 1),
    // after k_raycast_retry has walked those rays the reference's way: the listed slots only, results
    // taken as they are.  A photon that survives the step is appended to the next working set; one that
    // ends here is written to the caller's arrays (the only time they are touched).
    constexpr int BLOCK = PHYS_BLOCK_OF(FULL);
    __shared__ uint32_t s_counts[BLOCK / WAVE + 1];
    // The 512 slots of a round are dealt to the threads BY THE SURFACE THEY HIT (the material code of the winning
    // triangle's record): what a photon does at a black wall, at PMT glass, at the photocathode, at a mirror, a thin
    // film or a wavelength shifter are different, long branches, and a wave that holds all kinds executes them all.
    // Sorted, most waves hold one kind and skip the rest.  Only slot numbers move (through LDS).  In the ALL-MODELS
    // build only (-8 % at C5): with plain optics the step's divergence is in the bulk, not at the surface, and the four
    // barriers and the extra gather of the sort cost 2 ms per C3 step (profiles/r02/ab_physics_sort.txt).
    constexpr bool SORT = FULL && (PHYS_SORT != 0);
    __shared__ uint32_t s_class_count[SORT ? BLOCK / WAVE : 1][PHYS_CLASSES];
    __shared__ int32_t s_perm[SORT ? BLOCK : 1];
    // (fixup = 2, the literal walk: every slot, results taken as they are -- every ray took the reference's own loop)
    const int nthreads = fixup == 1 ? (int)st->retry : (int)st->n, renorm = (int)st->renorm, renorm_next = st->in_tail ? 0 : 1;
    unsigned long long nsteps = 0;
    // the grid is sized for an upper bound of the photon count: blocks stride over the slots
    for (int block_base = blockIdx.x * BLOCK; block_base < nthreads; block_base += gridDim.x * BLOCK) {
    int id = block_base + threadIdx.x;
    bool alive = false;
    uint32_t photon_id = 0;
    Photon p;
    uint32_t counter = 0;
    int last_hit_record = -1;
    int sorted_slot = (id < nthreads) ? id : -1;
    if (SORT && !fixup) {
        // class of this thread's own slot: 0 = nothing to do here (no slot, miss, NaN, retry), else 1 + surface kind
        uint32_t cls = 0;
        if (id < nthreads) {
            const int tri0 = hit_triangle[id];
            if (tri0 >= 0) {
                const uint32_t code = __float_as_uint(g.tri[TRI_STRIDE * (size_t)tri0].w);
                const int surface = convert(0xFF & (code >> 8));                       // -1: no surface (a material boundary)
                cls = 1u + (uint32_t)min(surface + 1, PHYS_CLASSES - 2);
            }
        }
        const unsigned lane = lane_id(), wave = threadIdx.x / WAVE;
        uint32_t my_rank = 0;
#pragma unroll
        for (uint32_t c = 0; c < PHYS_CLASSES; c++) {
            const unsigned long long m = __ballot(cls == c);
            if (cls == c) my_rank = (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) s_class_count[wave][c] = (uint32_t)__popcll(m);
        }
        __syncthreads();
        // start of (class, wave) in the sorted order -- classes in order, waves within a class: one wave scans the 64 counts
        static_assert(!SORT || PHYS_CLASSES * (BLOCK / WAVE) == WAVE, "one lane per (class, wave) pair");
        if (wave == 0) {
            const uint32_t c = lane / (BLOCK / WAVE), w = lane % (BLOCK / WAVE);
            const uint32_t k = s_class_count[w][c];
            uint32_t incl = k;
            for (int off = 1; off < WAVE; off <<= 1) { const uint32_t v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
            s_class_count[w][c] = incl - k;
        }
        __syncthreads();
        s_perm[s_class_count[wave][cls] + my_rank] = (id < nthreads) ? id : -1;
        __syncthreads();
        sorted_slot = s_perm[threadIdx.x];
        __syncthreads();               // (the tables are rewritten by the next round)
    }
    if (sorted_slot >= 0) {
        const int slot = fixup == 1 ? (int)retry_list[sorted_slot] : sorted_slot;
        // (the photon's record is asked for TOGETHER with its hit entry, not after it: one memory latency less in the
        //  chain of a round; the few slots that turn out to be HIT_RETRY read 64 bytes for nothing)
        const float4 *w = work_in + 4 * (size_t)slot;
        const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        int tri = hit_triangle[slot];
        const float hit_dist = hit_distance[slot];
        if (tri != HIT_RETRY) {
            photon_id = __float_as_uint(w3.w);
            p.position = mk3(w0.x, w0.y, w0.z);
            p.direction = mk3(w1.x, w1.y, w1.z);
            p.polarization = mk3(w2.x, w2.y, w2.z);
            if (renorm) {
                p.direction = p.direction / norm(p.direction);           // (the same arithmetic as k_ray_setup: the ray's)
                p.polarization = p.polarization / norm(p.polarization);
            }
            if (!fixup && tri >= 0) {
                // is the fast walk's winner one the reference is sure to find too?
                const float4 *t = g.tri + TRI_STRIDE * (size_t)tri;
                // (the plain build leaves the exact question to k_raycast_retry: it is rare there -- hits within ulps of
                //  a leaf box's upper face -- and its divisions cost registers; the all-models build asks it here, because
                //  the geometries it serves (faces ON the world box: every hit "near a face") would send everything round)
                const float4 ta = t[0], tb = t[1], tc = t[2];
                const bool regular = FULL ? record_hit_is_regular(g, ta, tb, tc, p.position, p.direction, hit_dist)
                                          : record_hit_is_plainly_regular(g, ta, tb, tc, p.position, p.direction, hit_dist);
                if (!regular) {
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
            p.last_hit_triangle = -1;                // (set by apply_hit_dev)
            p.evidx = 0;
            last_hit_record = __float_as_int(w3.z);
            cm_rng rng;
            cm_rng_init(&rng, seed, id_base + photon_id, __float_as_uint(w3.y));
            if (tri == HIT_NAN) {
                // the last hit stays what it was (propagate.cu:270-273)
                p.last_hit_triangle = last_hit_record >= 0 ? (int)g.dev_to_tri[last_hit_record] : -1;
                p.history |= CHROMA_NO_HIT | CHROMA_NAN_ABORT;
            } else {
                State s;
                apply_hit_dev(s, p, g, tri, hit_dist);
                if (tri != -1) step_after_hit<FULL>(p, s, rng, g, use_weights != 0, scatter_first);
                // (a photon scattered or absorbed in the bulk forgets the triangle, photon.h:232,262,283)
                last_hit_record = (p.last_hit_triangle < 0) ? -1 : tri;
            }
            counter = rng.counter;
            alive = (p.history & CHROMA_TERMINAL_MASK) == 0;
            if (!alive) {
                if (final_rec) {
                    // (chroma_propagate_hits: one 64-byte record at the photon's id -- a full sector instead of fifteen scattered
                    //  4-byte stores; k_finalize_hits fills the caller's arrays from it in a streaming pass and extracts the hits)
                    float4 *f = final_rec + 4 * (size_t)photon_id;
                    f[0] = make_float4(p.position.x, p.position.y, p.position.z, p.wavelength);
                    f[1] = make_float4(p.direction.x, p.direction.y, p.direction.z, p.time);
                    f[2] = make_float4(p.polarization.x, p.polarization.y, p.polarization.z, p.weight);
                    f[3] = make_float4(__uint_as_float(p.history), __uint_as_float(counter), __int_as_float(p.last_hit_triangle), __uint_as_float(epoch));
                } else {
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
        }
    }
    const uint32_t at = block_queue_append<BLOCK / WAVE>(output_queue, alive, photon_id, s_counts);
    if (alive) {
        float4 *w = work_out + 4 * (size_t)(at - 1u);
        w[0] = make_float4(p.position.x, p.position.y, p.position.z, p.wavelength);
        w[1] = make_float4(p.direction.x, p.direction.y, p.direction.z, p.time);
        w[2] = make_float4(p.polarization.x, p.polarization.y, p.polarization.z, p.weight);
        w[3] = make_float4(__uint_as_float(p.history), __uint_as_float(counter), __int_as_float(last_hit_record), __uint_as_float(photon_id));
        // the survivor's ray for the next step (see k_ray_setup): the next launch re-normalises unless the reference's
        // last launch has begun -- which k_step_begin of THIS step has already decided
        if (rays_next) make_ray_record(g, rays_next + 4 * (size_t)(at - 1u), p.position, p.direction, renorm_next, last_hit_record, literal_rays != 0);
    }
    __syncthreads();        // s_counts is reused by the next round
    }
    if (counters) {
        nsteps = wave_sum_u64(nsteps);
        if (lane_id() == 0 && nsteps) atomicAdd(&counters->photon_steps, nsteps);
    }
}
