// kernel_propagate_fused.h -- k_propagate: the lane-per-photon fused multi-step kernel (chroma_propagate_step, tracking mode, CHROMA_TAIL=fused).
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

// propagate (chroma/cuda/propagate.cu:217-319): up to max_steps steps per photon in one launch.
// The step loop is wave-uniform (intersect_mesh votes across the wave): lanes whose photon has
// finished simply sit out the remaining ray casts of their wave.
template <int LDS_N, bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) void
k_propagate(GeoView g, PhotonView pv, int first_photon, int nthreads, const uint32_t *input_queue,
            uint32_t *output_queue, uint64_t seed, uint64_t id_base, int max_steps, int use_weights,
            int scatter_first, DeviceCounters *counters)
{
    __shared__ uint32_t s_lds[TRAV_LDS_WORDS(LDS_N, PROP_BLOCK)];
    uint32_t *lds = s_lds + threadIdx.x;

    int id = blockIdx.x * PROP_BLOCK + threadIdx.x;
    bool alive = false, loaded = false;
    uint32_t photon_id = 0;
    LaneCounters cnt = {0, 0, 0, 0};
    Photon p;
    cm_rng rng;
    State s;

    if (id < nthreads) {
        photon_id = input_queue ? input_queue[first_photon + id] : (uint32_t)(first_photon + id);
        p.position = load3(pv.pos, photon_id);
        p.direction = load3(pv.dir, photon_id);
        p.direction = p.direction / norm(p.direction);
        p.polarization = load3(pv.pol, photon_id);
        p.polarization = p.polarization / norm(p.polarization);
        p.wavelength = pv.wavelengths[photon_id];
        p.time = pv.t[photon_id];
        p.last_hit_triangle = pv.last_hit_triangles[photon_id];
        p.history = pv.flags[photon_id];
        p.weight = pv.weights[photon_id];
        p.evidx = pv.evidx[photon_id];
        if (!(p.history & CHROMA_TERMINAL_MASK)) {          // propagate.cu:258: terminal photons are left untouched
            loaded = true;
            cm_rng_init(&rng, seed, id_base + photon_id, pv.rng_counters[photon_id]);
        }
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
            } else if (COUNT) cnt.steps++;
        }
        float distance;
        int triangle = intersect_mesh<LDS_N, PROP_BLOCK, COUNT>(g, p.position, p.direction, distance, p.last_hit_triangle,
                                                                lds, cnt, stepping);
        if (stepping) {
            apply_hit(s, p, g, triangle, distance);
            if (triangle == -1) {
                live = false;
            } else {
                live = step_after_hit(p, s, rng, g, use_weights != 0, scatter_first);
                scatter_first = 0;
            }
        }
    }

    if (loaded) {
        pv.rng_counters[photon_id] = rng.counter;
        store3(pv.pos, photon_id, p.position);
        store3(pv.dir, photon_id, p.direction);
        store3(pv.pol, photon_id, p.polarization);
        pv.wavelengths[photon_id] = p.wavelength;
        pv.t[photon_id] = p.time;
        pv.flags[photon_id] = p.history;
        pv.last_hit_triangles[photon_id] = p.last_hit_triangle;
        pv.weights[photon_id] = p.weight;
        pv.evidx[photon_id] = p.evidx;
        alive = (p.history & CHROMA_TERMINAL_MASK) == 0;
    }
    if (output_queue) wave_queue_append(output_queue, alive, photon_id);

    unsigned long long ov = wave_sum_u64(cnt.overflows);
    if (COUNT) {
        unsigned long long st = wave_sum_u64(cnt.steps), nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane_id() == 0) {
            atomicAdd(&counters->photon_steps, st);
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
        }
    }
    if (lane_id() == 0 && ov) atomicAdd(&counters->stack_overflows, ov);
}
