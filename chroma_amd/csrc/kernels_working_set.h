// kernels_working_set.h -- the dense working set of live photons: k_load_working, k_store_working.
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

// ---- the dense working set ----------------------------------------------------------------------------
// While a batch propagates, its live photons are kept as 64-byte records ordered by queue slot:
// {pos, wavelength} {dir, time} {pol, weight} {flags, draw counter, last hit record, photon id}.  Steps
// read and append these records (streaming), so their traffic follows the number of survivors; the
// caller's SoA arrays are read once (here) and written once per photon (when it ends, or at the end of
// the call).  Working through the arrays instead made steps 2..5 touch nearly every line of every array
// for a fraction of the photons.
// k_load_working also is the initial queue of GPUPhotons.propagate (chroma/gpu/photon.py:206-216: the
// ncopies clones of a photon next to each other); photons that are already terminal are left out -- and
// thereby untouched (propagate.cu:258).
#ifndef LOAD_STAGE_LDS
#define LOAD_STAGE_LDS 1
#endif
__global__ __launch_bounds__(PHYS_BLOCK) void
k_load_working(GeoView g, PhotonView pv, uint32_t *queue, float4 *work, uint64_t n, uint32_t ncopies, uint32_t true_n, float4 *rays,
               uint32_t *coherence, const uint32_t *order = nullptr, int literal_rays = 0)
{
    // (`literal_rays`: ray records for k_raycast_literal -- 1/d and -o/d in place of the fused slab constants)
    // (`order`: take the photons up in this order instead of by index -- propagate_order below; ncopies == 1 then)
    // (`rays`: also the ray records of the first step -- the first launch of a call always re-normalises)
    // (`order` and `coherence` serve the experiments of csrc/experimental/ only: the product passes NULL, and their code is not built)
    // (`coherence`: [0] += waves whose photons share an origin and lie within a cone of 50 mrad, [1] += waves looked at:
    //  what decides between k_raycast_packet and k_raycast_quad for the first step.  A heuristic: it steers speed only.)
    __shared__ uint32_t s_counts[PHYS_BLOCK / WAVE + 1];
#if LOAD_STAGE_LDS
    __shared__ float4 s_stage[PHYS_BLOCK / WAVE][WAVE * 4];
#endif
#if CHROMA_EXPERIMENTAL
    uint32_t coh_yes = 0, coh_all = 0;
#endif
    for (uint64_t block_base = (uint64_t)blockIdx.x * PHYS_BLOCK; block_base < n; block_base += (uint64_t)gridDim.x * PHYS_BLOCK) {
        uint64_t j = block_base + threadIdx.x;
        bool take = false;
        uint32_t photon_id = 0, flags = 0;
        if (j < n) {
#if CHROMA_EXPERIMENTAL
            photon_id = order ? order[j] : (uint32_t)(j / ncopies) + (uint32_t)(j % ncopies) * true_n;
#else
            photon_id = (uint32_t)(j / ncopies) + (uint32_t)(j % ncopies) * true_n;
#endif
            flags = pv.flags[photon_id];
            take = (flags & CHROMA_TERMINAL_MASK) == 0;
        }
        const uint32_t at = block_queue_append<PHYS_BLOCK / WAVE>(queue, take, photon_id, s_counts);
#if LOAD_STAGE_LDS
        // The survivors of a wave land in consecutive slots (block_queue_append), 64 bytes each -- but a lane's four
        // 16-byte stores are 64 bytes apart from its neighbours': 64 partial lines per store instruction.  The records go
        // through LDS instead and leave as whole kilobytes: store i of the wave writes bytes [1024 i, 1024 (i + 1)) of
        // the wave's span.
        const unsigned long long tm = __ballot(take);
        const uint32_t nsurv = (uint32_t)__popcll(tm), rnk = (uint32_t)__popcll(tm & ((1ull << lane_id()) - 1ull));
        const uint32_t first_slot = nsurv ? (uint32_t)__shfl(at, __ffsll((long long)tm) - 1) - 1u : 0u;
        float4 *st = s_stage[threadIdx.x / WAVE];
        v3 pos = mk3(0.f, 0.f, 0.f), dir = mk3(0.f, 0.f, 1.f);
        int lh = -1;
        if (take) {
            pos = load3(pv.pos, photon_id); dir = load3(pv.dir, photon_id);
            const v3 pol = load3(pv.pol, photon_id);
            lh = pv.last_hit_triangles[photon_id];
            lh = (lh >= 0 && (uint32_t)lh < g.ntriangles) ? (int)g.tri_to_dev[lh] : -1;
            float4 *w = st + 4 * rnk;
            w[0] = make_float4(pos.x, pos.y, pos.z, pv.wavelengths[photon_id]);
            w[1] = make_float4(dir.x, dir.y, dir.z, pv.t[photon_id]);
            w[2] = make_float4(pol.x, pol.y, pol.z, pv.weights[photon_id]);
            w[3] = make_float4(__uint_as_float(flags), __uint_as_float(pv.rng_counters[photon_id]), __int_as_float(lh), __uint_as_float(photon_id));
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t q = lane_id(); q < 4u * nsurv; q += WAVE) work[4 * (size_t)first_slot + q] = st[q];
        __builtin_amdgcn_wave_barrier();
        if (rays) {
            if (take) make_ray_record(g, st + 4 * rnk, pos, dir, 1, lh, literal_rays != 0);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t q = lane_id(); q < 4u * nsurv; q += WAVE) rays[4 * (size_t)first_slot + q] = st[q];
            __builtin_amdgcn_wave_barrier();
        }
        if (take) {
#else
        if (take) {
            v3 pos = load3(pv.pos, photon_id), dir = load3(pv.dir, photon_id), pol = load3(pv.pol, photon_id);
            int lh = pv.last_hit_triangles[photon_id];
            lh = (lh >= 0 && (uint32_t)lh < g.ntriangles) ? (int)g.tri_to_dev[lh] : -1;
            float4 *w = work + 4 * (size_t)(at - 1u);
            w[0] = make_float4(pos.x, pos.y, pos.z, pv.wavelengths[photon_id]);
            w[1] = make_float4(dir.x, dir.y, dir.z, pv.t[photon_id]);
            w[2] = make_float4(pol.x, pol.y, pol.z, pv.weights[photon_id]);
            w[3] = make_float4(__uint_as_float(flags), __uint_as_float(pv.rng_counters[photon_id]), __int_as_float(lh), __uint_as_float(photon_id));
            if (rays) make_ray_record(g, rays + 4 * (size_t)(at - 1u), pos, dir, 1, lh, literal_rays != 0);
#endif
#if CHROMA_EXPERIMENTAL
            if (coherence) {
                // against the wave's first taken lane (the lanes of a wave land in consecutive slots)
                const unsigned long long m = __ballot(true);
                const int first = __ffsll((long long)m) - 1;
                const float px = __shfl(pos.x, first), py = __shfl(pos.y, first), pz = __shfl(pos.z, first);
                const float qx = __shfl(dir.x, first), qy = __shfl(dir.y, first), qz = __shfl(dir.z, first);
                const float d2 = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z, q2 = qx * qx + qy * qy + qz * qz;
                const float c = dir.x * qx + dir.y * qy + dir.z * qz;
                const bool near = fabsf(pos.x - px) + fabsf(pos.y - py) + fabsf(pos.z - pz) < 1.0f && c > 0.0f && c * c > 0.9975f * d2 * q2;
                const unsigned long long ok = __ballot(near);
                if ((int)lane_id() == first && __popcll(m) >= 32) { coh_all++; coh_yes += (ok == m) ? 1u : 0u; }
            }
#endif
        }
        __syncthreads();
    }
#if CHROMA_EXPERIMENTAL
    if (coherence) {
        if (coh_all) { atomicAdd(&coherence[1], coh_all); if (coh_yes) atomicAdd(&coherence[0], coh_yes); }
    }
#endif
}

// the photons still alive when the call ends go back to the caller's arrays
__global__ void k_store_working(GeoView g, PhotonView pv, const uint32_t *queue, const float4 *work)
{
    const uint32_t n = queue[0] - 1u;
    for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < n; slot += gridDim.x * blockDim.x) {
        const float4 *w = work + 4 * (size_t)slot;
        const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        const uint32_t photon_id = __float_as_uint(w3.w);
        const int rec = __float_as_int(w3.z);
        store3(pv.pos, photon_id, mk3(w0.x, w0.y, w0.z));
        store3(pv.dir, photon_id, mk3(w1.x, w1.y, w1.z));
        store3(pv.pol, photon_id, mk3(w2.x, w2.y, w2.z));
        pv.wavelengths[photon_id] = w0.w;
        pv.t[photon_id] = w1.w;
        pv.weights[photon_id] = w2.w;
        pv.flags[photon_id] = __float_as_uint(w3.x);
        pv.rng_counters[photon_id] = __float_as_uint(w3.y);
        pv.last_hit_triangles[photon_id] = rec >= 0 ? (int)g.dev_to_tri[rec] : -1;
    }
}
