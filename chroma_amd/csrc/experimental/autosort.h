// experimental/autosort.h -- chroma_set_autosort: the engine orders a caller's unsorted photons by direction itself (profiles/r03/ab_autosort.txt)
// Built only into build_variants/libchroma_hip_experimental.so (-DCHROMA_EXPERIMENTAL=1, `make variants`): measured, parity-green,
// and NOT faster than the product path (the A/B records are named below).  The product library does not contain this code.
#pragma once

// ---- a point-like source in no particular order ----------------------------------------------------------------------
// The first launches of a call take a third of a C3 step, and how long they take depends on whether the rays of a wave walk
// the same part of the tree: 29 ms with the photons in direction order against 39 ms in generation order (item 4 of round 3;
// the reference's own benchmark sorts its photons before the clock starts, chroma/benchmark.py:80-82).  A caller's photons
// are not sorted.  Nothing in the RESULT depends on the order in which the working set takes the photons up -- streams are
// keyed by photon id, results are stored by photon id -- so chroma_propagate chooses that order itself when it pays: a sample
// of the input says "one origin, directions all over the place" (a bomb, a calibration source), and the call is large.  Then
// the photons are ordered by a 16-bit direction cell (bvh_device.hip) and k_load_working gathers through that order.
// Photons that already are coherent, or that come from many places (tracks: their order is the caller's locality), are
// taken as they come.  chroma_set_autosort / CHROMA_AUTOSORT=off|on|auto: never (default), for every large call, by the probe.
// MEASURED (profiles/r03/ab_autosort.txt, C3, 1e8 photons of a bomb in generation order): 164 ms per batch with the engine's
// ordering against 128 ms with the photons taken as they come (and 114 ms when the caller hands them over sorted): the codes,
// the radix sort of 1e8 pairs and above all k_load_working GATHERING ten arrays through a random permutation (12-byte reads
// that each pull a 64-byte sector) cost 50 ms to win 15.  So the switch is OFF by default -- an opt-in with its parity test,
// like the packet kernel -- and the sorted order stays what the reference makes it: the caller's preparation
// (chroma_photons_sort_direction / GPUPhotons.sort_by_direction, outside the clock as in chroma/benchmark.py:80-82).
__global__ void k_order_probe(PhotonView pv, uint64_t n, uint32_t nsamples, uint32_t *out /* [0] waves of one origin, [1] of those: coherent, [2] waves looked at */)
{
    const uint32_t s = blockIdx.x * (blockDim.x / WAVE) + threadIdx.x / WAVE;
    if (s >= nsamples) return;
    const uint64_t start = (n / nsamples) * s / WAVE * WAVE;
    const uint64_t i = start + lane_id();
    if (start + WAVE > n) return;
    const v3 pos = load3(pv.pos, i), dir = load3(pv.dir, i);
    const float px = __shfl(pos.x, 0), py = __shfl(pos.y, 0), pz = __shfl(pos.z, 0);
    const float qx = __shfl(dir.x, 0), qy = __shfl(dir.y, 0), qz = __shfl(dir.z, 0);
    const float d2 = dir.x * dir.x + dir.y * dir.y + dir.z * dir.z, q2 = qx * qx + qy * qy + qz * qz;
    const float c = dir.x * qx + dir.y * qy + dir.z * qz;
    const bool same = fabsf(pos.x - px) + fabsf(pos.y - py) + fabsf(pos.z - pz) < 1.0f;
    const bool cone = c > 0.0f && c * c > 0.9975f * d2 * q2;
    const unsigned long long all_same = __ballot(same), all_cone = __ballot(cone);
    if (lane_id() == 0) {
        atomicAdd(out + 2, 1u);
        if (all_same == ~0ull) { atomicAdd(out, 1u); if (all_cone == ~0ull) atomicAdd(out + 1, 1u); }
    }
}
#ifndef AUTOSORT_MIN
#define AUTOSORT_MIN (1u << 21)
#endif
// *d_order: nullptr (take the photons as they come) or a chroma_malloc'ed permutation the caller frees after k_load_working
static int propagate_order(chroma_ctx *ctx, const CallOpts &co, const PhotonView &pv, uint64_t nphotons, uint32_t ncopies, uint32_t **d_order)
{
    *d_order = nullptr;
    const int mode = co.autosort;
    if (mode == 0 || ncopies != 1 || nphotons < AUTOSORT_MIN) return CHROMA_OK;
    if (mode == 2) {
        const uint32_t nsamples = 1024;
        HIP_TRY(hipMemsetAsync(ctx->d_words + 8, 0, 12, ctx->stream));
        hipLaunchKernelGGL(k_order_probe, dim3(nsamples / 4), dim3(256), 0, ctx->stream, pv, nphotons, nsamples, ctx->d_words + 8);
        uint32_t h[3];
        HIP_TRY(hipMemcpyAsync(h, ctx->d_words + 8, sizeof h, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        // one origin in nine sampled waves of ten, and fewer than half of them already coherent
        if (h[2] == 0 || 10ull * h[0] < 9ull * h[2] || 2ull * h[1] >= h[0]) return CHROMA_OK;
    }
    void *p = nullptr;
    int rc = chroma_malloc(ctx, (size_t)nphotons * 4, &p);
    if (rc != CHROMA_OK) return rc;
    rc = chroma_internal_direction_order(ctx, pv.dir, (uint32_t)nphotons, (uint32_t *)p);
    if (rc != CHROMA_OK) { chroma_free(ctx, p); return rc; }
    *d_order = (uint32_t *)p;
    return CHROMA_OK;
}

