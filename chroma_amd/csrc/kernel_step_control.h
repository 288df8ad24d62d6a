// kernel_step_control.h -- one step as separate launches: hit codes, the device-side step block (k_step_begin), ray records (k_ray_setup).
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

// ---- one step as separate launches -----------------------------------------------------------------
// The ray cast needs few registers and benefits from many resident waves; the physics needs many
// registers and little time.  A ray-cast kernel writes (triangle record, distance) per queue slot,
// k_physics consumes them.  Together they perform exactly one iteration of the loop of k_propagate
// for every queued photon, with identical arithmetic (both re-normalise dir/pol on load like
// propagate.cu:248,250 when the step opens a launch in the reference's sense, see k_step_begin).
#define HIT_SKIP (-3)      // photon already terminal: untouched (propagate.cu:258)
#define HIT_NAN  (-2)      // NaN guard fired (propagate.cu:270-273)
#define HIT_RETRY (-4)     // the ray takes the literal reference walk (k_raycast_retry)

// ---- device-side step control ---------------------------------------------------------------------
// chroma_propagate enqueues its steps without waiting for any of them: how many photons a step has
// (the tail of its input queue), whether its launch re-normalises (the reference's launch policy,
// chroma/gpu/photon.py:225-252) and the ray-cast work counters live in StepState (top of this file),
// written by k_step_begin at the head of every step and read by the step's kernels.
// `first_n` (first step of a call only, else 0): the size of the caller's arrays.  The reference decides its
// FIRST launch on pos.size, photons that are already terminal included (gpu/photon.py:207,227), and every
// later one on the survivor count; a batch that is mostly terminal already therefore still gets a
// one-step launch first and is re-normalised again by the launch after it.
__global__ void k_step_begin(const uint32_t *in_queue, uint32_t *out_queue, StepState *st, uint32_t few, uint32_t first_n)
{
    const uint32_t n = in_queue[0] - 1u;
    st->n = n;
    uint32_t renorm = 1u;
    if (st->in_tail) renorm = 0u;
    else if ((first_n ? first_n : n) < few) st->in_tail = 1u;
    st->renorm = renorm;
    if (renorm && n) st->launches++;
    st->work = 0u;
    st->retry = 0u;
    out_queue[0] = 1u;
}

// ---- ray records --------------------------------------------------------------------------------------
// What a ray cast needs of a photon, prepared once per step by a streaming kernel instead of inside the
// persistent ray-cast kernels: there the set-up of a new ray (two dependent gathers, a normalisation,
// six IEEE divisions for the slab constants, the NaN and "moderate" checks) was ~300 instructions
// executed by the whole wave for the few rays being refilled -- a quarter of the kernel's VALU work.
// A record is 64 bytes at the queue slot: {origin, last hit record}, {direction, status},
// {a = scale/d}, {b = (world_origin - o)/d} (RayFast: blo = b - a, bhi = b + a).  Status 0 = cast; the
// other slots (NaN, 1/d not moderate) get their hit entry -- and their place in the retry list -- right
// here.  The photon comes from the dense working set (see k_load_working).
// the record of one ray at `r`; returns its status (0 = cast, HIT_NAN, HIT_RETRY)
// `literal`: the record of the exact walk (k_raycast_literal) carries the reference's own two per-ray constants, 1/d and
// -o/d (mesh.h:52-53), in place of the fused slab constants a and b.
__device__ inline int make_ray_record(const GeoView &g, float4 *r, v3 origin, v3 direction, int renorm, int last_hit, bool literal = false)
{
    int status;
    v3 a = mk3(0.f, 0.f, 0.f), b = mk3(0.f, 0.f, 0.f);
    if (renorm) direction = direction / norm(direction);
    if (cm_isnan(direction.x * direction.y * direction.z * origin.x * origin.y * origin.z)) {
        status = HIT_NAN;
    } else {
        v3 noid = (-origin) / direction;
        v3 inv_dir = 1.0f / direction;
        bool moderate = cm_fabsf(inv_dir.x) < 1e30f && cm_fabsf(inv_dir.y) < 1e30f && cm_fabsf(inv_dir.z) < 1e30f &&
                        cm_fabsf(noid.x) < 1e30f && cm_fabsf(noid.y) < 1e30f && cm_fabsf(noid.z) < 1e30f;
        if (!moderate) {
            status = HIT_RETRY;
        } else if (literal) {
            a = inv_dir;
            b = noid;
            status = 0;
        } else {
            a = ray_fast(g, noid, inv_dir, 1.0f).a;
            // b exactly as ray_fast forms it (blo = b - G a, bhi = b + G a are rebuilt by the kernels; G travels in r[2].w)
            b = mk3(cm_fmaf(g.world_origin[0], inv_dir.x, noid.x), cm_fmaf(g.world_origin[1], inv_dir.y, noid.y),
                    cm_fmaf(g.world_origin[2], inv_dir.z, noid.z));
            status = 0;
        }
    }
    r[0] = make_float4(origin.x, origin.y, origin.z, __int_as_float(last_hit));
    r[1] = make_float4(direction.x, direction.y, direction.z, __int_as_float(status));
    r[2] = make_float4(a.x, a.y, a.z, ray_growth(g, origin));
    r[3] = make_float4(b.x, b.y, b.z, 0.0f);
    return status;
}

// With the default walk this kernel does not run at all (round 2): k_load_working writes the records of the first
// step, k_physics those of every later one -- the photon is in their registers anyway, the launch policy of the next
// step is known (re-normalise unless the reference's last launch has begun: StepState::in_tail) -- and k_raycast_quad
// settles the few slots whose status is not 0 when it meets them (`settle`).  The cross-check walks keep it.
__global__ __launch_bounds__(256) void
k_ray_setup(GeoView g, const float4 *work, const StepState *st, float4 *rays,
            int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, uint32_t *retry_counter, int literal = 0)
{
    const int nthreads = (int)st->n, renorm = (int)st->renorm;
    for (int slot = blockIdx.x * blockDim.x + threadIdx.x; slot < nthreads; slot += gridDim.x * blockDim.x) {
        const float4 *w = work + 4 * (size_t)slot;
        const float4 w0 = w[0], w1 = w[1], w3 = w[3];
        const int status = make_ray_record(g, rays + 4 * (size_t)slot, mk3(w0.x, w0.y, w0.z), mk3(w1.x, w1.y, w1.z), renorm,
                                           __float_as_int(w3.z), literal != 0);
        if (status != 0) {
            hit_triangle[slot] = status;
            hit_distance[slot] = 0.0f;
            if (status == HIT_RETRY) retry_list[atomicAdd(retry_counter, 1u)] = (uint32_t)slot;
        }
    }
}

// ---- how a persistent ray-cast wave takes its rays (k_raycast_quad, k_raycast_literal) ---------------------------------------
// Most of a launch's rays are dealt out WITHOUT the work counter: the atomics of all waves on one word run at ~1e8 per second
// (every XCD's L2 hands them on to the memory side), and at 16-64 short rays per claim that rate IS the launch -- C5's 6-node
// tree: 0.19 ns per ray at any size; a small launch: two atomics per wave, 0.1 ms (profiles/r04/ab_static_claims.txt).
// Wave b of the W that take part owns the chunks b, b + W, b + 2W, ... of the first `eighths` / 8 of its share -- at least
// one -- and only the rest, what evens the waves out, goes through the counter; a launch of at most W chunks touches no counter
// at all.  `eighths` holds two shares: bits 0-3 for launches of big chunks (5: with 7 C3's big launches wait for their slowest
// waves, +5 %), bits 4-7 for the small ones (8).  0: every chunk through the counter (rounds 1-3).
struct WorkClaim {
    uint32_t n, chunk, left, stride, next_static, dyn_base;
    bool no_dynamic;
    __device__ WorkClaim(int nthreads, int chunk_, int eighths)
    {
        n = (uint32_t)nthreads; chunk = (uint32_t)chunk_;
        const uint32_t e = (uint32_t)((chunk_ == 16) ? (eighths >> 4) : (eighths & 15));
        const uint32_t nwaves = min((uint32_t)gridDim.x, (n + 15u) / 16u), nchunks = (n + chunk - 1u) / chunk;
        // (a share of 8/8 is the whole launch: every chunk owned, the last round of chunks by the first waves only -- a wave
        //  that must ask the counter just to learn that nothing is left costs the launch 6144 atomics, 70 us)
        left = e == 8u ? (nchunks + nwaves - 1u) / nwaves : e ? max(nchunks / nwaves * e / 8u, 1u) : 0u;
        stride = chunk * nwaves;
        dyn_base = left * stride;
        no_dynamic = e && dyn_base >= n;
        next_static = (uint32_t)blockIdx.x * chunk;
    }
    // first ray of the wave's next chunk (the caller clamps [base, base + chunk) to n); `exhausted`: there is none after it
    __device__ uint32_t next(uint32_t *counter, unsigned lane, bool &exhausted)
    {
        uint32_t base = 0;
        if (left) {
            base = next_static; next_static += stride; left--;
            if (!left && no_dynamic) exhausted = true;
        } else {
            if (lane == 0) base = atomicAdd(counter, chunk);
            base = dyn_base + (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
            if (base + chunk >= n) exhausted = true;
        }
        return base;
    }
};
