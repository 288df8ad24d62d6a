// chroma_hip.hip -- kernels and C ABI of libchroma_hip.so (gfx950 / MI355X only).
//
// One photon per lane, 64-lane workgroups (one wavefront each) for the propagate kernel so a
// workgroup retires as soon as its own photons are done; survivors are re-queued with one atomic
// per wave (ballot compaction).  See DESIGN.md for the data layout and the kernel inventory.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>
#include <algorithm>
#include <chrono>
#include <atomic>
#include <mutex>
#include <unordered_map>
#include <map>

#include <dlfcn.h>
#include <rccl/rccl.h>          // types and prototypes only: RCCL itself is found with dlopen at first use

#include "propagate_device.h"
#include "wide_build.h"
#include "host_utils.h"

// ---------------------------------------------------------------------------------------------------
// error handling
// ---------------------------------------------------------------------------------------------------
static thread_local std::string g_last_error;

static int set_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#include "ctx_access.h"
extern "C" int chroma_internal_set_error(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return set_error((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct StepState {
    uint32_t n;          // photons queued for this step
    uint32_t renorm;     // this step opens a launch in the reference's sense: re-normalise dir/pol on load
    uint32_t in_tail;    // the reference's last launch (all remaining steps) has begun
    uint32_t launches;   // launches in the reference's sense so far
    uint32_t work;       // next unclaimed ray of the persistent ray cast
    uint32_t retry;      // rays left for k_raycast_retry
    uint32_t pad[2];
};

struct chroma_ctx {
    int device;
    hipStream_t stream;
    // queue ping-pong buffers for chroma_propagate (n+1 words each)
    uint32_t *queue_a = nullptr, *queue_b = nullptr;
    size_t queue_capacity = 0;
    // (triangle, distance) per queue slot handed from k_raycast to k_physics
    int32_t *hit_triangle = nullptr;
    float *hit_distance = nullptr;
    uint32_t *retry_list = nullptr;        // [capacity] queue slots handed to k_raycast_retry
    float4 *rays = nullptr;                // [capacity][4] ray records (k_ray_setup / k_load_working / k_physics)
    float4 *rays_b = nullptr;              // the records of the NEXT step while k_physics writes them (default walk)
    float4 *work_a = nullptr, *work_b = nullptr;    // [capacity][4] the dense working sets that go with queue_a / queue_b
    // chroma_propagate_hits: photons that end in k_physics leave as one 64-byte record at their id (final_rec, stamped with the
    // call's epoch); k_finalize_hits fills the caller's arrays from the records and extracts the hits in one pass
    float4 *final_rec = nullptr; size_t final_capacity = 0; uint32_t final_epoch = 0;
    float4 *final_use = nullptr;           // final_rec while a call uses the records, else NULL (k_physics stores to the arrays)
    // small device scratch: [0..3] DeviceCounters, then misc words
    DeviceCounters *d_counters = nullptr;
    uint32_t *d_words = nullptr;        // 16 words
    uint32_t *h_words = nullptr;        // pinned mirror
    int counting = 0;
    StepState *d_step = nullptr;           // device-side step control block (k_step_begin)
    uint32_t *h_step = nullptr;            // pinned copy for the occasional read-back
    int physics_blocks = 256 * 8;          // grid cap of k_physics (blocks stride over the queue)
    std::vector<hipEvent_t> step_events;   // 4 per step when kernels are timed
    int persistent_waves = 256 * 20 * 4;   // grid of the persistent ray-cast kernel (set from the device at init)
    int wide_waves = 256 * 14;             // same for k_raycast_wide (11 KB of LDS per wave)
    uint2 *wide_spill = nullptr;           // [wide_waves][WIDE_SPILL][64] stack entries beyond the LDS part
    int coop_waves = 256 * 32;             // grid of k_raycast_coop (2 KB of LDS per wave: wave slots limit residency)
    int quad_waves = 256 * 24;             // grid of k_raycast_quad
    int pair_waves = 256 * 20;             // grid of k_raycast_pair (32 rays per wave)
    uint2 *coop_spill = nullptr;           // [coop_waves][8][COOP_SPILL]
    int ray_chunk = 256, coop_chunk = 64;  // rays a persistent wave takes from the queue per atomic (big batches)
    int fused_tail = 1;                    // 0 (CHROMA_TAIL=split): the last photons also take one launch set per step
    int split_tail = 1;                    // 0 (CHROMA_TAIL=fused): chroma_propagate launches the fused kernel only, as the reference does
    int autosort_mode = 0;                 // the order a large call takes its photons up in: 0 as they come (default: the index sort + gather cost more than they gain, profiles/r03/ab_autosort.txt), 1 by direction cell, 2 decided by a probe (propagate_order)
    int packet_mode = 0;                   // k_raycast_packet for the first step: 0 never (default: it is not faster, profiles/r03/ab_packet_first_step.txt), 1 always, 2 when the photons are coherent (CHROMA_PACKET=off|on|auto)
    int wide_walk = CHROMA_WALK_QUAD;      // CHROMA_WALK_*: reference tree | wide tree with 1, 8 or 4 (default) lanes per ray
    hipEvent_t ev_start = nullptr, ev_stop = nullptr, ev_mid = nullptr;
    // the one exchange of the path (per-channel hit arrays): an RCCL communicator over the node's GPUs
    ncclComm_t comm = nullptr;
    int comm_nranks = 1, comm_rank = 0;
    uint32_t *gather_buf = nullptr;        // [comm_nranks][n] words for the OR reduction (all-gather + local OR)
    size_t gather_capacity = 0;
    // ---- device-memory pool behind chroma_malloc / chroma_free (see there) ----
    struct PoolBlock { void *ptr; hipEvent_t ev; };
    std::mutex pool_mu;
    std::multimap<size_t, PoolBlock> pool;                 // free blocks by size
    std::unordered_map<void *, size_t> live;               // size of every block handed out
    std::vector<hipEvent_t> pool_events;                   // spare events
    size_t pool_bytes = 0, pool_limit = 0;
    uint64_t pool_hits = 0, pool_misses = 0;
    // ---- host -> device uploads: a second stream and a ring of pinned staging buffers (chroma_upload) ----
    hipStream_t copy_stream = nullptr;
    std::mutex stage_mu;
    static constexpr int STAGE_N = 3;
    static constexpr size_t STAGE_BYTES = 64u << 20;
    void *stage[STAGE_N] = {nullptr, nullptr, nullptr};
    hipEvent_t stage_ev[STAGE_N] = {nullptr, nullptr, nullptr};
    // the same ring for device -> host copies, with its own lock: a hit download does not queue behind the next batch's upload
    std::mutex stage_down_mu;
    void *stage_down[STAGE_N] = {nullptr, nullptr, nullptr};
    hipEvent_t stage_down_ev[STAGE_N] = {nullptr, nullptr, nullptr};
};

extern "C" hipStream_t chroma_internal_stream(chroma_ctx *ctx) { return ctx->stream; }
extern "C" int chroma_internal_device(chroma_ctx *ctx) { return ctx->device; }

struct chroma_geometry {
    chroma_ctx *ctx;
    GeoView view;
    std::vector<void *> allocations;
    void *d_vertices = nullptr, *d_triangles = nullptr, *d_material_codes = nullptr, *d_colors = nullptr;
    void *d_nodes_api = nullptr;       // nodes exactly as passed in (GPUGeometry.nodes)
    size_t nvertices = 0, ntriangles = 0, nnodes = 0, nwide = 0, nrecords = 0;
    uint32_t stack_need = 0, wide_depth = 0, wide_stack_need = 0;
    size_t device_bytes = 0;
};

// hipMalloc for the library's own working buffers: when the device is out of memory, everything parked in the pool
// behind chroma_malloc / chroma_free is given back first (defined next to the pool)
static hipError_t ctx_malloc(chroma_ctx *ctx, void **ptr, size_t bytes);

// ---------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------
#define PROP_BLOCK 64
#ifndef STACK_LDS
#define STACK_LDS 24      // traversal stack entries kept in LDS (6 KB per wave); deeper ones spill to scratch
#endif
#ifndef RAY_WAVES
#define RAY_WAVES 1        // __launch_bounds__ waves/SIMD hint for the ray-cast kernel
#endif

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

// ---- persistent ray cast with lane refill ---------------------------------------------------------
// One ray per lane, but a lane that finishes its ray takes the next one from the queue (one atomic
// per wave per refill), so the 64 lanes of a wave stay busy although their rays need very
// different numbers of node visits (measured: 26 % of the lanes active without refill).
// Traversal is the walk of intersect_mesh (same visit order, postponed triangle tests); the stack
// lives in LDS only.  The rare rays this kernel cannot take -- a component of 1/d that is not
// "moderate" (exactly or nearly axis-parallel) or a stack deeper than RAY_LDS_STACK -- are marked
// HIT_RETRY and done by k_raycast_retry with the general code.
#ifndef RAY_LDS_STACK
#define RAY_LDS_STACK 24
#endif
#ifndef RAY_REFILL_MIN
#define RAY_REFILL_MIN 12     // refill once this many lanes are idle
#endif

template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK, RAY_WAVES) void
k_raycast_persistent(GeoView g, const float4 *rays, int first_photon, StepState *st,
                     int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, DeviceCounters *counters)
{
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * PROP_BLOCK >= nthreads) return;
    uint32_t *work_counter = &st->work, *retry_counter = &st->retry;
    __shared__ uint32_t s_lds[(RAY_LDS_STACK + TRAV_PENDING) * PROP_BLOCK];
    uint32_t *stack = s_lds + threadIdx.x;
    uint32_t *pending = stack + RAY_LDS_STACK * PROP_BLOCK;
    const unsigned lane = lane_id();
    LaneCounters cnt = {0, 0, 0, 0};

    // per-lane ray state
    bool has_ray = false, active = false;
    int slot = 0;
    v3 origin = mk3(0.f, 0.f, 0.f), direction = mk3(0.f, 0.f, 1.f);
    RayFast rf;
    rf.a = rf.blo = rf.bhi = mk3(0.f, 0.f, 0.f);
    int last_hit = -1, triangle_index = -1;
    float min_distance = -1.0f;
    uint32_t cur = 1, end = 0;
    int sp = 0, npend = 0;
    bool exhausted = false;     // wave-uniform: the queue has no more rays

    for (;;) {
        // ---- refill idle lanes
        unsigned long long idle_mask = __ballot(!has_ray);
        int n_idle = __popcll(idle_mask);
        if (!exhausted && (n_idle >= RAY_REFILL_MIN || n_idle == WAVE)) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(work_counter, (uint32_t)n_idle);
            base = __shfl(base, 0);
            if (base + (uint32_t)n_idle >= (uint32_t)nthreads) exhausted = true;
            if (!has_ray) {
                uint32_t idx = base + (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
                if (idx < (uint32_t)nthreads) {
                    slot = first_photon + (int)idx;
                    const float4 *r = rays + 4 * (size_t)slot;
                    const float4 r0 = r[0], r1 = r[1];
                    if (__float_as_int(r1.w) == 0) {             // (other slots were settled by k_ray_setup)
                        const float4 r2 = r[2], r3 = r[3];
                        origin = mk3(r0.x, r0.y, r0.z);
                        direction = mk3(r1.x, r1.y, r1.z);
                        last_hit = __float_as_int(r0.w);
                        rf.a = mk3(r2.x, r2.y, r2.z);
                        const v3 bb = mk3(r3.x, r3.y, r3.z);
                        rf.blo = bb - r2.w * rf.a;
                        rf.bhi = bb + r2.w * rf.a;
                        triangle_index = -1;
                        min_distance = -1.0f;
                        sp = 0;
                        npend = 0;
                        uint4 root = g.nodes[0];
                        has_ray = true;
                        if (node_passes(box_tmin_fast(rf, root), min_distance)) {
                            active = true;
                            cur = root.w & ~CHROMA_NCHILD_MASK;
                            end = cur + (root.w >> CHROMA_CHILD_BITS) - 1;
                        } else {
                            active = false;      // misses the world box: result -1 written below
                        }
                    }
                }
            }
        }
        if (!__any(has_ray)) {
            if (exhausted) break;
            continue;
        }

        // ---- node phase: one node per active lane per iteration; it ends when a lane's FIFO of
        // postponed leaves is full, or enough lanes have finished to make a refill worthwhile
        const int stop_at = exhausted ? 0 : max(0, __popcll(__ballot(active)) - RAY_REFILL_MIN);
        do {
            if (active) {
                if (cur > end) {
                    if (sp == 0) {
                        active = false;
                    } else {
                        sp--;
                        uint32_t w = stack[sp * PROP_BLOCK];
                        cur = w & ~CHROMA_NCHILD_MASK;
                        end = cur + (w >> CHROMA_CHILD_BITS) - 1;
                    }
                }
                if (active) {
                    uint4 nd = g.nodes[cur];
                    cur++;
                    if (COUNT) cnt.nodes++;
                    float tmin = box_tmin_fast(rf, nd);
                    if (node_passes(tmin, min_distance)) {
                        uint32_t nd_child = nd.w & ~CHROMA_NCHILD_MASK;
                        if ((nd.w >> CHROMA_CHILD_BITS) == 0) {
                            if ((int)nd_child != last_hit) {
                                pending[npend * PROP_BLOCK] = nd_child;
                                npend++;
                            }
                        } else if (sp >= RAY_LDS_STACK) {
                            // deeper than the LDS stack: hand the whole ray to the retry kernel
                            active = false;
                            npend = 0;
                            triangle_index = HIT_RETRY;
                        } else {
                            stack[sp * PROP_BLOCK] = nd.w;
                            sp++;
                        }
                    }
                }
            }
        } while (!__any(npend >= TRAV_PENDING) && __popcll(__ballot(active)) > stop_at);

        // ---- leaf phase: postponed triangle tests, oldest first
        for (int j = 0; __any(j < npend); j++) {
            if (j < npend) {
                uint32_t tri = pending[j * PROP_BLOCK];
                if (COUNT) cnt.tris++;
                const float4 *t = g.tri + TRI_STRIDE * (size_t)tri;
                float4 a = t[0], b = t[1], c = t[2];
                float distance;
                if (intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance)) {
                    if (triangle_index == -1 || distance < min_distance) {
                        triangle_index = (int)tri;
                        min_distance = distance;
                    }
                }
            }
        }
        npend = 0;

        // ---- retire finished rays
        if (has_ray && !active) {
            hit_triangle[slot] = triangle_index;                 // record index, or a HIT_* code
            hit_distance[slot] = min_distance;
            if (triangle_index == HIT_RETRY) retry_list[atomicAdd(retry_counter, 1u)] = (uint32_t)slot;
            has_ray = false;
        }
    }

    if (COUNT) {
        unsigned long long st = wave_sum_u64(cnt.steps), nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane == 0) {
            atomicAdd(&counters->photon_steps, st);
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
        }
    }
}

// ---- persistent ray cast over the derived 8-wide tree ---------------------------------------------
// Same frame as k_raycast_persistent (one ray per lane, lanes refilled from the queue), but a node
// visit is one 128-byte line: eight child boxes tested with the fast slab test, triangle children
// noted for the leaf phase, the nearest inner child walked next and the others pushed with their
// box distance so that a popped entry farther than the best hit costs nothing.  The visiting order
// is NOT the reference's; the result is, because the walk is conservative and exact ties between
// triangles are broken by the reference's test order (`rank`, see csrc/wide_build.cpp).
// Rays this kernel cannot take (1/d not moderate, more than WIDE_STACK entries) go to
// k_raycast_retry as before.
#ifndef WIDE_STACK
#define WIDE_STACK 16        // (node, distance) entries per lane in LDS
#endif
#ifndef WIDE_PENDING
#define WIDE_PENDING 12      // postponed triangle tests per lane in LDS
#endif
#ifndef WIDE_FLUSH
#define WIDE_FLUSH 5         // run the leaf phase once a lane holds this many (a visit adds up to 8)
#endif
#ifndef WIDE_SPILL
#define WIDE_SPILL 112       // further entries per lane in global memory (rarely touched)
#endif
#define WIDE_NONE 0xFFFFFFFFu

template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK, RAY_WAVES) void
k_raycast_wide(GeoView g, const float4 *rays, int first_photon, StepState *st,
               int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, uint2 *spill_base, DeviceCounters *counters,
               int big_chunk)
{
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * PROP_BLOCK >= nthreads) return;
    uint32_t *work_counter = &st->work, *retry_counter = &st->retry;
    // rays taken from the queue per atomic: many for big batches (a hot word serves only ~88 atomics/us),
    // one wave-load when every wave gets only a few rounds anyway
    const int chunk = ((long long)nthreads > 4ll * big_chunk * (long long)gridDim.x) ? big_chunk : PROP_BLOCK;
    static_assert(WIDE_FLUSH - 1 + 8 <= WIDE_PENDING, "a node visit must fit the FIFO");
    static_assert(PROP_BLOCK == WAVE, "one wave per workgroup: blockIdx.x names the wave's spill area");
    // stack entries beyond the LDS part live in this wave's slice of a global buffer, [entry][lane]
    uint2 *spill = spill_base + (size_t)blockIdx.x * WIDE_SPILL * PROP_BLOCK + threadIdx.x;
    __shared__ uint32_t s_lds[(2 * WIDE_STACK + WIDE_PENDING) * PROP_BLOCK];
    uint32_t *stack_n = s_lds + threadIdx.x;
    float *stack_t = (float *)(stack_n + WIDE_STACK * PROP_BLOCK);
    uint32_t *pending = stack_n + 2 * WIDE_STACK * PROP_BLOCK;
    const unsigned lane = lane_id();
    LaneCounters cnt = {0, 0, 0, 0};

    bool has_ray = false, active = false;
    int slot = 0;
    v3 origin = mk3(0.f, 0.f, 0.f), direction = mk3(0.f, 0.f, 1.f);
    RayFast rf;
    rf.a = rf.blo = rf.bhi = mk3(0.f, 0.f, 0.f);
    int last_hit = -1, triangle_index = -1;
    uint32_t best_rank = 0;
    float min_distance = -1.0f;
    uint32_t cur = WIDE_NONE;
    int sp = 0, npend = 0;
    // the wave's share of the queue, [loc_next, loc_end), taken `chunk` rays per atomic: a hot word
    // serves only ~88 atomics/us, far fewer than the refills 1e8 rays need
    uint32_t loc_next = 0, loc_end = 0;
    bool exhausted = false;

    for (;;) {
        // ---- refill idle lanes
        unsigned long long idle_mask = __ballot(!has_ray);
        int n_idle = __popcll(idle_mask);
        bool more = !exhausted || loc_next < loc_end;
        if (more && (n_idle >= RAY_REFILL_MIN || n_idle == WAVE)) {
            if (loc_next >= loc_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (uint32_t)chunk);
                base = __shfl(base, 0);
                if (base + (uint32_t)chunk >= (uint32_t)nthreads) exhausted = true;
                loc_next = min(base, (uint32_t)nthreads);
                loc_end = min(base + (uint32_t)chunk, (uint32_t)nthreads);
            }
            uint32_t idx = loc_next + (uint32_t)__popcll(idle_mask & ((1ull << lane) - 1ull));
            loc_next = min(loc_end, loc_next + (uint32_t)n_idle);
            if (!has_ray) {
                if (idx < loc_end) {
                    slot = first_photon + (int)idx;
                    const float4 *r = rays + 4 * (size_t)slot;
                    const float4 r0 = r[0], r1 = r[1];
                    if (__float_as_int(r1.w) == 0) {             // (other slots were settled by k_ray_setup)
                        const float4 r2 = r[2], r3 = r[3];
                        origin = mk3(r0.x, r0.y, r0.z);
                        direction = mk3(r1.x, r1.y, r1.z);
                        last_hit = __float_as_int(r0.w);
                        rf.a = mk3(r2.x, r2.y, r2.z);
                        const v3 bb = mk3(r3.x, r3.y, r3.z);
                        rf.blo = bb - r2.w * rf.a;
                        rf.bhi = bb + r2.w * rf.a;
                        triangle_index = -1;
                        min_distance = -1.0f;
                        sp = 0;
                        npend = 0;
                        cur = 0;                 // the wide root holds the children of the reference root
                        has_ray = true;
                        active = true;
                    }
                }
            }
        }
        if (!__any(has_ray)) {
            if (exhausted && loc_next >= loc_end) break;
            continue;
        }

        // ---- node phase: one wide node per active lane per iteration
        more = !exhausted || loc_next < loc_end;
        const int stop_at = more ? max(0, __popcll(__ballot(active)) - RAY_REFILL_MIN) : 0;
        do {
            if (active && cur == WIDE_NONE) {
                // next entry that can still hold a nearer hit
                while (sp > 0) {
                    sp--;
                    uint32_t n; float t;
                    if (sp < WIDE_STACK) { n = stack_n[sp * PROP_BLOCK]; t = stack_t[sp * PROP_BLOCK]; }
                    else { uint2 e = spill[(size_t)(sp - WIDE_STACK) * PROP_BLOCK]; n = e.x; t = __uint_as_float(e.y); }
                    if (min_distance < 0.0f || !(t > min_distance)) { cur = n; break; }
                }
                if (cur == WIDE_NONE) active = false;
            }
            if (active) {
                const uint4 *wn = g.wnodes + 8 * (size_t)cur;
                uint4 c[8];
#pragma unroll
                for (int j = 0; j < 8; j++) c[j] = wn[j];
                if (COUNT) cnt.nodes += 8;
                uint32_t nxt = WIDE_NONE;
                float nxt_t = 0.0f;
#pragma unroll
                for (int j = 0; j < 8; j++) {
                    float t = box_tmin_fast(rf, c[j]);
                    uint32_t w = c[j].w;
                    if (w != WIDE_NONE && node_passes(t, min_distance)) {
                        if (w & 0x80000000u) {
                            uint32_t tri = w & 0x7FFFFFFFu;
                            if ((int)tri != last_hit) {
                                pending[npend * PROP_BLOCK] = tri;
                                npend++;
                            }
                        } else if (nxt == WIDE_NONE) {
                            nxt = w; nxt_t = t;
                        } else {
                            uint32_t pw = w; float pt = t;
                            if (t < nxt_t) { pw = nxt; pt = nxt_t; nxt = w; nxt_t = t; }
                            if (sp < WIDE_STACK) {
                                stack_n[sp * PROP_BLOCK] = pw;
                                stack_t[sp * PROP_BLOCK] = pt;
                                sp++;
                            } else if (sp < WIDE_STACK + WIDE_SPILL) {
                                spill[(size_t)(sp - WIDE_STACK) * PROP_BLOCK] = make_uint2(pw, __float_as_uint(pt));
                                sp++;
                            } else {                                 // cannot happen: the host checked the tree's need
                                triangle_index = HIT_RETRY;
                            }
                        }
                    }
                }
                cur = nxt;
                if (triangle_index == HIT_RETRY) { active = false; npend = 0; cur = WIDE_NONE; sp = 0; }
            }
        } while (!__any(npend >= WIDE_FLUSH) && __popcll(__ballot(active)) > stop_at);

        // ---- leaf phase: postponed triangle tests
        for (int j = 0; __any(j < npend); j++) {
            if (j < npend) {
                uint32_t tri = pending[j * PROP_BLOCK];
                if (COUNT) cnt.tris++;
                const float4 *t = g.tri + TRI_STRIDE * (size_t)tri;
                float4 a = t[0], b = t[1], cc = t[2];
                float distance;
                if (intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(cc.x, cc.y, cc.z), distance)) {
                    uint32_t rank = __float_as_uint(cc.w);
                    if (triangle_index == -1 || distance < min_distance || (distance == min_distance && rank < best_rank)) {
                        triangle_index = (int)tri;
                        min_distance = distance;
                        best_rank = rank;
                    }
                }
            }
        }
        npend = 0;

        // ---- retire finished rays
        if (has_ray && !active) {
            hit_triangle[slot] = triangle_index;                 // record index, or a HIT_* code
            hit_distance[slot] = min_distance;
            if (triangle_index == HIT_RETRY) retry_list[atomicAdd(retry_counter, 1u)] = (uint32_t)slot;
            has_ray = false;
        }
    }

    if (COUNT) {
        unsigned long long st = wave_sum_u64(cnt.steps), nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane == 0) {
            atomicAdd(&counters->photon_steps, st);
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
        }
    }
}

// ---- cooperative ray cast over the 8-wide tree: eight lanes per ray ---------------------------------
// A wavefront carries 8 rays; the 8 lanes of a group each own ONE of the eight child entries of the
// node their ray is visiting.  A node visit is therefore one coalesced 128-byte read per group (one
// dwordx4 per lane, 8 lines per wave instruction instead of 64), one slab test per lane, and a few
// group-wide operations: ballots give the set of children hit, DPP min-reductions pick the nearest
// inner child, and every other hit lane writes its own (node, distance) entry at its own stack slot,
// so nothing in the visit is serial.  Triangle tests are shared the same way: up to 8 postponed
// triangles of a ray are tested at once, one per lane, and reduced by (distance, rank).
// The per-ray state (origin, direction, slab constants, best hit, stack pointer) is replicated in
// the 8 lanes of the group and stays identical because every lane computes it from the same
// ballots and broadcasts.  LDS: 8 groups x (24 stack entries + 16 postponed triangles) = 2 KB per
// wave, so residency is limited by wave slots only.  Results are those of k_raycast_wide (and of
// the reference): same conservative tree, same tie-break.
#ifndef COOP_STACK
#define COOP_STACK 24
#endif
#define COOP_PENDING 16
#define COOP_STRIDE (2 * COOP_STACK + COOP_PENDING + 1)     // words per group, +1 staggers the banks
#ifndef COOP_SPILL
#define COOP_SPILL 104       // stack entries per ray beyond the LDS part (global memory)
#endif
#ifndef COOP_REFILL_MIN
#define COOP_REFILL_MIN 2    // refill once this many of the 8 groups are idle
#endif

// group-wide (8 lanes) minimum with DPP: quad_perm [1,0,3,2], quad_perm [2,3,0,1], row_half_mirror;
// every lane of the group ends with the result
__device__ inline float group8_min(float v)
{
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));
    return v;
}
__device__ inline uint32_t group8_min_u32(uint32_t v)
{
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));
    v = min(v, (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false));
    return v;
}

#ifndef COOP_WAVES_PER_EU
#define COOP_WAVES_PER_EU 7
#endif
template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) __attribute__((amdgpu_waves_per_eu(COOP_WAVES_PER_EU, COOP_WAVES_PER_EU))) void
k_raycast_coop(GeoView g, const float4 *rays, int first_photon, StepState *st,
               int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, uint2 *spill_base, DeviceCounters *counters,
               int big_chunk)
{
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * 8 >= nthreads) return;
    uint32_t *work_counter = &st->work, *retry_counter = &st->retry;
    const int chunk = ((long long)nthreads > 4ll * big_chunk * (long long)gridDim.x) ? big_chunk : 8;
    static_assert(PROP_BLOCK == WAVE, "one wave per workgroup");
    __shared__ uint32_t s_lds[8 * COOP_STRIDE];
    const unsigned lane = lane_id();
    const unsigned j = lane & 7u, gshift = lane & ~7u, grp = lane >> 3;
    const uint32_t below = (1u << j) - 1u;
    uint32_t *stack_n = s_lds + grp * COOP_STRIDE;
    float *stack_t = (float *)(stack_n + COOP_STACK);
    uint32_t *pending = stack_n + 2 * COOP_STACK;
    uint2 *spill = spill_base + ((size_t)blockIdx.x * 8 + grp) * COOP_SPILL;
    LaneCounters cnt = {0, 0, 0, 0};
    const float inf = cm_inff();

    // per-ray state, identical in the 8 lanes of a group
    bool has_ray = false, active = false;
    int slot = 0;
    v3 origin = mk3(0.f, 0.f, 0.f), direction = mk3(0.f, 0.f, 1.f);
    RayFast rf;
    rf.a = rf.blo = rf.bhi = mk3(0.f, 0.f, 0.f);
    int last_hit = -1, triangle_index = -1;
    uint32_t best_rank = 0;
    float min_distance = -1.0f;
    uint32_t cur = WIDE_NONE;
    int sp = 0, npend = 0;
    // the wave's share of the queue: [loc_next, loc_end) taken `chunk` rays at a time
    uint32_t loc_next = 0, loc_end = 0;
    bool exhausted = false;

    for (;;) {
        // ---- refill idle groups
        unsigned long long idle_mask = __ballot(!has_ray && j == 0);
        int n_idle = __popcll(idle_mask);
        bool more = !exhausted || loc_next < loc_end;
        if (more && (n_idle >= COOP_REFILL_MIN || n_idle == 8)) {
            if (loc_next >= loc_end) {
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (uint32_t)chunk);
                base = __shfl(base, 0);
                if (base + (uint32_t)chunk >= (uint32_t)nthreads) exhausted = true;
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
                    origin = mk3(r0.x, r0.y, r0.z);
                    direction = mk3(r1.x, r1.y, r1.z);
                    last_hit = __float_as_int(r0.w);
                    rf.a = mk3(r2.x, r2.y, r2.z);
                    const v3 bb = mk3(r3.x, r3.y, r3.z);
                    rf.blo = bb - r2.w * rf.a;
                    rf.bhi = bb + r2.w * rf.a;
                    triangle_index = -1;
                    min_distance = -1.0f;
                    sp = 0;
                    npend = 0;
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

        // ---- node phase: every active group visits one node per iteration
        more = !exhausted || loc_next < loc_end;
        const int stop_at = more ? max(0, (int)__popcll(__ballot(active && j == 0)) - (int)COOP_REFILL_MIN) : 0;
        do {
            if (active && cur == WIDE_NONE) {
                // next entry that can still hold a nearer hit
                while (sp > 0) {
                    sp--;
                    uint32_t n; float t;
                    if (sp < COOP_STACK) { n = stack_n[sp]; t = stack_t[sp]; }
                    else { uint2 e = spill[sp - COOP_STACK]; n = e.x; t = __uint_as_float(e.y); }
                    if (min_distance < 0.0f || !(t > min_distance)) { cur = n; break; }
                }
                if (cur == WIDE_NONE) active = false;
            }
            if (active) {
                const uint4 e = g.wnodes[8 * (size_t)cur + j];
                if (COUNT && j == 0) cnt.nodes += 8;
                const float t = box_tmin_fast(rf, e);
                const uint32_t w = e.w;
                const bool pass = (w != WIDE_NONE) && node_passes(t, min_distance);
                const bool isleaf = (w & 0x80000000u) != 0u;
                const bool leaf = pass && isleaf && (int)(w & 0x7FFFFFFFu) != last_hit;
                const bool inner = pass && !isleaf;
                const uint32_t gl = (uint32_t)(__ballot(leaf) >> gshift) & 0xFFu;
                const uint32_t gi = (uint32_t)(__ballot(inner) >> gshift) & 0xFFu;
                if (leaf) pending[npend + __popc(gl & below)] = w & 0x7FFFFFFFu;
                npend += __popc(gl);
                cur = WIDE_NONE;
                if (gi) {
                    const float tm = group8_min(inner ? t : inf);
                    const uint32_t gn = (uint32_t)(__ballot(inner && t == tm) >> gshift) & 0xFFu;
                    const uint32_t nj = (uint32_t)__ffs((int)gn) - 1u;          // lane of the nearest inner child
                    const uint32_t others = gi & ~(1u << nj);
                    if (inner && j != nj) {
                        int pos = sp + __popc(others & below);
                        if (pos < COOP_STACK) { stack_n[pos] = w; stack_t[pos] = t; }
                        else if (pos < COOP_STACK + COOP_SPILL) { spill[pos - COOP_STACK] = make_uint2(w, __float_as_uint(t)); if (COUNT) cnt.spills++; }
                    }
                    sp += __popc(others);
                    cur = (uint32_t)__shfl((int)w, (int)(gshift + nj));
                    if (sp > COOP_STACK + COOP_SPILL) {          // cannot happen: the host checked the tree's need
                        triangle_index = HIT_RETRY;
                        active = false; npend = 0; cur = WIDE_NONE; sp = 0;
                    }
                }
            }
        } while (!__any(npend >= 8) && __popcll(__ballot(active && j == 0)) > stop_at);
        __builtin_amdgcn_wave_barrier();      // (scheduling fence: the lanes of a group exchange data through LDS)

        // ---- leaf phase: up to 8 postponed triangles of a ray at once, one per lane
        while (__any(npend > 0)) {
            if (npend > 0) {
                const int take = min(npend, 8);
                bool hit = false;
                float distance = inf;
                uint32_t tri = 0, rank = 0xFFFFFFFFu;
                if ((int)j < take) {
                    tri = pending[j];
                    if (COUNT) cnt.tris++;
                    const float4 *tp = g.tri + TRI_STRIDE * (size_t)tri;
                    float4 a = tp[0], b = tp[1], c = tp[2];
                    hit = intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance);
                    rank = __float_as_uint(c.w);
                }
                const float dm = group8_min(hit ? distance : inf);
                if (dm < inf) {
                    const bool cand = hit && distance == dm;
                    const uint32_t rm = group8_min_u32(cand ? rank : 0xFFFFFFFFu);
                    const uint32_t gw = (uint32_t)(__ballot(cand && rank == rm) >> gshift) & 0xFFu;
                    const uint32_t wj = (uint32_t)__ffs((int)gw) - 1u;
                    const int wtri = __shfl((int)tri, (int)(gshift + wj));
                    if (triangle_index == -1 || dm < min_distance || (dm == min_distance && rm < best_rank)) {
                        triangle_index = wtri;
                        min_distance = dm;
                        best_rank = rm;
                    }
                }
                if (npend > 8) {                      // keep the rest: move entries 8.. down
                    uint32_t mv = pending[j + 8];
                    if ((int)j + 8 < npend) pending[j] = mv;
                }
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
        unsigned long long st = wave_sum_u64(cnt.steps), nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        unsigned long long sx = wave_sum_u64(cnt.spills);
        if (lane == 0) {
            atomicAdd(&counters->photon_steps, st);
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
            if (sx) atomicAdd(&counters->stack_spills, sx);
        }
    }
}

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

#ifndef QUAD_OVERFLOW_RESET
#define QUAD_OVERFLOW_RESET 0
#endif
#ifndef QUAD_POP_TWO_ARMS
#define QUAD_POP_TWO_ARMS 0
#endif
template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) __attribute__((amdgpu_waves_per_eu(QUAD_WAVES_PER_EU, QUAD_WAVES_PER_EU))) void
k_raycast_quad(GeoView g, const float4 *rays, int first_photon, StepState *st,
               int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, uint2 *spill_base, DeviceCounters *counters,
               int big_chunk, int settle, const uint32_t *skip = nullptr)
{
    // (`settle`: nobody has written the hit entries of the slots whose ray record says "not to be cast" yet)
    // (`skip`: the step has been given to k_raycast_packet, launched before this kernel)
    if (skip && *skip != 0u) return;
    const int nthreads = (int)st->n;
    if ((long long)blockIdx.x * 16 >= nthreads) return;
    uint32_t *work_counter = &st->work, *retry_counter = &st->retry;
    const int chunk = ((long long)nthreads > 4ll * big_chunk * (long long)gridDim.x) ? big_chunk : 16;
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
                uint32_t base = 0;
                if (lane == 0) base = atomicAdd(work_counter, (uint32_t)chunk);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);       // (wave-uniform from here: scalar registers)
                if (base + (uint32_t)chunk >= (uint32_t)nthreads) exhausted = true;
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
#if QUAD_POP_TWO_ARMS
            if (!__any(sp > QUAD_STACK)) {
                // (node, distance) read together, the entry kept or dropped by a select: no branch inside the loop
                while (active && cur == WIDE_NONE) {
                    if (sp == 0) { active = false; break; }
                    sp--;
                    const uint32_t n = stack_n[sp];
                    const float t = stack_t[sp];
                    cur = (t > prune_t) ? WIDE_NONE : n;
                }
            } else
            if (active && cur == WIDE_NONE) {
                while (sp > 0) {
                    sp--;
                    uint32_t n; float t;
                    if (sp < QUAD_STACK) { n = stack_n[sp]; t = stack_t[sp]; }
                    else { uint2 se = spill[sp - QUAD_STACK]; n = se.x; t = __uint_as_float(se.y); }
                    if (!(t > prune_t)) { cur = n; break; }
                }
                if (cur == WIDE_NONE) active = false;
            }
#else
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
#endif
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
#if QUAD_OVERFLOW_RESET
                        if (sp > QUAD_STACK + COOP_SPILL) {          // cannot happen: the host checked the tree's need
                            triangle_index = HIT_RETRY;
                            active = false; npend = 0; cur = WIDE_NONE; sp = 0;
                        }
#else
                        // A stack deeper than LDS part + spill area cannot happen: chroma_geometry_create works the tree's need out
                        // and the launch code only picks this walk when it fits.  Rounds 1-3 nevertheless reset the ray's whole
                        // state here -- a merge of five loop-carried values with an arm that never runs, which cost the arm that
                        // always runs eight register copies per node visit.  The guards above already keep every write inside
                        // the two areas; clamping the depth keeps every later read inside them too, and the overflow is counted
                        // (stats.stack_overflows, which the tests hold at zero).
                        if (sp > QUAD_STACK + COOP_SPILL) { atomicAdd(&counters->stack_overflows, 1ull); sp = QUAD_STACK + COOP_SPILL; }
#endif
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

// ---- fused tail: all remaining steps of the last few photons, eight lanes per photon ---------------
// Once fewer than 64*16*8 photons are alive the reference finishes them in ONE launch
// (chroma/gpu/photon.py:227-230).  Per-step launches are a poor fit for that tail -- a few thousand
// rays, each a chain of ~30 dependent fetches, ~70 steps deep -- so it gets its own kernel: a group of
// 8 lanes owns one photon for all its remaining steps, casts its rays cooperatively (coop_cast: the
// walk of k_raycast_coop without the refill) and runs the physics redundantly in its 8 lanes (same
// inputs, same arithmetic, so the lanes stay identical; lane 0 of the group stores).  No launch or
// queue round trip between steps: the tail takes as long as its longest photon, not 70 launches.
// Rays the wide walk cannot take go through the general walk on the group's first lane.
template <bool COUNT>
__device__ inline int coop_cast(const GeoView &g, v3 origin, v3 direction, int last_hit, bool on, float &min_distance,
                                uint32_t *stack_n, float *stack_t, uint32_t *pending, uint2 *spill,
                                unsigned j, unsigned gshift, uint32_t below, LaneCounters &cnt)
{
    const float inf = cm_inff();
    int triangle_index = -1;
    uint32_t best_rank = 0;
    min_distance = -1.0f;
    uint32_t cur = WIDE_NONE;
    int sp = 0, npend = 0;
    bool active = false;
    RayFast rf;
    rf.a = rf.blo = rf.bhi = mk3(0.f, 0.f, 0.f);
    if (on) {
        v3 noid = (-origin) / direction;
        v3 inv_dir = 1.0f / direction;
        bool moderate = cm_fabsf(inv_dir.x) < 1e30f && cm_fabsf(inv_dir.y) < 1e30f && cm_fabsf(inv_dir.z) < 1e30f &&
                        cm_fabsf(noid.x) < 1e30f && cm_fabsf(noid.y) < 1e30f && cm_fabsf(noid.z) < 1e30f;
        if (!moderate) {
            triangle_index = HIT_RETRY;
        } else {
            rf = ray_fast(g, noid, inv_dir, ray_growth(g, origin));
            cur = 0;
            active = true;
        }
    }
    while (__any(active || npend > 0)) {
        // node phase
        while (__any(active) && !__any(npend >= 8)) {
            if (active && cur == WIDE_NONE) {
                while (sp > 0) {
                    sp--;
                    uint32_t n; float t;
                    if (sp < COOP_STACK) { n = stack_n[sp]; t = stack_t[sp]; }
                    else { uint2 se = spill[sp - COOP_STACK]; n = se.x; t = __uint_as_float(se.y); }
                    if (min_distance < 0.0f || !(t > min_distance)) { cur = n; break; }
                }
                if (cur == WIDE_NONE) active = false;
            }
            if (active) {
                const uint4 e = g.wnodes[8 * (size_t)cur + j];
                if (COUNT && j == 0) cnt.nodes += 8;
                const float t = box_tmin_fast(rf, e);
                const uint32_t w = e.w;
                const bool pass = (w != WIDE_NONE) && node_passes(t, min_distance);
                const bool isleaf = (w & 0x80000000u) != 0u;
                const bool leaf = pass && isleaf && (int)(w & 0x7FFFFFFFu) != last_hit;
                const bool inner = pass && !isleaf;
                const uint32_t gl = (uint32_t)(__ballot(leaf) >> gshift) & 0xFFu;
                const uint32_t gi = (uint32_t)(__ballot(inner) >> gshift) & 0xFFu;
                if (leaf) pending[npend + __popc(gl & below)] = w & 0x7FFFFFFFu;
                npend += __popc(gl);
                cur = WIDE_NONE;
                if (gi) {
                    const float tm = group8_min(inner ? t : inf);
                    const uint32_t gn = (uint32_t)(__ballot(inner && t == tm) >> gshift) & 0xFFu;
                    const uint32_t nj = (uint32_t)__ffs((int)gn) - 1u;
                    const uint32_t others = gi & ~(1u << nj);
                    if (inner && j != nj) {
                        int pos = sp + __popc(others & below);
                        if (pos < COOP_STACK) { stack_n[pos] = w; stack_t[pos] = t; }
                        else if (pos < COOP_STACK + COOP_SPILL) { spill[pos - COOP_STACK] = make_uint2(w, __float_as_uint(t)); if (COUNT) cnt.spills++; }
                    }
                    sp += __popc(others);
                    cur = (uint32_t)__shfl((int)w, (int)(gshift + nj));
                    if (sp > COOP_STACK + COOP_SPILL) {          // cannot happen: the host checked the tree's need
                        triangle_index = HIT_RETRY;
                        active = false; npend = 0; cur = WIDE_NONE; sp = 0;
                    }
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // leaf phase
        while (__any(npend > 0)) {
            if (npend > 0) {
                const int take = min(npend, 8);
                bool hit = false;
                float distance = inf;
                uint32_t tri = 0, rank = 0xFFFFFFFFu;
                if ((int)j < take) {
                    tri = pending[j];
                    if (COUNT) cnt.tris++;
                    const float4 *tp = g.tri + TRI_STRIDE * (size_t)tri;
                    float4 a = tp[0], b = tp[1], c = tp[2];
                    hit = intersect_triangle(origin, direction, mk3(a.x, a.y, a.z), mk3(b.x, b.y, b.z), mk3(c.x, c.y, c.z), distance);
                    rank = __float_as_uint(c.w);
                }
                const float dm = group8_min(hit ? distance : inf);
                if (dm < inf) {
                    const bool cand = hit && distance == dm;
                    const uint32_t rm = group8_min_u32(cand ? rank : 0xFFFFFFFFu);
                    const uint32_t gw = (uint32_t)(__ballot(cand && rank == rm) >> gshift) & 0xFFu;
                    const uint32_t wj = (uint32_t)__ffs((int)gw) - 1u;
                    const int wtri = __shfl((int)tri, (int)(gshift + wj));
                    if (triangle_index == -1 || dm < min_distance || (dm == min_distance && rm < best_rank)) {
                        triangle_index = wtri;
                        min_distance = dm;
                        best_rank = rm;
                    }
                }
                if (npend > 8) {
                    uint32_t mv = pending[j + 8];
                    if ((int)j + 8 < npend) pending[j] = mv;
                }
                npend -= take;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    return triangle_index;
}

template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) void
k_tail_coop(GeoView g, PhotonView pv, const StepState *st, const float4 *work_in,
            uint64_t seed, uint64_t id_base, int max_steps, int use_weights, int scatter_first, uint2 *spill_base,
            DeviceCounters *counters)
{
    __shared__ uint32_t s_coop[8 * COOP_STRIDE];
    __shared__ uint32_t s_walk[TRAV_LDS_WORDS(STACK_LDS, PROP_BLOCK)];
    const int nthreads = (int)st->n, renorm = (int)st->renorm;
    if ((long long)blockIdx.x * 8 >= nthreads) return;
    const unsigned lane = lane_id();
    const unsigned j = lane & 7u, gshift = lane & ~7u, grp = lane >> 3;
    const uint32_t below = (1u << j) - 1u;
    uint32_t *stack_n = s_coop + grp * COOP_STRIDE;
    float *stack_t = (float *)(stack_n + COOP_STACK);
    uint32_t *pending = stack_n + 2 * COOP_STACK;
    uint2 *spill = spill_base + ((size_t)blockIdx.x * 8 + grp) * COOP_SPILL;
    LaneCounters cnt = {0, 0, 0, 0};

    const int id = (int)blockIdx.x * 8 + (int)grp;          // one photon per group
    bool loaded = false;
    uint32_t photon_id = 0;
    int last_hit_dev = -1;
    Photon p;
    cm_rng rng;
    State s;
    if (id < nthreads) {
        const float4 *w = work_in + 4 * (size_t)id;
        const float4 w0 = w[0], w1 = w[1], w2 = w[2], w3 = w[3];
        photon_id = __float_as_uint(w3.w);
        p.position = mk3(w0.x, w0.y, w0.z);
        p.direction = mk3(w1.x, w1.y, w1.z);
        p.polarization = mk3(w2.x, w2.y, w2.z);
        if (renorm) {
            p.direction = p.direction / norm(p.direction);
            p.polarization = p.polarization / norm(p.polarization);
        }
        p.wavelength = w0.w;
        p.time = w1.w;
        p.weight = w2.w;
        p.history = __float_as_uint(w3.x);
        last_hit_dev = __float_as_int(w3.z);
        p.last_hit_triangle = last_hit_dev >= 0 ? (int)g.dev_to_tri[last_hit_dev] : -1;
        p.evidx = 0;
        loaded = true;                                      // (the working set holds live photons only)
        cm_rng_init(&rng, seed, id_base + photon_id, __float_as_uint(w3.y));
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
            } else if (COUNT && j == 0) cnt.steps++;
        }
        float distance;
        int record = coop_cast<COUNT>(g, p.position, p.direction, last_hit_dev, stepping, distance, stack_n, stack_t, pending,
                                      spill, j, gshift, below, cnt);
        // the reference's own walk for the rays the wide walk cannot take, and for winners that are not
        // regular (record_hit_is_regular): first lane of the group, then shared
        bool general = stepping && record == HIT_RETRY;
        if (stepping && record >= 0) {
            const float4 *t = g.tri + TRI_STRIDE * (size_t)record;
            general = !record_hit_is_regular(g, t[0], t[1], t[2], p.position, p.direction, distance);
        }
        if (__any(general)) {
            float d2 = 0.0f;
            int r2 = intersect_mesh_dev<STACK_LDS, PROP_BLOCK, COUNT>(g, p.position, p.direction, d2, last_hit_dev,
                                                                       s_walk + threadIdx.x, cnt, general && j == 0);
            r2 = __shfl(r2, (int)gshift);
            d2 = __shfl(d2, (int)gshift);
            if (general) { record = r2; distance = d2; }
        }
        if (stepping) {
            apply_hit_dev(s, p, g, record, distance);
            if (record == -1) {
                live = false;
                last_hit_dev = -1;
            } else {
                live = step_after_hit(p, s, rng, g, use_weights != 0, scatter_first);
                scatter_first = 0;
                last_hit_dev = (p.last_hit_triangle < 0) ? -1 : record;
            }
        }
    }

    if (loaded && j == 0) {                                 // the call ends with this kernel: everything goes back
        pv.rng_counters[photon_id] = rng.counter;
        store3(pv.pos, photon_id, p.position);
        store3(pv.dir, photon_id, p.direction);
        store3(pv.pol, photon_id, p.polarization);
        pv.wavelengths[photon_id] = p.wavelength;
        pv.t[photon_id] = p.time;
        pv.flags[photon_id] = p.history;
        pv.last_hit_triangles[photon_id] = p.last_hit_triangle;
        pv.weights[photon_id] = p.weight;
    }

    if (COUNT) {
        unsigned long long sts = wave_sum_u64(cnt.steps), nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        unsigned long long sx = wave_sum_u64(cnt.spills);
        if (lane == 0) {
            atomicAdd(&counters->photon_steps, sts);
            atomicAdd(&counters->nodes_visited, nd);
            atomicAdd(&counters->triangles_tested, tr);
            if (sx) atomicAdd(&counters->stack_spills, sx);
        }
    }
}

// Second pass for the rays the fast walks hand over (their queue slots are listed in retry_list):
// 1/d not moderate, a winner that is not regular (record_hit_is_regular), a stack deeper than the spill.
// They take the literal reference walk, intersect_mesh_strict.  ~1e-4 of the rays.
// ALL (the walk CHROMA_WALK_LITERAL): every queued ray takes the literal walk -- the one mode whose answer is the
// reference's on EVERY ray, the erratic Moeller-Trumbore hits of DESIGN.md section 3.1 included, because nothing about
// the order of box and triangle tests differs from mesh.h:42-118.  (Slots k_ray_setup settled as NaN keep their entry.)
template <bool COUNT, bool ALL = false>
__global__ __launch_bounds__(PROP_BLOCK) void
k_raycast_retry(GeoView g, const float4 *rays, const StepState *st,
                int32_t *hit_triangle, float *hit_distance, const uint32_t *retry_list, DeviceCounters *counters)
{
    const int nretry = ALL ? (int)st->n : (int)st->retry;
    __shared__ uint32_t s_lds[TRAV_LDS_WORDS(STACK_LDS, PROP_BLOCK)];
    if (nretry == 0) return;
    LaneCounters cnt = {0, 0, 0, 0};
    const int stride = gridDim.x * PROP_BLOCK;
    // (the loop bound is wave-uniform: intersect_mesh_dev votes across the wave)
    for (int k0 = blockIdx.x * PROP_BLOCK; k0 < nretry; k0 += stride) {
        const int k = k0 + (int)threadIdx.x;
        bool walk = false;
        int slot = 0, last_hit = -1;
        v3 position = mk3(0.f, 0.f, 0.f), direction = mk3(0.f, 0.f, 1.f);
        if (k < nretry) {
            slot = ALL ? k : (int)retry_list[k];
            const float4 *r = rays + 4 * (size_t)slot;
            const float4 r0 = r[0], r1 = r[1];
            position = mk3(r0.x, r0.y, r0.z); direction = mk3(r1.x, r1.y, r1.z);         // (normalised by k_ray_setup)
            last_hit = __float_as_int(r0.w);
        }
        if (ALL) {
            walk = k < nretry && __float_as_int(rays[4 * (size_t)slot + 1].w) != HIT_NAN;
        } else if (k < nretry) {
            // a slot k_physics listed because the cheap test could not vouch for the fast walk's winner still holds
            // that winner: the exact question first (the leaf box by the reference's rule, the reference's slab
            // test); only a winner the reference may really miss is walked again
            const int rec = hit_triangle[slot];
            walk = true;
            if (rec >= 0) {
                const float4 *t = g.tri + TRI_STRIDE * (size_t)rec;
                walk = !record_hit_is_exactly_regular(g, t[0], t[1], t[2], position, direction, hit_distance[slot]);
            }
        }
        float dist;
        int found = intersect_mesh_dev<STACK_LDS, PROP_BLOCK, COUNT>(g, position, direction, dist, last_hit, s_lds + threadIdx.x, cnt, walk);
        if (walk) {
            hit_triangle[slot] = found;
            hit_distance[slot] = dist;
        }
    }
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

#include "raycast_literal.h"

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
#ifndef PHYS_STAGE_LDS
#define PHYS_STAGE_LDS 0     // experiment: survivors' records through LDS as in k_load_working -- no gain here (profiles/r03/ab_lds_staged_stores.txt)
#endif
template <bool FULL>
__global__ __launch_bounds__(PHYS_BLOCK_OF(FULL)) __attribute__((amdgpu_waves_per_eu(FULL ? PHYS_WAVES_PER_EU : PHYS_PLAIN_WAVES_PER_EU))) void
k_physics(GeoView g, PhotonView pv, StepState *st, const float4 *work_in, uint32_t *output_queue, float4 *work_out,
          const int32_t *hit_triangle, const float *hit_distance, uint64_t seed, uint64_t id_base,
          int use_weights, int scatter_first, uint32_t *retry_list, int fixup, DeviceCounters *counters, float4 *rays_next,
          float4 *final_rec = nullptr, uint32_t epoch = 0u)
{
    // Two passes per step.  Main pass (fixup = 0): every slot of the working set; a slot the ray cast
    // handed to the strict walk (HIT_RETRY) is left alone, and so is a hit that is not REGULAR
    // (record_hit_is_regular, propagate_device.h): its slot joins retry_list.  Fix-up pass (fixup = 1),
    // after k_raycast_retry has walked those rays the reference's way: the listed slots only, results
    // taken as they are.  A photon that survives the step is appended to the next working set; one that
    // ends here is written to the caller's arrays (the only time they are touched).
    constexpr int BLOCK = PHYS_BLOCK_OF(FULL);
    __shared__ uint32_t s_counts[BLOCK / WAVE + 1];
    // (survivor records leave through LDS in the plain build: see the end of the round; the all-models build has no registers to spare)
    constexpr bool STAGE = (PHYS_STAGE_LDS != 0) && !FULL;
    __shared__ float4 s_stage[STAGE ? BLOCK / WAVE : 1][STAGE ? WAVE * 4 : 1];
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
    if constexpr (STAGE) {
        // the survivors of a wave take consecutive slots: their 64-byte records (and then their ray records) leave through
        // LDS as whole kilobytes instead of as 64 partial lines per store instruction (see k_load_working)
        const unsigned long long tm = __ballot(alive);
        const uint32_t nsurv = (uint32_t)__popcll(tm), rnk = (uint32_t)__popcll(tm & ((1ull << lane_id()) - 1ull));
        const uint32_t first_slot = nsurv ? (uint32_t)__shfl(at, __ffsll((long long)tm) - 1) - 1u : 0u;
        float4 *stg = s_stage[threadIdx.x / WAVE];
        if (alive) {
            float4 *w = stg + 4 * rnk;
            w[0] = make_float4(p.position.x, p.position.y, p.position.z, p.wavelength);
            w[1] = make_float4(p.direction.x, p.direction.y, p.direction.z, p.time);
            w[2] = make_float4(p.polarization.x, p.polarization.y, p.polarization.z, p.weight);
            w[3] = make_float4(__uint_as_float(p.history), __uint_as_float(counter), __int_as_float(last_hit_record), __uint_as_float(photon_id));
        }
        __builtin_amdgcn_wave_barrier();
        for (uint32_t q = lane_id(); q < 4u * nsurv; q += WAVE) work_out[4 * (size_t)first_slot + q] = stg[q];
        __builtin_amdgcn_wave_barrier();
        if (rays_next) {
            // the survivor's ray for the next step (see k_ray_setup): the next launch re-normalises unless the reference's
            // last launch has begun -- which k_step_begin of THIS step has already decided
            if (alive) make_ray_record(g, stg + 4 * rnk, p.position, p.direction, renorm_next, last_hit_record);
            __builtin_amdgcn_wave_barrier();
            for (uint32_t q = lane_id(); q < 4u * nsurv; q += WAVE) rays_next[4 * (size_t)first_slot + q] = stg[q];
            __builtin_amdgcn_wave_barrier();
        }
    } else
    if (alive) {
        float4 *w = work_out + 4 * (size_t)(at - 1u);
        w[0] = make_float4(p.position.x, p.position.y, p.position.z, p.wavelength);
        w[1] = make_float4(p.direction.x, p.direction.y, p.direction.z, p.time);
        w[2] = make_float4(p.polarization.x, p.polarization.y, p.polarization.z, p.weight);
        w[3] = make_float4(__uint_as_float(p.history), __uint_as_float(counter), __int_as_float(last_hit_record), __uint_as_float(photon_id));
        // the survivor's ray for the next step (see k_ray_setup): the next launch re-normalises unless the reference's
        // last launch has begun -- which k_step_begin of THIS step has already decided
        if (rays_next) make_ray_record(g, rays_next + 4 * (size_t)(at - 1u), p.position, p.direction, renorm_next, last_hit_record);
    }
    __syncthreads();        // s_counts is reused by the next round
    }
    if (counters) {
        nsteps = wave_sum_u64(nsteps);
        if (lane_id() == 0 && nsteps) atomicAdd(&counters->photon_steps, nsteps);
    }
}

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
               uint32_t *coherence, const uint32_t *order = nullptr)
{
    // (`order`: take the photons up in this order instead of by index -- propagate_order below; ncopies == 1 then)
    // (`rays`: also the ray records of the first step -- the first launch of a call always re-normalises)
    // (`coherence`: [0] += waves whose photons share an origin and lie within a cone of 50 mrad, [1] += waves looked at:
    //  what decides between k_raycast_packet and k_raycast_quad for the first step.  A heuristic: it steers speed only.)
    __shared__ uint32_t s_counts[PHYS_BLOCK / WAVE + 1];
#if LOAD_STAGE_LDS
    __shared__ float4 s_stage[PHYS_BLOCK / WAVE][WAVE * 4];
#endif
    uint32_t coh_yes = 0, coh_all = 0;
    for (uint64_t block_base = (uint64_t)blockIdx.x * PHYS_BLOCK; block_base < n; block_base += (uint64_t)gridDim.x * PHYS_BLOCK) {
        uint64_t j = block_base + threadIdx.x;
        bool take = false;
        uint32_t photon_id = 0, flags = 0;
        if (j < n) {
            photon_id = order ? order[j] : (uint32_t)(j / ncopies) + (uint32_t)(j % ncopies) * true_n;
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
            if (take) make_ray_record(g, st + 4 * rnk, pos, dir, 1, lh);
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
            if (rays) make_ray_record(g, rays + 4 * (size_t)(at - 1u), pos, dir, 1, lh);
#endif
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
        }
        __syncthreads();
    }
    if (coherence) {
        if (coh_all) { atomicAdd(&coherence[1], coh_all); if (coh_yes) atomicAdd(&coherence[0], coh_yes); }
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

// initial queue of GPUPhotons.propagate (chroma/gpu/photon.py:206-216): slot 0 unused counter,
// then photon ids with the ncopies clones of a photon next to each other.
__global__ void k_init_queue(uint32_t *queue, uint64_t n, uint32_t ncopies, uint32_t true_n)
{
    uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j == 0) queue[0] = (uint32_t)n + 1u;      // slot 0 = tail index, as after a step that queued all n
    if (j < n) {
        uint32_t copy = (uint32_t)(j % ncopies);
        uint32_t idx = (uint32_t)(j / ncopies);
        queue[1 + j] = idx + copy * true_n;
    }
}

__global__ void k_set_word(uint32_t *p, uint32_t v) { *p = v; }


// OR of (flags & mask) over all photons -> one word (abort warning, photon.py:254)
__global__ void k_flags_or(const uint32_t *flags, uint64_t n, uint32_t mask, uint32_t *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) acc |= flags[i] & mask;
    if (__ballot(acc != 0)) {
        for (int off = 32; off > 0; off >>= 1) acc |= __shfl_down(acc, off);
        if (lane_id() == 0 && acc) atomicOr(out, acc);
    }
}

__device__ inline void copy_photon(const PhotonView &src, size_t i, const PhotonView &dst, size_t o)
{
    store3(dst.pos, o, load3(src.pos, i));
    store3(dst.dir, o, load3(src.dir, i));
    store3(dst.pol, o, load3(src.pol, i));
    dst.wavelengths[o] = src.wavelengths[i];
    dst.t[o] = src.t[i];
    dst.flags[o] = src.flags[i];
    dst.last_hit_triangles[o] = src.last_hit_triangles[i];
    dst.weights[o] = src.weights[i];
    dst.evidx[o] = src.evidx[i];
    if (dst.rng_counters && src.rng_counters) dst.rng_counters[o] = src.rng_counters[i];
}

// photon_duplicate (chroma/cuda/propagate.cu:13-52)
__global__ void k_photon_duplicate(PhotonView pv, int first_photon, int nthreads, int copies, int stride)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nthreads) return;
    size_t photon_id = (size_t)first_photon + id;
    for (int i = 1; i <= copies; i++) copy_photon(pv, photon_id, pv, photon_id + (size_t)stride * i);
}

// count_photons (propagate.cu:54-79): grid-stride, one atomic per block
__global__ __launch_bounds__(256) void
k_count_photons(const uint32_t *flags, int first_photon, int nthreads, uint32_t target_flag, uint32_t *counter)
{
    __shared__ uint32_t s_total;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nthreads; id += (long long)gridDim.x * blockDim.x)
        mine += (flags[first_photon + id] & target_flag) != 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if (lane_id() == 0 && mine) atomicAdd(&s_total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_total) atomicAdd(counter, s_total);
}

__device__ inline uint32_t wave_reserve(uint32_t *counter, bool pred, bool &any)
{
    unsigned long long mask = __ballot(pred);
    any = mask != 0ull;
    if (!any) return 0;
    unsigned lane = lane_id();
    unsigned leader = (unsigned)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    return base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
}

// copy_photons (propagate.cu:81-114): one atomic per block of COPY_ITEMS * 256 photons, as k_copy_hits below
__global__ __launch_bounds__(256) void
k_copy_photons(PhotonView src, PhotonView dst, int first_photon, int nthreads, uint32_t target_flag, uint32_t *counter)
{
    __shared__ uint32_t s_wave[256 / WAVE + 1];
    const long long base = (long long)blockIdx.x * (16 * 256);
    const unsigned lane = lane_id(), wave = threadIdx.x / WAVE;
    uint32_t take = 0, mine = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const long long id = base + (long long)k * 256 + threadIdx.x;
        if (id < nthreads && (src.flags[first_photon + id] & target_flag)) { take |= 1u << k; mine++; }
    }
    uint32_t incl = mine;
    for (int off = 1; off < WAVE; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
    if (lane == WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (unsigned w = 0; w < 256 / WAVE; w++) { uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
        s_wave[256 / WAVE] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t off = s_wave[256 / WAVE] + s_wave[wave] + incl - mine;
#pragma unroll
    for (int k = 0; k < 16; k++)
        if (take & (1u << k)) copy_photon(src, (size_t)first_photon + (size_t)(base + (long long)k * 256 + threadIdx.x), dst, off++);
}

// copy_photon_queue (propagate.cu:116-144)
__global__ void k_copy_photon_queue(PhotonView src, PhotonView dst, int first_photon, int nthreads, const uint32_t *queue)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nthreads) return;
    size_t offset = (size_t)first_photon + id;
    copy_photon(src, queue[offset], dst, offset);
}

__device__ inline int hit_channel(const GeoView &g, uint32_t history, int triangle_id, uint32_t detection_state)
{
    if (!(history & detection_state)) return -1;
    if (triangle_id <= -1) return -1;
    uint32_t solid_id = g.solid_id_map[triangle_id];
    return g.solid_id_to_channel_index[solid_id];
}

// count_photon_hits (propagate.cu:147-174): grid-stride, one atomic per block
__global__ __launch_bounds__(256) void
k_count_hits(GeoView g, const uint32_t *flags, const int32_t *last_hit, int first_photon, int nphotons,
             uint32_t detection_state, uint32_t *counter)
{
    __shared__ uint32_t s_total;
    if (threadIdx.x == 0) s_total = 0;
    __syncthreads();
    uint32_t mine = 0;
    for (long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x; id < nphotons; id += (long long)gridDim.x * blockDim.x)
        mine += hit_channel(g, flags[first_photon + id], last_hit[first_photon + id], detection_state) >= 0;
    for (int off = 32; off > 0; off >>= 1) mine += __shfl_down(mine, off);
    if (lane_id() == 0 && mine) atomicAdd(&s_total, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_total) atomicAdd(counter, s_total);
}

// copy_photon_hits (propagate.cu:176-214).  A block looks at COPY_ITEMS * 256 photons and reserves its
// output span with ONE atomic (the reference's one atomic per detected photon -- or one per wave -- on a
// single word costs 18 ms for 1e8 photons: a hot word serves ~88 atomics per microsecond).
#define COPY_ITEMS 16
__global__ __launch_bounds__(256) void
k_copy_hits(GeoView g, PhotonView src, PhotonView dst, int32_t *channels, int first_photon, int nphotons,
            uint32_t detection_state, uint32_t *counter)
{
    __shared__ uint32_t s_wave[256 / WAVE + 1];
    const long long base = (long long)blockIdx.x * (COPY_ITEMS * 256);
    const unsigned lane = lane_id(), wave = threadIdx.x / WAVE;
    int ch[COPY_ITEMS];
    uint32_t mine = 0;
#pragma unroll
    for (int k = 0; k < COPY_ITEMS; k++) {
        const long long id = base + (long long)k * 256 + threadIdx.x;
        ch[k] = -1;
        if (id < nphotons) ch[k] = hit_channel(g, src.flags[first_photon + id], src.last_hit_triangles[first_photon + id], detection_state);
        mine += ch[k] >= 0;
    }
    // exclusive prefix of `mine` over the block: wave scan, then the waves' totals through LDS
    uint32_t incl = mine;
    for (int off = 1; off < WAVE; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
    if (lane == WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (unsigned w = 0; w < 256 / WAVE; w++) { uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
        s_wave[256 / WAVE] = total ? atomicAdd(counter, total) : 0u;
    }
    __syncthreads();
    uint32_t off = s_wave[256 / WAVE] + s_wave[wave] + incl - mine;
#pragma unroll
    for (int k = 0; k < COPY_ITEMS; k++) {
        if (ch[k] >= 0) {
            copy_photon(src, (size_t)first_photon + (size_t)(base + (long long)k * 256 + threadIdx.x), dst, off);
            channels[off] = ch[k];
            off++;
        }
    }
}

// per-channel hit count + earliest time (float bits; non-negative times only, cuda/daq.cu:5-20)
__global__ void k_channel_hits(GeoView g, const uint32_t *flags, const int32_t *last_hit, const float *t, uint64_t n,
                               uint32_t detection_state, uint32_t *hit_count, uint32_t *earliest)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int ch = -1;
    uint32_t tb = 0xFFFFFFFFu;
    if (i < n) {
        ch = hit_channel(g, flags[i], last_hit[i], detection_state);
        if (ch >= 0 && earliest) tb = __float_as_uint(t[i]);
    }
    // The hits of a wave that fall on ONE channel are added with one atomic (a detector of few channels -- the stress
    // geometry has one -- otherwise serialises every hit on a hot word: 4.4 ms for 3.9e5 hits); with thousands of
    // channels the lanes of a wave hardly ever agree, and each adds its own.
    const unsigned long long hitters = __ballot(ch >= 0);
    if (!hitters) return;
    const int first = __builtin_amdgcn_readlane(ch, (int)__builtin_ctzll(hitters));
    if (__ballot(ch >= 0 && ch != first) == 0ull) {
        uint32_t m = tb;
        for (int off = 32; off > 0; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off));
        if (lane_id() == (unsigned)__builtin_ctzll(hitters)) {
            atomicAdd(&hit_count[first], (uint32_t)__popcll(hitters));
            if (earliest) atomicMin(&earliest[first], m);
        }
    } else if (ch >= 0) {
        atomicAdd(&hit_count[ch], 1u);
        if (earliest) atomicMin(&earliest[ch], tb);
    }
}

// ---- the end of a chroma_propagate_hits call: ONE pass over the photons ---------------------------------------------------
// What the reference does in four passes after propagate -- the abort-flag reduction (gpu/photon.py:254), count_photon_hits,
// copy_photon_hits (propagate.cu:147-214) and, for the detector's channel arrays, a DAQ-like reduction -- happens here while a
// photon's final state is in registers anyway: a photon that ended in k_physics during this call left a 64-byte record at
// final_rec[id] (stamped with the call's epoch), which is unpacked into the caller's ten arrays (coalesced: every array gets
// whole lines); any other photon (terminal before the call, finished by the tail kernel, or still alive at max_steps) is read
// from the arrays.  Detected photons that belong to a channel are counted, compacted into `dst` with their
// channel (one atomic per block of COPY_ITEMS * 256 photons, as k_copy_hits: the order of the blocks is the order of their atomics), and bump the per-channel count / earliest-time arrays.
// final_rec == NULL: everything comes from the arrays (the fused form of k_count_hits + k_copy_hits + k_channel_hits).
struct HitsOut {
    PhotonView dst; int32_t *channels; uint32_t capacity;
    uint32_t *hit_count, *earliest;
    uint32_t detection_state; int want;
};
__global__ __launch_bounds__(256) void
k_finalize_hits(GeoView g, PhotonView pv, const float4 *final_rec, uint32_t epoch, uint64_t n, HitsOut h,
                uint32_t *words /* [0] number of hits, [2] OR of the NAN_ABORT bits */)
{
    __shared__ uint32_t s_wave[256 / WAVE + 1];
    const long long base = (long long)blockIdx.x * (COPY_ITEMS * 256);
    const unsigned lane = lane_id(), wave = threadIdx.x / WAVE;
    int ch[COPY_ITEMS];
    uint32_t mine = 0, from_record = 0, aborts = 0;
#pragma unroll
    for (int k = 0; k < COPY_ITEMS; k++) {
        const long long id = base + (long long)k * 256 + threadIdx.x;
        ch[k] = -1;
        uint32_t tb = 0xFFFFFFFFu;
        if (id < (long long)n) {
            uint32_t flags; int lh = -1; float t = 0.f;
            bool have = false;
            float4 f3 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (final_rec) { f3 = final_rec[4 * (size_t)id + 3]; have = __float_as_uint(f3.w) == epoch; }
            if (have) {
                const float4 *f = final_rec + 4 * (size_t)id;
                const float4 f0 = f[0], f1 = f[1], f2 = f[2];
                store3(pv.pos, (size_t)id, mk3(f0.x, f0.y, f0.z));
                store3(pv.dir, (size_t)id, mk3(f1.x, f1.y, f1.z));
                store3(pv.pol, (size_t)id, mk3(f2.x, f2.y, f2.z));
                pv.wavelengths[id] = f0.w;
                pv.t[id] = f1.w;
                pv.weights[id] = f2.w;
                flags = __float_as_uint(f3.x);
                pv.flags[id] = flags;
                pv.rng_counters[id] = __float_as_uint(f3.y);
                lh = __float_as_int(f3.z);
                pv.last_hit_triangles[id] = lh;
                t = f1.w;
                from_record |= 1u << k;
            } else {
                flags = pv.flags[id];
                if (h.want && (flags & h.detection_state)) { lh = pv.last_hit_triangles[id]; t = pv.t[id]; }
            }
            aborts |= flags & CHROMA_NAN_ABORT;
            if (h.want) {
                ch[k] = hit_channel(g, flags, lh, h.detection_state);
                if (ch[k] >= 0) { mine++; tb = __float_as_uint(t); }
            }
        }
        if (h.want && h.hit_count) {
            // (the hits of a wave that fall on ONE channel are added with one atomic: see k_channel_hits)
            const int c = ch[k];
            const unsigned long long hitters = __ballot(c >= 0);
            if (hitters) {
                const int first = __builtin_amdgcn_readlane(c, (int)__builtin_ctzll(hitters));
                if (__ballot(c >= 0 && c != first) == 0ull) {
                    uint32_t m = tb;
                    for (int off = 32; off > 0; off >>= 1) m = min(m, (uint32_t)__shfl_xor((int)m, off));
                    if (lane == (unsigned)__builtin_ctzll(hitters)) {
                        atomicAdd(&h.hit_count[first], (uint32_t)__popcll(hitters));
                        if (h.earliest) atomicMin(&h.earliest[first], m);
                    }
                } else if (c >= 0) {
                    atomicAdd(&h.hit_count[c], 1u);
                    if (h.earliest) atomicMin(&h.earliest[c], tb);
                }
            }
        }
    }
    if (__ballot(aborts != 0u)) {
        for (int off = 32; off > 0; off >>= 1) aborts |= __shfl_down(aborts, off);
        if (lane == 0 && aborts) atomicOr(words + 2, aborts);
    }
    if (!h.want) return;
    // exclusive prefix of `mine` over the block: wave scan, then the waves' totals through LDS (as k_copy_hits)
    uint32_t incl = mine;
    for (int off = 1; off < WAVE; off <<= 1) { uint32_t v = __shfl_up(incl, off); if ((int)lane >= off) incl += v; }
    if (lane == WAVE - 1) s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (unsigned w = 0; w < 256 / WAVE; w++) { uint32_t c = s_wave[w]; s_wave[w] = total; total += c; }
        s_wave[256 / WAVE] = total ? atomicAdd(words, total) : 0u;
    }
    __syncthreads();
    if (!h.channels) return;
    uint32_t off = s_wave[256 / WAVE] + s_wave[wave] + incl - mine;
#pragma unroll
    for (int k = 0; k < COPY_ITEMS; k++) {
        if (ch[k] >= 0) {
            if (off < h.capacity) {
                const size_t id = (size_t)(base + (long long)k * 256 + threadIdx.x);
                if (from_record & (1u << k)) {
                    // (64 contiguous bytes instead of nine sparse reads of the arrays just written)
                    const float4 *f = final_rec + 4 * id;
                    const float4 f0 = f[0], f1 = f[1], f2 = f[2], f3 = f[3];
                    store3(h.dst.pos, off, mk3(f0.x, f0.y, f0.z));
                    store3(h.dst.dir, off, mk3(f1.x, f1.y, f1.z));
                    store3(h.dst.pol, off, mk3(f2.x, f2.y, f2.z));
                    h.dst.wavelengths[off] = f0.w;
                    h.dst.t[off] = f1.w;
                    h.dst.flags[off] = __float_as_uint(f3.x);
                    h.dst.last_hit_triangles[off] = __float_as_int(f3.z);
                    h.dst.weights[off] = f2.w;
                    h.dst.evidx[off] = pv.evidx[id];
                } else {
                    copy_photon(pv, id, h.dst, off);
                }
                h.channels[off] = ch[k];
            }
            off++;
        }
    }
}

// ---- DAQ (chroma/cuda/daq.cu) ------------------------------------------------------------------
// interp (interpolate.h:32-57) as used by sample_cdf(rng, n, cdf_x, cdf_y) (random.h:26-31)
__device__ inline float interp_table(float x, int n, const float *xp, const float *fp)
{
    int lower = 0;
    int upper = n - 1;
    if (x <= xp[lower]) return fp[lower];
    if (x >= xp[upper]) return fp[upper];
    while (lower < upper - 1) {
        int half = (lower + upper) / 2;
        if (x < xp[half]) upper = half; else lower = half;
    }
    float df = fp[upper] - fp[lower];
    float dx = xp[upper] - xp[lower];
    return fp[lower] + df * (x - xp[lower]) / dx;
}

__global__ void k_daq_reset(float maxtime, uint32_t n, uint32_t *time_ints, uint32_t *q_ints, uint32_t *histories)
{
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) {
        time_ints[id] = __float_as_uint(maxtime);
        q_ints[id] = 0u;
        histories[id] = 0u;
    }
}

// run_daq (daq.cu:35-86)
__global__ void k_run_daq(GeoView g, chroma_daq_tables tab, int first_photon, int nphotons, uint32_t detection_state,
                          const float *photon_times, const uint32_t *photon_histories, const int32_t *last_hit_triangles,
                          const float *weights, uint64_t seed, uint64_t id_base, uint32_t acquisition, float global_weight,
                          uint32_t *earliest_time_int, uint32_t *channel_q_int, uint32_t *channel_histories)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= nphotons) return;
    int photon_id = id + first_photon;
    int triangle_id = last_hit_triangles[photon_id];
    if (triangle_id <= -1) return;
    uint32_t history = photon_histories[photon_id];
    int channel_index = g.solid_id_to_channel_index[g.solid_id_map[triangle_id]];
    if (channel_index < 0 || !(history & detection_state)) return;
    cm_rng rng;
    cm_rng_init(&rng, seed, id_base + (uint64_t)photon_id, 0);
    rng.stream = 1u + acquisition;
    float weight = weights[photon_id] * global_weight;
    if (cm_rng_uniform(&rng) < weight) {
        float time = photon_times[photon_id] + interp_table(cm_rng_uniform(&rng), tab.time_cdf_len, tab.d_time_cdf_y, tab.d_time_cdf_x);
        float charge = interp_table(cm_rng_uniform(&rng), tab.charge_cdf_len, tab.d_charge_cdf_y, tab.d_charge_cdf_x);
        uint32_t charge_int = (uint32_t)cm_roundf(charge / tab.charge_unit);
        atomicMin(earliest_time_int + channel_index, __float_as_uint(time));
        atomicAdd(channel_q_int + channel_index, charge_int);
        atomicOr(channel_histories + channel_index, history);
    }
}

// run_daq_many (daq.cu:88-150): ndaq independent acquisitions of the same photons side by side, copy i
// in channels [i * stride, (i + 1) * stride); a copy adds a unit normal jitter to the hit time.  The
// reference gives a photon a block and its copies the block's threads; here a thread is one (photon,
// copy) pair and copy i draws from words 8 i ... of the photon's DAQ stream, so copies are independent
// and the result does not depend on the launch shape.
__global__ void k_run_daq_many(GeoView g, chroma_daq_tables tab, int first_photon, int nphotons, uint32_t detection_state,
                               const float *photon_times, const uint32_t *photon_histories, const int32_t *last_hit_triangles,
                               const float *weights, uint64_t seed, uint64_t id_base, uint32_t acquisition, float global_weight,
                               int ndaq, int channel_stride,
                               uint32_t *earliest_time_int, uint32_t *channel_q_int, uint32_t *channel_histories)
{
    long long id = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= (long long)nphotons * ndaq) return;
    int photon_id = (int)(id / ndaq) + first_photon, copy = (int)(id % ndaq);
    int triangle_id = last_hit_triangles[photon_id];
    if (triangle_id <= -1) return;
    uint32_t history = photon_histories[photon_id];
    int channel_index = g.solid_id_to_channel_index[g.solid_id_map[triangle_id]];
    if (channel_index < 0 || !(history & detection_state)) return;
    cm_rng rng;
    cm_rng_init(&rng, seed, id_base + (uint64_t)photon_id, 8u * (uint32_t)copy);
    rng.stream = 1u + acquisition;
    float weight = weights[photon_id] * global_weight;
    int channel_offset = channel_index + copy * channel_stride;
    if (cm_rng_uniform(&rng) < weight) {
        float jitter = cm_rng_normal(&rng);
        float time = photon_times[photon_id] + jitter +
                     interp_table(cm_rng_uniform(&rng), tab.time_cdf_len, tab.d_time_cdf_y, tab.d_time_cdf_x);
        float charge = interp_table(cm_rng_uniform(&rng), tab.charge_cdf_len, tab.d_charge_cdf_y, tab.d_charge_cdf_x);
        uint32_t charge_int = (uint32_t)cm_roundf(charge / tab.charge_unit);
        atomicMin(earliest_time_int + channel_offset, __float_as_uint(time));
        atomicAdd(channel_q_int + channel_offset, charge_int);
        atomicOr(channel_histories + channel_offset, history);
    }
}
__global__ void k_daq_convert(uint32_t n, float charge_unit, const uint32_t *time_ints, const uint32_t *q_ints, float *t_out, float *q_out)
{
    uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id < n) {
        t_out[id] = __uint_as_float(time_ints[id]);
        q_out[id] = (float)q_ints[id] * charge_unit;
    }
}

// distance_to_mesh (chroma/cuda/mesh.h:124-151)
template <int LDS_N, bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) void
k_distance_to_mesh(GeoView g, int nthreads, const float *origin, const float *direction, const int32_t *last_hit_in,
                   float *distance_out, int32_t *triangle_out, DeviceCounters *counters)
{
    __shared__ uint32_t s_lds[TRAV_LDS_WORDS(LDS_N, PROP_BLOCK)];
    int id = blockIdx.x * PROP_BLOCK + threadIdx.x;
    LaneCounters cnt = {0, 0, 0, 0};
    bool on = id < nthreads;
    v3 o = mk3(0.f, 0.f, 0.f), d = mk3(0.f, 0.f, 1.f);
    if (on) {
        o = load3(origin, id);
        d = load3(direction, id);
        d = d / norm(d);
    }
    float dist;
    const int last_hit = (on && last_hit_in) ? last_hit_in[id] : -1;
    int tri = intersect_mesh<LDS_N, PROP_BLOCK, COUNT>(g, o, d, dist, last_hit, s_lds + threadIdx.x, cnt, on);
    if (on) {
        if (tri != -1) distance_out[id] = dist;
        if (triangle_out) triangle_out[id] = tri;
    }
    unsigned long long ov = wave_sum_u64(cnt.overflows);
    if (COUNT) {
        unsigned long long nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane_id() == 0) { atomicAdd(&counters->nodes_visited, nd); atomicAdd(&counters->triangles_tested, tr); }
    }
    if (lane_id() == 0 && ov) atomicAdd(&counters->stack_overflows, ov);
}

// ---- render (chroma/cuda/render.cu:37-181) ---------------------------------------------------------------
// One lane per ray: EVERY triangle along the ray (no nearest-hit pruning: intersect_node without a distance,
// render.cu:107), the `alpha_depth` nearest kept as a list sorted by distance -- an equal distance goes in FRONT of
// the ones already there (searchsorted/insert, sorting.h:64-98), so the order of discovery is part of the
// result -- then composited front to back over the background colour.  The walk is therefore the reference's
// own: its tree, its child order, its box arithmetic; the lists live in the caller's arrays (GPURays.dx /
// .color / .dxlen), which is what lets a second render continue the first (keep_last_render).
__device__ inline uint32_t render_searchsorted(uint32_t n, const float *arr, float x)       // sorting.h:64-87
{
    uint32_t jl = 0, ju = n;
    const bool ascnd = arr[n - 1] >= arr[0];
    while (ju - jl > 1) {
        const uint32_t jm = (ju + jl) >> 1;
        if ((x > arr[jm]) == ascnd) jl = jm; else ju = jm;
    }
    return ((x <= arr[0]) == ascnd) ? 0u : ju;
}

template <int LDS_N>
__global__ __launch_bounds__(PROP_BLOCK) void
k_render(GeoView g, const uint32_t *colors, int nthreads, const float *origin_in, const float *direction_in, uint32_t alpha_depth,
         uint32_t *pixels, float *dx_all, uint32_t *dxlen, float4 *color_all, uint32_t bg_color, DeviceCounters *counters)
{
    __shared__ uint32_t s_lds[LDS_N * PROP_BLOCK];
    const int id = blockIdx.x * PROP_BLOCK + threadIdx.x;
    if (id >= nthreads) return;                    // (lanes are independent: no wave-wide votes below)
    const v3 origin = load3(origin_in, id), direction = load3(direction_in, id);      // as given: NOT normalised (render.cu:57-58)
    uint32_t n = dxlen[id];
    const v3 noid = (-origin) / direction;
    const v3 inv_dir = 1.0f / direction;
    const v3 wo = mk3(g.world_origin[0], g.world_origin[1], g.world_origin[2]);
    const float ws = g.world_scale;
#define R_LO(nd) mk3(wo.x + (float)((nd).x & 0xFFFFu) * ws, wo.y + (float)((nd).y & 0xFFFFu) * ws, wo.z + (float)((nd).z & 0xFFFFu) * ws)
#define R_HI(nd) mk3(wo.x + (float)((nd).x >> 16) * ws, wo.y + (float)((nd).y >> 16) * ws, wo.z + (float)((nd).z >> 16) * ws)
    const uint4 root = g.nodes[0];
    if (n < 1 && box_tmin(origin, noid, inv_dir, R_LO(root), R_HI(root), ws) < 0.0f) {
        pixels[id] = bg_color;
        return;
    }
    TravStack<LDS_N, PROP_BLOCK> stack;
    stack.lds = s_lds + threadIdx.x;
    int sp = 0;
    bool overflow = false;
    stack.put(sp++, root.w);
    float *dx = dx_all + (size_t)id * alpha_depth;
    float4 *color_a = color_all + (size_t)id * alpha_depth;
    while (sp > 0 && !overflow) {
        const uint32_t w = stack.get(--sp);
        const uint32_t first = w & ~CHROMA_NCHILD_MASK, nchild = w >> CHROMA_CHILD_BITS;
        for (uint32_t i = first; i < first + nchild; i++) {
            const uint4 nd = g.nodes[i];
            if (box_tmin(origin, noid, inv_dir, R_LO(nd), R_HI(nd), ws) < 0.0f) continue;
            const uint32_t child = nd.w & ~CHROMA_NCHILD_MASK;
            if ((nd.w >> CHROMA_CHILD_BITS) != 0) {
                if (sp >= LDS_N + STACK_SCRATCH) { overflow = true; break; }      // cannot happen when the host check passed
                stack.put(sp++, nd.w);
                continue;
            }
            const float4 *t = g.tri + TRI_STRIDE * (size_t)child;                  // leaf: the triangle record (device order)
            const float4 a = t[0], b = t[1], c = t[2];
            const v3 v0 = mk3(a.x, a.y, a.z), v1 = mk3(b.x, b.y, b.z), v2 = mk3(c.x, c.y, c.z);
            float distance;
            if (!intersect_triangle(origin, direction, v0, v1, v2, distance)) continue;
            // get_color (render.cu:11-32)
            const v3 normal = normalize(cross(v1 - v0, v2 - v1));
            float cos_theta = dot(normal, -direction);
            if (cos_theta < 0.0f) cos_theta = -cos_theta;
            const uint32_t rgba = colors[__float_as_uint(b.w)];
            const float4 color = make_float4((float)(0xffu & (rgba >> 16)) * cos_theta, (float)(0xffu & (rgba >> 8)) * cos_theta,
                                             (float)(0xffu & rgba) * cos_theta, (float)(255u - (0xffu & (rgba >> 24))) / 255.0f);
            if (n < 1) {
                dx[0] = distance;
                color_a[0] = color;
            } else {
                const uint32_t j = render_searchsorted(n, dx, distance);
                if (j <= alpha_depth - 1u) {
                    for (uint32_t k = alpha_depth - 1u; k > j; k--) { dx[k] = dx[k - 1]; color_a[k] = color_a[k - 1]; }     // sorting.h:89-98
                    dx[j] = distance;
                    color_a[j] = color;
                }
            }
            if (n < alpha_depth) n++;
        }
    }
#undef R_LO
#undef R_HI
    if (overflow) atomicAdd(&counters->stack_overflows, 1ull);
    if (n < 1) {
        pixels[id] = bg_color;
        return;
    }
    dxlen[id] = n;
    float scale = 1.0f, fr = 0.0f, fg = 0.0f, fb = 0.0f;
    for (uint32_t i = 0; i < n; i++) {
        const float4 ci = color_a[i];
        const float alpha = ci.w;
        fr += scale * ci.x * alpha;
        fg += scale * ci.y * alpha;
        fb += scale * ci.z * alpha;
        scale *= (1.0f - alpha);
    }
    // (the reference divides by the double literal 255.0 here, render.cu:163)
    const float alpha = (float)((double)(float)((bg_color & 0xFF000000u) >> 24) / 255.0);
    fr += scale * (float)((bg_color & 0xFF0000u) >> 16) * alpha;
    fg += scale * (float)((bg_color & 0xFF00u) >> 8) * alpha;
    fb += scale * (float)(bg_color & 0xFFu) * alpha;
    scale *= (1.0f - alpha);
    const uint32_t av = (n < alpha_depth) ? cm_f2u32(cm_floorf(255.0f * (1.0f - scale))) : 255u;
    const uint32_t red = cm_f2u32(cm_floorf(fr / (1.0f - scale)));
    const uint32_t green = cm_f2u32(cm_floorf(fg / (1.0f - scale)));
    const uint32_t blue = cm_f2u32(cm_floorf(fb / (1.0f - scale)));
    pixels[id] = av << 24 | red << 16 | green << 8 | blue;
}

// chroma/cuda/transform.cu: translate / rotate / rotate_around_point of a point array
__global__ void k_rays_transform(int n, float *a, int mode, float phi, float ax, float ay, float az, float px, float py, float pz)
{
    int id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n) return;
    v3 p = load3(a, id);
    const v3 axis = mk3(ax, ay, az), point = mk3(px, py, pz);
    if (mode == 0) p = p + point;                                        // translate by `point`
    else if (mode == 1) p = rotate(p, phi, axis);
    else { p = p - point; p = rotate(p, phi, axis); p = p + point; }
    store3(a, id, p);
}

// isotropic photon bomb (chroma/benchmark.py:77-83 with chroma/sample.py:16-30's formulas)
__global__ void k_generate_bomb(PhotonView pv, uint64_t n, uint64_t seed, uint64_t id_base, float px, float py, float pz,
                                float wl_lo, float wl_hi)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    cm_rng rng;
    cm_rng_init(&rng, seed, 0xB0B0000000000000ull + id_base + i, 0);
    v3 dir = uniform_sphere(rng);
    v3 aux = uniform_sphere(rng);
    v3 pol = cross(aux, dir);
    pol = pol / norm(pol);
    float wl = (wl_hi > wl_lo) ? uniform(rng, wl_lo, wl_hi) : wl_lo;
    store3(pv.pos, i, mk3(px, py, pz));
    store3(pv.dir, i, dir);
    store3(pv.pol, i, pol);
    pv.wavelengths[i] = wl;
    pv.t[i] = 0.0f;
    pv.flags[i] = 0u;
    pv.last_hit_triangles[i] = -1;
    pv.weights[i] = 1.0f;
    pv.evidx[i] = 0u;
    pv.rng_counters[i] = 0u;
}

// chroma_probe: single device functions of the path, one call per element (tests pin them on the oracle
// and on the reference's own headers compiled for gfx950 by the test infrastructure)
__global__ void k_probe(int fn, uint64_t n, const float *x, const float *tab_x, const float *tab_f, uint32_t ntab,
                        float start, float step, float *out)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (fn == 0) {
        GeoView g;
        g.wavelength_n = ntab; g.wavelength_start = start; g.wavelength_step = step;
        out[i] = interp_property(g, x[i], tab_f);
    } else if (fn == 1) {
        out[i] = interp_idx(x[i], (int)ntab, tab_x);
    } else if (fn == 2) {
        out[i] = interp_table(x[i], (int)ntab, tab_x, tab_f);
    } else {
        const float *p = x + 7 * i;
        v3 r = rotate(mk3(p[0], p[1], p[2]), p[3], mk3(p[4], p[5], p[6]));
        float *o = out + 5 * i;
        o[0] = r.x; o[1] = r.y; o[2] = r.z; o[3] = cm_cosf(p[3]); o[4] = cm_sinf(p[3]);
    }
}

// ---------------------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------------------
static PhotonView to_view(const chroma_photon_arrays *a)
{
    PhotonView v;
    v.pos = a->pos; v.dir = a->dir; v.pol = a->pol; v.wavelengths = a->wavelengths; v.t = a->t;
    v.flags = a->flags; v.last_hit_triangles = a->last_hit_triangles; v.weights = a->weights;
    v.evidx = a->evidx; v.rng_counters = a->rng_counters;
    return v;
}

static int check_photons(const chroma_photon_arrays *a, bool need_rng)
{
    if (!a || !a->pos || !a->dir || !a->pol || !a->wavelengths || !a->t || !a->flags || !a->last_hit_triangles ||
        !a->weights || !a->evidx || (need_rng && !a->rng_counters))
        return set_error(CHROMA_ERR_INVALID, "photon arrays: null pointer");
    return CHROMA_OK;
}


template <bool COUNT>
static int launch_propagate_t(chroma_ctx *ctx, chroma_geometry *geom, PhotonView pv, int first, int nthreads,
                              const uint32_t *in_q, uint32_t *out_q, chroma_rng rng, int max_steps, int use_weights,
                              int scatter_first)
{
    dim3 grid((unsigned)((nthreads + PROP_BLOCK - 1) / PROP_BLOCK)), block(PROP_BLOCK);
    uint32_t need = geom->stack_need;
#define LAUNCH(N)                                                                                         \
    hipLaunchKernelGGL((k_propagate<N, COUNT>), grid, block, 0, ctx->stream, geom->view, pv, first, nthreads, \
                       in_q, out_q, rng.seed, rng.photon_id_base, max_steps, use_weights, scatter_first, ctx->d_counters)
    if (need <= STACK_LDS + STACK_SCRATCH) LAUNCH(STACK_LDS);
    else return set_error(CHROMA_ERR_STACK, "BVH needs %u traversal stack entries, more than the %d supported", need, STACK_LDS + STACK_SCRATCH);
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

static int launch_propagate(chroma_ctx *ctx, chroma_geometry *geom, PhotonView pv, int first, int nthreads,
                            const uint32_t *in_q, uint32_t *out_q, chroma_rng rng, int max_steps, int use_weights,
                            int scatter_first)
{
    if (nthreads <= 0) return CHROMA_OK;
    if (ctx->counting)
        return launch_propagate_t<true>(ctx, geom, pv, first, nthreads, in_q, out_q, rng, max_steps, use_weights, scatter_first);
    return launch_propagate_t<false>(ctx, geom, pv, first, nthreads, in_q, out_q, rng, max_steps, use_weights, scatter_first);
}

// the per-ray slices of global memory for stack entries beyond the LDS part: every cooperative walk indexes
// it with (wave * rays-per-wave + ray) * COOP_SPILL, so it is sized for the largest grid of any of them
static size_t spill_entries(const chroma_ctx *ctx)
{
    size_t rays = std::max(std::max((size_t)ctx->coop_waves * 8, (size_t)ctx->quad_waves * 16), (size_t)ctx->pair_waves * 32);
    return rays * COOP_SPILL;
}

// one step for many photons: ray cast and physics as two launches
// One step as ray set-up + ray cast + physics (+ the strict walk and the physics of the few rays that
// need it), all reading the photon count and the launch policy from ctx->d_step (k_step_begin).
// `n_upper` bounds the count and sizes the grids; `in_q`/`out_q` are whole queues (slot 0 = tail) and
// `work_in`/`work_out` the working sets that go with them.  With `ev` (5 events): [0] step start,
// [3] ray-cast kernel start, [1] its end, [4] end of the main physics pass, [2] step end.
// The walk whose steps chain their ray records from kernel to kernel (k_load_working -> k_raycast_quad -> k_physics ->
// k_raycast_quad ...) instead of running k_ray_setup: the default one.
static bool step_uses_quad_walk(const chroma_ctx *ctx, const chroma_geometry *geom)
{
    return geom->view.wnodes != nullptr && ctx->wide_walk == CHROMA_WALK_QUAD && geom->wide_stack_need <= QUAD_STACK + COOP_SPILL;
}

// `rays_ready`: the records of this step are in ctx->rays already (written by k_load_working or by the k_physics of
// the step before).  With the default walk the records of the next step go to ctx->rays_b, and the two are swapped.
static int launch_split_step(chroma_ctx *ctx, chroma_geometry *geom, PhotonView pv, long long n_upper, const uint32_t *in_q,
                             uint32_t *out_q, const float4 *work_in, float4 *work_out, chroma_rng rng, int use_weights,
                             int scatter_first, hipEvent_t *ev = nullptr, uint32_t first_n = 0, bool rays_ready = false, bool packet = false)
{
    // (`packet`: the first step of a call whose k_load_working looked at the photons' coherence: k_raycast_packet is
    //  launched before k_raycast_quad, and the word k_packet_decide wrote tells the two which of them has the step)
    if (n_upper <= 0) return CHROMA_OK;
    uint32_t need = geom->stack_need;
    if (need > STACK_LDS + STACK_SCRATCH)
        return set_error(CHROMA_ERR_STACK, "BVH needs %u traversal stack entries, more than the %d supported", need, STACK_LDS + STACK_SCRATCH);
    const bool have_wide = geom->view.wnodes != nullptr;
    if (ctx->wide_walk == CHROMA_WALK_LITERAL || ctx->wide_walk == CHROMA_WALK_LITERAL_LANE) {
        // the reference's own loop for every ray (mesh.h:42-118 as it stands: its tree, its order, its box arithmetic,
        // every triangle tested the moment its leaf box is entered), then the physics on the results as they are.
        // LITERAL: k_raycast_literal (four lanes per ray, persistent waves; raycast_literal.h) + the strict lane-per-ray
        // loop for the few rays whose 1/d is not moderate; LITERAL_LANE: the strict loop for every ray (the cross-check).
        const bool lane_walk = ctx->wide_walk == CHROMA_WALK_LITERAL_LANE;
        StepState *st = ctx->d_step;
        hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1), 0, ctx->stream, in_q, out_q, st,
                           use_weights ? 0xFFFFFFFFu : (uint32_t)(PROP_BLOCK * 16 * 8), first_n);
        if (ev) HIP_TRY(hipEventRecord(ev[0], ctx->stream));
        if (!lane_walk && !ctx->coop_spill) {
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(ctx_malloc(ctx, (void **)&ctx->coop_spill, spill_entries(ctx) * sizeof(uint2)));
        }
        unsigned sblocks = (unsigned)std::min<long long>((n_upper + 255) / 256, (long long)ctx->physics_blocks * 4);
        hipLaunchKernelGGL(k_ray_setup, dim3(sblocks), dim3(256), 0, ctx->stream, geom->view, work_in, st, ctx->rays,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, &st->retry, lane_walk ? 0 : 1);
        if (ev) { HIP_TRY(hipEventRecord(ev[3], ctx->stream)); HIP_TRY(hipEventRecord(ev[5], ctx->stream)); }
        if (lane_walk) {
            const unsigned lblocks = (unsigned)std::min<long long>((n_upper + PROP_BLOCK - 1) / PROP_BLOCK, (long long)ctx->persistent_waves);
            if (ctx->counting)
                hipLaunchKernelGGL((k_raycast_retry<true, true>), dim3(lblocks), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
            else
                hipLaunchKernelGGL((k_raycast_retry<false, true>), dim3(lblocks), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
        } else {
            const unsigned lwaves = (unsigned)std::min<long long>((n_upper + 15) / 16, (long long)ctx->quad_waves);
            const unsigned rblocks = (unsigned)std::min<long long>((n_upper + PROP_BLOCK - 1) / PROP_BLOCK, 8 * 256);
            if (ctx->counting) {
                hipLaunchKernelGGL((k_raycast_literal<true>), dim3(lwaves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk);
                hipLaunchKernelGGL((k_raycast_retry<true>), dim3(rblocks), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
            } else {
                hipLaunchKernelGGL((k_raycast_literal<false>), dim3(lwaves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk);
                hipLaunchKernelGGL((k_raycast_retry<false>), dim3(rblocks), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
            }
        }
        if (ev) HIP_TRY(hipEventRecord(ev[1], ctx->stream));
        const bool deal = PHYS_DEAL != 0 && geom->view.plain_optics != 0;
        const int pb = deal ? PHYS_DEAL_BLOCK : PHYS_BLOCK_OF(geom->view.plain_optics == 0);
        unsigned pblocks = (unsigned)std::min<long long>((n_upper + pb - 1) / pb, std::max<long long>(1, (long long)ctx->physics_blocks * PHYS_BLOCK / pb));
        DeviceCounters *pc = ctx->counting ? ctx->d_counters : nullptr;
        if (deal)
            hipLaunchKernelGGL(k_physics_deal, dim3(pblocks), dim3(PHYS_DEAL_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in, out_q, work_out,
                               ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights, scatter_first,
                               ctx->retry_list, 2, pc, (float4 *)nullptr);
        else if (geom->view.plain_optics != 0)
            hipLaunchKernelGGL((k_physics<false>), dim3(pblocks), dim3(PHYS_BLOCK_OF(false)), 0, ctx->stream, geom->view, pv, st, work_in, out_q, work_out,
                               ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights, scatter_first,
                               ctx->retry_list, 2, pc, (float4 *)nullptr, ctx->final_use, ctx->final_epoch);
        else
            hipLaunchKernelGGL((k_physics<true>), dim3(pblocks), dim3(PHYS_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in, out_q, work_out,
                               ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights, scatter_first,
                               ctx->retry_list, 2, pc, (float4 *)nullptr, ctx->final_use, ctx->final_epoch);
        if (ev) { HIP_TRY(hipEventRecord(ev[4], ctx->stream)); HIP_TRY(hipEventRecord(ev[2], ctx->stream)); }
        HIP_TRY(hipGetLastError());
        return CHROMA_OK;
    }
    const bool pair = ctx->wide_walk == CHROMA_WALK_PAIR && have_wide && geom->wide_stack_need <= PAIR_STACK + COOP_SPILL;
    const bool quad = !pair && (ctx->wide_walk == CHROMA_WALK_QUAD || ctx->wide_walk == CHROMA_WALK_PAIR) && have_wide &&
                      geom->wide_stack_need <= QUAD_STACK + COOP_SPILL;
    const bool coop = !pair && !quad && (ctx->wide_walk == CHROMA_WALK_COOP || ctx->wide_walk == CHROMA_WALK_QUAD) && have_wide &&
                      geom->wide_stack_need <= COOP_STACK + COOP_SPILL;
    const bool wide = !pair && !coop && !quad && ctx->wide_walk != CHROMA_WALK_REFERENCE && have_wide && geom->wide_stack_need <= WIDE_STACK + WIDE_SPILL;
    if (wide && !ctx->wide_spill) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(ctx_malloc(ctx, (void **)&ctx->wide_spill, (size_t)ctx->wide_waves * WIDE_SPILL * PROP_BLOCK * sizeof(uint2)));
    }
    if ((coop || quad || pair) && !ctx->coop_spill) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(ctx_malloc(ctx, (void **)&ctx->coop_spill, spill_entries(ctx) * sizeof(uint2)));
    }
    // persistent ray cast: enough waves to fill the chip, each pulling rays from the queue
    unsigned waves = pair ? (unsigned)std::min<long long>((n_upper + 31) / 32, (long long)ctx->pair_waves)
                   : quad ? (unsigned)std::min<long long>((n_upper + 15) / 16, (long long)ctx->quad_waves)
                   : coop ? (unsigned)std::min<long long>((n_upper + 7) / 8, (long long)ctx->coop_waves)
                          : (unsigned)std::min<long long>((n_upper + PROP_BLOCK - 1) / PROP_BLOCK,
                                                          (long long)(wide ? ctx->wide_waves : ctx->persistent_waves));
    dim3 grid(waves), block(PROP_BLOCK);
    StepState *st = ctx->d_step;
    // (with weights the reference runs ALL steps in one launch: every count is "few")
    hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1), 0, ctx->stream, in_q, out_q, st,
                       use_weights ? 0xFFFFFFFFu : (uint32_t)(PROP_BLOCK * 16 * 8), first_n);
    if (ev) HIP_TRY(hipEventRecord(ev[0], ctx->stream));
    const bool chained = quad && step_uses_quad_walk(ctx, geom);
    if (!(chained && rays_ready)) {
        unsigned sblocks = (unsigned)std::min<long long>((n_upper + 255) / 256, (long long)ctx->physics_blocks * 4);
        hipLaunchKernelGGL(k_ray_setup, dim3(sblocks), dim3(256), 0, ctx->stream, geom->view, work_in, st, ctx->rays,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, &st->retry);
    }
    const int settle = (chained && rays_ready) ? 1 : 0;
    float4 *rays_next = chained ? ctx->rays_b : nullptr;
    if (ev) HIP_TRY(hipEventRecord(ev[3], ctx->stream));        // the ray-cast kernels proper are timed from here
    const bool offer_packet = packet && chained && rays_ready;
    const uint32_t *skip_quad = offer_packet ? ctx->d_words + 4 : nullptr;
    if (offer_packet) {
        const unsigned pwaves = (unsigned)std::min<long long>((n_upper + WAVE - 1) / WAVE, (long long)ctx->quad_waves);
        if (ctx->counting)
            hipLaunchKernelGGL((k_raycast_packet<true>), dim3(pwaves), block, 0, ctx->stream, geom->view, ctx->rays, st, ctx->hit_triangle,
                               ctx->hit_distance, ctx->retry_list, ctx->d_counters, ctx->d_words + 4);
        else
            hipLaunchKernelGGL((k_raycast_packet<false>), dim3(pwaves), block, 0, ctx->stream, geom->view, ctx->rays, st, ctx->hit_triangle,
                               ctx->hit_distance, ctx->retry_list, ctx->d_counters, ctx->d_words + 4);
    }
    if (ev) HIP_TRY(hipEventRecord(ev[5], ctx->stream));        // (k_raycast_packet before, the step's other ray cast after)
#define RAYCAST_LAUNCH(COUNT)                                                                                          \
    do {                                                                                                               \
        if (pair)                                                                                                      \
            hipLaunchKernelGGL((k_raycast_pair<COUNT>), grid, block, 0, ctx->stream, geom->view, ctx->rays, 0, st,      \
                               ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk); \
        else if (quad)                                                                                                 \
            hipLaunchKernelGGL((k_raycast_quad<COUNT>), grid, block, 0, ctx->stream, geom->view, ctx->rays, 0, st,      \
                               ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk, settle, skip_quad); \
        else if (coop)                                                                                                 \
            hipLaunchKernelGGL((k_raycast_coop<COUNT>), grid, block, 0, ctx->stream, geom->view, ctx->rays, 0, st,      \
                               ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk); \
        else if (wide)                                                                                                 \
            hipLaunchKernelGGL((k_raycast_wide<COUNT>), grid, block, 0, ctx->stream, geom->view, ctx->rays, 0, st,      \
                               ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->wide_spill, ctx->d_counters, ctx->ray_chunk); \
        else                                                                                                           \
            hipLaunchKernelGGL((k_raycast_persistent<COUNT>), grid, block, 0, ctx->stream, geom->view, ctx->rays, 0, st, \
                               ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);                \
        if (ev) HIP_TRY(hipEventRecord(ev[1], ctx->stream));                                                            \
    } while (0)
    if (ctx->counting) RAYCAST_LAUNCH(true); else RAYCAST_LAUNCH(false);
#undef RAYCAST_LAUNCH
    // physics for every slot whose hit is regular; then the strict walk and the physics of the rest
    const bool plain = geom->view.plain_optics != 0;      // (no re-emitting component, default surface model only)
    const bool deal = PHYS_DEAL != 0 && plain && !ctx->final_use;          // (photons of a block dealt by what happens to them: k_physics_deal)
    const int pb = deal ? PHYS_DEAL_BLOCK : PHYS_BLOCK_OF(!plain);
    unsigned pblocks = (unsigned)std::min<long long>((n_upper + pb - 1) / pb, std::max<long long>(1, (long long)ctx->physics_blocks * PHYS_BLOCK / pb));
    DeviceCounters *pc = ctx->counting ? ctx->d_counters : nullptr;
    if (deal)
        hipLaunchKernelGGL(k_physics_deal, dim3(pblocks), dim3(PHYS_DEAL_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in, out_q, work_out,
                           ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights, scatter_first,
                           ctx->retry_list, 0, pc, rays_next);
    else if (plain)
        hipLaunchKernelGGL((k_physics<false>), dim3(pblocks), dim3(PHYS_BLOCK_OF(false)), 0, ctx->stream, geom->view, pv, st, work_in, out_q, work_out,
                           ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights, scatter_first,
                           ctx->retry_list, 0, pc, rays_next, ctx->final_use, ctx->final_epoch);
    else
        hipLaunchKernelGGL((k_physics<true>), dim3(pblocks), dim3(PHYS_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in, out_q, work_out,
                           ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights, scatter_first,
                           ctx->retry_list, 0, pc, rays_next, ctx->final_use, ctx->final_epoch);
    if (ev) HIP_TRY(hipEventRecord(ev[4], ctx->stream));          // end of the main physics pass
    // (both passes stride over the list and leave at once when it is short -- the usual case -- but a plain geometry
    //  with faces on the world box lists a good part of its hits for the exact check: grids for that)
    const unsigned rblocks = (unsigned)std::min<long long>((n_upper + PROP_BLOCK - 1) / PROP_BLOCK, 8 * 256);
    if (ctx->counting)
        hipLaunchKernelGGL((k_raycast_retry<true>), dim3(rblocks), block, 0, ctx->stream, geom->view, ctx->rays, st,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
    else
        hipLaunchKernelGGL((k_raycast_retry<false>), dim3(rblocks), block, 0, ctx->stream, geom->view, ctx->rays, st,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
    // (the list is ~1e-3 of the slots with plain optics: an eighth of the grid strides over it in a round or two, and a
    //  launch of 2048 blocks that find nothing to do costs 0.07 ms, 29 times per batch)
    const unsigned fblocks = plain ? std::max(std::min(pblocks, 64u), pblocks / 8) : pblocks;
    if (deal)
        hipLaunchKernelGGL(k_physics_deal, dim3(fblocks), dim3(PHYS_DEAL_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in, out_q,
                           work_out, ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights,
                           scatter_first, ctx->retry_list, 1, pc, rays_next);
    else if (plain)
        hipLaunchKernelGGL((k_physics<false>), dim3(fblocks), dim3(PHYS_BLOCK_OF(false)), 0, ctx->stream, geom->view, pv, st, work_in, out_q,
                           work_out, ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights,
                           scatter_first, ctx->retry_list, 1, pc, rays_next, ctx->final_use, ctx->final_epoch);
    else
        hipLaunchKernelGGL((k_physics<true>), dim3(fblocks), dim3(PHYS_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in, out_q,
                           work_out, ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights,
                           scatter_first, ctx->retry_list, 1, pc, rays_next, ctx->final_use, ctx->final_epoch);
    if (ev) HIP_TRY(hipEventRecord(ev[2], ctx->stream));
    HIP_TRY(hipGetLastError());
    if (chained) std::swap(ctx->rays, ctx->rays_b);       // (what k_physics wrote is the next step's input)
    return CHROMA_OK;
}

// All remaining steps of the last photons in one launch (k_tail_coop).  Returns CHROMA_OK and sets
// *done when the geometry has a wide tree the kernel can walk; otherwise leaves *done false.
static int launch_tail(chroma_ctx *ctx, chroma_geometry *geom, PhotonView pv, long long n_upper, const uint32_t *in_q,
                       uint32_t *out_q, const float4 *work_in, chroma_rng rng, int nsteps, int use_weights, int scatter_first,
                       hipEvent_t *ev, bool *done, uint32_t first_n = 0)
{
    *done = false;
    if (!geom->view.wnodes || geom->wide_stack_need > COOP_STACK + COOP_SPILL || geom->stack_need > STACK_LDS + STACK_SCRATCH)
        return CHROMA_OK;
    if (!ctx->coop_spill) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(ctx_malloc(ctx, (void **)&ctx->coop_spill, spill_entries(ctx) * sizeof(uint2)));
    }
    unsigned waves = (unsigned)std::min<long long>((n_upper + 7) / 8, (long long)ctx->coop_waves);
    if ((long long)waves * 8 < n_upper) return CHROMA_OK;          // (cannot happen below 8192 photons)
    StepState *st = ctx->d_step;
    hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1), 0, ctx->stream, in_q, out_q, st,
                       use_weights ? 0xFFFFFFFFu : (uint32_t)(PROP_BLOCK * 16 * 8), first_n);
    if (ev) { HIP_TRY(hipEventRecord(ev[0], ctx->stream)); HIP_TRY(hipEventRecord(ev[1], ctx->stream)); }
    if (ctx->counting)
        hipLaunchKernelGGL((k_tail_coop<true>), dim3(waves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in,
                           rng.seed, rng.photon_id_base, nsteps, use_weights, scatter_first, ctx->coop_spill, ctx->d_counters);
    else
        hipLaunchKernelGGL((k_tail_coop<false>), dim3(waves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in,
                           rng.seed, rng.photon_id_base, nsteps, use_weights, scatter_first, ctx->coop_spill, ctx->d_counters);
    if (ev) HIP_TRY(hipEventRecord(ev[2], ctx->stream));
    HIP_TRY(hipGetLastError());
    *done = true;
    return CHROMA_OK;
}

template <class T>
static int upload(chroma_geometry *g, const T *host, size_t count, const T **dev_out)
{
    *dev_out = nullptr;
    size_t bytes = std::max(count, (size_t)1) * sizeof(T);
    void *d = nullptr;
    HIP_TRY(ctx_malloc(g->ctx, &d, bytes));
    g->allocations.push_back(d);
    g->device_bytes += bytes;
    if (count && host) HIP_TRY(hipMemcpy(d, host, count * sizeof(T), hipMemcpyHostToDevice));
    else HIP_TRY(hipMemset(d, 0, bytes));
    *dev_out = (const T *)d;
    return CHROMA_OK;
}

// chroma_geometry_create's two derived arrays, made on the device from what has just been uploaded
__global__ void k_traversal_nodes(const uint4 *nodes, uint32_t nnodes, const uint32_t *tri_to_dev, uint32_t ntriangles, uint4 *out)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnodes) return;
    uint4 n = nodes[i];
    if ((n.w >> CHROMA_CHILD_BITS) == 0) { const uint32_t t = n.w & ~CHROMA_NCHILD_MASK; n.w = t < ntriangles ? tri_to_dev[t] : n.w; }
    out[i] = n;
}
__global__ void k_triangle_records(const float *vertices, const uint32_t *triangles, const uint32_t *codes, const uint32_t *rank,
                                   const uint32_t *dev_to_tri, uint32_t nrecords, float4 *tri)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= nrecords) return;
    const uint32_t t = dev_to_tri[k];
    const uint32_t extra[3] = {codes[t], t, rank[t]};
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float *vv = vertices + 3 * (size_t)triangles[3 * (size_t)t + c];
        tri[(size_t)TRI_STRIDE * k + c] = make_float4(vv[0], vv[1], vv[2], __uint_as_float(extra[c]));
    }
}
// Worst-case number of simultaneously live stack entries of the depth-first walk in
// intersect_mesh for this tree (every box test succeeding).  Children always have larger
// indices than their parent (layers are stored root first), so one backward sweep suffices.
// Most entries a walk's stack can hold at once, for the two trees of a geometry, from the arrays AS UPLOADED.
// need(node) = max over its inner children c, in push order, of (inner children before c) + need(c) [reference walk, mesh.h:68-110],
// need(node) = inner children - 1 + max need(child) [nearest-first wide walk].  Children follow their parents in both arrays, so
// the values are the least fixed point of these rules: every pass over the array only raises entries, and after (depth of the
// tree) passes nothing changes -- ~30 passes of a few milliseconds instead of a second-long backward sweep on one host core.
__global__ void k_stack_need_ref(const uint4 *nodes, uint32_t nnodes, uint32_t *need, uint32_t *changed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnodes) return;
    const uint32_t w = nodes[i].w, nchild = w >> CHROMA_CHILD_BITS, first = w & ~CHROMA_NCHILD_MASK;
    if (nchild == 0) return;
    uint32_t best = 0;
    if ((uint64_t)first + nchild > nnodes || first <= i) best = 0xFFFFu;
    else {
        uint32_t rank = 0;
        for (uint32_t j = 0; j < nchild; j++)
            if ((nodes[first + j].w >> CHROMA_CHILD_BITS) != 0) { best = max(best, rank + need[first + j]); rank++; }
        best = min(max(best, rank), 0xFFFFu);
    }
    if (best != need[i]) { need[i] = best; *changed = 1u; }
}
__global__ void k_stack_need_wide(const uint4 *wnodes, uint32_t nwide, uint32_t *need, uint32_t *changed)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nwide) return;
    uint32_t inner = 0, below = 0;
    for (int j = 0; j < 8; j++) {
        const uint32_t w = wnodes[8 * (size_t)i + j].w;
        if (w == 0xFFFFFFFFu || (w & 0x80000000u)) continue;
        inner++;
        if (w < nwide && w > i) below = max(below, need[w]);
    }
    const uint32_t v = min(0xFFFFu, inner ? inner - 1u + below : 0u);
    if (v != need[i]) { need[i] = v; *changed = 1u; }
}
// runs `pass` until an entry no longer changes; returns need[0]
template <class Pass>
static int stack_need_fixed_point(chroma_ctx *ctx, size_t n, Pass pass, uint32_t *result)
{
    uint32_t *d_need = nullptr, *d_changed = nullptr;
    HIP_TRY(hipMalloc(&d_need, std::max<size_t>(n, 1) * 4));
    if (hipMalloc(&d_changed, 4) != hipSuccess) { hipFree(d_need); return set_error(CHROMA_ERR_INTERNAL, "out of device memory"); }
    hipError_t e = hipMemsetAsync(d_need, 0, std::max<size_t>(n, 1) * 4, ctx->stream);
    uint32_t changed = 1, h_need = 0;
    for (int it = 0; e == hipSuccess && changed && it < 8192; it++) {
        e = hipMemsetAsync(d_changed, 0, 4, ctx->stream);
        for (int k = 0; k < 4; k++) pass(d_need, d_changed);                    // (four passes per question)
        if (e == hipSuccess) e = hipMemcpyAsync(&changed, d_changed, 4, hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    }
    if (e == hipSuccess) e = hipMemcpy(&h_need, d_need, 4, hipMemcpyDeviceToHost);
    hipFree(d_need); hipFree(d_changed);
    if (e != hipSuccess) return set_error((int)e, "stack need: %s", hipGetErrorString(e));
    if (changed) return set_error(CHROMA_ERR_INVALID, "stack need: the tree does not settle (a child range that points back?)");
    *result = h_need;
    return CHROMA_OK;
}

// ---- distance_to_mesh through the fast ray cast --------------------------------------------------------
// mesh.h:124-151 asks for the nearest triangle along free rays.  Same pipeline as a propagation step:
// ray records, k_raycast_quad, the check that the reference tests the winner (record_hit_is_regular),
// the literal reference walk for the rays that fail it or that the fast walk cannot take.
__global__ void k_rays_from_arrays(GeoView g, int n, const float *origin_in, const float *direction_in, const int32_t *last_hit_in,
                                   float4 *rays, int32_t *hit_triangle, float *hit_distance, uint32_t *retry_list, StepState *st)
{
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    int last_hit = last_hit_in ? last_hit_in[slot] : -1;          // a triangle id (mesh.h:82) -> its record
    last_hit = (last_hit >= 0 && (uint32_t)last_hit < g.ntriangles) ? (int)g.tri_to_dev[last_hit] : -1;
    v3 origin = load3(origin_in, slot), direction = load3(direction_in, slot);
    direction = direction / norm(direction);
    v3 noid = (-origin) / direction;
    v3 inv_dir = 1.0f / direction;
    bool moderate = cm_fabsf(inv_dir.x) < 1e30f && cm_fabsf(inv_dir.y) < 1e30f && cm_fabsf(inv_dir.z) < 1e30f &&
                    cm_fabsf(noid.x) < 1e30f && cm_fabsf(noid.y) < 1e30f && cm_fabsf(noid.z) < 1e30f;
    int status = moderate ? 0 : HIT_RETRY;                 // (a NaN ray is not moderate: the literal walk answers)
    v3 a = mk3(0.f, 0.f, 0.f), b = mk3(0.f, 0.f, 0.f);
    if (moderate) {
        a = ray_fast(g, noid, inv_dir, 1.0f).a;
        b = mk3(cm_fmaf(g.world_origin[0], inv_dir.x, noid.x), cm_fmaf(g.world_origin[1], inv_dir.y, noid.y),
                cm_fmaf(g.world_origin[2], inv_dir.z, noid.z));
    }
    float4 *r = rays + 4 * (size_t)slot;
    r[0] = make_float4(origin.x, origin.y, origin.z, __int_as_float(last_hit));
    r[1] = make_float4(direction.x, direction.y, direction.z, __int_as_float(status));
    r[2] = make_float4(a.x, a.y, a.z, ray_growth(g, origin));
    r[3] = make_float4(b.x, b.y, b.z, 0.0f);
    if (status != 0) {
        hit_triangle[slot] = status;
        hit_distance[slot] = 0.0f;
        retry_list[atomicAdd(&st->retry, 1u)] = (uint32_t)slot;
    }
}
__global__ void k_step_set(StepState *st, uint32_t n) { st->n = n; st->renorm = 0u; st->in_tail = 0u; st->launches = 0u; st->work = 0u; st->retry = 0u; }

// results of the fast cast: checked, translated to triangle ids, or handed to the literal walk
__global__ void k_distance_finish(GeoView g, int n, const float4 *rays, const int32_t *hit_triangle, const float *hit_distance,
                                  float *distance_out, int32_t *triangle_out, uint32_t *retry_list, StepState *st)
{
    int slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= n) return;
    int rec = hit_triangle[slot];
    if (rec == HIT_RETRY) return;                          // already listed
    if (rec >= 0) {
        const float4 *r = rays + 4 * (size_t)slot;
        const float4 r0 = r[0], r1 = r[1];
        const float4 *t = g.tri + TRI_STRIDE * (size_t)rec;
        const float4 a = t[0], b = t[1], c = t[2];
        const float dist = hit_distance[slot];
        if (!record_hit_is_regular(g, a, b, c, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), dist)) {
            retry_list[atomicAdd(&st->retry, 1u)] = (uint32_t)slot;
            return;
        }
        distance_out[slot] = dist;
        if (triangle_out) triangle_out[slot] = (int32_t)__float_as_uint(b.w);
    } else if (triangle_out) {
        triangle_out[slot] = -1;                           // a miss leaves the distance untouched (mesh.h:145-148)
    }
}
template <bool COUNT>
__global__ __launch_bounds__(PROP_BLOCK) void
k_distance_retry(GeoView g, const float4 *rays, const StepState *st, const uint32_t *retry_list, float *distance_out,
                 int32_t *triangle_out, DeviceCounters *counters)
{
    __shared__ uint32_t s_lds[TRAV_LDS_WORDS(STACK_LDS, PROP_BLOCK)];
    const int nretry = (int)st->retry;
    LaneCounters cnt = {0, 0, 0, 0};
    for (int k = blockIdx.x * PROP_BLOCK + threadIdx.x; k < nretry; k += gridDim.x * PROP_BLOCK) {
        const int slot = (int)retry_list[k];
        const float4 *r = rays + 4 * (size_t)slot;
        const float4 r0 = r[0], r1 = r[1];
        float dist;
        int rec = intersect_mesh_dev<STACK_LDS, PROP_BLOCK, COUNT>(g, mk3(r0.x, r0.y, r0.z), mk3(r1.x, r1.y, r1.z), dist,
                                                                   __float_as_int(r0.w), s_lds + threadIdx.x, cnt, true);
        if (rec >= 0) distance_out[slot] = dist;
        if (triangle_out) triangle_out[slot] = rec >= 0 ? (int32_t)g.dev_to_tri[rec] : -1;
    }
    unsigned long long ov = wave_sum_u64(cnt.overflows);
    if (COUNT) {
        unsigned long long nd = wave_sum_u64(cnt.nodes), tr = wave_sum_u64(cnt.tris);
        if (lane_id() == 0) { atomicAdd(&counters->nodes_visited, nd); atomicAdd(&counters->triangles_tested, tr); }
    }
    if (lane_id() == 0 && ov) atomicAdd(&counters->stack_overflows, ov);
}

// ---------------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------------
extern "C" {

const char *chroma_last_error(void) { return g_last_error.c_str(); }
const char *chroma_version(void) { return "chroma_hip 0.1 (gfx950)"; }

int chroma_device_count(int *count)
{
    if (!count) return set_error(CHROMA_ERR_INVALID, "null count");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { *count = 0; return set_error(CHROMA_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *count = n;
    return CHROMA_OK;
}

int chroma_init(int device, chroma_ctx **out)
{
    if (!out) return set_error(CHROMA_ERR_INVALID, "null ctx");
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n == 0)
        return set_error(CHROMA_ERR_NO_DEVICE, "no HIP device available (libchroma_hip needs an MI355X/gfx950 GPU)");
    if (device < 0) device = 0;
    if (device >= n) return set_error(CHROMA_ERR_INVALID, "device %d out of range (%d devices)", device, n);
    HIP_TRY(hipSetDevice(device));
    chroma_ctx *ctx = new chroma_ctx;
    ctx->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&ctx->copy_stream, hipStreamNonBlocking));
    {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        ctx->pool_limit = (size_t)(0.4 * (double)total_b);
        if (const char *e = getenv("CHROMA_POOL_MB")) ctx->pool_limit = (size_t)std::max(0ll, atoll(e)) << 20;
    }
    HIP_TRY(hipMalloc((void **)&ctx->d_counters, sizeof(DeviceCounters)));
    HIP_TRY(hipMemset(ctx->d_counters, 0, sizeof(DeviceCounters)));
    HIP_TRY(hipMalloc((void **)&ctx->d_words, 16 * sizeof(uint32_t)));
    HIP_TRY(hipMemset(ctx->d_words, 0, 16 * sizeof(uint32_t)));
    HIP_TRY(hipHostMalloc((void **)&ctx->h_words, 16 * sizeof(uint32_t), hipHostMallocDefault));
    HIP_TRY(hipMalloc((void **)&ctx->d_step, sizeof(StepState)));
    HIP_TRY(hipMemset(ctx->d_step, 0, sizeof(StepState)));
    HIP_TRY(hipHostMalloc((void **)&ctx->h_step, sizeof(StepState), hipHostMallocDefault));
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        int per_cu = 20;                     // LDS-limited residency of k_raycast_persistent (8 KB per wave)
        if (const char *e = getenv("CHROMA_RAY_WAVES_PER_CU")) per_cu = std::max(1, atoi(e));
        ctx->persistent_waves = prop.multiProcessorCount * per_cu;
        ctx->physics_blocks = prop.multiProcessorCount * 8;          // (for blocks of PHYS_BLOCK threads)
        int wide_per_cu = 14;                // LDS-limited residency of k_raycast_wide
        if (const char *e = getenv("CHROMA_WIDE_WAVES_PER_CU")) wide_per_cu = std::max(1, atoi(e));
        ctx->wide_waves = prop.multiProcessorCount * wide_per_cu;
        int coop_per_cu = 28;                // 71 VGPRs (amdgpu_waves_per_eu 7): 7 waves per SIMD
        if (const char *e = getenv("CHROMA_COOP_WAVES_PER_CU")) coop_per_cu = std::max(1, atoi(e));
        ctx->coop_waves = prop.multiProcessorCount * coop_per_cu;
        int quad_per_cu = 4 * QUAD_WAVES_PER_EU;
        if (const char *e = getenv("CHROMA_QUAD_WAVES_PER_CU")) quad_per_cu = std::max(1, atoi(e));
        ctx->quad_waves = prop.multiProcessorCount * quad_per_cu;
        int pair_per_cu = 4 * PAIR_WAVES_PER_EU;
        if (const char *e = getenv("CHROMA_PAIR_WAVES_PER_CU")) pair_per_cu = std::max(1, atoi(e));
        ctx->pair_waves = prop.multiProcessorCount * pair_per_cu;
        if (const char *e = getenv("CHROMA_WALK"))
            ctx->wide_walk = !strcmp(e, "reference") ? CHROMA_WALK_REFERENCE : !strcmp(e, "wide") ? CHROMA_WALK_WIDE
                           : !strcmp(e, "coop") ? CHROMA_WALK_COOP : !strcmp(e, "pair") ? CHROMA_WALK_PAIR
                           : (!strcmp(e, "literal") || !strcmp(e, "exact")) ? CHROMA_WALK_LITERAL
                           : !strcmp(e, "literal_lane") ? CHROMA_WALK_LITERAL_LANE : CHROMA_WALK_QUAD;
        if (const char *e = getenv("CHROMA_PACKET")) ctx->packet_mode = !strcmp(e, "on") ? 1 : !strcmp(e, "auto") ? 2 : 0;
        if (const char *e = getenv("CHROMA_AUTOSORT")) ctx->autosort_mode = !strcmp(e, "on") || !strcmp(e, "1") ? 1 : !strcmp(e, "off") || !strcmp(e, "0") ? 0 : 2;
        if (const char *e = getenv("CHROMA_RAY_CHUNK")) ctx->ray_chunk = std::max(64, atoi(e));
        if (const char *e = getenv("CHROMA_COOP_CHUNK")) ctx->coop_chunk = std::max(8, atoi(e));
        if (const char *e = getenv("CHROMA_TAIL")) {      // coop (default) | split | fused (the lane-per-photon k_propagate)
            ctx->split_tail = (strcmp(e, "fused") != 0);
            ctx->fused_tail = (strcmp(e, "split") != 0 && strcmp(e, "fused") != 0);
        }
    }
    HIP_TRY(hipEventCreate(&ctx->ev_start));
    HIP_TRY(hipEventCreate(&ctx->ev_stop));
    HIP_TRY(hipEventCreate(&ctx->ev_mid));
    *out = ctx;
    return CHROMA_OK;
}

static void pool_release_all(chroma_ctx *ctx);
int chroma_shutdown(chroma_ctx *ctx)
{
    if (!ctx) return CHROMA_OK;
    hipSetDevice(ctx->device);
    hipStreamSynchronize(ctx->stream);
    hipStreamSynchronize(ctx->copy_stream);
    chroma_comm_destroy(ctx);
    { std::lock_guard<std::mutex> lock(ctx->pool_mu); pool_release_all(ctx); for (hipEvent_t e : ctx->pool_events) hipEventDestroy(e); ctx->pool_events.clear(); }
    for (int i = 0; i < chroma_ctx::STAGE_N; i++) { if (ctx->stage[i]) hipHostFree(ctx->stage[i]); if (ctx->stage_ev[i]) hipEventDestroy(ctx->stage_ev[i]); }
    for (int i = 0; i < chroma_ctx::STAGE_N; i++) { if (ctx->stage_down[i]) hipHostFree(ctx->stage_down[i]); if (ctx->stage_down_ev[i]) hipEventDestroy(ctx->stage_down_ev[i]); }
    hipStreamDestroy(ctx->copy_stream);
    if (ctx->queue_a) hipFree(ctx->queue_a);
    if (ctx->queue_b) hipFree(ctx->queue_b);
    if (ctx->wide_spill) hipFree(ctx->wide_spill);
    if (ctx->coop_spill) hipFree(ctx->coop_spill);
    if (ctx->d_step) hipFree(ctx->d_step);
    if (ctx->h_step) hipHostFree(ctx->h_step);
    for (hipEvent_t e : ctx->step_events) hipEventDestroy(e);
    if (ctx->hit_triangle) hipFree(ctx->hit_triangle);
    if (ctx->hit_distance) hipFree(ctx->hit_distance);
    if (ctx->retry_list) hipFree(ctx->retry_list);
    if (ctx->rays) hipFree(ctx->rays);
    if (ctx->rays_b) hipFree(ctx->rays_b);
    if (ctx->work_a) hipFree(ctx->work_a);
    if (ctx->work_b) hipFree(ctx->work_b);
    hipFree(ctx->d_counters);
    hipFree(ctx->d_words);
    hipHostFree(ctx->h_words);
    hipEventDestroy(ctx->ev_start);
    hipEventDestroy(ctx->ev_stop);
    hipEventDestroy(ctx->ev_mid);
    hipStreamDestroy(ctx->stream);
    delete ctx;
    return CHROMA_OK;
}

int chroma_synchronize(chroma_ctx *ctx)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CHROMA_OK;
}

int chroma_mem_info(chroma_ctx *ctx, size_t *free_bytes, size_t *total_bytes)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    HIP_TRY(hipSetDevice(ctx->device));
    size_t f = 0, t = 0;
    HIP_TRY(hipMemGetInfo(&f, &t));
    if (free_bytes) *free_bytes = f;
    if (total_bytes) *total_bytes = t;
    return CHROMA_OK;
}

int chroma_device_name(chroma_ctx *ctx, char *buf, size_t buflen)
{
    if (!ctx || !buf || !buflen) return set_error(CHROMA_ERR_INVALID, "bad argument");
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, ctx->device));
    snprintf(buf, buflen, "%s (%s, %d CUs)", prop.name, prop.gcnArchName, prop.multiProcessorCount);
    return CHROMA_OK;
}

// ---- device memory: a pool -----------------------------------------------------------------------------------
// Simulation builds a GPUPhotons per event batch: ten arrays allocated, used for one propagate, dropped.  hipMalloc and
// hipFree each cost ~0.1-1 ms for blocks of hundreds of MB and hipFree synchronises the device, so blocks are kept
// instead: chroma_free parks a block (with an event recorded on the context's stream: work already queued on it may
// still use the block), chroma_malloc hands a parked block of exactly the requested size back once that event has
// completed -- no waiting, no new allocation.  Capped at CHROMA_POOL_MB (default: 40 % of the device's memory);
// chroma_pool_trim releases everything parked (also done by itself when hipMalloc runs out of memory).
static size_t pool_round(size_t nbytes) { return (std::max(nbytes, (size_t)4) + 255) & ~(size_t)255; }

static void pool_release_all(chroma_ctx *ctx)       // (pool_mu held)
{
    for (auto &kv : ctx->pool) { hipEventSynchronize(kv.second.ev); hipFree(kv.second.ptr); ctx->pool_events.push_back(kv.second.ev); }
    ctx->pool.clear();
    ctx->pool_bytes = 0;
}

static hipError_t ctx_malloc(chroma_ctx *ctx, void **ptr, size_t bytes)
{
    hipError_t e = hipMalloc(ptr, bytes);
    if (e == hipErrorOutOfMemory) {
        (void)hipGetLastError();
        std::lock_guard<std::mutex> lock(ctx->pool_mu);
        if (!ctx->pool.empty()) { pool_release_all(ctx); e = hipMalloc(ptr, bytes); }
    }
    return e;
}

int chroma_malloc(chroma_ctx *ctx, size_t nbytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return set_error(CHROMA_ERR_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(ctx->device));
    const size_t size = pool_round(nbytes);
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    auto range = ctx->pool.equal_range(size);
    for (auto it = range.first; it != range.second; ++it) {
        // (still in use by queued work: hipErrorNotReady is not an error here, and must not stay behind as the thread's
        //  "last error" for the next hipGetLastError() after a kernel launch to find)
        if (hipEventQuery(it->second.ev) != hipSuccess) { (void)hipGetLastError(); continue; }
        *d_ptr = it->second.ptr;
        ctx->pool_events.push_back(it->second.ev);
        ctx->pool.erase(it);
        ctx->pool_bytes -= size;
        ctx->live[*d_ptr] = size;
        ctx->pool_hits++;
        return CHROMA_OK;
    }
    hipError_t e = hipMalloc(d_ptr, size);
    if (e == hipErrorOutOfMemory && !ctx->pool.empty()) {
        (void)hipGetLastError();
        pool_release_all(ctx);
        e = hipMalloc(d_ptr, size);
    }
    if (e != hipSuccess) return set_error((int)e, "hipMalloc(%zu bytes) failed: %s", size, hipGetErrorString(e));
    ctx->live[*d_ptr] = size;
    ctx->pool_misses++;
    return CHROMA_OK;
}

int chroma_free(chroma_ctx *ctx, void *d_ptr)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (!d_ptr) return CHROMA_OK;
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    auto it = ctx->live.find(d_ptr);
    if (it == ctx->live.end()) {                   // not one of ours (should not happen): the old behaviour
        HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(d_ptr));
        return CHROMA_OK;
    }
    const size_t size = it->second;
    ctx->live.erase(it);
    if (ctx->pool_bytes + size > ctx->pool_limit) {
        HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(d_ptr));
        return CHROMA_OK;
    }
    hipEvent_t ev;
    if (!ctx->pool_events.empty()) { ev = ctx->pool_events.back(); ctx->pool_events.pop_back(); }
    else HIP_TRY(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(ev, ctx->stream));
    ctx->pool.emplace(size, chroma_ctx::PoolBlock{d_ptr, ev});
    ctx->pool_bytes += size;
    return CHROMA_OK;
}

int chroma_pool_trim(chroma_ctx *ctx)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    pool_release_all(ctx);
    return CHROMA_OK;
}

int chroma_pool_stats(chroma_ctx *ctx, uint64_t *parked_bytes, uint64_t *reused, uint64_t *allocated)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    if (parked_bytes) *parked_bytes = ctx->pool_bytes;
    if (reused) *reused = ctx->pool_hits;
    if (allocated) *allocated = ctx->pool_misses;
    return CHROMA_OK;
}

// ---- host -> device ------------------------------------------------------------------------------------------
// A copy from pageable host memory runs at ~11 GB/s through the runtime's own bounce buffer (one thread).  Large copies
// are staged here instead: the host threads copy 64 MB pieces into a ring of PINNED buffers in parallel and each piece
// goes to the device by DMA while the next is being staged.  (r03: 11.2 -> 16 GB/s with 32 MB pieces and 64 threads on
// a 16-core quota; the thread count now follows the quota.)
static int staged_htod(chroma_ctx *ctx, hipStream_t stream, void *d_dst, const void *h_src, size_t nbytes)
{
    std::lock_guard<std::mutex> lock(ctx->stage_mu);
    for (int i = 0; i < chroma_ctx::STAGE_N; i++)
        if (!ctx->stage[i]) {
            HIP_TRY(hipHostMalloc(&ctx->stage[i], chroma_ctx::STAGE_BYTES, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming));
        }
    size_t off = 0;
    int k = 0;
    while (off < nbytes) {
        const size_t len = std::min(chroma_ctx::STAGE_BYTES, nbytes - off);
        HIP_TRY(hipEventSynchronize(ctx->stage_ev[k]));            // (the DMA that last read this buffer is done)
        char *dst = (char *)ctx->stage[k];
        const char *src = (const char *)h_src + off;
        chroma_host::parallel_for(len, [&](size_t a, size_t b) { memcpy(dst + a, src + a, b - a); }, 1u << 20);
        HIP_TRY(hipMemcpyAsync((char *)d_dst + off, dst, len, hipMemcpyHostToDevice, stream));
        HIP_TRY(hipEventRecord(ctx->stage_ev[k], stream));
        off += len;
        k = (k + 1) % chroma_ctx::STAGE_N;
    }
    HIP_TRY(hipStreamSynchronize(stream));
    return CHROMA_OK;
}

int chroma_memcpy_htod(chroma_ctx *ctx, void *d_dst, const void *h_src, size_t nbytes)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (nbytes == 0) return CHROMA_OK;
    if (nbytes >= (8u << 20)) return staged_htod(ctx, ctx->stream, d_dst, h_src, nbytes);
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CHROMA_OK;
}

// The same copy on the context's SECOND stream: not ordered with the work queued on the main stream, so that the
// photons of the next event batch can go up while the current batch propagates (Simulation, one thread ahead).  The
// destination must not be in use by queued work: a block fresh from chroma_malloc never is.  Returns when the data is
// on the device.
int chroma_upload(chroma_ctx *ctx, void *d_dst, const void *h_src, size_t nbytes)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (nbytes == 0) return CHROMA_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    if (nbytes >= (8u << 20)) return staged_htod(ctx, ctx->copy_stream, d_dst, h_src, nbytes);
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, ctx->copy_stream));
    HIP_TRY(hipStreamSynchronize(ctx->copy_stream));
    return CHROMA_OK;
}

// ---- device -> host: the same ring the other way round -- a piece comes down by DMA into a pinned buffer while the host
// threads copy the previous one out to (pageable, possibly never touched) destination memory in parallel
static int staged_dtoh(chroma_ctx *ctx, hipStream_t stream, void *h_dst, const void *d_src, size_t nbytes)
{
    std::lock_guard<std::mutex> lock(ctx->stage_down_mu);
    for (int i = 0; i < chroma_ctx::STAGE_N; i++)
        if (!ctx->stage_down[i]) {
            HIP_TRY(hipHostMalloc(&ctx->stage_down[i], chroma_ctx::STAGE_BYTES, hipHostMallocDefault));
            HIP_TRY(hipEventCreateWithFlags(&ctx->stage_down_ev[i], hipEventDisableTiming));
        }
    const size_t npieces = (nbytes + chroma_ctx::STAGE_BYTES - 1) / chroma_ctx::STAGE_BYTES;
    auto issue = [&](size_t i) -> hipError_t {
        const size_t off = i * chroma_ctx::STAGE_BYTES, len = std::min(chroma_ctx::STAGE_BYTES, nbytes - off);
        const int k = (int)(i % chroma_ctx::STAGE_N);
        hipError_t e = hipMemcpyAsync(ctx->stage_down[k], (const char *)d_src + off, len, hipMemcpyDeviceToHost, stream);
        return e != hipSuccess ? e : hipEventRecord(ctx->stage_down_ev[k], stream);
    };
    HIP_TRY(issue(0));
    for (size_t i = 0; i < npieces; i++) {
        if (i + 1 < npieces) HIP_TRY(issue(i + 1));               // (its buffer was copied out two pieces ago)
        const size_t off = i * chroma_ctx::STAGE_BYTES, len = std::min(chroma_ctx::STAGE_BYTES, nbytes - off);
        const int k = (int)(i % chroma_ctx::STAGE_N);
        HIP_TRY(hipEventSynchronize(ctx->stage_down_ev[k]));
        const char *src = (const char *)ctx->stage_down[k];
        char *dst = (char *)h_dst + off;
        chroma_host::parallel_for(len, [&](size_t a, size_t b) { memcpy(dst + a, src + a, b - a); }, 1u << 20);
    }
    return CHROMA_OK;
}
// for the other translation units of the library (ctx_access.h): a large download on the context's stream
extern "C" int chroma_internal_dtoh(chroma_ctx *ctx, void *h_dst, const void *d_src, size_t nbytes)
{
    if (nbytes == 0) return CHROMA_OK;
    if (nbytes >= (8u << 20)) return staged_dtoh(ctx, ctx->stream, h_dst, d_src, nbytes);
    HIP_TRY(hipMemcpyAsync(h_dst, d_src, nbytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CHROMA_OK;
}
extern "C" int chroma_internal_htod(chroma_ctx *ctx, void *d_dst, const void *h_src, size_t nbytes)
{
    if (nbytes == 0) return CHROMA_OK;
    if (nbytes >= (8u << 20)) return staged_htod(ctx, ctx->stream, d_dst, h_src, nbytes);
    HIP_TRY(hipMemcpyAsync(d_dst, h_src, nbytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    return CHROMA_OK;
}

int chroma_memcpy_dtoh(chroma_ctx *ctx, void *h_dst, const void *d_src, size_t nbytes)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    return chroma_internal_dtoh(ctx, h_dst, d_src, nbytes);
}

int chroma_memcpy_dtod(chroma_ctx *ctx, void *d_dst, const void *d_src, size_t nbytes)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (nbytes == 0) return CHROMA_OK;
    HIP_TRY(hipMemcpyAsync(d_dst, d_src, nbytes, hipMemcpyDeviceToDevice, ctx->stream));
    return CHROMA_OK;
}

int chroma_memset32(chroma_ctx *ctx, void *d_dst, uint32_t value, size_t count)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (count == 0) return CHROMA_OK;
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)d_dst, (int)value, count, ctx->stream));
    return CHROMA_OK;
}

// ---- geometry -------------------------------------------------------------------------------------
int chroma_geometry_create(chroma_ctx *ctx, const chroma_geometry_desc *d, chroma_geometry **out)
{
    if (!ctx || !d || !out) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (!d->vertices || !d->triangles || !d->material_codes || !d->nodes || d->nnodes == 0 || d->ntriangles == 0)
        return set_error(CHROMA_ERR_INVALID, "geometry: missing mesh or BVH arrays");
    if (d->wavelength_n < 2 || d->nmaterials == 0 || d->nmaterials > 127 || d->nsurfaces > 127)
        return set_error(CHROMA_ERR_INVALID, "geometry: bad optics table sizes (8-bit signed material/surface indices)");
    if (!d->mat_refractive_index || !d->mat_absorption_length || !d->mat_scattering_length || !d->mat_num_comp || !d->mat_comp_offset)
        return set_error(CHROMA_ERR_INVALID, "geometry: missing material tables");
    // host-side shape checks the kernels rely on (all cores; the first offender in index order is reported)
    {
        using chroma_host::parallel_for;
        std::atomic<size_t> bad_tri(SIZE_MAX), bad_node(SIZE_MAX), bad_code(SIZE_MAX);
        auto note = [](std::atomic<size_t> &slot, size_t i) { size_t cur = slot.load(); while (i < cur && !slot.compare_exchange_weak(cur, i)) {} };
        parallel_for((size_t)d->ntriangles * 3, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) if (d->triangles[i] >= d->nvertices) { note(bad_tri, i); break; }
        });
        if (bad_tri != SIZE_MAX) { size_t i = bad_tri; return set_error(CHROMA_ERR_INVALID, "triangle %zu references vertex %u >= %u", i / 3, d->triangles[i], d->nvertices); }
        parallel_for((size_t)d->nnodes, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                uint32_t w = d->nodes[4 * i + 3];
                uint32_t nchild = w >> CHROMA_CHILD_BITS, child = w & ~CHROMA_NCHILD_MASK;
                bool bad = nchild == 0 ? child >= d->ntriangles : ((size_t)child + nchild > d->nnodes || child <= i);
                if (bad) { note(bad_node, i); break; }
            }
        });
        if (bad_node != SIZE_MAX) {
            size_t i = bad_node;
            uint32_t w = d->nodes[4 * i + 3], nchild = w >> CHROMA_CHILD_BITS, child = w & ~CHROMA_NCHILD_MASK;
            if (nchild == 0) return set_error(CHROMA_ERR_INVALID, "leaf node %zu references triangle %u >= %u", i, child, d->ntriangles);
            return set_error(CHROMA_ERR_INVALID, "node %zu has a bad child range [%u, %u)", i, child, child + nchild);
        }
        parallel_for((size_t)d->ntriangles, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) {
                uint32_t code = d->material_codes[i];
                int inner = (int8_t)(code >> 24), outer = (int8_t)(code >> 16), surf = (int8_t)(code >> 8);
                bool bad = inner < 0 || outer < 0 || inner >= (int)d->nmaterials || outer >= (int)d->nmaterials || surf < -1 || surf >= (int)d->nsurfaces;
                if (!bad && d->nsolids && d->solid_id_map && d->solid_id_map[i] >= d->nsolids) bad = true;
                if (bad) { note(bad_code, i); break; }
            }
        });
        if (bad_code != SIZE_MAX) {
            size_t i = bad_code;
            uint32_t code = d->material_codes[i];
            int inner = (int8_t)(code >> 24), outer = (int8_t)(code >> 16), surf = (int8_t)(code >> 8);
            if (inner < 0 || outer < 0 || inner >= (int)d->nmaterials || outer >= (int)d->nmaterials || surf < -1 || surf >= (int)d->nsurfaces)
                return set_error(CHROMA_ERR_INVALID, "triangle %zu has material code 0x%08x outside the tables", i, code);
            return set_error(CHROMA_ERR_INVALID, "triangle %zu has solid id %u >= %u", i, d->solid_id_map[i], d->nsolids);
        }
    }
    for (uint32_t m = 0; m < d->nmaterials; m++)
        if (d->mat_num_comp[m] && d->mat_comp_offset[m] + d->mat_num_comp[m] > d->ncomp_total)
            return set_error(CHROMA_ERR_INVALID, "material %u: component rows out of range", m);
    for (uint32_t s = 0; s < d->nsurfaces; s++) {
        if (d->surf_model[s] == CHROMA_SURFACE_DICHROIC) {
            int di = d->surf_dichroic_index ? d->surf_dichroic_index[s] : -1;
            if (di < 0 || di >= (int)d->ndichroic || d->dichroic_nangles[di] < 2 ||
                d->dichroic_offset[di] + d->dichroic_nangles[di] > d->ndichroic_angles_total)
                return set_error(CHROMA_ERR_INVALID, "surface %u: dichroic tables missing or out of range", s);
        }
    }

    const bool timing = getenv("CHROMA_TIMING") != nullptr;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char *what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[chroma_geometry_create] %-28s %.2f s\n", what, std::chrono::duration<double>(now - t_phase).count());
        t_phase = now;
    };
    phase("validation");
    HIP_TRY(hipSetDevice(ctx->device));
    chroma_geometry *g = new chroma_geometry;
    g->ctx = ctx;
    g->nvertices = d->nvertices; g->ntriangles = d->ntriangles; g->nnodes = d->nnodes;
    GeoView &v = g->view;
    memset(&v, 0, sizeof v);
    int rc;
#define UP(field, src, count) if ((rc = upload(g, src, (size_t)(count), &v.field)) != CHROMA_OK) { chroma_geometry_destroy(g); return rc; }
    // nodes as passed in (what GPUGeometry.nodes shows)
    { const uint4 *p; if ((rc = upload(g, (const uint4 *)d->nodes, d->nnodes, &p)) != CHROMA_OK) { chroma_geometry_destroy(g); return rc; } g->d_nodes_api = (void *)p; }
    // derived 8-wide tree, device triangle order and reference test ranks (csrc/wide_build.cpp)
    chroma_host::WideTree wt;
    const bool wide_given = d->wide_nodes != nullptr;
    if (wide_given) {
        if (!d->wide_tri_to_record || !d->wide_record_to_tri || !d->wide_rank || d->nwide == 0 || d->nrecords == 0) {
            chroma_geometry_destroy(g);
            return set_error(CHROMA_ERR_INVALID, "geometry: a supplied wide tree needs its nodes, both record maps and the ranks");
        }
    } else {
        // the default topology is built where the reference builds its tree: on the device (csrc/wide_device.hip);
        // the others, and CHROMA_WIDE_BUILD=host, on the host cores (csrc/wide_build.cpp) -- "levels" gives the same tree either way
        std::string werr;
        const int topology = chroma_host::wide_topology_from_env();
        const char *where = getenv("CHROMA_WIDE_BUILD");
        if (topology == chroma_host::WIDE_TOPOLOGY_LEVELS && !(where && !strcmp(where, "host"))) {
            void *h = nullptr;
            rc = chroma_wide_build_device(ctx, d->nodes, d->nnodes, d->ntriangles, &h, nullptr, nullptr, nullptr);
            if (rc == (int)hipErrorOutOfMemory) {
                // (the builder's scratch -- ~150 bytes per triangle -- did not fit beside what the caller keeps on the card:
                //  give the pool's parked blocks back and try once more; then the host cores build the SAME tree)
                (void)hipGetLastError();
                chroma_pool_trim(ctx);
                rc = chroma_wide_build_device(ctx, d->nodes, d->nnodes, d->ntriangles, &h, nullptr, nullptr, nullptr);
            }
            if (rc == (int)hipErrorOutOfMemory) {
                (void)hipGetLastError();
                fprintf(stderr, "chroma_geometry_create: no room on the device for the tree builder's scratch: building the same tree on the host cores\n");
                if (chroma_host::build_wide_tree(d->nodes, d->nnodes, d->ntriangles, wt, werr, topology) != 0) {
                    chroma_geometry_destroy(g);
                    return set_error(CHROMA_ERR_INVALID, "%s", werr.c_str());
                }
            } else if (rc != CHROMA_OK) { chroma_geometry_destroy(g); return rc; }
            else {
                wt = std::move(*(chroma_host::WideTree *)h);
                delete (chroma_host::WideTree *)h;
            }
        } else if (chroma_host::build_wide_tree(d->nodes, d->nnodes, d->ntriangles, wt, werr, topology) != 0) {
            chroma_geometry_destroy(g);
            return set_error(CHROMA_ERR_INVALID, "%s", werr.c_str());
        }
    }
    phase(wide_given ? "nodes upload" : "nodes upload + wide tree");
    const uint32_t *wide_nodes = wide_given ? d->wide_nodes : wt.wnodes.data();
    const uint32_t *tri_to_dev = wide_given ? d->wide_tri_to_record : wt.tri_to_dev.data();
    const uint32_t *dev_to_tri = wide_given ? d->wide_record_to_tri : wt.dev_to_tri.data();
    const uint32_t *tri_rank = wide_given ? d->wide_rank : wt.rank.data();
    const size_t nwide = wide_given ? (size_t)d->nwide : wt.nwide;
    const size_t nrecords = wide_given ? (size_t)d->nrecords : wt.dev_to_tri.size();
    {   // the walks index the wide nodes and the records with what this tree holds: check it before any upload
        std::string werr;
        if (chroma_host::validate_wide_tree(wide_nodes, nwide, tri_to_dev, d->ntriangles, dev_to_tri, nrecords, werr) != 0) {
            chroma_geometry_destroy(g);
            return set_error(CHROMA_ERR_INVALID, "%s", werr.c_str());
        }
    }
    phase("wide tree index checks");
    { const uint4 *p; if ((rc = upload(g, (const uint4 *)wide_nodes, nwide * 8, &p)) != CHROMA_OK) { chroma_geometry_destroy(g); return rc; } v.wnodes = p; }
    v.nwide = (uint32_t)nwide;
    g->nwide = nwide; g->wide_depth = wt.depth; g->nrecords = nrecords;
    {
        const uint4 *dw = v.wnodes;
        const uint32_t nw = (uint32_t)nwide;
        hipStream_t st = ctx->stream;
        if ((rc = stack_need_fixed_point(ctx, nwide, [&](uint32_t *need, uint32_t *changed) {
                 hipLaunchKernelGGL(k_stack_need_wide, dim3((nw + 255) / 256), dim3(256), 0, st, dw, nw, need, changed); }, &g->wide_stack_need)) != CHROMA_OK) {
            chroma_geometry_destroy(g);
            return rc;
        }
    }
    { chroma_host::WordBuffer().swap(wt.wnodes); }
    UP(tri_to_dev, tri_to_dev, d->ntriangles);
    UP(dev_to_tri, dev_to_tri, nrecords);
    // traversal copy of the nodes: leaf child -> device triangle index (a pass over the array uploaded above)
    {
        void *dn = nullptr;
        size_t bytes = (size_t)d->nnodes * 16;
        hipError_t e = hipMalloc(&dn, bytes);
        if (e != hipSuccess) { chroma_geometry_destroy(g); return set_error((int)e, "hipMalloc(%zu) for nodes: %s", bytes, hipGetErrorString(e)); }
        g->allocations.push_back(dn);
        g->device_bytes += bytes;
        hipLaunchKernelGGL(k_traversal_nodes, dim3((unsigned)((d->nnodes + 255) / 256)), dim3(256), 0, ctx->stream, (const uint4 *)g->d_nodes_api, (uint32_t)d->nnodes,
                           v.tri_to_dev, d->ntriangles, (uint4 *)dn);
        v.nodes = (const uint4 *)dn;
    }
    phase("wide nodes + traversal copy");
    // API-visible copies of the mesh arrays (GPUGeometry.vertices/.triangles/.material_codes/.colors)
    { const float *p; if ((rc = upload(g, d->vertices, (size_t)d->nvertices * 3, &p)) != CHROMA_OK) { chroma_geometry_destroy(g); return rc; } g->d_vertices = (void *)p; }
    { const uint32_t *p; if ((rc = upload(g, d->triangles, (size_t)d->ntriangles * 3, &p)) != CHROMA_OK) { chroma_geometry_destroy(g); return rc; } g->d_triangles = (void *)p; }
    { const uint32_t *p; if ((rc = upload(g, d->material_codes, d->ntriangles, &p)) != CHROMA_OK) { chroma_geometry_destroy(g); return rc; } g->d_material_codes = (void *)p; }
    // 48-byte triangle records in device order: gathered on the device from those arrays (+ the ranks, uploaded for this only)
    {
        void *dtri = nullptr;
        size_t bytes = nrecords * (16 * TRI_STRIDE);
        hipError_t e = hipMalloc(&dtri, bytes);
        if (e != hipSuccess) { chroma_geometry_destroy(g); return set_error((int)e, "hipMalloc(%zu) for triangle records: %s", bytes, hipGetErrorString(e)); }
        g->allocations.push_back(dtri);
        g->device_bytes += bytes;
        uint32_t *d_rank = nullptr;
        e = hipMalloc((void **)&d_rank, std::max<size_t>(d->ntriangles, 1) * 4);
        if (e != hipSuccess) { chroma_geometry_destroy(g); return set_error((int)e, "hipMalloc for triangle ranks: %s", hipGetErrorString(e)); }
        rc = chroma_internal_htod(ctx, d_rank, tri_rank, (size_t)d->ntriangles * 4);
        if (rc == CHROMA_OK) {
            hipLaunchKernelGGL(k_triangle_records, dim3((unsigned)((nrecords + 255) / 256)), dim3(256), 0, ctx->stream, (const float *)g->d_vertices, (const uint32_t *)g->d_triangles,
                               (const uint32_t *)g->d_material_codes, d_rank, v.dev_to_tri, (uint32_t)nrecords, (float4 *)dtri);
            e = hipStreamSynchronize(ctx->stream);
            if (e != hipSuccess) rc = set_error((int)e, "triangle records: %s", hipGetErrorString(e));
        }
        hipFree(d_rank);
        if (rc != CHROMA_OK) { chroma_geometry_destroy(g); return rc; }
        v.tri = (const float4 *)dtri;
    }
    phase("triangle records");
    { const uint32_t *p; if ((rc = upload(g, d->colors, d->colors ? d->ntriangles : 0, &p)) != CHROMA_OK) { chroma_geometry_destroy(g); return rc; } g->d_colors = (void *)p; }
    UP(solid_id_map, d->solid_id_map, d->solid_id_map ? d->ntriangles : 0);
    size_t wn = d->wavelength_n;
    UP(mat_refractive_index, d->mat_refractive_index, d->nmaterials * wn);
    UP(mat_absorption_length, d->mat_absorption_length, d->nmaterials * wn);
    UP(mat_scattering_length, d->mat_scattering_length, d->nmaterials * wn);
    UP(mat_num_comp, d->mat_num_comp, d->nmaterials);
    UP(mat_comp_offset, d->mat_comp_offset, d->nmaterials);
    UP(comp_reemission_prob, d->comp_reemission_prob, d->ncomp_total * wn);
    UP(comp_reemission_wvl_cdf, d->comp_reemission_wvl_cdf, d->ncomp_total * wn);
    UP(comp_absorption_length, d->comp_absorption_length, d->ncomp_total * wn);
    UP(comp_reemission_time_cdf, d->comp_reemission_time_cdf, (size_t)d->ncomp_total * d->time_n);
    UP(surf_detect, d->surf_detect, d->nsurfaces * wn);
    UP(surf_absorb, d->surf_absorb, d->nsurfaces * wn);
    UP(surf_reemit, d->surf_reemit, d->nsurfaces * wn);
    UP(surf_reflect_diffuse, d->surf_reflect_diffuse, d->nsurfaces * wn);
    UP(surf_reflect_specular, d->surf_reflect_specular, d->nsurfaces * wn);
    UP(surf_eta, d->surf_eta, d->nsurfaces * wn);
    UP(surf_k, d->surf_k, d->nsurfaces * wn);
    UP(surf_reemission_cdf, d->surf_reemission_cdf, d->nsurfaces * wn);
    {
        std::vector<SurfaceInfo> info(std::max<uint32_t>(d->nsurfaces, 1));
        for (uint32_t s = 0; s < d->nsurfaces; s++)
            info[s] = SurfaceInfo{d->surf_model[s], d->surf_transmissive[s], d->surf_thickness[s],
                                  d->surf_dichroic_index ? d->surf_dichroic_index[s] : -1};
        UP(surf_info, info.data(), info.size());
    }
    UP(dichroic_nangles, d->dichroic_nangles, d->ndichroic);
    UP(dichroic_offset, d->dichroic_offset, d->ndichroic);
    UP(dichroic_angles, d->dichroic_angles, d->ndichroic_angles_total);
    UP(dichroic_reflect, d->dichroic_reflect, d->ndichroic_angles_total * wn);
    UP(dichroic_transmit, d->dichroic_transmit, d->ndichroic_angles_total * wn);
    UP(solid_id_to_channel_index, d->solid_id_to_channel_index, d->nsolids);
#undef UP
    memcpy(v.world_origin, d->world_origin, sizeof v.world_origin);
    v.world_scale = d->world_scale;
    {   // ~16 ulp of the largest world coordinate (record_hit_is_regular)
        float maxabs = 0.0f;
        for (int a = 0; a < 3; a++)
            maxabs = std::max(maxabs, std::max(fabsf(d->world_origin[a]), fabsf(d->world_origin[a] + 65535.0f * d->world_scale)));
        v.suspect_margin = 2e-6f * maxabs;
        // growth of the boxes in the fast slab test (ray_growth, propagate_device.h): four times the bound on what the fused
        // evaluation can differ from the reference's, at least a quarter of a quantum, at most the whole quantum of rounds 1-2
        // (CHROMA_SLAB_GROW overrides: A/B runs)
        const double bound = ldexp(1.0, -24) * (10.0 * 65534.0 + 2.0 * (double)maxabs / std::max((double)d->world_scale, 1e-30));
        v.slab_grow = (float)std::min(1.0, std::max(0.25, 4.0 * bound));
        if (const char *e = getenv("CHROMA_SLAB_GROW")) v.slab_grow = (float)std::min(1.0, std::max(0.0625, atof(e)));
    }
    v.wavelength_n = d->wavelength_n; v.wavelength_start = d->wavelength_start; v.wavelength_step = d->wavelength_step;
    v.time_n = d->time_n; v.time_start = d->time_start; v.time_step = d->time_step;
    v.nnodes = d->nnodes; v.ntriangles = d->ntriangles; v.nsolids = d->nsolids; v.nchannels = d->nchannels;
    v.plain_optics = 1u;
    for (uint32_t m = 0; m < d->nmaterials; m++) if (d->mat_num_comp[m]) v.plain_optics = 0u;
    for (uint32_t k = 0; k < d->nsurfaces; k++) if (d->surf_model[k] != CHROMA_SURFACE_DEFAULT) v.plain_optics = 0u;
    if (getenv("CHROMA_FULL_PHYSICS")) v.plain_optics = 0u;          // (A/B: the all-models kernel on a plain geometry)

    phase("mesh arrays + tables");
    {
        const uint4 *dn = (const uint4 *)g->d_nodes_api;
        const uint32_t nn = (uint32_t)d->nnodes;
        hipStream_t st = ctx->stream;
        uint32_t need = 0;
        if ((rc = stack_need_fixed_point(ctx, d->nnodes, [&](uint32_t *nd, uint32_t *changed) {
                 hipLaunchKernelGGL(k_stack_need_ref, dim3((nn + 255) / 256), dim3(256), 0, st, dn, nn, nd, changed); }, &need)) != CHROMA_OK) {
            chroma_geometry_destroy(g);
            return rc;
        }
        g->stack_need = std::max<uint32_t>(1, need);
    }
    phase("stack need");
    if (g->stack_need > STACK_LDS + STACK_SCRATCH) {
        uint32_t need = g->stack_need;
        chroma_geometry_destroy(g);
        return set_error(CHROMA_ERR_STACK, "BVH needs %u traversal stack entries, more than the %d supported", need, STACK_LDS + STACK_SCRATCH);
    }
    *out = g;
    return CHROMA_OK;
}

int chroma_geometry_destroy(chroma_geometry *g)
{
    if (!g) return CHROMA_OK;
    hipSetDevice(g->ctx->device);
    hipStreamSynchronize(g->ctx->stream);
    for (void *p : g->allocations) hipFree(p);
    delete g;
    return CHROMA_OK;
}

int chroma_geometry_device_ptr(chroma_geometry *g, const char *name, void **d_ptr, size_t *nbytes)
{
    if (!g || !name || !d_ptr) return set_error(CHROMA_ERR_INVALID, "bad argument");
    std::string n(name);
    size_t bytes = 0; void *p = nullptr;
    if (n == "nodes") { p = g->d_nodes_api; bytes = g->nnodes * 16; }
    else if (n == "vertices") { p = g->d_vertices; bytes = g->nvertices * 12; }
    else if (n == "triangles") { p = g->d_triangles; bytes = g->ntriangles * 12; }
    else if (n == "material_codes") { p = g->d_material_codes; bytes = g->ntriangles * 4; }
    else if (n == "colors") { p = g->d_colors; bytes = g->ntriangles * 4; }
    else if (n == "solid_id_map") { p = (void *)g->view.solid_id_map; bytes = g->ntriangles * 4; }
    else if (n == "solid_id_to_channel_index") { p = (void *)g->view.solid_id_to_channel_index; bytes = (size_t)g->view.nsolids * 4; }
    else if (n == "triangle_records") { p = (void *)g->view.tri; bytes = g->nrecords * (16 * TRI_STRIDE); }
    else if (n == "wide_nodes") { p = (void *)g->view.wnodes; bytes = g->nwide * 128; }
    else if (n == "tri_to_dev") { p = (void *)g->view.tri_to_dev; bytes = g->ntriangles * 4; }
    else if (n == "dev_to_tri") { p = (void *)g->view.dev_to_tri; bytes = g->nrecords * 4; }
    else return set_error(CHROMA_ERR_INVALID, "unknown geometry array '%s'", name);
    *d_ptr = p;
    if (nbytes) *nbytes = bytes;
    return CHROMA_OK;
}

int chroma_geometry_stack_need(chroma_geometry *g, uint32_t *entries)
{
    if (!g || !entries) return set_error(CHROMA_ERR_INVALID, "bad argument");
    *entries = g->stack_need;
    return CHROMA_OK;
}

// ---- kernel-level entry points ------------------------------------------------------------------------
int chroma_propagate_step(chroma_ctx *ctx, chroma_geometry *geom, int32_t first_photon, int32_t nthreads,
                          const uint32_t *d_input_queue, uint32_t *d_output_queue, chroma_rng rng,
                          const chroma_photon_arrays *photons, int32_t max_steps, int32_t use_weights,
                          int32_t scatter_first)
{
    if (!ctx || !geom) return set_error(CHROMA_ERR_INVALID, "bad argument");
    int rc = check_photons(photons, true);
    if (rc) return rc;
    if (first_photon < 0 || nthreads < 0) return set_error(CHROMA_ERR_INVALID, "negative photon range");
    return launch_propagate(ctx, geom, to_view(photons), first_photon, nthreads, d_input_queue, d_output_queue, rng,
                            max_steps, use_weights, scatter_first);
}

int chroma_photon_duplicate(chroma_ctx *ctx, int32_t first_photon, int32_t nthreads,
                            const chroma_photon_arrays *photons, int32_t copies, int32_t stride)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    int rc = check_photons(photons, false);
    if (rc) return rc;
    if (nthreads <= 0 || copies <= 0) return CHROMA_OK;
    hipLaunchKernelGGL(k_photon_duplicate, dim3((nthreads + 255) / 256), dim3(256), 0, ctx->stream, to_view(photons),
                       first_photon, nthreads, copies, stride);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

static int read_word(chroma_ctx *ctx, int slot, uint32_t *out)
{
    HIP_TRY(hipMemcpyAsync(ctx->h_words + slot, ctx->d_words + slot, 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    *out = ctx->h_words[slot];
    return CHROMA_OK;
}

int chroma_count_photons(chroma_ctx *ctx, int32_t first_photon, int32_t nthreads, uint32_t target_flag,
                         const uint32_t *d_flags, uint32_t *count)
{
    if (!ctx || !d_flags || !count) return set_error(CHROMA_ERR_INVALID, "bad argument");
    HIP_TRY(hipMemsetAsync(ctx->d_words, 0, 4, ctx->stream));
    if (nthreads > 0) {
        hipLaunchKernelGGL(k_count_photons, dim3((unsigned)std::min((nthreads + 255) / 256, 4096)), dim3(256), 0, ctx->stream, d_flags, first_photon,
                           nthreads, target_flag, ctx->d_words);
        HIP_TRY(hipGetLastError());
    }
    return read_word(ctx, 0, count);
}

int chroma_copy_photons(chroma_ctx *ctx, int32_t first_photon, int32_t nthreads, uint32_t target_flag,
                        const chroma_photon_arrays *src, const chroma_photon_arrays *dst, uint32_t *ncopied)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    int rc = check_photons(src, false); if (rc) return rc;
    rc = check_photons(dst, false); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ctx->d_words, 0, 4, ctx->stream));
    if (nthreads > 0) {
        hipLaunchKernelGGL(k_copy_photons, dim3((unsigned)(((long long)nthreads + 16 * 256 - 1) / (16 * 256))), dim3(256), 0, ctx->stream, to_view(src), to_view(dst),
                           first_photon, nthreads, target_flag, ctx->d_words);
        HIP_TRY(hipGetLastError());
    }
    uint32_t n = 0;
    rc = read_word(ctx, 0, &n);
    if (ncopied) *ncopied = n;
    return rc;
}

int chroma_copy_photon_queue(chroma_ctx *ctx, int32_t first_photon, int32_t nthreads, const uint32_t *d_queue,
                             const chroma_photon_arrays *src, const chroma_photon_arrays *dst)
{
    if (!ctx || !d_queue) return set_error(CHROMA_ERR_INVALID, "bad argument");
    int rc = check_photons(src, false); if (rc) return rc;
    rc = check_photons(dst, false); if (rc) return rc;
    if (nthreads <= 0) return CHROMA_OK;
    hipLaunchKernelGGL(k_copy_photon_queue, dim3((nthreads + 255) / 256), dim3(256), 0, ctx->stream, to_view(src), to_view(dst),
                       first_photon, nthreads, d_queue);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

int chroma_count_photon_hits(chroma_ctx *ctx, chroma_geometry *geom, int32_t first_photon, int32_t nphotons,
                             uint32_t detection_state, const chroma_photon_arrays *photons, uint32_t *count)
{
    if (!ctx || !geom || !count) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (!geom->view.nsolids) return set_error(CHROMA_ERR_INVALID, "geometry has no detector channel map");
    int rc = check_photons(photons, false); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ctx->d_words, 0, 4, ctx->stream));
    if (nphotons > 0) {
        hipLaunchKernelGGL(k_count_hits, dim3((unsigned)std::min((nphotons + 255) / 256, 4096)), dim3(256), 0, ctx->stream, geom->view, photons->flags,
                           photons->last_hit_triangles, first_photon, nphotons, detection_state, ctx->d_words);
        HIP_TRY(hipGetLastError());
    }
    return read_word(ctx, 0, count);
}

int chroma_copy_photon_hits(chroma_ctx *ctx, chroma_geometry *geom, int32_t first_photon, int32_t nphotons,
                            uint32_t detection_state, const chroma_photon_arrays *src, const chroma_photon_arrays *dst,
                            int32_t *d_channels, uint32_t *ncopied)
{
    if (!ctx || !geom || !d_channels) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (!geom->view.nsolids) return set_error(CHROMA_ERR_INVALID, "geometry has no detector channel map");
    int rc = check_photons(src, false); if (rc) return rc;
    rc = check_photons(dst, false); if (rc) return rc;
    HIP_TRY(hipMemsetAsync(ctx->d_words, 0, 4, ctx->stream));
    if (nphotons > 0) {
        hipLaunchKernelGGL(k_copy_hits, dim3((unsigned)(((long long)nphotons + COPY_ITEMS * 256 - 1) / (COPY_ITEMS * 256))), dim3(256), 0, ctx->stream, geom->view, to_view(src),
                           to_view(dst), d_channels, first_photon, nphotons, detection_state, ctx->d_words);
        HIP_TRY(hipGetLastError());
    }
    uint32_t n = 0;
    rc = read_word(ctx, 0, &n);
    if (ncopied) *ncopied = n;
    return rc;
}

static int ensure_queues(chroma_ctx *ctx, size_t n);
static int distance_to_mesh_fast(chroma_ctx *ctx, chroma_geometry *geom, int32_t n, const float *d_origin,
                                 const float *d_direction, const int32_t *d_last_hit, float *d_distance, int32_t *d_triangle);

int chroma_distance_to_mesh(chroma_ctx *ctx, chroma_geometry *geom, int32_t nthreads, const float *d_origin,
                            const float *d_direction, float *d_distance, int32_t *d_triangle)
{
    return chroma_intersect_mesh(ctx, geom, nthreads, d_origin, d_direction, nullptr, d_distance, d_triangle);
}

int chroma_intersect_mesh(chroma_ctx *ctx, chroma_geometry *geom, int32_t nthreads, const float *d_origin,
                          const float *d_direction, const int32_t *d_last_hit, float *d_distance, int32_t *d_triangle)
{
    if (!ctx || !geom || !d_origin || !d_direction || !d_distance) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (nthreads <= 0) return CHROMA_OK;
    uint32_t need = geom->stack_need;
    if (need > STACK_LDS + STACK_SCRATCH)
        return set_error(CHROMA_ERR_STACK, "BVH needs %u traversal stack entries, more than the %d supported", need, STACK_LDS + STACK_SCRATCH);
    if (geom->view.wnodes && geom->wide_stack_need <= COOP_STACK + COOP_SPILL && ctx->wide_walk != CHROMA_WALK_REFERENCE &&
        ctx->wide_walk != CHROMA_WALK_LITERAL && ctx->wide_walk != CHROMA_WALK_LITERAL_LANE)
        return distance_to_mesh_fast(ctx, geom, nthreads, d_origin, d_direction, d_last_hit, d_distance, d_triangle);
    dim3 grid((unsigned)((nthreads + PROP_BLOCK - 1) / PROP_BLOCK)), block(PROP_BLOCK);
#define LAUNCH(N, C) hipLaunchKernelGGL((k_distance_to_mesh<N, C>), grid, block, 0, ctx->stream, geom->view, nthreads, \
                                        d_origin, d_direction, d_last_hit, d_distance, d_triangle, ctx->d_counters)
    if (ctx->counting) LAUNCH(STACK_LDS, true); else LAUNCH(STACK_LDS, false);
#undef LAUNCH
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

static int distance_to_mesh_fast(chroma_ctx *ctx, chroma_geometry *geom, int32_t n, const float *d_origin,
                                 const float *d_direction, const int32_t *d_last_hit, float *d_distance, int32_t *d_triangle)
{
    HIP_TRY(hipSetDevice(ctx->device));
    int rc = ensure_queues(ctx, (size_t)n); if (rc) return rc;
    if (!ctx->coop_spill)
        HIP_TRY(ctx_malloc(ctx, (void **)&ctx->coop_spill, spill_entries(ctx) * sizeof(uint2)));
    StepState *st = ctx->d_step;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_step_set, dim3(1), dim3(1), 0, ctx->stream, st, (uint32_t)n);
    hipLaunchKernelGGL(k_rays_from_arrays, dim3(blocks), dim3(256), 0, ctx->stream, geom->view, (int)n, d_origin, d_direction,
                       d_last_hit, ctx->rays, ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, st);
    const unsigned waves = (unsigned)std::min<long long>(((long long)n + 15) / 16, (long long)ctx->quad_waves);
    if (ctx->counting)
        hipLaunchKernelGGL((k_raycast_quad<true>), dim3(waves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, 0, st,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk, 0);
    else
        hipLaunchKernelGGL((k_raycast_quad<false>), dim3(waves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, 0, st,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk, 0);
    hipLaunchKernelGGL(k_distance_finish, dim3(blocks), dim3(256), 0, ctx->stream, geom->view, (int)n, ctx->rays, ctx->hit_triangle,
                       ctx->hit_distance, d_distance, d_triangle, ctx->retry_list, st);
    if (ctx->counting)
        hipLaunchKernelGGL((k_distance_retry<true>), dim3(256), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                           ctx->retry_list, d_distance, d_triangle, ctx->d_counters);
    else
        hipLaunchKernelGGL((k_distance_retry<false>), dim3(256), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                           ctx->retry_list, d_distance, d_triangle, ctx->d_counters);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

// ---- fused host loops -----------------------------------------------------------------------------------
static int ensure_queues(chroma_ctx *ctx, size_t n)
{
    if (ctx->queue_capacity >= n + 1) return CHROMA_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->queue_a) hipFree(ctx->queue_a);
    if (ctx->queue_b) hipFree(ctx->queue_b);
    if (ctx->hit_triangle) hipFree(ctx->hit_triangle);
    if (ctx->hit_distance) hipFree(ctx->hit_distance);
    if (ctx->retry_list) hipFree(ctx->retry_list);
    if (ctx->rays) hipFree(ctx->rays);
    if (ctx->rays_b) hipFree(ctx->rays_b);
    if (ctx->work_a) hipFree(ctx->work_a);
    if (ctx->work_b) hipFree(ctx->work_b);
    ctx->work_a = ctx->work_b = nullptr;
    ctx->queue_a = ctx->queue_b = nullptr;
    ctx->hit_triangle = nullptr; ctx->hit_distance = nullptr; ctx->retry_list = nullptr;
    ctx->rays = nullptr;
    ctx->rays_b = nullptr;
    ctx->queue_capacity = 0;
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->queue_a, (n + 1) * sizeof(uint32_t)));
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->queue_b, (n + 1) * sizeof(uint32_t)));
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->hit_triangle, (n + 1) * sizeof(int32_t)));
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->hit_distance, (n + 1) * sizeof(float)));
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->retry_list, (n + 1) * sizeof(uint32_t)));
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->rays, (n + 1) * 4 * sizeof(float4)));
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->rays_b, (n + 1) * 4 * sizeof(float4)));
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->work_a, (n + 1) * 4 * sizeof(float4)));
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->work_b, (n + 1) * 4 * sizeof(float4)));
    ctx->queue_capacity = n + 1;
    return CHROMA_OK;
}

int chroma_propagate_stats_read(chroma_ctx *ctx, chroma_propagate_stats *stats)
{
    if (!ctx || !stats) return set_error(CHROMA_ERR_INVALID, "bad argument");
    DeviceCounters c;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(hipMemcpy(&c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(ctx->d_counters, 0, sizeof c));
    stats->photon_steps += c.photon_steps;
    stats->nodes_visited += c.nodes_visited;
    stats->triangles_tested += c.triangles_tested;
    stats->stack_overflows += c.stack_overflows;
    stats->stack_spills += c.stack_spills;
    stats->packet_rays += c.packet_rays;
    stats->packet_nodes_visited += c.packet_nodes;
    stats->packet_triangles_tested += c.packet_tris;
    return CHROMA_OK;
}

int chroma_set_counting(chroma_ctx *ctx, int32_t enabled)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    ctx->counting = enabled ? 1 : 0;
    return CHROMA_OK;
}

int chroma_set_walk(chroma_ctx *ctx, int32_t mode)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (mode != CHROMA_WALK_REFERENCE && mode != CHROMA_WALK_WIDE && mode != CHROMA_WALK_COOP && mode != CHROMA_WALK_QUAD &&
        mode != CHROMA_WALK_PAIR && mode != CHROMA_WALK_LITERAL && mode != CHROMA_WALK_LITERAL_LANE)
        return set_error(CHROMA_ERR_INVALID, "unknown walk mode %d", mode);
    ctx->wide_walk = mode;
    return CHROMA_OK;
}

int chroma_set_packet(chroma_ctx *ctx, int32_t mode)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (mode < 0 || mode > 2) return set_error(CHROMA_ERR_INVALID, "unknown packet mode %d", mode);
    ctx->packet_mode = mode;
    return CHROMA_OK;
}

int chroma_set_autosort(chroma_ctx *ctx, int32_t mode)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (mode < 0 || mode > 2) return set_error(CHROMA_ERR_INVALID, "unknown autosort mode %d", mode);
    ctx->autosort_mode = mode;
    return CHROMA_OK;
}

int chroma_set_tail(chroma_ctx *ctx, int32_t mode)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (mode != CHROMA_TAIL_COOP && mode != CHROMA_TAIL_SPLIT && mode != CHROMA_TAIL_FUSED)
        return set_error(CHROMA_ERR_INVALID, "unknown tail mode %d", mode);
    ctx->split_tail = mode != CHROMA_TAIL_FUSED;
    ctx->fused_tail = mode == CHROMA_TAIL_COOP;
    return CHROMA_OK;
}

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
static int propagate_order(chroma_ctx *ctx, const PhotonView &pv, uint64_t nphotons, uint32_t ncopies, uint32_t **d_order)
{
    *d_order = nullptr;
    const int mode = ctx->autosort_mode;
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

// the photons' final records (chroma_propagate_hits): 64 bytes per photon of the largest batch seen, zeroed once -- a record
// belongs to a call when it carries that call's epoch, and epochs start at 1
static int ensure_final_records(chroma_ctx *ctx, size_t n)
{
    if (ctx->final_capacity >= n) return CHROMA_OK;
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (ctx->final_rec) hipFree(ctx->final_rec);
    ctx->final_rec = nullptr; ctx->final_capacity = 0;
    HIP_TRY(ctx_malloc(ctx, (void **)&ctx->final_rec, n * 4 * sizeof(float4)));
    HIP_TRY(hipMemsetAsync(ctx->final_rec, 0, n * 4 * sizeof(float4), ctx->stream));
    ctx->final_capacity = n;
    ctx->final_epoch = 0;
    return CHROMA_OK;
}

static int propagate_impl(chroma_ctx *ctx, chroma_geometry *geom, const chroma_photon_arrays *photons, uint64_t nphotons,
                          uint32_t ncopies, chroma_rng rng, int32_t max_steps, int32_t use_weights, int32_t scatter_first,
                          int32_t time_kernels, chroma_propagate_stats *stats, int32_t *aborted, chroma_hits_request *hr)
{
    if (!ctx || !geom) return set_error(CHROMA_ERR_INVALID, "bad argument");
    int rc = check_photons(photons, true); if (rc) return rc;
    if (hr) {
        hr->nhits = 0;
        if (!geom->view.nsolids) return set_error(CHROMA_ERR_INVALID, "geometry has no detector channel map");
        if (hr->dst) { rc = check_photons(hr->dst, false); if (rc) return rc; }
        if ((hr->dst != nullptr) != (hr->d_channels != nullptr)) return set_error(CHROMA_ERR_INVALID, "flat hits need both dst and d_channels");
    }
    if (nphotons >= 0x7fffffffull) return set_error(CHROMA_ERR_INVALID, "at most 2^31-2 photons per call");
    if (ncopies == 0 || nphotons % ncopies) return set_error(CHROMA_ERR_INVALID, "nphotons must be a multiple of ncopies");
    if (aborted) *aborted = 0;
    if (nphotons == 0 || max_steps <= 0) return CHROMA_OK;
    HIP_TRY(hipSetDevice(ctx->device));
    rc = ensure_queues(ctx, nphotons); if (rc) return rc;
    PhotonView pv = to_view(photons);
    uint32_t *in_q = ctx->queue_a, *out_q = ctx->queue_b;
    float4 *work_in = ctx->work_a, *work_out = ctx->work_b;
    // (final records: with a hit request, or for every call under CHROMA_FINAL_RECORDS=1 -- an A/B switch)
    static const bool records_always = getenv("CHROMA_FINAL_RECORDS") && atoi(getenv("CHROMA_FINAL_RECORDS")) != 0;
    const bool use_records = (hr != nullptr || records_always) && ctx->split_tail != 0;
    ctx->final_use = nullptr;
    if (use_records) {
        rc = ensure_final_records(ctx, nphotons); if (rc) return rc;
        ctx->final_epoch++;
        if (ctx->final_epoch == 0u) {            // (wrapped: no stale record may look current)
            HIP_TRY(hipMemsetAsync(ctx->final_rec, 0, ctx->final_capacity * 4 * sizeof(float4), ctx->stream));
            ctx->final_epoch = 1u;
        }
        ctx->final_use = ctx->final_rec;
    }
    struct FinalGuard { chroma_ctx *c; ~FinalGuard() { c->final_use = nullptr; } } final_guard{ctx};

    double kernel_ms = 0.0, raycast_ms = 0.0, physics_ms = 0.0, packet_ms = 0.0;
    uint64_t launches = 0, raycast_launches = 0, physics_launches = 0, packet_launches = 0;
    bool packet_offered = false;
    uint64_t reordered = 0;
    // Launch policy of the reference (chroma/gpu/photon.py:225-252): one step per launch while many
    // photons are alive, and ONE launch for all remaining steps once fewer than 64*16*8 are left (or
    // with weights).  A launch re-normalises dir/pol when it loads a photon (propagate.cu:248,250), so
    // the policy is part of the arithmetic.  Here every step is a ray cast + physics pair that gets the
    // whole chip; a step that the reference would run inside its last launch skips the re-normalisation
    // instead (same numbers; with weights that is every step but the first).  The policy is evaluated ON
    // THE DEVICE (k_step_begin), so the steps are enqueued back to back; the host looks at the survivor
    // count only now and then, to stop early, to shrink the grids and to hand the last photons to the
    // fused tail kernel.  The live photons travel in the dense working set (k_load_working).
    const bool device_steps = ctx->split_tail != 0;
    if (device_steps) {
        HIP_TRY(hipMemsetAsync(ctx->d_step, 0, sizeof(StepState), ctx->stream));
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, ctx->stream, in_q, 1u);
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, ctx->stream, out_q, 1u);
        {
            unsigned blocks = (unsigned)std::min<uint64_t>((nphotons + PHYS_BLOCK - 1) / PHYS_BLOCK, (uint64_t)ctx->physics_blocks);
            const bool probe = step_uses_quad_walk(ctx, geom) && ctx->packet_mode != 0 && geom->wide_stack_need <= PACKET_STACK;
            packet_offered = probe;
            HIP_TRY(hipMemsetAsync(ctx->d_words + 4, 0, 12, ctx->stream));          // [4] use_packet, [5] coherent waves, [6] waves
            uint32_t *d_order = nullptr;
            if (step_uses_quad_walk(ctx, geom)) { rc = propagate_order(ctx, pv, nphotons, ncopies, &d_order); if (rc) return rc; }
            hipLaunchKernelGGL(k_load_working, dim3(blocks), dim3(PHYS_BLOCK), 0, ctx->stream, geom->view, pv, in_q, work_in,
                               (uint64_t)nphotons, ncopies, (uint32_t)(nphotons / ncopies),
                               step_uses_quad_walk(ctx, geom) ? ctx->rays : nullptr, (probe && ctx->packet_mode == 2) ? ctx->d_words + 5 : nullptr,
                               (const uint32_t *)d_order);
            if (d_order) { chroma_free(ctx, d_order); reordered = nphotons; }      // (parked until the stream has passed this point)
            if (probe)
                hipLaunchKernelGGL(k_packet_decide, dim3(1), dim3(1), 0, ctx->stream, ctx->d_words + 5, ctx->d_words + 4, (uint64_t)nphotons, ctx->packet_mode);
        }
        HIP_TRY(hipGetLastError());
        const int nev = time_kernels ? 6 * max_steps : 0;
        while ((int)ctx->step_events.size() < nev) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); ctx->step_events.push_back(e); }
        long long n_upper = (long long)nphotons;
        int step = 0, next_check = 1, steps_timed = 0;
        bool done = false, tail_done = false;
        const long long few = (long long)PROP_BLOCK * 16 * 8;
        const bool fused_tail = ctx->fused_tail && (ctx->wide_walk == CHROMA_WALK_COOP || ctx->wide_walk == CHROMA_WALK_QUAD ||
                                                    ctx->wide_walk == CHROMA_WALK_PAIR);    // (the cross-check walks keep per-step launches)
        int tail_step = -1;                  // the step at which the fused tail was launched
        while (step < max_steps && !done) {
            if (fused_tail && n_upper < few) {
                // the reference's last launch: all remaining steps at once, 8 lanes per photon
                bool launched = false;
                rc = launch_tail(ctx, geom, pv, n_upper, in_q, out_q, work_in, rng, max_steps - step, use_weights,
                                 step == 0 ? scatter_first : 0, time_kernels ? ctx->step_events.data() + 6 * step : nullptr, &launched,
                                 step == 0 ? (uint32_t)nphotons : 0u);
                if (rc) return rc;
                if (launched) {
                    if (time_kernels) { tail_step = step; steps_timed = step + 1; }
                    step = max_steps;
                    tail_done = true;            // (it wrote every photon it held back to the caller's arrays)
                    break;
                }
            }
            rc = launch_split_step(ctx, geom, pv, n_upper, in_q, out_q, work_in, work_out, rng, use_weights,
                                   step == 0 ? scatter_first : 0, time_kernels ? ctx->step_events.data() + 6 * step : nullptr,
                                   step == 0 ? (uint32_t)nphotons : 0u, step_uses_quad_walk(ctx, geom), step == 0 && packet_offered);
            if (rc) return rc;
            if (time_kernels) steps_timed = step + 1;
            step++;
            std::swap(in_q, out_q);
            std::swap(work_in, work_out);
            if (step == next_check && step < max_steps) {
                // survivors = tail - 1 of what is now the input queue
                HIP_TRY(hipMemcpyAsync(ctx->h_words + 1, in_q, 4, hipMemcpyDeviceToHost, ctx->stream));
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                n_upper = (long long)ctx->h_words[1] - 1;
                if (n_upper <= 0) done = true;
                // look every step once the tail is near, so that it starts when the reference's does
                next_check = (fused_tail && n_upper < 16 * few) ? step + 1 : (step < 8) ? step * 2 : step + 8;
            }
        }
        if (!tail_done && !done) {
            // max_steps reached with photons still alive: they go back to the caller's arrays
            unsigned blocks = (unsigned)std::min<long long>((n_upper + 255) / 256, 4096);
            hipLaunchKernelGGL(k_store_working, dim3(std::max(blocks, 1u)), dim3(256), 0, ctx->stream, geom->view, pv, in_q, work_in);
        }
        HIP_TRY(hipMemcpyAsync(ctx->h_step, ctx->d_step, sizeof(StepState), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        launches = ((const StepState *)ctx->h_step)->launches;
        for (int k = 0; k < steps_timed; k++) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, ctx->step_events[6 * k], ctx->step_events[6 * k + 2]));
            kernel_ms += ms;
            if (k == tail_step) continue;             // the fused tail is not a ray-cast launch
            HIP_TRY(hipEventElapsedTime(&ms, ctx->step_events[6 * k + 5], ctx->step_events[6 * k + 1]));
            raycast_ms += ms;
            raycast_launches++;
            if (k == 0 && packet_offered) {           // the first step's k_raycast_packet launch (an empty one when the photons are not coherent)
                HIP_TRY(hipEventElapsedTime(&ms, ctx->step_events[6 * k + 3], ctx->step_events[6 * k + 5]));
                packet_ms += ms;
                packet_launches++;
            }
            HIP_TRY(hipEventElapsedTime(&ms, ctx->step_events[6 * k + 1], ctx->step_events[6 * k + 4]));
            physics_ms += ms;                         // the main pass of k_physics (not the fix-up pass)
            physics_launches++;
        }
    } else {
        // CHROMA_TAIL=fused: the lane-per-photon kernel with the reference's own launch shapes
        hipLaunchKernelGGL(k_init_queue, dim3((unsigned)((nphotons + 255) / 256)), dim3(256), 0, ctx->stream, in_q,
                           (uint64_t)nphotons, ncopies, (uint32_t)(nphotons / ncopies));
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, ctx->stream, out_q, 1u);
        HIP_TRY(hipGetLastError());
        uint64_t n = nphotons;
        int step = 0;
        while (step < max_steps) {
            const bool few = n < (uint64_t)PROP_BLOCK * 16 * 8;
            int nsteps = (few || use_weights) ? (max_steps - step) : 1;
            if (time_kernels) HIP_TRY(hipEventRecord(ctx->ev_start, ctx->stream));
            rc = launch_propagate(ctx, geom, pv, 0, (int)n, in_q + 1, out_q, rng, nsteps, use_weights, scatter_first);
            if (rc) return rc;
            launches++;
            if (time_kernels) {
                HIP_TRY(hipEventRecord(ctx->ev_stop, ctx->stream));
                HIP_TRY(hipEventSynchronize(ctx->ev_stop));
                float ms = 0.f;
                HIP_TRY(hipEventElapsedTime(&ms, ctx->ev_start, ctx->ev_stop));
                kernel_ms += ms;
            }
            step += nsteps;
            scatter_first = 0;
            if (step < max_steps) {
                std::swap(in_q, out_q);
                // survivors = tail - 1 (one 4-byte read per step, as photon.py:250)
                HIP_TRY(hipMemcpyAsync(ctx->h_words + 1, in_q, 4, hipMemcpyDeviceToHost, ctx->stream));
                hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, ctx->stream, out_q, 1u);
                HIP_TRY(hipStreamSynchronize(ctx->stream));
                n = (uint64_t)ctx->h_words[1] - 1;
                if (n == 0) break;
            }
        }
    }
    uint32_t word = 0;
    if (use_records || hr) {
        // one pass: records -> the caller's arrays, abort word, hit count + compaction + per-channel arrays (k_finalize_hits)
        HitsOut ho; memset(&ho, 0, sizeof ho);
        if (hr) {
            ho.want = 1;
            ho.detection_state = hr->detection_state;
            if (hr->dst) { ho.dst = to_view(hr->dst); ho.channels = hr->d_channels; ho.capacity = hr->capacity; }
            ho.hit_count = hr->d_hit_count; ho.earliest = hr->d_hit_count ? hr->d_earliest_time_bits : nullptr;
        }
        HIP_TRY(hipMemsetAsync(ctx->d_words, 0, 12, ctx->stream));
        const unsigned blocks = (unsigned)((nphotons + COPY_ITEMS * 256 - 1) / (COPY_ITEMS * 256));
        hipLaunchKernelGGL(k_finalize_hits, dim3(blocks), dim3(256), 0, ctx->stream, geom->view, pv, (const float4 *)ctx->final_use, ctx->final_epoch,
                           (uint64_t)nphotons, ho, ctx->d_words);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(ctx->h_words, ctx->d_words, 12, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        word = ctx->h_words[2];
        if (hr) hr->nhits = ctx->h_words[0];
    } else {
        // abort warning word (photon.py:254-255)
        HIP_TRY(hipMemsetAsync(ctx->d_words + 2, 0, 4, ctx->stream));
        {
            unsigned blocks = (unsigned)std::min<uint64_t>((nphotons + 255) / 256, 4096);
            hipLaunchKernelGGL(k_flags_or, dim3(blocks), dim3(256), 0, ctx->stream, photons->flags, (uint64_t)nphotons,
                               CHROMA_NAN_ABORT, ctx->d_words + 2);
            HIP_TRY(hipGetLastError());
        }
        rc = read_word(ctx, 2, &word); if (rc) return rc;
    }
    if (aborted) *aborted = (word & CHROMA_NAN_ABORT) ? 1 : 0;
    if (stats) {
        rc = chroma_propagate_stats_read(ctx, stats); if (rc) return rc;
        stats->launches += launches;
        stats->kernel_ms += kernel_ms;
        stats->raycast_ms += raycast_ms;
        stats->raycast_launches += raycast_launches;
        stats->physics_ms += physics_ms;
        stats->physics_launches += physics_launches;
        stats->packet_ms += packet_ms;
        stats->packet_launches += packet_launches;
        stats->reordered += reordered;
        if (stats->stack_overflows) return set_error(CHROMA_ERR_STACK, "traversal stack overflowed for %llu rays", (unsigned long long)stats->stack_overflows);
    } else {
        chroma_propagate_stats tmp; memset(&tmp, 0, sizeof tmp);
        rc = chroma_propagate_stats_read(ctx, &tmp); if (rc) return rc;
        if (tmp.stack_overflows) return set_error(CHROMA_ERR_STACK, "traversal stack overflowed for %llu rays", (unsigned long long)tmp.stack_overflows);
    }
    return CHROMA_OK;
}

int chroma_propagate(chroma_ctx *ctx, chroma_geometry *geom, const chroma_photon_arrays *photons, uint64_t nphotons,
                     uint32_t ncopies, chroma_rng rng, int32_t max_steps, int32_t use_weights, int32_t scatter_first,
                     int32_t time_kernels, chroma_propagate_stats *stats, int32_t *aborted)
{
    return propagate_impl(ctx, geom, photons, nphotons, ncopies, rng, max_steps, use_weights, scatter_first, time_kernels, stats, aborted, nullptr);
}

int chroma_propagate_hits(chroma_ctx *ctx, chroma_geometry *geom, const chroma_photon_arrays *photons, uint64_t nphotons,
                          uint32_t ncopies, chroma_rng rng, int32_t max_steps, int32_t use_weights, int32_t scatter_first,
                          int32_t time_kernels, chroma_propagate_stats *stats, int32_t *aborted, chroma_hits_request *hits)
{
    if (!hits) return set_error(CHROMA_ERR_INVALID, "bad argument");
    return propagate_impl(ctx, geom, photons, nphotons, ncopies, rng, max_steps, use_weights, scatter_first, time_kernels, stats, aborted, hits);
}

int chroma_channel_hits(chroma_ctx *ctx, chroma_geometry *geom, uint64_t nphotons, uint32_t detection_state,
                        const chroma_photon_arrays *photons, uint32_t *d_hit_count, uint32_t *d_earliest_time_bits)
{
    if (!ctx || !geom || !d_hit_count) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (!geom->view.nsolids) return set_error(CHROMA_ERR_INVALID, "geometry has no detector channel map");
    int rc = check_photons(photons, false); if (rc) return rc;
    if (nphotons == 0) return CHROMA_OK;
    hipLaunchKernelGGL(k_channel_hits, dim3((unsigned)((nphotons + 255) / 256)), dim3(256), 0, ctx->stream, geom->view,
                       photons->flags, photons->last_hit_triangles, photons->t, (uint64_t)nphotons, detection_state,
                       d_hit_count, d_earliest_time_bits);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

int chroma_daq_reset(chroma_ctx *ctx, float maxtime, uint32_t nchannels, uint32_t *d_earliest_time_int,
                     uint32_t *d_channel_q_int, uint32_t *d_channel_histories)
{
    if (!ctx || !d_earliest_time_int || !d_channel_q_int || !d_channel_histories) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (nchannels == 0) return CHROMA_OK;
    hipLaunchKernelGGL(k_daq_reset, dim3((nchannels + 255) / 256), dim3(256), 0, ctx->stream, maxtime, nchannels,
                       d_earliest_time_int, d_channel_q_int, d_channel_histories);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

int chroma_daq_acquire(chroma_ctx *ctx, chroma_geometry *geom, const chroma_daq_tables *tables, int32_t first_photon,
                       int32_t nphotons, uint32_t detection_state, const chroma_photon_arrays *photons, chroma_rng rng,
                       uint32_t acquisition, float global_weight, uint32_t *d_earliest_time_int,
                       uint32_t *d_channel_q_int, uint32_t *d_channel_histories)
{
    if (!ctx || !geom || !tables || !d_earliest_time_int || !d_channel_q_int || !d_channel_histories)
        return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (!geom->view.nsolids) return set_error(CHROMA_ERR_INVALID, "geometry has no detector channel map");
    if (tables->time_cdf_len < 2 || tables->charge_cdf_len < 2 || !tables->d_time_cdf_x || !tables->d_time_cdf_y ||
        !tables->d_charge_cdf_x || !tables->d_charge_cdf_y || !(tables->charge_unit > 0.0f))
        return set_error(CHROMA_ERR_INVALID, "DAQ tables: need two CDFs of at least 2 points and a positive charge unit");
    int rc = check_photons(photons, false); if (rc) return rc;
    if (nphotons <= 0) return CHROMA_OK;
    hipLaunchKernelGGL(k_run_daq, dim3((nphotons + 255) / 256), dim3(256), 0, ctx->stream, geom->view, *tables, first_photon,
                       nphotons, detection_state, photons->t, photons->flags, photons->last_hit_triangles, photons->weights,
                       rng.seed, rng.photon_id_base, acquisition, global_weight, d_earliest_time_int, d_channel_q_int,
                       d_channel_histories);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

int chroma_daq_acquire_many(chroma_ctx *ctx, chroma_geometry *geom, const chroma_daq_tables *tables, int32_t first_photon,
                            int32_t nphotons, uint32_t detection_state, const chroma_photon_arrays *photons, chroma_rng rng,
                            uint32_t acquisition, float global_weight, int32_t ndaq, int32_t channel_stride,
                            uint32_t *d_earliest_time_int, uint32_t *d_channel_q_int, uint32_t *d_channel_histories)
{
    if (!ctx || !geom || !tables || !d_earliest_time_int || !d_channel_q_int || !d_channel_histories)
        return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (!geom->view.nsolids) return set_error(CHROMA_ERR_INVALID, "geometry has no detector channel map");
    if (ndaq < 1 || channel_stride < (int32_t)geom->view.nchannels)
        return set_error(CHROMA_ERR_INVALID, "ndaq must be positive and the channel stride at least the number of channels");
    if (tables->time_cdf_len < 2 || tables->charge_cdf_len < 2 || !tables->d_time_cdf_x || !tables->d_time_cdf_y ||
        !tables->d_charge_cdf_x || !tables->d_charge_cdf_y || !(tables->charge_unit > 0.0f))
        return set_error(CHROMA_ERR_INVALID, "DAQ tables: need two CDFs of at least 2 points and a positive charge unit");
    int rc = check_photons(photons, false); if (rc) return rc;
    if (nphotons <= 0) return CHROMA_OK;
    const long long total = (long long)nphotons * ndaq;
    hipLaunchKernelGGL(k_run_daq_many, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, ctx->stream, geom->view, *tables,
                       first_photon, nphotons, detection_state, photons->t, photons->flags, photons->last_hit_triangles,
                       photons->weights, rng.seed, rng.photon_id_base, acquisition, global_weight, ndaq, channel_stride,
                       d_earliest_time_int, d_channel_q_int, d_channel_histories);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

int chroma_daq_convert(chroma_ctx *ctx, uint32_t nchannels, float charge_unit, const uint32_t *d_earliest_time_int,
                       const uint32_t *d_channel_q_int, float *d_earliest_time, float *d_channel_q)
{
    if (!ctx || !d_earliest_time_int || !d_channel_q_int || !d_earliest_time || !d_channel_q)
        return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (nchannels == 0) return CHROMA_OK;
    hipLaunchKernelGGL(k_daq_convert, dim3((nchannels + 255) / 256), dim3(256), 0, ctx->stream, nchannels, charge_unit,
                       d_earliest_time_int, d_channel_q_int, d_earliest_time, d_channel_q);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

int chroma_render(chroma_ctx *ctx, chroma_geometry *geom, int32_t nthreads, const float *d_origin, const float *d_direction,
                  uint32_t alpha_depth, uint32_t *d_pixels, float *d_dx, uint32_t *d_dxlen, float *d_color, uint32_t bg_color)
{
    if (!ctx || !geom || !d_origin || !d_direction || !d_pixels || !d_dx || !d_dxlen || !d_color)
        return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (alpha_depth < 1) return set_error(CHROMA_ERR_INVALID, "alpha_depth must be at least 1");
    if (nthreads <= 0) return CHROMA_OK;
    if (geom->stack_need > STACK_LDS + STACK_SCRATCH)
        return set_error(CHROMA_ERR_STACK, "BVH needs %u traversal stack entries, more than the %d supported", geom->stack_need, STACK_LDS + STACK_SCRATCH);
    hipLaunchKernelGGL((k_render<STACK_LDS>), dim3((unsigned)((nthreads + PROP_BLOCK - 1) / PROP_BLOCK)), dim3(PROP_BLOCK), 0, ctx->stream,
                       geom->view, (const uint32_t *)geom->d_colors, (int)nthreads, d_origin, d_direction, alpha_depth, d_pixels, d_dx,
                       d_dxlen, (float4 *)d_color, bg_color, ctx->d_counters);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

static int rays_transform(chroma_ctx *ctx, int32_t n, float *d_a, int mode, float phi, const float axis[3], const float point[3])
{
    if (!ctx || !d_a) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (n <= 0) return CHROMA_OK;
    const float zero[3] = {0.f, 0.f, 0.f};
    if (!axis) axis = zero;
    if (!point) point = zero;
    hipLaunchKernelGGL(k_rays_transform, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)n, d_a, mode, phi,
                       axis[0], axis[1], axis[2], point[0], point[1], point[2]);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}
int chroma_points_translate(chroma_ctx *ctx, int32_t n, float *d_a, const float v[3]) { return rays_transform(ctx, n, d_a, 0, 0.f, nullptr, v); }
int chroma_points_rotate(chroma_ctx *ctx, int32_t n, float *d_a, float phi, const float axis[3]) { return rays_transform(ctx, n, d_a, 1, phi, axis, nullptr); }
int chroma_points_rotate_around_point(chroma_ctx *ctx, int32_t n, float *d_a, float phi, const float axis[3], const float point[3])
{ return rays_transform(ctx, n, d_a, 2, phi, axis, point); }

int chroma_probe(chroma_ctx *ctx, int32_t fn, uint64_t n, const float *d_x, const float *d_tab_x, const float *d_tab_f,
                 uint32_t ntab, float start, float step, float *d_out)
{
    if (!ctx || !d_x || !d_out || fn < 0 || fn > 3) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if ((fn == 0 && (!d_tab_f || ntab < 2)) || (fn == 1 && (!d_tab_x || ntab < 2)) || (fn == 2 && (!d_tab_x || !d_tab_f || ntab < 2)))
        return set_error(CHROMA_ERR_INVALID, "probe %d: table missing", fn);
    if (n == 0) return CHROMA_OK;
    hipLaunchKernelGGL(k_probe, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, (int)fn, n, d_x, d_tab_x, d_tab_f,
                       ntab, start, step, d_out);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

int chroma_generate_bomb(chroma_ctx *ctx, const chroma_photon_arrays *photons, uint64_t nphotons, uint64_t seed,
                         uint64_t id_base, const float pos[3], float wavelength_lo, float wavelength_hi)
{
    if (!ctx || !pos) return set_error(CHROMA_ERR_INVALID, "bad argument");
    int rc = check_photons(photons, true); if (rc) return rc;
    if (nphotons == 0) return CHROMA_OK;
    hipLaunchKernelGGL(k_generate_bomb, dim3((unsigned)((nphotons + 255) / 256)), dim3(256), 0, ctx->stream, to_view(photons),
                       (uint64_t)nphotons, seed, id_base, pos[0], pos[1], pos[2], wavelength_lo, wavelength_hi);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

// ---- the hit reduction across GPUs ----------------------------------------------------
// Photons never interact and every GPU holds the whole geometry, so a batch sharded over the GPUs of a
// node needs exactly one exchange: its per-channel arrays.  That exchange is RCCL on the library's own
// stream, on the device arrays the hit kernels filled -- nothing is staged through the host.  RCCL is
// found with dlopen when the first communicator call is made (a process that already holds an RCCL, e.g.
// torch's, gets that one through the shared-object name), so single-GPU users never load it.
struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;

static int rccl_load()
{
    if (g_rccl.handle) return CHROMA_OK;
    void *h = nullptr;
    for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
        h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
        if (h) break;
    }
    if (!h) return set_error(CHROMA_ERR_INVALID, "RCCL not found (dlopen librccl.so.1): %s", dlerror());
#define SYM(field, name) \
    do { *(void **)(&g_rccl.field) = dlsym(h, name); \
         if (!g_rccl.field) { dlclose(h); return set_error(CHROMA_ERR_INVALID, "RCCL: symbol %s missing", name); } } while (0)
    SYM(GetUniqueId, "ncclGetUniqueId"); SYM(CommInitRank, "ncclCommInitRank"); SYM(CommDestroy, "ncclCommDestroy");
    SYM(AllReduce, "ncclAllReduce"); SYM(AllGather, "ncclAllGather"); SYM(GroupStart, "ncclGroupStart");
    SYM(GroupEnd, "ncclGroupEnd"); SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
    g_rccl.handle = h;
    return CHROMA_OK;
}
#define RCCL_TRY(expr)                                                                             \
    do {                                                                                           \
        ncclResult_t r_ = (expr);                                                                  \
        if (r_ != ncclSuccess)                                                                     \
            return set_error(CHROMA_ERR_INVALID, "%s failed: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

__global__ void k_or_gathered(uint32_t *out, const uint32_t *gathered, uint32_t n, int nranks)
{
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t acc = 0;
    for (int r = 0; r < nranks; r++) acc |= gathered[(size_t)r * n + i];
    out[i] = acc;
}

int chroma_comm_unique_id(uint8_t id[CHROMA_COMM_ID_BYTES])
{
    if (!id) return set_error(CHROMA_ERR_INVALID, "null id");
    static_assert(CHROMA_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "id size");
    int rc = rccl_load(); if (rc) return rc;
    ncclUniqueId u;
    RCCL_TRY(g_rccl.GetUniqueId(&u));
    memcpy(id, u.internal, NCCL_UNIQUE_ID_BYTES);
    return CHROMA_OK;
}

int chroma_comm_init(chroma_ctx *ctx, int32_t nranks, int32_t rank, const uint8_t id[CHROMA_COMM_ID_BYTES])
{
    if (!ctx || !id || nranks < 1 || rank < 0 || rank >= nranks) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (ctx->comm) return set_error(CHROMA_ERR_INVALID, "this context already has a communicator");
    int rc = rccl_load(); if (rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId u;
    memcpy(u.internal, id, NCCL_UNIQUE_ID_BYTES);
    RCCL_TRY(g_rccl.CommInitRank(&ctx->comm, nranks, u, rank));
    ctx->comm_nranks = nranks;
    ctx->comm_rank = rank;
    return CHROMA_OK;
}

int chroma_comm_destroy(chroma_ctx *ctx)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (ctx->comm) {
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        RCCL_TRY(g_rccl.CommDestroy(ctx->comm));
        ctx->comm = nullptr;
        ctx->comm_nranks = 1;
        ctx->comm_rank = 0;
    }
    if (ctx->gather_buf) { hipFree(ctx->gather_buf); ctx->gather_buf = nullptr; ctx->gather_capacity = 0; }
    return CHROMA_OK;
}

// hit_count: sum; earliest-time bit patterns: min (non-negative times order like their bits,
// chroma/cuda/daq.cu:5-20).  In place, on the library's stream; without a communicator the arrays
// already are the whole job's.
int chroma_allreduce_hits(chroma_ctx *ctx, uint32_t *d_hit_count, uint32_t *d_earliest_time_bits, uint32_t nchannels)
{
    if (!ctx || !d_hit_count) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (!ctx->comm || nchannels == 0) return CHROMA_OK;
    // (the first error is kept and the group is ALWAYS closed: an early return between GroupStart and GroupEnd would
    //  leave the group open for every later RCCL call of the process -- torch's included, the library is shared)
    RCCL_TRY(g_rccl.GroupStart());
    ncclResult_t first = g_rccl.AllReduce(d_hit_count, d_hit_count, nchannels, ncclUint32, ncclSum, ctx->comm, ctx->stream);
    if (first == ncclSuccess && d_earliest_time_bits)
        first = g_rccl.AllReduce(d_earliest_time_bits, d_earliest_time_bits, nchannels, ncclUint32, ncclMin, ctx->comm, ctx->stream);
    const ncclResult_t end = g_rccl.GroupEnd();
    if (first == ncclSuccess) first = end;
    if (first != ncclSuccess) return set_error(CHROMA_ERR_INVALID, "chroma_allreduce_hits: %s", g_rccl.GetErrorString(first));
    return CHROMA_OK;
}

// The three integer arrays a DAQ acquisition accumulates (chroma/cuda/daq.cu:73-75) over sharded photons:
// earliest time bits (min), integer charge (sum), channel histories (bitwise OR -- not an RCCL reduction:
// all-gather, then OR locally).
int chroma_allreduce_daq(chroma_ctx *ctx, uint32_t *d_earliest_time_int, uint32_t *d_channel_q_int,
                         uint32_t *d_channel_histories, uint32_t nchannels)
{
    if (!ctx || !d_earliest_time_int || !d_channel_q_int || !d_channel_histories) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (!ctx->comm || nchannels == 0) return CHROMA_OK;
    const size_t need = (size_t)ctx->comm_nranks * nchannels;
    if (ctx->gather_capacity < need) {
        HIP_TRY(hipSetDevice(ctx->device));
        HIP_TRY(hipStreamSynchronize(ctx->stream));
        if (ctx->gather_buf) hipFree(ctx->gather_buf);
        ctx->gather_buf = nullptr; ctx->gather_capacity = 0;
        HIP_TRY(hipMalloc((void **)&ctx->gather_buf, need * sizeof(uint32_t)));
        ctx->gather_capacity = need;
    }
    RCCL_TRY(g_rccl.GroupStart());
    ncclResult_t first = g_rccl.AllReduce(d_earliest_time_int, d_earliest_time_int, nchannels, ncclUint32, ncclMin, ctx->comm, ctx->stream);
    if (first == ncclSuccess)
        first = g_rccl.AllReduce(d_channel_q_int, d_channel_q_int, nchannels, ncclUint32, ncclSum, ctx->comm, ctx->stream);
    if (first == ncclSuccess)
        first = g_rccl.AllGather(d_channel_histories, ctx->gather_buf, nchannels, ncclUint32, ctx->comm, ctx->stream);
    const ncclResult_t end = g_rccl.GroupEnd();          // (always: see chroma_allreduce_hits)
    if (first == ncclSuccess) first = end;
    if (first != ncclSuccess) return set_error(CHROMA_ERR_INVALID, "chroma_allreduce_daq: %s", g_rccl.GetErrorString(first));
    hipLaunchKernelGGL(k_or_gathered, dim3((nchannels + 255) / 256), dim3(256), 0, ctx->stream, d_channel_histories,
                       ctx->gather_buf, nchannels, ctx->comm_nranks);
    HIP_TRY(hipGetLastError());
    return CHROMA_OK;
}

}  // extern "C"
