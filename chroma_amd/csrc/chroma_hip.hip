// chroma_hip.hip -- the C ABI of libchroma_hip.so (gfx950 / MI355X only): contexts, device memory, geometry upload, the host
// side of chroma_propagate* (step loop, launch policy) and of every other entry point of include/chroma_hip.h.
//
// The kernels live in one header per family, included below in dependency order (one translation unit: the families share
// device helpers and launch-time constants):
//   kernel_propagate_fused.h      k_propagate -- lane-per-photon fused multi-step kernel
//   kernel_step_control.h         hit codes, k_step_begin, ray records, k_ray_setup
//   kernels_raycast_crosscheck.h  k_raycast_persistent / _wide / _coop -- cross-check walks (+ the eight-lane helpers)
//   kernel_raycast_quad.h         k_raycast_quad -- the DEFAULT ray cast
//   kernel_raycast_pair.h         k_raycast_pair -- cross-check walk
//   kernel_raycast_literal.h      k_raycast_literal -- the EXACT walk (mesh.h:42-118 for every ray), and its cast for the tail kernel
//   kernel_tail_coop.h            k_tail_coop -- the last photons' remaining steps in one launch (default and exact walk)
//   kernel_raycast_retry.h        k_raycast_retry -- the strict loop for rays the fast walks hand over
//   kernel_physics.h              k_physics
//   kernels_working_set.h         k_load_working, k_store_working
//   kernels_photons_hits.h        photon-array kernels, hit extraction, k_finalize_hits
//   kernels_daq_render.h          DAQ, distance_to_mesh, render, transforms, bomb generator, probe
//   experimental/*.h              measured-and-not-faster kernels: ONLY in build_variants/libchroma_hip_experimental.so
// See DESIGN.md for the data layout and what bounds each kernel.
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

#ifndef CHROMA_EXPERIMENTAL
#define CHROMA_EXPERIMENTAL 0      // 1: build_variants/libchroma_hip_experimental.so (csrc/experimental/: packet ray cast, dealt physics, autosort)
#endif
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
    int claim_static = 5 | 8 << 4;                  // eighths of a wave's share of a launch's rays that it takes without the counter (k_raycast_quad; CHROMA_CLAIM_STATIC)
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
    std::mutex call_mu;                                    // one chroma_propagate* call at a time per context (see propagate_impl)
    std::mutex pool_mu;
    std::multimap<size_t, PoolBlock> pool;                 // free blocks by size
    std::unordered_map<void *, size_t> live;               // size of every block handed out
    std::vector<hipEvent_t> pool_events;                   // spare events
    size_t pool_bytes = 0, pool_limit = 0;
    uint64_t pool_hits = 0, pool_misses = 0;
    // ---- host -> device uploads: a second stream and a ring of pinned staging buffers (chroma_upload) ----
    hipStream_t copy_stream = nullptr;
    hipStream_t aux_stream = nullptr;          // k_finalize_hits beside the tail kernel (launch_tail)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
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

// What one call does, fixed when it starts: the context's settings (chroma_set_walk / _tail / _packet / _autosort / _counting,
// the CHROMA_* environment) overridden by the call's own chroma_propagate_options.  Every function below a public entry
// point reads THIS, never the context's mutable settings, so a call's behaviour cannot change under it.
struct CallOpts {
    int walk, packet, autosort, counting;
    int fused_tail, split_tail;
};
static CallOpts call_opts(const chroma_ctx *ctx)
{
    return CallOpts{ctx->wide_walk, ctx->packet_mode, ctx->autosort_mode, ctx->counting, ctx->fused_tail, ctx->split_tail};
}

// hipMalloc for the library's own working buffers: when the device is out of memory, everything parked in the pool
// behind chroma_malloc / chroma_free is given back first (defined next to the pool)
static hipError_t ctx_malloc(chroma_ctx *ctx, void **ptr, size_t bytes);
extern "C" hipError_t chroma_internal_malloc(chroma_ctx *ctx, void **ptr, size_t bytes) { return ctx_malloc(ctx, ptr, bytes); }

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

#include "kernel_propagate_fused.h"

#include "kernel_step_control.h"

#include "kernels_raycast_crosscheck.h"

#include "kernel_raycast_quad.h"

#include "kernel_raycast_pair.h"

#if CHROMA_EXPERIMENTAL
#include "experimental/raycast_packet.h"
#endif

#include "kernel_raycast_literal.h"

#include "kernel_tail_coop.h"

#include "kernel_raycast_retry.h"

#include "kernel_physics.h"

#if CHROMA_EXPERIMENTAL
#include "experimental/physics_deal.h"
#else
#define PHYS_DEAL 0
#define PHYS_DEAL_BLOCK 512
#endif

#include "kernels_working_set.h"

#include "kernels_photons_hits.h"

#include "kernels_daq_render.h"

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
static int launch_propagate_t(chroma_ctx *ctx, const CallOpts &co, chroma_geometry *geom, PhotonView pv, int first, int nthreads,
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

static int launch_propagate(chroma_ctx *ctx, const CallOpts &co, chroma_geometry *geom, PhotonView pv, int first, int nthreads,
                            const uint32_t *in_q, uint32_t *out_q, chroma_rng rng, int max_steps, int use_weights,
                            int scatter_first)
{
    if (nthreads <= 0) return CHROMA_OK;
    if (co.counting)
        return launch_propagate_t<true>(ctx, co, geom, pv, first, nthreads, in_q, out_q, rng, max_steps, use_weights, scatter_first);
    return launch_propagate_t<false>(ctx, co, geom, pv, first, nthreads, in_q, out_q, rng, max_steps, use_weights, scatter_first);
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
// `work_in`/`work_out` the working sets that go with them.  With `ev` (SIX events per step, indices 0..5): [0] step start,
// [3] ray-cast kernels start, [5] start of the step's own ray cast (after an experimental packet launch), [1] its end,
// [4] end of the main physics pass, [2] step end.
// The walk whose steps chain their ray records from kernel to kernel (k_load_working -> k_raycast_quad -> k_physics ->
// k_raycast_quad ...) instead of running k_ray_setup: the default one.
static bool step_uses_quad_walk(const CallOpts &co, const chroma_geometry *geom)
{
    return geom->view.wnodes != nullptr && co.walk == CHROMA_WALK_QUAD && geom->wide_stack_need <= QUAD_STACK + COOP_SPILL;
}

// k_physics for one pass of a step.  `fixup`: 0 the main pass over every slot, 1 the slots k_raycast_retry has walked again
// (a short list: a small grid), 2 every slot with the ray cast's results taken as they are (the exact walk).
static void launch_physics(chroma_ctx *ctx, const CallOpts &co, chroma_geometry *geom, const PhotonView &pv, long long n_upper,
                           const float4 *work_in, uint32_t *out_q, float4 *work_out, chroma_rng rng, int use_weights, int scatter_first,
                           int fixup, float4 *rays_next, int literal_rays = 0)
{
    StepState *st = ctx->d_step;
    const bool plain = geom->view.plain_optics != 0;      // (no re-emitting component, default surface model only)
    bool deal = false;
#if PHYS_DEAL
    deal = plain && !ctx->final_use;                      // (photons of a block dealt by what happens to them: experimental/physics_deal.h)
#endif
    const int pb = deal ? PHYS_DEAL_BLOCK : PHYS_BLOCK_OF(!plain);
    unsigned blocks = (unsigned)std::min<long long>((n_upper + pb - 1) / pb, std::max<long long>(1, (long long)ctx->physics_blocks * PHYS_BLOCK / pb));
    // (the retry list is ~1e-3 of the slots with plain optics: an eighth of the grid strides over it in a round or two, and a
    //  launch of 2048 blocks that find nothing to do costs 0.07 ms, 29 times per batch; a plain geometry with faces on the
    //  world box lists a good part of its hits for the exact check, so not less than that)
    if (fixup == 1 && plain) blocks = std::max(std::min(blocks, 64u), blocks / 8);
    DeviceCounters *pc = co.counting ? ctx->d_counters : nullptr;
#if PHYS_DEAL
    if (deal) {
        hipLaunchKernelGGL(k_physics_deal, dim3(blocks), dim3(PHYS_DEAL_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in, out_q, work_out,
                           ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights, scatter_first,
                           ctx->retry_list, fixup, pc, rays_next);
        return;
    }
#endif
    if (plain)
        hipLaunchKernelGGL((k_physics<false>), dim3(blocks), dim3(PHYS_BLOCK_OF(false)), 0, ctx->stream, geom->view, pv, st, work_in, out_q, work_out,
                           ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights, scatter_first,
                           ctx->retry_list, fixup, pc, rays_next, ctx->final_use, ctx->final_epoch, literal_rays);
    else
        hipLaunchKernelGGL((k_physics<true>), dim3(blocks), dim3(PHYS_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in, out_q, work_out,
                           ctx->hit_triangle, ctx->hit_distance, rng.seed, rng.photon_id_base, use_weights, scatter_first,
                           ctx->retry_list, fixup, pc, rays_next, ctx->final_use, ctx->final_epoch, literal_rays);
}

// `rays_ready`: the records of this step are in ctx->rays already (written by k_load_working or by the k_physics of
// the step before).  With the default walk the records of the next step go to ctx->rays_b, and the two are swapped.
static int launch_split_step(chroma_ctx *ctx, const CallOpts &co, chroma_geometry *geom, PhotonView pv, long long n_upper, const uint32_t *in_q,
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
    if (co.walk == CHROMA_WALK_LITERAL || co.walk == CHROMA_WALK_LITERAL_LANE) {
        // the reference's own loop for every ray (mesh.h:42-118 as it stands: its tree, its order, its box arithmetic,
        // every triangle tested the moment its leaf box is entered), then the physics on the results as they are.
        // LITERAL: k_raycast_literal (four lanes per ray, persistent waves; raycast_literal.h) + the strict lane-per-ray
        // loop for the few rays whose 1/d is not moderate; LITERAL_LANE: the strict loop for every ray (the cross-check).
        const bool lane_walk = co.walk == CHROMA_WALK_LITERAL_LANE;
        StepState *st = ctx->d_step;
        hipLaunchKernelGGL(k_step_begin, dim3(1), dim3(1), 0, ctx->stream, in_q, out_q, st,
                           use_weights ? 0xFFFFFFFFu : (uint32_t)(PROP_BLOCK * 16 * 8), first_n);
        if (ev) HIP_TRY(hipEventRecord(ev[0], ctx->stream));
        if (!lane_walk && !ctx->coop_spill) {
            HIP_TRY(hipSetDevice(ctx->device));
            HIP_TRY(ctx_malloc(ctx, (void **)&ctx->coop_spill, spill_entries(ctx) * sizeof(uint2)));
        }
        // (the exact walk chains its ray records from kernel to kernel like the default walk: k_load_working writes the first
        //  step's, k_physics the next step's while the photon is in its registers; LITERAL_LANE keeps the k_ray_setup pass)
        const bool chained_lit = !lane_walk && rays_ready;
        if (!chained_lit) {
            unsigned sblocks = (unsigned)std::min<long long>((n_upper + 255) / 256, (long long)ctx->physics_blocks * 4);
            hipLaunchKernelGGL(k_ray_setup, dim3(sblocks), dim3(256), 0, ctx->stream, geom->view, work_in, st, ctx->rays,
                               ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, &st->retry, lane_walk ? 0 : 1);
        }
        if (ev) { HIP_TRY(hipEventRecord(ev[3], ctx->stream)); HIP_TRY(hipEventRecord(ev[5], ctx->stream)); }
        if (lane_walk) {
            const unsigned lblocks = (unsigned)std::min<long long>((n_upper + PROP_BLOCK - 1) / PROP_BLOCK, (long long)ctx->persistent_waves);
            if (co.counting)
                hipLaunchKernelGGL((k_raycast_retry<true, true>), dim3(lblocks), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
            else
                hipLaunchKernelGGL((k_raycast_retry<false, true>), dim3(lblocks), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
        } else {
            const unsigned lwaves = (unsigned)std::min<long long>((n_upper + 15) / 16, (long long)ctx->quad_waves);
            const unsigned rblocks = (unsigned)std::min<long long>((n_upper + PROP_BLOCK - 1) / PROP_BLOCK, 8 * 256);
            if (co.counting) {
                hipLaunchKernelGGL((k_raycast_literal<true>), dim3(lwaves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk, chained_lit ? 1 : 0, ctx->retry_list, ctx->claim_static);
                hipLaunchKernelGGL((k_raycast_retry<true>), dim3(rblocks), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
            } else {
                hipLaunchKernelGGL((k_raycast_literal<false>), dim3(lwaves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk, chained_lit ? 1 : 0, ctx->retry_list, ctx->claim_static);
                hipLaunchKernelGGL((k_raycast_retry<false>), dim3(rblocks), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, st,
                                   ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
            }
        }
        if (ev) HIP_TRY(hipEventRecord(ev[1], ctx->stream));
        launch_physics(ctx, co, geom, pv, n_upper, work_in, out_q, work_out, rng, use_weights, scatter_first, 2, lane_walk ? nullptr : ctx->rays_b, 1);
        if (ev) { HIP_TRY(hipEventRecord(ev[4], ctx->stream)); HIP_TRY(hipEventRecord(ev[2], ctx->stream)); }
        HIP_TRY(hipGetLastError());
        if (!lane_walk) std::swap(ctx->rays, ctx->rays_b);       // (what k_physics wrote is the next step's input)
        return CHROMA_OK;
    }
    const bool pair = co.walk == CHROMA_WALK_PAIR && have_wide && geom->wide_stack_need <= PAIR_STACK + COOP_SPILL;
    const bool quad = !pair && (co.walk == CHROMA_WALK_QUAD || co.walk == CHROMA_WALK_PAIR) && have_wide &&
                      geom->wide_stack_need <= QUAD_STACK + COOP_SPILL;
    const bool coop = !pair && !quad && (co.walk == CHROMA_WALK_COOP || co.walk == CHROMA_WALK_QUAD) && have_wide &&
                      geom->wide_stack_need <= COOP_STACK + COOP_SPILL;
    const bool wide = !pair && !coop && !quad && co.walk != CHROMA_WALK_REFERENCE && have_wide && geom->wide_stack_need <= WIDE_STACK + WIDE_SPILL;
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
    const bool chained = quad && step_uses_quad_walk(co, geom);
    if (!(chained && rays_ready)) {
        unsigned sblocks = (unsigned)std::min<long long>((n_upper + 255) / 256, (long long)ctx->physics_blocks * 4);
        hipLaunchKernelGGL(k_ray_setup, dim3(sblocks), dim3(256), 0, ctx->stream, geom->view, work_in, st, ctx->rays,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, &st->retry);
    }
    const int settle = (chained && rays_ready) ? 1 : 0;
    float4 *rays_next = chained ? ctx->rays_b : nullptr;
    if (ev) HIP_TRY(hipEventRecord(ev[3], ctx->stream));        // the ray-cast kernels proper are timed from here
    const uint32_t *skip_quad = nullptr;
#if CHROMA_EXPERIMENTAL
    const bool offer_packet = packet && chained && rays_ready;
    skip_quad = offer_packet ? ctx->d_words + 4 : nullptr;
    if (offer_packet) {
        const unsigned pwaves = (unsigned)std::min<long long>((n_upper + WAVE - 1) / WAVE, (long long)ctx->quad_waves);
        if (co.counting)
            hipLaunchKernelGGL((k_raycast_packet<true>), dim3(pwaves), block, 0, ctx->stream, geom->view, ctx->rays, st, ctx->hit_triangle,
                               ctx->hit_distance, ctx->retry_list, ctx->d_counters, ctx->d_words + 4);
        else
            hipLaunchKernelGGL((k_raycast_packet<false>), dim3(pwaves), block, 0, ctx->stream, geom->view, ctx->rays, st, ctx->hit_triangle,
                               ctx->hit_distance, ctx->retry_list, ctx->d_counters, ctx->d_words + 4);
    }
#else
    (void)packet;
#endif
    if (ev) HIP_TRY(hipEventRecord(ev[5], ctx->stream));        // (k_raycast_packet before, the step's other ray cast after)
#define RAYCAST_LAUNCH(COUNT)                                                                                          \
    do {                                                                                                               \
        if (pair)                                                                                                      \
            hipLaunchKernelGGL((k_raycast_pair<COUNT>), grid, block, 0, ctx->stream, geom->view, ctx->rays, 0, st,      \
                               ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk); \
        else if (quad)                                                                                                 \
            hipLaunchKernelGGL((k_raycast_quad<COUNT>), grid, block, 0, ctx->stream, geom->view, ctx->rays, 0, st,      \
                               ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk, settle, skip_quad, ctx->claim_static); \
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
    if (co.counting) RAYCAST_LAUNCH(true); else RAYCAST_LAUNCH(false);
#undef RAYCAST_LAUNCH
    // physics for every slot whose hit is regular; then the strict walk and the physics of the rest
    launch_physics(ctx, co, geom, pv, n_upper, work_in, out_q, work_out, rng, use_weights, scatter_first, 0, rays_next);
    if (ev) HIP_TRY(hipEventRecord(ev[4], ctx->stream));          // end of the main physics pass
    // (both passes stride over the list and leave at once when it is short -- the usual case -- but a plain geometry
    //  with faces on the world box lists a good part of its hits for the exact check: grids for that)
    const unsigned rblocks = (unsigned)std::min<long long>((n_upper + PROP_BLOCK - 1) / PROP_BLOCK, 8 * 256);
    if (co.counting)
        hipLaunchKernelGGL((k_raycast_retry<true>), dim3(rblocks), block, 0, ctx->stream, geom->view, ctx->rays, st,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
    else
        hipLaunchKernelGGL((k_raycast_retry<false>), dim3(rblocks), block, 0, ctx->stream, geom->view, ctx->rays, st,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->d_counters);
    launch_physics(ctx, co, geom, pv, n_upper, work_in, out_q, work_out, rng, use_weights, scatter_first, 1, rays_next);
    if (ev) HIP_TRY(hipEventRecord(ev[2], ctx->stream));
    HIP_TRY(hipGetLastError());
    if (chained) std::swap(ctx->rays, ctx->rays_b);       // (what k_physics wrote is the next step's input)
    return CHROMA_OK;
}

// All remaining steps of the last photons in one launch (k_tail_coop).  Returns CHROMA_OK and sets
// *done when the geometry has a wide tree the kernel can walk; otherwise leaves *done false.
static int launch_tail(chroma_ctx *ctx, const CallOpts &co, chroma_geometry *geom, PhotonView pv, long long n_upper, const uint32_t *in_q,
                       uint32_t *out_q, const float4 *work_in, chroma_rng rng, int nsteps, int use_weights, int scatter_first,
                       hipEvent_t *ev, bool *done, uint32_t first_n = 0, const HitsOut *beside = nullptr, uint64_t nphotons = 0)
{
    // (`beside`: a call that ends in k_finalize_hits -- that pass runs on the context's auxiliary stream WHILE the tail kernel
    //  finishes the last photons, which are stamped first so that it leaves them to the tail kernel; the two meet again
    //  before the call reads its result words)
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
    HitsOut ho; memset(&ho, 0, sizeof ho);
    uint32_t *words = nullptr;
    if (beside) {
        ho = *beside;
        words = ctx->d_words;
        const uint32_t tail_mark = ctx->final_epoch | 0x80000000u;
        HIP_TRY(hipMemsetAsync(ctx->d_words, 0, 12, ctx->stream));
        if (ctx->final_use)
            hipLaunchKernelGGL(k_mark_tail, dim3(32), dim3(256), 0, ctx->stream, in_q, ctx->final_use, tail_mark);
        HIP_TRY(hipEventRecord(ctx->ev_fork, ctx->stream));
        HIP_TRY(hipStreamWaitEvent(ctx->aux_stream, ctx->ev_fork, 0));
        const unsigned blocks = (unsigned)((nphotons + COPY_ITEMS * 256 - 1) / (COPY_ITEMS * 256));
        hipLaunchKernelGGL(k_finalize_hits, dim3(blocks), dim3(256), 0, ctx->aux_stream, geom->view, pv, (const float4 *)ctx->final_use, ctx->final_epoch,
                           nphotons, ho, ctx->d_words, tail_mark);
        HIP_TRY(hipEventRecord(ctx->ev_join, ctx->aux_stream));
    }
    if (ev) { HIP_TRY(hipEventRecord(ev[0], ctx->stream)); HIP_TRY(hipEventRecord(ev[1], ctx->stream)); }
#define TAIL_LAUNCH(C, L) hipLaunchKernelGGL((k_tail_coop<C, L>), dim3(waves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, pv, st, work_in, \
                                             rng.seed, rng.photon_id_base, nsteps, use_weights, scatter_first, ctx->coop_spill, ctx->d_counters, ho, words)
    if (co.walk == CHROMA_WALK_LITERAL) { if (co.counting) TAIL_LAUNCH(true, true); else TAIL_LAUNCH(false, true); }
    else { if (co.counting) TAIL_LAUNCH(true, false); else TAIL_LAUNCH(false, false); }
#undef TAIL_LAUNCH
    if (ev) HIP_TRY(hipEventRecord(ev[2], ctx->stream));
    if (beside) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_join, 0));
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
    HIP_TRY(ctx_malloc(ctx, (void **)&d_need, std::max<size_t>(n, 1) * 4));
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
    HIP_TRY(hipStreamCreateWithFlags(&ctx->aux_stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
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
#if CHROMA_EXPERIMENTAL
        if (const char *e = getenv("CHROMA_PACKET")) ctx->packet_mode = !strcmp(e, "on") ? 1 : !strcmp(e, "auto") ? 2 : 0;
        if (const char *e = getenv("CHROMA_AUTOSORT")) ctx->autosort_mode = !strcmp(e, "on") || !strcmp(e, "1") ? 1 : !strcmp(e, "off") || !strcmp(e, "0") ? 0 : 2;
#endif
        if (const char *e = getenv("CHROMA_RAY_CHUNK")) ctx->ray_chunk = std::max(64, atoi(e));
        if (const char *e = getenv("CHROMA_COOP_CHUNK")) ctx->coop_chunk = std::max(8, atoi(e));
        if (const char *e = getenv("CHROMA_CLAIM_STATIC")) { int big = 0, small = 0; if (sscanf(e, "%d:%d", &big, &small) < 2) small = big; ctx->claim_static = std::min(8, std::max(0, big)) | std::min(8, std::max(0, small)) << 4; }
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
    if (ctx->aux_stream) { hipStreamSynchronize(ctx->aux_stream); hipStreamDestroy(ctx->aux_stream); }
    if (ctx->ev_fork) hipEventDestroy(ctx->ev_fork);
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
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
    // (a Python __del__ or the prefetch worker may call this from a thread whose current device is another GPU's)
    HIP_TRY(hipSetDevice(ctx->device));
    std::lock_guard<std::mutex> lock(ctx->pool_mu);
    auto it = ctx->live.find(d_ptr);
    if (it == ctx->live.end()) {                   // not one of ours (should not happen): the old behaviour
        HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(d_ptr));
        return CHROMA_OK;
    }
    const size_t size = it->second;
    ctx->live.erase(it);
    bool park = ctx->pool_bytes + size <= ctx->pool_limit;
    hipEvent_t ev = nullptr;
    if (park) {
        // a block is parked behind an event on the context's stream; should the event not come about, the block is
        // simply freed (after the stream has drained) -- it must never be left neither parked nor freed
        if (!ctx->pool_events.empty()) { ev = ctx->pool_events.back(); ctx->pool_events.pop_back(); }
        else if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ev = nullptr; park = false; }
        if (park && hipEventRecord(ev, ctx->stream) != hipSuccess) { ctx->pool_events.push_back(ev); park = false; }
    }
    if (!park) {
        (void)hipGetLastError();
        HIP_TRY(hipStreamSynchronize(ctx->stream)); HIP_TRY(hipFree(d_ptr));
        return CHROMA_OK;
    }
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
        hipError_t e = ctx_malloc(ctx, &dn, bytes);
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
        hipError_t e = ctx_malloc(ctx, &dtri, bytes);
        if (e != hipSuccess) { chroma_geometry_destroy(g); return set_error((int)e, "hipMalloc(%zu) for triangle records: %s", bytes, hipGetErrorString(e)); }
        g->allocations.push_back(dtri);
        g->device_bytes += bytes;
        uint32_t *d_rank = nullptr;
        e = ctx_malloc(ctx, (void **)&d_rank, std::max<size_t>(d->ntriangles, 1) * 4);
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
    return launch_propagate(ctx, call_opts(ctx), geom, to_view(photons), first_photon, nthreads, d_input_queue, d_output_queue, rng,
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
    std::lock_guard<std::mutex> call_lock(ctx->call_mu);          // (the fast path uses the context's queues and ray records, as a propagate call does)
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
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk, 0, nullptr, ctx->claim_static);
    else
        hipLaunchKernelGGL((k_raycast_quad<false>), dim3(waves), dim3(PROP_BLOCK), 0, ctx->stream, geom->view, ctx->rays, 0, st,
                           ctx->hit_triangle, ctx->hit_distance, ctx->retry_list, ctx->coop_spill, ctx->d_counters, ctx->coop_chunk, 0, nullptr, ctx->claim_static);
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
#if !CHROMA_EXPERIMENTAL
    if (mode != 0) return set_error(CHROMA_ERR_INVALID, "the packet ray cast is an experiment that is not part of this build (csrc/experimental/: build_variants/libchroma_hip_experimental.so)");
#endif
    ctx->packet_mode = mode;
    return CHROMA_OK;
}

int chroma_set_autosort(chroma_ctx *ctx, int32_t mode)
{
    if (!ctx) return set_error(CHROMA_ERR_INVALID, "null ctx");
    if (mode < 0 || mode > 2) return set_error(CHROMA_ERR_INVALID, "unknown autosort mode %d", mode);
#if !CHROMA_EXPERIMENTAL
    if (mode != 0) return set_error(CHROMA_ERR_INVALID, "the engine-side direction sort is an experiment that is not part of this build (csrc/experimental/: build_variants/libchroma_hip_experimental.so)");
#endif
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

#if CHROMA_EXPERIMENTAL
#include "experimental/autosort.h"
#else
// (product build: a call takes its photons as they come -- the engine-side direction sort lives in experimental/autosort.h)
static int propagate_order(chroma_ctx *, const CallOpts &, const PhotonView &, uint64_t, uint32_t, uint32_t **d_order) { *d_order = nullptr; return CHROMA_OK; }
#endif

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
                          uint32_t ncopies, chroma_rng rng, const chroma_propagate_options &opt,
                          chroma_propagate_stats *stats, int32_t *aborted, chroma_hits_request *hr)
{
    if (!ctx || !geom) return set_error(CHROMA_ERR_INVALID, "bad argument");
    int rc = check_photons(photons, true); if (rc) return rc;
    // what this call does: the context's settings as they are NOW, overridden by the call's own options
    CallOpts co = call_opts(ctx);
    if (opt.walk >= 0) {
        if (opt.walk > CHROMA_WALK_LITERAL_LANE) return set_error(CHROMA_ERR_INVALID, "unknown walk mode %d", opt.walk);
        co.walk = opt.walk;
    }
    if (opt.tail >= 0) {
        if (opt.tail > CHROMA_TAIL_FUSED) return set_error(CHROMA_ERR_INVALID, "unknown tail mode %d", opt.tail);
        co.split_tail = opt.tail != CHROMA_TAIL_FUSED;
        co.fused_tail = opt.tail == CHROMA_TAIL_COOP;
    }
    if (opt.counting >= 0) co.counting = opt.counting ? 1 : 0;
    const int32_t max_steps = opt.max_steps, use_weights = opt.use_weights, time_kernels = opt.time_kernels;
    int32_t scatter_first = opt.scatter_first;
    // one call at a time per context: the queues, working sets, step block and final records are the context's own
    std::lock_guard<std::mutex> call_lock(ctx->call_mu);
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
    const bool use_records = (hr != nullptr || records_always) && co.split_tail != 0;
    ctx->final_use = nullptr;
    if (use_records) {
        rc = ensure_final_records(ctx, nphotons); if (rc) return rc;
        ctx->final_epoch++;
        if (ctx->final_epoch == 0u || ctx->final_epoch >= 0x7FFFFFFFu) {            // (wrapped -- the top bit marks the tail kernel's photons --: no stale record may look current)
            HIP_TRY(hipMemsetAsync(ctx->final_rec, 0, ctx->final_capacity * 4 * sizeof(float4), ctx->stream));
            ctx->final_epoch = 1u;
        }
        ctx->final_use = ctx->final_rec;
    }
    struct FinalGuard { chroma_ctx *c; ~FinalGuard() { c->final_use = nullptr; } } final_guard{ctx};
    // where the call's last pass (k_finalize_hits) puts the hits, if it runs at all
    const bool finalize = use_records || hr != nullptr;
    bool finalized = false;              // it has been launched already, beside the tail kernel (launch_tail)
    HitsOut ho; memset(&ho, 0, sizeof ho);
    if (hr) {
        ho.want = 1;
        ho.detection_state = hr->detection_state;
        if (hr->dst) { ho.dst = to_view(hr->dst); ho.channels = hr->d_channels; ho.capacity = hr->capacity; }
        ho.hit_count = hr->d_hit_count; ho.earliest = hr->d_hit_count ? hr->d_earliest_time_bits : nullptr;
    }

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
    const bool device_steps = co.split_tail != 0;
    // the walks whose steps chain their ray records from kernel to kernel (k_load_working -> ray cast -> k_physics -> ...)
    const bool chains_rays = step_uses_quad_walk(co, geom) || co.walk == CHROMA_WALK_LITERAL;
    if (device_steps) {
        HIP_TRY(hipMemsetAsync(ctx->d_step, 0, sizeof(StepState), ctx->stream));
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, ctx->stream, in_q, 1u);
        hipLaunchKernelGGL(k_set_word, dim3(1), dim3(1), 0, ctx->stream, out_q, 1u);
        {
            unsigned blocks = (unsigned)std::min<uint64_t>((nphotons + PHYS_BLOCK - 1) / PHYS_BLOCK, (uint64_t)ctx->physics_blocks);
#if CHROMA_EXPERIMENTAL
            const bool probe = step_uses_quad_walk(co, geom) && co.packet != 0 && geom->wide_stack_need <= PACKET_STACK;
#else
            const bool probe = false;
#endif
            packet_offered = probe;
            HIP_TRY(hipMemsetAsync(ctx->d_words + 4, 0, 12, ctx->stream));          // [4] use_packet, [5] coherent waves, [6] waves
            uint32_t *d_order = nullptr;
            if (step_uses_quad_walk(co, geom)) { rc = propagate_order(ctx, co, pv, nphotons, ncopies, &d_order); if (rc) return rc; }
            hipLaunchKernelGGL(k_load_working, dim3(blocks), dim3(PHYS_BLOCK), 0, ctx->stream, geom->view, pv, in_q, work_in,
                               (uint64_t)nphotons, ncopies, (uint32_t)(nphotons / ncopies),
                               chains_rays ? ctx->rays : nullptr, (probe && co.packet == 2) ? ctx->d_words + 5 : nullptr,
                               (const uint32_t *)d_order, co.walk == CHROMA_WALK_LITERAL ? 1 : 0);
            if (d_order) { chroma_free(ctx, d_order); reordered = nphotons; }      // (parked until the stream has passed this point)
#if CHROMA_EXPERIMENTAL
            if (probe)
                hipLaunchKernelGGL(k_packet_decide, dim3(1), dim3(1), 0, ctx->stream, ctx->d_words + 5, ctx->d_words + 4, (uint64_t)nphotons, co.packet);
#endif
        }
        HIP_TRY(hipGetLastError());
        const int nev = time_kernels ? 6 * max_steps : 0;
        while ((int)ctx->step_events.size() < nev) { hipEvent_t e; HIP_TRY(hipEventCreate(&e)); ctx->step_events.push_back(e); }
        long long n_upper = (long long)nphotons;
        int step = 0, next_check = 1, steps_timed = 0;
        bool done = false, tail_done = false;
        const long long few = (long long)PROP_BLOCK * 16 * 8;
        const bool fused_tail = co.fused_tail && (co.walk == CHROMA_WALK_COOP || co.walk == CHROMA_WALK_QUAD ||
                                                    co.walk == CHROMA_WALK_PAIR || co.walk == CHROMA_WALK_LITERAL);    // (the cross-check walks keep per-step launches)
        int tail_step = -1;                  // the step at which the fused tail was launched
        while (step < max_steps && !done) {
            if (fused_tail && n_upper < few) {
                // the reference's last launch: all remaining steps at once, 8 lanes per photon
                bool launched = false;
                rc = launch_tail(ctx, co, geom, pv, n_upper, in_q, out_q, work_in, rng, max_steps - step, use_weights,
                                 step == 0 ? scatter_first : 0, time_kernels ? ctx->step_events.data() + 6 * step : nullptr, &launched,
                                 step == 0 ? (uint32_t)nphotons : 0u, finalize ? &ho : nullptr, (uint64_t)nphotons);
                if (rc) return rc;
                if (launched) {
                    finalized = finalize;
                    if (time_kernels) { tail_step = step; steps_timed = step + 1; }
                    step = max_steps;
                    tail_done = true;            // (it wrote every photon it held back to the caller's arrays)
                    break;
                }
            }
            rc = launch_split_step(ctx, co, geom, pv, n_upper, in_q, out_q, work_in, work_out, rng, use_weights,
                                   step == 0 ? scatter_first : 0, time_kernels ? ctx->step_events.data() + 6 * step : nullptr,
                                   step == 0 ? (uint32_t)nphotons : 0u, chains_rays, step == 0 && packet_offered);
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
            rc = launch_propagate(ctx, co, geom, pv, 0, (int)n, in_q + 1, out_q, rng, nsteps, use_weights, scatter_first);
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
    if (finalize) {
        // one pass: records -> the caller's arrays, abort word, hit count + compaction + per-channel arrays (k_finalize_hits);
        // a call that ended in the tail kernel has run it beside that kernel already (launch_tail)
        if (!finalized) {
            HIP_TRY(hipMemsetAsync(ctx->d_words, 0, 12, ctx->stream));
            const unsigned blocks = (unsigned)((nphotons + COPY_ITEMS * 256 - 1) / (COPY_ITEMS * 256));
            hipLaunchKernelGGL(k_finalize_hits, dim3(blocks), dim3(256), 0, ctx->stream, geom->view, pv, (const float4 *)ctx->final_use, ctx->final_epoch,
                               (uint64_t)nphotons, ho, ctx->d_words, 0u);
            HIP_TRY(hipGetLastError());
        }
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

static chroma_propagate_options default_options(int32_t max_steps, int32_t use_weights, int32_t scatter_first, int32_t time_kernels)
{
    chroma_propagate_options o; memset(&o, 0, sizeof o);
    o.max_steps = max_steps; o.use_weights = use_weights; o.scatter_first = scatter_first; o.time_kernels = time_kernels;
    o.walk = o.tail = o.counting = -1;
    return o;
}

int chroma_propagate(chroma_ctx *ctx, chroma_geometry *geom, const chroma_photon_arrays *photons, uint64_t nphotons,
                     uint32_t ncopies, chroma_rng rng, int32_t max_steps, int32_t use_weights, int32_t scatter_first,
                     int32_t time_kernels, chroma_propagate_stats *stats, int32_t *aborted)
{
    return propagate_impl(ctx, geom, photons, nphotons, ncopies, rng, default_options(max_steps, use_weights, scatter_first, time_kernels), stats, aborted, nullptr);
}

int chroma_propagate_hits(chroma_ctx *ctx, chroma_geometry *geom, const chroma_photon_arrays *photons, uint64_t nphotons,
                          uint32_t ncopies, chroma_rng rng, int32_t max_steps, int32_t use_weights, int32_t scatter_first,
                          int32_t time_kernels, chroma_propagate_stats *stats, int32_t *aborted, chroma_hits_request *hits)
{
    if (!hits) return set_error(CHROMA_ERR_INVALID, "bad argument");
    return propagate_impl(ctx, geom, photons, nphotons, ncopies, rng, default_options(max_steps, use_weights, scatter_first, time_kernels), stats, aborted, hits);
}

int chroma_propagate_opt(chroma_ctx *ctx, chroma_geometry *geom, const chroma_photon_arrays *photons, uint64_t nphotons,
                         uint32_t ncopies, chroma_rng rng, const chroma_propagate_options *options,
                         chroma_propagate_stats *stats, int32_t *aborted, chroma_hits_request *hits)
{
    if (!options) return set_error(CHROMA_ERR_INVALID, "bad argument");
    return propagate_impl(ctx, geom, photons, nphotons, ncopies, rng, *options, stats, aborted, hits);
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

int chroma_color_solids(chroma_ctx *ctx, chroma_geometry *geom, int32_t first_triangle, int32_t ntriangles, const uint8_t *d_solid_hit,
                        const uint32_t *d_solid_colors, uint32_t nsolids)
{
    if (!ctx || !geom || !d_solid_hit || !d_solid_colors) return set_error(CHROMA_ERR_INVALID, "bad argument");
    if (!geom->d_colors || !geom->view.solid_id_map) return set_error(CHROMA_ERR_INVALID, "geometry was created without colors / solid_id_map");
    if (first_triangle < 0 || ntriangles < 0 || (uint64_t)first_triangle + (uint64_t)ntriangles > (uint64_t)geom->ntriangles)
        return set_error(CHROMA_ERR_INVALID, "triangles %d .. %lld of %llu", first_triangle, (long long)first_triangle + ntriangles, (unsigned long long)geom->ntriangles);
    if (ntriangles == 0) return CHROMA_OK;
    hipLaunchKernelGGL(k_color_solids, dim3((unsigned)((ntriangles + 255) / 256)), dim3(256), 0, ctx->stream, (int)first_triangle, (int)ntriangles,
                       geom->view.solid_id_map, d_solid_hit, d_solid_colors, nsolids, (uint32_t *)geom->d_colors);
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
    if (!ctx || !d_x || !d_out || fn < 0 || fn > 4) return set_error(CHROMA_ERR_INVALID, "bad argument");
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
