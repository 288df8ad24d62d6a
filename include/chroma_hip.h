/* chroma_hip.h -- C ABI of libchroma_hip.so, the MI355X (gfx950) photon-propagation engine.
 *
 * This is the drop-in boundary for chroma's propagate path.  In the reference the
 * boundary is PyCUDA: Python JIT-compiles the .cu files under chroma/cuda and launches kernels by
 * name (chroma/gpu/tools.py:14-54).  Every entry point below names the reference
 * kernel or host routine it replaces (paths relative to the reference root).
 *
 * Conventions
 *   - plain C: opaque handles, raw device/host pointers, sizes; no C++ or torch types.
 *   - every function returns 0 (CHROMA_OK) or a negative chroma error / positive
 *     hipError_t; chroma_last_error() returns a thread-local message.
 *   - "d_" arguments are device pointers obtained from chroma_malloc(); all other
 *     pointers are host pointers.
 *   - all work is issued on the context's stream; entry points that return values
 *     to the host synchronise that stream, the others are asynchronous.
 *   - photon arrays are structure-of-arrays exactly as chroma/cuda/propagate.cu:217-226
 *     takes them: float3 arrays are packed [n][3] floats (12-byte stride).
 */
#ifndef CHROMA_HIP_H
#define CHROMA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CHROMA_OK                 0
#define CHROMA_ERR_INVALID       -1   /* bad argument / shape mismatch             */
#define CHROMA_ERR_NO_DEVICE     -2   /* no usable HIP device                      */
#define CHROMA_ERR_STACK         -3   /* BVH needs a deeper traversal stack        */
#define CHROMA_ERR_INTERNAL      -4

/* History bits: chroma/cuda/photon.h:49-64 == chroma/event.py:5-17 */
#define CHROMA_NO_HIT            (1u << 0)
#define CHROMA_BULK_ABSORB       (1u << 1)
#define CHROMA_SURFACE_DETECT    (1u << 2)
#define CHROMA_SURFACE_ABSORB    (1u << 3)
#define CHROMA_RAYLEIGH_SCATTER  (1u << 4)
#define CHROMA_REFLECT_DIFFUSE   (1u << 5)
#define CHROMA_REFLECT_SPECULAR  (1u << 6)
#define CHROMA_SURFACE_REEMIT    (1u << 7)
#define CHROMA_SURFACE_TRANSMIT  (1u << 8)
#define CHROMA_BULK_REEMIT       (1u << 9)
#define CHROMA_CHERENKOV         (1u << 10)
#define CHROMA_SCINTILLATION     (1u << 11)
#define CHROMA_NAN_ABORT         (1u << 31)
/* chroma/cuda/propagate.cu:258,315 */
#define CHROMA_TERMINAL_MASK (CHROMA_NO_HIT | CHROMA_BULK_ABSORB | CHROMA_SURFACE_DETECT | \
                              CHROMA_SURFACE_ABSORB | CHROMA_NAN_ABORT)

/* surface models: chroma/cuda/geometry_types.h:22 */
#define CHROMA_SURFACE_DEFAULT  0
#define CHROMA_SURFACE_COMPLEX  1
#define CHROMA_SURFACE_WLS      2
#define CHROMA_SURFACE_DICHROIC 3

/* packed BVH node: chroma/cuda/geometry_types.h:57-59 */
#define CHROMA_CHILD_BITS  28
#define CHROMA_NCHILD_MASK 0xF0000000u

typedef struct chroma_ctx chroma_ctx;
typedef struct chroma_geometry chroma_geometry;

/* Flat, pointer-free description of a flattened geometry + BVH + optics tables.
 * Replaces the pointer-linked Material/Surface/DichroicProps/Geometry/Detector
 * structs assembled byte-by-byte by chroma/gpu/geometry.py:47-253 and
 * chroma/gpu/detector.py:17-40 (struct layouts chroma/cuda/geometry_types.h:4-82,
 * chroma/cuda/detector.h:4-22).  All pointers are HOST pointers; the library
 * copies what it needs.  Tables are row-major [index][wavelength_n]. */
typedef struct chroma_geometry_desc {
    /* mesh (chroma/gpu/geometry.py:191-209) */
    const float    *vertices;        /* [nvertices][3]                               */
    const uint32_t *triangles;       /* [ntriangles][3] vertex indices               */
    const uint32_t *material_codes;  /* [ntriangles] inner<<24|outer<<16|surface<<8  */
    const uint32_t *solid_id_map;    /* [ntriangles]                                 */
    const uint32_t *colors;          /* [ntriangles] or NULL                         */
    uint32_t nvertices, ntriangles;
    /* BVH (chroma/bvh/bvh.py, chroma/cuda/geometry.h:31-47) */
    const uint32_t *nodes;           /* [nnodes][4] packed x,y,z,w                   */
    uint32_t nnodes;
    float world_origin[3];
    float world_scale;
    /* common grids (chroma/gpu/geometry.py:15-30) */
    uint32_t wavelength_n; float wavelength_start, wavelength_step;
    uint32_t time_n;       float time_start, time_step;
    /* materials (chroma/gpu/geometry.py:47-103) */
    uint32_t nmaterials;
    const float    *mat_refractive_index;   /* [nmaterials][wavelength_n] */
    const float    *mat_absorption_length;
    const float    *mat_scattering_length;
    const uint32_t *mat_num_comp;           /* [nmaterials]                          */
    const uint32_t *mat_comp_offset;        /* [nmaterials] first row in comp tables */
    uint32_t ncomp_total;
    const float    *comp_reemission_prob;     /* [ncomp_total][wavelength_n] */
    const float    *comp_reemission_wvl_cdf;  /* [ncomp_total][wavelength_n] */
    const float    *comp_absorption_length;   /* [ncomp_total][wavelength_n] */
    const float    *comp_reemission_time_cdf; /* [ncomp_total][time_n]       */
    /* surfaces (chroma/gpu/geometry.py:108-189); a None surface keeps its slot, zero-filled */
    uint32_t nsurfaces;
    const float    *surf_detect;            /* [nsurfaces][wavelength_n] each */
    const float    *surf_absorb;
    const float    *surf_reemit;
    const float    *surf_reflect_diffuse;
    const float    *surf_reflect_specular;
    const float    *surf_eta;
    const float    *surf_k;
    const float    *surf_reemission_cdf;
    const uint32_t *surf_model;             /* [nsurfaces] */
    const uint32_t *surf_transmissive;
    const float    *surf_thickness;
    const int32_t  *surf_dichroic_index;    /* [nsurfaces] index into dichroic tables or -1 */
    uint32_t ndichroic;
    const uint32_t *dichroic_nangles;       /* [ndichroic]                 */
    const uint32_t *dichroic_offset;        /* [ndichroic] first angle row */
    uint32_t ndichroic_angles_total;
    const float    *dichroic_angles;        /* [ndichroic_angles_total]                */
    const float    *dichroic_reflect;       /* [ndichroic_angles_total][wavelength_n]  */
    const float    *dichroic_transmit;      /* [ndichroic_angles_total][wavelength_n]  */
    /* detector part (chroma/gpu/detector.py:17-40); nsolids == 0 for a plain Geometry */
    const int32_t  *solid_id_to_channel_index; /* [nsolids] */
    uint32_t nsolids, nchannels;
    /* OPTIONAL: the derived traversal tree of `nodes`, as chroma_wide_build returned it for exactly
     * these nodes (a cache, or one process of a node building it for the others).  wide_nodes == NULL:
     * chroma_geometry_create derives it (16 s of all-core work at 170 M triangles).  A supplied tree goes
     * through the same index checks (chroma_wide_validate) before anything is uploaded. */
    const uint32_t *wide_nodes;             /* [nwide][8][4] */
    const uint32_t *wide_tri_to_record;     /* [ntriangles]  */
    const uint32_t *wide_record_to_tri;     /* [nrecords]    */
    const uint32_t *wide_rank;              /* [ntriangles]  */
    uint64_t nwide, nrecords;
} chroma_geometry_desc;

/* The nine photon arrays of chroma/cuda/propagate.cu:220-225 plus the per-photon
 * Philox draw counter that replaces curandState (propagate.cu:219,241,303). */
typedef struct chroma_photon_arrays {
    float    *pos;                 /* [n][3] mm          */
    float    *dir;                 /* [n][3]             */
    float    *pol;                 /* [n][3]             */
    float    *wavelengths;         /* [n] nm             */
    float    *t;                   /* [n] ns             */
    uint32_t *flags;               /* [n] history bits   */
    int32_t  *last_hit_triangles;  /* [n]                */
    float    *weights;             /* [n]                */
    uint32_t *evidx;               /* [n]                */
    uint32_t *rng_counters;        /* [n] uniforms already drawn by each photon */
} chroma_photon_arrays;

/* Replaces the curandState array from get_rng_states (chroma/gpu/tools.py:75-84):
 * the stream of photon i is Philox4x32-10 keyed by `seed`, indexed by
 * `photon_id_base + i` (see include/chroma_math.h). */
typedef struct chroma_rng {
    uint64_t seed;
    uint64_t photon_id_base;
} chroma_rng;

typedef struct chroma_propagate_stats {
    uint64_t photon_steps;       /* loop iterations that ran a ray cast (propagate.cu:264-275)  */
    uint64_t nodes_visited;      /* get_node calls in the child loop (cuda/mesh.h:75-76)        */
    uint64_t triangles_tested;   /* intersect_triangle calls (cuda/mesh.h:84-86)                */
    uint64_t launches;           /* propagate kernel launches                                   */
    uint64_t stack_overflows;    /* rays whose traversal stack overflowed (must be 0)           */
    double   kernel_ms;          /* HIP-event time of the propagate kernel launches, if timed   */
    double   raycast_ms;         /* of which: the ray-cast kernel (k_raycast_persistent)         */
    uint64_t raycast_launches;   /* number of ray-cast launches timed in raycast_ms              */
    uint64_t stack_spills;       /* stack entries pushed beyond the LDS part of a fast walk's stack
                                    (counting mode only; the deep-stack path of the 4- and 8-lane walks) */
    double   physics_ms;         /* HIP-event time of the main k_physics pass of every timed step            */
    uint64_t physics_launches;
    double   packet_ms;          /* HIP-event time of k_raycast_packet (the first step of a call with coherent photons);
                                    raycast_ms / raycast_launches then are k_raycast_quad's alone               */
    uint64_t packet_launches;
    uint64_t packet_rays, packet_nodes_visited, packet_triangles_tested;   /* counting mode: the packet kernel's share */
    uint64_t reordered;          /* photons of calls that took them up in direction order (chroma_set_autosort) */
} chroma_propagate_stats;

const char *chroma_last_error(void);
const char *chroma_version(void);

/* ---- context: replaces create_cuda_context (chroma/gpu/tools.py:121-142) ---- */
int chroma_device_count(int *count);
int chroma_init(int device, chroma_ctx **ctx);
int chroma_shutdown(chroma_ctx *ctx);
int chroma_synchronize(chroma_ctx *ctx);
int chroma_mem_info(chroma_ctx *ctx, size_t *free_bytes, size_t *total_bytes);
int chroma_device_name(chroma_ctx *ctx, char *buf, size_t buflen);

/* ---- device memory: replaces pycuda.gpuarray allocation / get / set ---- */
int chroma_malloc(chroma_ctx *ctx, size_t nbytes, void **d_ptr);
int chroma_free(chroma_ctx *ctx, void *d_ptr);
/* (chroma_free parks the block in a pool and chroma_malloc reuses a parked block of the same size once the work that
 *  was queued when it was freed has completed: a GPUPhotons per event batch costs no hipMalloc / hipFree after the
 *  first.  chroma_pool_trim gives everything parked back to the device; chroma_pool_stats: bytes parked, allocations
 *  served from the pool, allocations that went to hipMalloc.) */
int chroma_pool_trim(chroma_ctx *ctx);
int chroma_pool_stats(chroma_ctx *ctx, uint64_t *parked_bytes, uint64_t *reused, uint64_t *allocated);
/* (copies of 8 MB and more are staged through pinned buffers by all host threads, piece by piece, DMA overlapping the
 *  staging of the next piece.)  Synchronous: ordered after the work queued on the context's stream. */
int chroma_memcpy_htod(chroma_ctx *ctx, void *d_dst, const void *h_src, size_t nbytes);
/* The same on the context's second stream -- NOT ordered with the queued work, so that the next event batch can be
 * uploaded (by another host thread) while the current one propagates: replaces the copies of GPUPhotons.__init__
 * (chroma/gpu/photon.py:13-94) in Simulation's batch loop (chroma/sim.py:58-139).  d_dst must not be in use by queued
 * work (a block fresh from chroma_malloc is not).  Returns when the data is on the device. */
int chroma_upload(chroma_ctx *ctx, void *d_dst, const void *h_src, size_t nbytes);
int chroma_memcpy_dtoh(chroma_ctx *ctx, void *h_dst, const void *d_src, size_t nbytes);
int chroma_memcpy_dtod(chroma_ctx *ctx, void *d_dst, const void *d_src, size_t nbytes);
int chroma_memset32(chroma_ctx *ctx, void *d_dst, uint32_t value, size_t count);

/* ---- geometry: replaces GPUGeometry.__init__ / GPUDetector.__init__ ---- */
int chroma_geometry_create(chroma_ctx *ctx, const chroma_geometry_desc *desc, chroma_geometry **geom);
int chroma_geometry_destroy(chroma_geometry *geom);
/* device pointers of the uploaded arrays, for the GPUGeometry attributes
 * (vertices, triangles, nodes, material_codes, colors, solid_id_map; gpu/geometry.py:191-209) */
int chroma_geometry_device_ptr(chroma_geometry *geom, const char *name, void **d_ptr, size_t *nbytes);
/* worst-case traversal stack entries this BVH can need (host-side tree walk) */
int chroma_geometry_stack_need(chroma_geometry *geom, uint32_t *entries);

/* ---- kernel-level entry points (one per reference kernel) ---- */

/* `propagate` (chroma/cuda/propagate.cu:217-319): up to max_steps steps for the photons
 * input_queue[first_photon .. first_photon+nthreads); survivors are appended to
 * d_output_queue (slot 0 = tail index, initial value 1, propagate.cu:315-318).
 * Asynchronous.  `stats` may be NULL; if given it is ACCUMULATED into on the next
 * synchronising call (chroma_propagate_stats_read). */
int chroma_propagate_step(chroma_ctx *ctx, chroma_geometry *geom,
                          int32_t first_photon, int32_t nthreads,
                          const uint32_t *d_input_queue, uint32_t *d_output_queue,
                          chroma_rng rng, const chroma_photon_arrays *photons,
                          int32_t max_steps, int32_t use_weights, int32_t scatter_first);

/* `photon_duplicate` (chroma/cuda/propagate.cu:13-52) */
int chroma_photon_duplicate(chroma_ctx *ctx, int32_t first_photon, int32_t nthreads,
                            const chroma_photon_arrays *photons, int32_t copies, int32_t stride);

/* `count_photons` + `copy_photons` (chroma/cuda/propagate.cu:54-114) */
int chroma_count_photons(chroma_ctx *ctx, int32_t first_photon, int32_t nthreads,
                         uint32_t target_flag, const uint32_t *d_flags, uint32_t *count);
int chroma_copy_photons(chroma_ctx *ctx, int32_t first_photon, int32_t nthreads,
                        uint32_t target_flag, const chroma_photon_arrays *src,
                        const chroma_photon_arrays *dst, uint32_t *ncopied);

/* `copy_photon_queue` (chroma/cuda/propagate.cu:116-144) */
int chroma_copy_photon_queue(chroma_ctx *ctx, int32_t first_photon, int32_t nthreads,
                             const uint32_t *d_queue, const chroma_photon_arrays *src,
                             const chroma_photon_arrays *dst);

/* `count_photon_hits` + `copy_photon_hits` (chroma/cuda/propagate.cu:147-214) */
int chroma_count_photon_hits(chroma_ctx *ctx, chroma_geometry *geom, int32_t first_photon,
                             int32_t nphotons, uint32_t detection_state,
                             const chroma_photon_arrays *photons, uint32_t *count);
int chroma_copy_photon_hits(chroma_ctx *ctx, chroma_geometry *geom, int32_t first_photon,
                            int32_t nphotons, uint32_t detection_state,
                            const chroma_photon_arrays *src, const chroma_photon_arrays *dst,
                            int32_t *d_channels, uint32_t *ncopied);

/* `distance_to_mesh` (chroma/cuda/mesh.h:124-151); d_triangle may be NULL.
 * Rays that miss leave d_distance untouched, as in the reference. */
int chroma_distance_to_mesh(chroma_ctx *ctx, chroma_geometry *geom, int32_t nthreads,
                            const float *d_origin, const float *d_direction,
                            float *d_distance, int32_t *d_triangle);
/* The device function `intersect_mesh` (chroma/cuda/mesh.h:42-118) over a ray bundle, with its
 * `last_hit_triangle` argument: ray i never hits triangle d_last_hit[i] (mesh.h:82; NULL or -1: no
 * exclusion).  chroma_distance_to_mesh is this call with d_last_hit = NULL.  Same fast walk, check and
 * literal-walk fallback as a propagation step. */
int chroma_intersect_mesh(chroma_ctx *ctx, chroma_geometry *geom, int32_t nthreads,
                          const float *d_origin, const float *d_direction, const int32_t *d_last_hit,
                          float *d_distance, int32_t *d_triangle);

/* ---- fused host loops ---- */

/* GPUPhotons.propagate (chroma/gpu/photon.py:193-259) for n photons, whole loop on the
 * device side: queue ping-pong, survivor compaction, no per-step host copy of the photon
 * arrays.  ncopies/true_nphotons give the interleaved initial queue order of
 * photon.py:209-212.  Synchronises.  `stats` may be NULL.  `aborted` (may be NULL)
 * receives 1 if any photon carries NAN_ABORT afterwards (photon.py:254-255). */
int chroma_propagate(chroma_ctx *ctx, chroma_geometry *geom, const chroma_photon_arrays *photons,
                     uint64_t nphotons, uint32_t ncopies, chroma_rng rng,
                     int32_t max_steps, int32_t use_weights, int32_t scatter_first,
                     int32_t time_kernels, chroma_propagate_stats *stats, int32_t *aborted);

/* propagate_hit in ONE call: chroma_propagate followed -- in the same pass that finishes the propagation -- by what the
 * reference does afterwards in separate kernels: the abort-flag reduction (chroma/gpu/photon.py:254-255), `count_photon_hits`
 * + `copy_photon_hits` (chroma/cuda/propagate.cu:147-214, GPUPhotons.get_flat_hits, chroma/gpu/photon.py:96-175) and the
 * per-channel hit count / earliest time of chroma_channel_hits.  Photons that end during the call travel as one 64-byte
 * record each and a single streaming kernel fills the caller's arrays from them while it extracts the hits; the arrays end up
 * exactly as chroma_propagate leaves them, and the flat hits are the SET chroma_copy_photon_hits would copy afterwards (their order
 * is unspecified, as in the reference: blocks of 4096 photons land in the order their atomics do) -- tests/test_gpu_hits.py.
 *   dst / d_channels   device arrays for up to `capacity` flat hits and their channels (both or neither)
 *   d_hit_count, d_earliest_time_bits   per-channel arrays (device, nchannels), ACCUMULATED into as by chroma_channel_hits;
 *                      d_hit_count may be NULL (then neither is touched), d_earliest_time_bits may be NULL
 *   nhits (out)        detected photons that belong to a channel; if it exceeds `capacity` only `capacity` of them were
 *                      written -- the photon arrays are final, chroma_copy_photon_hits with larger buffers gets all */
typedef struct chroma_hits_request {
    uint32_t detection_state;
    uint32_t capacity;
    const chroma_photon_arrays *dst;
    int32_t *d_channels;
    uint32_t *d_hit_count;
    uint32_t *d_earliest_time_bits;
    uint32_t nhits;
} chroma_hits_request;
int chroma_propagate_hits(chroma_ctx *ctx, chroma_geometry *geom, const chroma_photon_arrays *photons,
                          uint64_t nphotons, uint32_t ncopies, chroma_rng rng,
                          int32_t max_steps, int32_t use_weights, int32_t scatter_first,
                          int32_t time_kernels, chroma_propagate_stats *stats, int32_t *aborted,
                          chroma_hits_request *hits);

/* The general form of the two calls above: what a call does is an ARGUMENT, not a setting of the context -- which walk the ray
 * cast takes (GPUPhotons.propagate(exact=True) passes CHROMA_WALK_LITERAL here), how the tail runs, whether the kernels count
 * their work.  A field of -1 takes the context's setting (chroma_set_walk / chroma_set_tail / chroma_set_counting or the CHROMA_*
 * environment) as it is when the call starts; nothing a concurrent call or setter does changes a call under way.  Calls on one
 * context run one at a time (the context's queues and working sets are its own): two threads may share a handle.
 * `hits` may be NULL (then this is chroma_propagate).  Zero the structure, then set what you need: reserved fields must be 0. */
typedef struct chroma_propagate_options {
    int32_t max_steps, use_weights, scatter_first, time_kernels;      /* as the arguments of chroma_propagate */
    int32_t walk;          /* CHROMA_WALK_* or -1 */
    int32_t tail;          /* CHROMA_TAIL_* or -1 */
    int32_t counting;      /* 0, 1 or -1 */
    int32_t reserved[5];
} chroma_propagate_options;
int chroma_propagate_opt(chroma_ctx *ctx, chroma_geometry *geom, const chroma_photon_arrays *photons,
                         uint64_t nphotons, uint32_t ncopies, chroma_rng rng,
                         const chroma_propagate_options *options,
                         chroma_propagate_stats *stats, int32_t *aborted, chroma_hits_request *hits);

/* Per-channel reduction of detected photons: hit count and earliest hit time
 * (float bits, valid for t >= 0 as in chroma/cuda/daq.cu:5-20).  The arrays
 * (length nchannels, device) are ACCUMULATED into: zero / 0x7f800000-fill them first.
 * This is the quantity all-reduced across GPUs. */
int chroma_channel_hits(chroma_ctx *ctx, chroma_geometry *geom, uint64_t nphotons,
                        uint32_t detection_state, const chroma_photon_arrays *photons,
                        uint32_t *d_hit_count, uint32_t *d_earliest_time_bits);

/* ---- DAQ --------------------------------------------------------
 * `run_daq` (chroma/cuda/daq.cu:35-86) for photons [first_photon, first_photon + nphotons): a
 * photon with `detection_state` set whose last hit triangle belongs to a channel contributes, with
 * probability weight * global_weight, a hit time (photon time + a draw from the time CDF) and a
 * charge (a draw from the charge CDF, quantised to charge_unit) to its channel: atomicMin on the
 * time bits, atomicAdd on the integer charge, atomicOr on the history.  The three draws come from
 * the photon's Philox stream number 1 + acquisition (include/chroma_math.h), not from per-thread
 * curandStates.  The CDF tables are DEVICE arrays of cdf_len floats each (chroma/cuda/detector.h).
 * The channel arrays are accumulated into: reset them first (chroma_daq_reset). */
typedef struct chroma_daq_tables {
    const float *d_time_cdf_x, *d_time_cdf_y;      int32_t time_cdf_len;
    const float *d_charge_cdf_x, *d_charge_cdf_y;  int32_t charge_cdf_len;
    float charge_unit;
} chroma_daq_tables;
/* `reset_earliest_time_int` (daq.cu:25-33) + zeroing of charge and history */
int chroma_daq_reset(chroma_ctx *ctx, float maxtime, uint32_t nchannels, uint32_t *d_earliest_time_int,
                     uint32_t *d_channel_q_int, uint32_t *d_channel_histories);
int chroma_daq_acquire(chroma_ctx *ctx, chroma_geometry *geom, const chroma_daq_tables *tables,
                       int32_t first_photon, int32_t nphotons, uint32_t detection_state,
                       const chroma_photon_arrays *photons, chroma_rng rng, uint32_t acquisition,
                       float global_weight, uint32_t *d_earliest_time_int, uint32_t *d_channel_q_int,
                       uint32_t *d_channel_histories);
/* `run_daq_many` (chroma/cuda/daq.cu:88-150; GPUDaq(ndaq > 1), chroma/gpu/daq.py:85-99): `ndaq`
 * independent acquisitions of the same photons side by side -- copy i accumulates into channels
 * [i * channel_stride, (i + 1) * channel_stride) of arrays of ndaq * channel_stride entries -- each
 * adding a unit normal jitter to the hit time (the reference's curand_normal; here Box-Muller on the
 * photon's stream, include/chroma_math.h).  Copy i of a photon draws from words 8 i ... of Philox
 * stream 1 + acquisition of that photon. */
int chroma_daq_acquire_many(chroma_ctx *ctx, chroma_geometry *geom, const chroma_daq_tables *tables,
                            int32_t first_photon, int32_t nphotons, uint32_t detection_state,
                            const chroma_photon_arrays *photons, chroma_rng rng, uint32_t acquisition,
                            float global_weight, int32_t ndaq, int32_t channel_stride,
                            uint32_t *d_earliest_time_int, uint32_t *d_channel_q_int, uint32_t *d_channel_histories);
/* `convert_sortable_int_to_float` + `convert_charge_int_to_float` (daq.cu:152-173) */
int chroma_daq_convert(chroma_ctx *ctx, uint32_t nchannels, float charge_unit, const uint32_t *d_earliest_time_int,
                       const uint32_t *d_channel_q_int, float *d_earliest_time, float *d_channel_q);

/* Isotropic photon bomb generated on the device (the benchmark source of
 * chroma/benchmark.py:77-83, formulas chroma/sample.py:16-30), photon i drawn from
 * Philox stream (seed, 0xB0B0000000000000 + id_base + i).  wavelength_hi <= wavelength_lo
 * gives a mono-energetic bomb. */
int chroma_generate_bomb(chroma_ctx *ctx, const chroma_photon_arrays *photons, uint64_t nphotons,
                         uint64_t seed, uint64_t id_base, const float pos[3],
                         float wavelength_lo, float wavelength_hi);

/* tools.argsort_direction (chroma/tools.py:175-193) and the reordering it serves, for a photon set on the device: the
 * photons are put in the order of a 32-bit Morton code of (theta, phi) of their directions (stable), every array of
 * the set gathered accordingly.  The reference's own benchmark does this to its photons before it starts the clock
 * (chroma/benchmark.py:80-82); bench.py does the same to its bomb.  Slot i afterwards holds the photon of rank i. */
int chroma_photons_sort_direction(chroma_ctx *ctx, const chroma_photon_arrays *photons, uint64_t nphotons);

/* Flat hits (what chroma_copy_photon_hits / chroma_propagate_hits left in `hits` and `d_channels`) put in (evidx, channel) order, in
 * place and stably: the split by event and by channel that the reference's callers do next with one mask over all hits per event and
 * per channel (chroma/sim.py:118-123, chroma/gpu/photon.py:96-105) is then a matter of slices.  The reference leaves the order of
 * flat hits unspecified (its compaction goes through an atomic); this is one valid order. */
int chroma_hits_sort(chroma_ctx *ctx, const chroma_photon_arrays *hits, int32_t *d_channels, uint64_t nhits);

/* `render` (chroma/cuda/render.cu:37-181): every triangle along each ray, the `alpha_depth` nearest kept as
 * a per-ray list sorted by distance (d_dx [n][alpha_depth], d_color [n][alpha_depth][4], d_dxlen [n]: in and
 * out, so that a second call continues the first -- GPURays.render(keep_last_render=True)), composited
 * front to back over bg_color into d_pixels [n] (0xAARRGGBB).  Directions are used as given (not normalised). */
int chroma_render(chroma_ctx *ctx, chroma_geometry *geom, int32_t nthreads, const float *d_origin, const float *d_direction,
                  uint32_t alpha_depth, uint32_t *d_pixels, float *d_dx, uint32_t *d_dxlen, float *d_color, uint32_t bg_color);
/* `color_solids` (chroma/cuda/mesh.h:153-166; host: GPUGeometry.color_solids, chroma/gpu/geometry.py:283-298): triangle t of
 * [first_triangle, first_triangle + ntriangles) takes d_solid_colors[solid_id_map[t]] where d_solid_hit[solid_id_map[t]] is set
 * (one byte per solid, as numpy bool); both arrays hold `nsolids` entries, a triangle of a solid beyond them keeps its colour.
 * Writes the geometry's `colors` array, the one chroma_render reads. */
int chroma_color_solids(chroma_ctx *ctx, chroma_geometry *geom, int32_t first_triangle, int32_t ntriangles, const uint8_t *d_solid_hit,
                        const uint32_t *d_solid_colors, uint32_t nsolids);
/* `translate`, `rotate`, `rotate_around_point` (chroma/cuda/transform.cu:9-53) on a float3 array */
int chroma_points_translate(chroma_ctx *ctx, int32_t n, float *d_a, const float v[3]);
int chroma_points_rotate(chroma_ctx *ctx, int32_t n, float *d_a, float phi, const float axis[3]);
int chroma_points_rotate_around_point(chroma_ctx *ctx, int32_t n, float *d_a, float phi, const float axis[3], const float point[3]);

/* ---- the hit reduction across the GPUs of a node ------------------------------
 * The reference has no multi-GPU mode.  Here one process drives one GPU, photons are sharded by
 * global id, the geometry is replicated, and the ONE exchange per batch is the per-channel arrays:
 * RCCL over xGMI, on the library's stream, in place on the device arrays chroma_channel_hits /
 * chroma_daq_acquire filled.  Rendezvous: one rank calls chroma_comm_unique_id and hands the 128
 * bytes to the others by any means (bench.py: torch.distributed broadcast); every rank then calls
 * chroma_comm_init.  Without a communicator the reductions are the identity (single GPU). */
#define CHROMA_COMM_ID_BYTES 128
int chroma_comm_unique_id(uint8_t id[CHROMA_COMM_ID_BYTES]);
int chroma_comm_init(chroma_ctx *ctx, int32_t nranks, int32_t rank, const uint8_t id[CHROMA_COMM_ID_BYTES]);
int chroma_comm_destroy(chroma_ctx *ctx);
/* hit_count: sum; earliest-time bit patterns: min (valid for t >= 0, chroma/cuda/daq.cu:5-20); d_earliest may be NULL */
int chroma_allreduce_hits(chroma_ctx *ctx, uint32_t *d_hit_count, uint32_t *d_earliest_time_bits, uint32_t nchannels);
/* DAQ accumulators of sharded photons (chroma/cuda/daq.cu:73-75): time bits min, integer charge sum, histories OR */
int chroma_allreduce_daq(chroma_ctx *ctx, uint32_t *d_earliest_time_int, uint32_t *d_channel_q_int,
                         uint32_t *d_channel_histories, uint32_t nchannels);

/* Test probe: ONE call per element of a single device function of the propagate path, so that tests
 * can pin the engine's own device code on the CPU oracle and on the reference's headers compiled for
 * gfx950 (oracle/ref_headers_driver.hip).  All pointers are device pointers.
 *   fn 0  interp_property (chroma/cuda/geometry.h:64-75)   d_x[n], d_tab_f[ntab], grid (start, step)
 *   fn 1  interp_idx (chroma/cuda/interpolate.h:5-29)      d_x[n], d_tab_x[ntab]
 *   fn 2  interp (chroma/cuda/interpolate.h:32-57)         d_x[n], d_tab_x[ntab], d_tab_f[ntab]
 *   fn 3  rotate (chroma/cuda/rotate.h:22-28)              d_x[7 n] = a.xyz, phi, axis.xyz -> d_out[5 n] = r.xyz, cos phi, sin phi
 *   fn 4  the float3 algebra (chroma/cuda/linalg.h)        d_x[7 n] = a.xyz, b.xyz, c -> d_out[32 n] = -a, a+b, a-b, a*c, c*a, a/c, c/a,
 *                                                          cross(a,b), dot(a,b), norm(a), normalize(a), a/b */
int chroma_probe(chroma_ctx *ctx, int32_t fn, uint64_t n, const float *d_x, const float *d_tab_x, const float *d_tab_f,
                 uint32_t ntab, float start, float step, float *d_out);

/* ---- host-side BVH construction -------------------------------------------------------
 * Recursive-grid builder: replaces make_recursive_grid_bvh (chroma/bvh/grid.py:11-95) and the
 * kernels it drives -- make_leaves, make_parents_detailed, copy_and_offset, collapse_child
 * (chroma/cuda/bvh.cu:149-203,270-308,365-384,530-543; hosts chroma/gpu/bvh.py:18-130,239-267).
 * Runs on the host cores (no device needed).  Two-phase: build returns a handle and the sizes,
 * fetch copies nodes ([nnodes][4] uint32) and layer bounds ([nlayers+1] uint64), free releases. */
int chroma_bvh_build(const float *vertices, uint32_t nvertices, const uint32_t *triangles, uint32_t ntriangles,
                     const float world_origin[3], float world_scale, int32_t target_degree,
                     void **handle, uint64_t *nnodes, uint32_t *nlayers);
/* The same builder ON THE DEVICE of `ctx` (csrc/bvh_device.hip), as in the reference, where these steps ARE device
 * kernels: leaf boxes + 48-bit Morton codes (bvh.cu:149-203), a device radix sort (grid.py:26-28), group boundaries by
 * one histogram pass and two scans (grid.py:37-76), parent unions per layer (bvh.cu:270-308), concatenate + offset
 * (bvh.cu:365-384), collapse (bvh.cu:530-543).  Host arrays in, the same kind of handle out, and the SAME node array
 * bit for bit as chroma_bvh_build (tests/test_gpu_bvh.py). */
int chroma_bvh_build_device(chroma_ctx *ctx, const float *vertices, uint32_t nvertices, const uint32_t *triangles,
                            uint32_t ntriangles, const float world_origin[3], float world_scale, int32_t target_degree,
                            void **handle, uint64_t *nnodes, uint32_t *nlayers);
int chroma_bvh_fetch(void *handle, uint32_t *nodes_out, uint64_t *layer_bounds_out);
/* zero-copy access to the arrays owned by the handle (valid until chroma_bvh_free) */
int chroma_bvh_data(void *handle, const uint32_t **nodes, const uint64_t **layer_bounds);
int chroma_bvh_free(void *handle);

/* ---- derived traversal tree ------------------------------------------------------------
 * chroma_geometry_create derives, from the reference-format nodes it is given, the tree the ray
 * cast actually walks on MI355X: 8-wide nodes of 128 bytes (one L2 line) whose entries are boxes
 * of the reference tree (chroma/cuda/geometry_types.h:57-67 format, w = child wide node, or
 * 0x80000000 | record index of a triangle, or 0xFFFFFFFF for an empty slot), the order of the
 * 48-byte triangle records, and each triangle's position in the test order of the reference
 * walk (chroma/cuda/mesh.h:68-110), which breaks exact distance ties the way the reference does.
 * These entry points expose that host-side step on its own (no device needed) for inspection
 * and tests; arrays stay owned by the handle until chroma_wide_free.
 *   wnodes      [nwide][8][4] uint32      tri_to_record [ntriangles] uint32
 *   record_to_tri [nrecords] uint32       rank [ntriangles] uint32 (0xFFFFFFFF: under no leaf) */
int chroma_wide_build(const uint32_t *nodes, uint64_t nnodes, uint32_t ntriangles, void **handle,
                      uint64_t *nwide, uint64_t *nrecords, uint32_t *depth);
/* The same step ON THE DEVICE of `ctx` (csrc/wide_device.hip): HIP kernels over one level of the tree at a time -- the
 * place where the reference builds its own tree (chroma/gpu/bvh.py, chroma/cuda/bvh.cu run on the GPU).  The result is the
 * tree CHROMA_TREE=levels makes on the host, bit for bit (tests/test_gpu_wide.py); same handle, same accessors. */
int chroma_wide_build_device(chroma_ctx *ctx, const uint32_t *nodes, uint64_t nnodes, uint32_t ntriangles, void **handle,
                             uint64_t *nwide, uint64_t *nrecords, uint32_t *depth);
int chroma_wide_data(void *handle, const uint32_t **wnodes, const uint32_t **tri_to_record,
                     const uint32_t **record_to_tri, const uint32_t **rank);
int chroma_wide_free(void *handle);
/* Index checks chroma_geometry_create runs on the derived tree before it uploads anything (every
 * inner child word names a later wide node, every leaf word a record < nrecords, every record a
 * triangle < ntriangles, every triangle a record that names it back): CHROMA_OK or CHROMA_ERR_INVALID.
 * The device buffers are sized from exactly these counts, so a tree that passes cannot index past them. */
int chroma_wide_validate(const uint32_t *wnodes, uint64_t nwide, const uint32_t *tri_to_record, uint32_t ntriangles,
                         const uint32_t *record_to_tri, uint64_t nrecords);

/* Merge identical vertices of a flattened mesh: replaces Mesh.remove_duplicate_vertices
 * (chroma/geometry.py:58-67, np.unique on a structured view) for large meshes.  The survivors
 * are written to `unique_out` ([nvertices][3] capacity) in lexicographic (x, y, z) order and
 * `triangle_indices` ([nindices], may be NULL) is remapped in place.  Host-side, all cores. */
int chroma_dedupe_vertices(const float *vertices, uint64_t nvertices, uint32_t *triangle_indices,
                           uint64_t nindices, float *unique_out, uint64_t *nunique);

/* accumulated counters of chroma_propagate_step launches since the last read */
int chroma_propagate_stats_read(chroma_ctx *ctx, chroma_propagate_stats *stats);
/* enable (1) / disable (0) node/triangle visit counting in the propagate kernels */
int chroma_set_counting(chroma_ctx *ctx, int32_t enabled);

/* Which tree the one-step ray cast walks, and how.
 *
 * The fast walks (QUAD, the default; PAIR, COOP, WIDE) go through the derived 8-wide tree nearest-first.  They agree
 * with EACH OTHER on every ray, and with the reference (chroma/cuda/mesh.h:42-118) on every ray whose winning
 * triangle's Moeller-Trumbore hit lies inside that triangle's own leaf box -- i.e. every hit that is geometrically one.
 * They do NOT reproduce the reference on rays for which mesh.h:82-101 accepts a numerically erratic hit: a ray almost
 * inside a triangle's plane (determinant above the FLT_EPSILON cut of intersect.h:58 but tiny) can "hit" hundreds of
 * millimetres in FRONT of the triangle's box; the reference's depth-first order may meet that triangle before any
 * nearer one and keep the bogus distance, while a nearest-first walk has pruned the box.
 * Measured: 0 of 2.4e8 random bomb photons on the 10k- and 29k-PMT detectors, ~2e-6 of rays aimed exactly at mesh
 * vertices / edge midpoints / centroids (profiles/r02/parity_sweep_*.txt; tests/test_gpu_exact_walk.py checks the
 * invariant "a ray on which QUAD and LITERAL differ is one whose reference hit lies outside its leaf box").
 *
 * LITERAL is the reference's loop as it stands -- its tree, its child order, its float box arithmetic, every triangle
 * tested the moment its leaf box is entered -- for EVERY ray, one ray per lane: the one mode that returns the
 * reference's triangle on every ray, the erratic ones included; several times slower (bench.py reports its rate as
 * config.exact_walk_photons_per_s).  GPUPhotons.propagate(exact=True), Simulation(exact=True) and chroma-sim --exact
 * select it per call.  REFERENCE walks the reference's tree in the reference's order but postpones triangle tests by up
 * to 8 per lane (a superset of the reference's tests in the same order): a cross-check walk, equal to LITERAL except
 * where an erratic hit belongs to a triangle the reference had already pruned.  chroma_intersect_mesh /
 * chroma_distance_to_mesh run the literal loop under both REFERENCE and LITERAL.
 * Env CHROMA_WALK=literal|reference|wide|coop|quad|pair sets the initial mode of a context. */
#define CHROMA_WALK_REFERENCE 0
#define CHROMA_WALK_WIDE      1
#define CHROMA_WALK_COOP      2
#define CHROMA_WALK_QUAD      3   /* the wide tree with four lanes per ray, two child entries per lane */
#define CHROMA_WALK_PAIR      4   /* the wide tree with two lanes per ray, four child entries per lane  */
#define CHROMA_WALK_LITERAL   5   /* chroma/cuda/mesh.h:42-118 literally, for every ray: exact (k_raycast_literal: four lanes per ray) */
#define CHROMA_WALK_LITERAL_LANE 6 /* the same loop with one lane per ray and no refill (intersect_mesh_strict): LITERAL's cross-check */
int chroma_set_walk(chroma_ctx *ctx, int32_t mode);

/* The first step of a chroma_propagate call can go to k_raycast_packet: 64 rays per wavefront walk the wide tree as ONE
 * packet (one stack, scalar node fetches, every triangle tested by all lanes whose ray enters its box) -- the same
 * results as the default walk, lane by lane, whatever the rays; much faster when the photons of neighbouring slots are
 * coherent (a direction-sorted bomb as chroma/benchmark.py:80-82 prepares it, a Cherenkov cone), much slower when they
 * are not.  AUTO: k_load_working looks at the photons (same origin, within 50 mrad, per wave of 64) and the packet kernel
 * takes the step when three quarters of the waves are coherent.  OFF is the default: measured on the 29k-PMT detector the
 * packet kernel is no faster than the default walk even on perfectly coherent rays (31.3 against 29.5 ms per 1e8 rays,
 * profiles/r03/ab_packet_first_step.txt: the union of 64 paths costs what the shared bookkeeping saves), so it stays an
 * opt-in with its parity tests.  Env CHROMA_PACKET=off|on|auto. */
#define CHROMA_PACKET_OFF  0
#define CHROMA_PACKET_ON   1
#define CHROMA_PACKET_AUTO 2
int chroma_set_packet(chroma_ctx *ctx, int32_t mode);

/* The ORDER in which a large chroma_propagate call (>= 2^21 photons, ncopies 1, default walk) takes its photons up.  Nothing
 * in the result depends on it (streams are keyed by photon id, results stored by photon id), the time does: rays of
 * neighbouring slots that walk the same part of the tree make the first launches of a call a quarter faster -- which is why
 * the reference's benchmark sorts its photons with tools.argsort_direction before the clock starts (chroma/benchmark.py:80-82).
 * AUTO: a sample of the input decides -- one origin and directions all over the place (a bomb or a calibration
 * source in generation order) are ordered by a 16-bit direction cell on the device (one radix sort of indices; the photon
 * arrays stay as they are); photons that are coherent already, or that come from many places, are taken as they come.
 * ON: every large call.  OFF (default): never -- measured at 1e8 photons on the 29k-PMT detector the index sort and the
 * gather through it cost 50 ms to win 15 (profiles/r03/ab_autosort.txt), so ordering stays the caller's preparation
 * (chroma_photons_sort_direction, as the reference's benchmark does it before its clock starts).
 * Env CHROMA_AUTOSORT=off|on|auto.  stats.reordered counts the photons so taken. */
#define CHROMA_AUTOSORT_OFF  0
#define CHROMA_AUTOSORT_ON   1
#define CHROMA_AUTOSORT_AUTO 2
int chroma_set_autosort(chroma_ctx *ctx, int32_t mode);

/* How chroma_propagate finishes a batch and, with FUSED, how it runs it at all (same results; for
 * tests and benchmarks).  COOP: per-step launch sets, then ONE cooperative launch for all remaining
 * steps once fewer than 8192 photons are alive.  SPLIT: per-step launch sets to the end.  FUSED: the
 * reference's own shape (chroma/gpu/photon.py:225-252) -- the lane-per-photon kernel of
 * chroma_propagate_step, one step per launch, then all remaining steps in one.
 * Env CHROMA_TAIL=coop|split|fused sets the initial mode of a context. */
#define CHROMA_TAIL_COOP  0
#define CHROMA_TAIL_SPLIT 1
#define CHROMA_TAIL_FUSED 2
int chroma_set_tail(chroma_ctx *ctx, int32_t mode);

#ifdef __cplusplus
}
#endif
#endif /* CHROMA_HIP_H */
