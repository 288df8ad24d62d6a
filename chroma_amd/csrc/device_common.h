// device_common.h -- device-side types shared by the HIP kernels of libchroma_hip.so.
//
// Written for gfx950 (CDNA4) only: 64-wide wavefronts, one photon per lane.
// Arithmetic follows include/chroma_math.h (the numeric contract) and is compiled with
// -ffp-contract=off so that the CPU oracle reproduces every result bit for bit.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#define CM_FN __host__ __device__ static inline
#include "../../include/chroma_math.h"
#include "../../include/chroma_hip.h"

#define WAVE 64

// ---- float3 algebra (operation order as chroma/cuda/linalg.h) -------------------------
struct v3 { float x, y, z; };
__device__ inline v3 mk3(float x, float y, float z) { return v3{x, y, z}; }
__device__ inline v3 operator-(v3 a) { return mk3(-a.x, -a.y, -a.z); }
__device__ inline v3 operator+(v3 a, v3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ inline v3 operator-(v3 a, v3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ inline v3 operator*(v3 a, float c) { return mk3(a.x * c, a.y * c, a.z * c); }
__device__ inline v3 operator*(float c, v3 a) { return mk3(c * a.x, c * a.y, c * a.z); }
__device__ inline v3 operator/(v3 a, float c) { return mk3(a.x / c, a.y / c, a.z / c); }
__device__ inline v3 operator/(v3 a, v3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
__device__ inline v3 operator/(float c, v3 a) { return mk3(c / a.x, c / a.y, c / a.z); }
__device__ inline float dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ inline v3 cross(v3 a, v3 b)
{ return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
__device__ inline float norm(v3 a) { return cm_sqrtf(dot(a, a)); }
__device__ inline v3 normalize(v3 a) { return a / norm(a); }

__device__ inline v3 load3(const float *p, size_t i) { return mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ inline void store3(float *p, size_t i, v3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }

// ---- device view of an uploaded geometry ------------------------------------------------
// Layout in HBM (see DESIGN.md "Data layout"):
//   nodes   uint4[nnodes]            16 B packed node, as chroma/cuda/geometry_types.h:57-67
//   wnodes  uint4[nwide][8]          128 B wide node = one L2 line: eight child entries in the node
//                                    format above, w = child wide node | 0x80000000+device triangle
//                                    | 0xFFFFFFFF (empty); derived from `nodes` (csrc/wide_build.cpp)
//   tri     float4[nrecords][3]      48 B record: v0.xyz|material_code, v1.xyz|triangle id, v2.xyz|rank
//                                    (one aligned 48-B gather instead of the reference's
//                                     12-B index fetch + three 12-B vertex gathers).  `rank` is the
//                                    triangle's position in the reference's test order (tie-break of
//                                    the wide walk).  Records are stored in the order the wide tree
//                                    lists its triangles, not in triangle-id order: the triangles of
//                                    one wide node -- tested together -- sit in adjacent 128-B lines.
//                                    Both trees hold that "device" index in their leaves;
//                                    tri_to_dev / dev_to_tri translate at ray start and end.
//   tables  float[...]               optics tables, row-major [row][wavelength_n]
struct SurfaceInfo { uint32_t model; uint32_t transmissive; float thickness; int32_t dichroic_index; };

// float4s per triangle record: {v0, material code} {v1, triangle id} {v2, rank}.  4 pads a record to
// 64 bytes, so that none straddles two 128-byte lines (a quarter of the 48-byte ones do).
#ifndef TRI_STRIDE
#define TRI_STRIDE 3
#endif
struct GeoView {
    const uint4  *nodes;             // traversal copy: leaf child = device triangle index
    const uint4  *wnodes;            // derived 8-wide tree, 8 entries (128 B) per node
    const float4 *tri;               // [device triangle index][3]
    const uint32_t *tri_to_dev, *dev_to_tri;
    // materials
    const float *mat_refractive_index, *mat_absorption_length, *mat_scattering_length;
    const uint32_t *mat_num_comp, *mat_comp_offset;
    const float *comp_reemission_prob, *comp_reemission_wvl_cdf, *comp_absorption_length, *comp_reemission_time_cdf;
    // surfaces
    const float *surf_detect, *surf_absorb, *surf_reemit, *surf_reflect_diffuse, *surf_reflect_specular,
                *surf_eta, *surf_k, *surf_reemission_cdf;
    const SurfaceInfo *surf_info;
    const uint32_t *dichroic_nangles, *dichroic_offset;
    const float *dichroic_angles, *dichroic_reflect, *dichroic_transmit;
    // hits
    const uint32_t *solid_id_map;
    const int32_t  *solid_id_to_channel_index;
    float world_origin[3];
    float world_scale;
    float suspect_margin;            // see record_hit_is_regular (propagate_device.h)
    float slab_grow;                 // quanta by which the fast slab test grows a box on every side (ray_growth, propagate_device.h)
    uint32_t wavelength_n; float wavelength_start, wavelength_step;
    uint32_t time_n;       float time_start, time_step;
    uint32_t nnodes, ntriangles, nsolids, nchannels, nwide;
    uint32_t plain_optics;           // no re-emitting material component, every surface of the default model
};

struct PhotonView {   // device pointers of chroma_photon_arrays
    float *pos, *dir, *pol, *wavelengths, *t;
    uint32_t *flags; int32_t *last_hit_triangles; float *weights; uint32_t *evidx; uint32_t *rng_counters;
};

struct DeviceCounters {   // accumulated with one atomic per wave
    unsigned long long photon_steps, nodes_visited, triangles_tested, stack_overflows;
    unsigned long long stack_spills;   // stack entries a fast walk pushed beyond its LDS part (counting builds)
    unsigned long long packet_rays, packet_nodes, packet_tris;      // k_raycast_packet's share of the two counts above (counting builds)
};

// ---- hits ---------------------------------------------------------------------------------------------------------
// the channel a photon was detected on, or -1 (propagate.cu:157-171: the flag, a last hit triangle, a solid with a channel)
__device__ inline int hit_channel(const GeoView &g, uint32_t history, int triangle_id, uint32_t detection_state)
{
    if (!(history & detection_state)) return -1;
    if (triangle_id <= -1) return -1;
    uint32_t solid_id = g.solid_id_map[triangle_id];
    return g.solid_id_to_channel_index[solid_id];
}

// where the hits of a chroma_propagate_hits call go (k_finalize_hits, and k_tail_coop for the photons it finishes)
struct HitsOut {
    PhotonView dst; int32_t *channels; uint32_t capacity;
    uint32_t *hit_count, *earliest;
    uint32_t detection_state; int want;
};

// ---- wave-level helpers --------------------------------------------------------------------
__device__ inline unsigned lane_id() { return __lane_id(); }

// Append `value` for every lane with pred set: one atomic per wave, lanes keep their order.
// queue[0] is the tail index (initially 1), as chroma/cuda/propagate.cu:315-318.
__device__ inline void wave_queue_append(uint32_t *queue, bool pred, uint32_t value)
{
    unsigned long long mask = __ballot(pred);
    if (mask == 0ull) return;
    unsigned lane = lane_id();
    unsigned leader = (unsigned)__ffsll((long long)mask) - 1u;
    uint32_t base = 0;
    if (lane == leader) base = atomicAdd(queue, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    if (pred) {
        unsigned rank = (unsigned)__popcll(mask & ((1ull << lane) - 1ull));
        queue[base + rank] = value;
    }
}

// Block-level variant: one atomic per BLOCK (a hot word takes only ~88 atomics/us, so one per
// wave is too many for 1e8 photons).  Waves publish their counts in LDS, wave 0 reserves the
// block's span, every lane then writes at its own offset.  All threads of the block must call it.
template <int MAX_WAVES>
__device__ inline uint32_t block_queue_append(uint32_t *queue, bool pred, uint32_t value, uint32_t *s_counts /* [MAX_WAVES+1] */)
{
    unsigned long long mask = __ballot(pred);
    unsigned lane = lane_id();
    unsigned wave = threadIdx.x / WAVE, nwaves = (blockDim.x + WAVE - 1) / WAVE;
    if (lane == 0) s_counts[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t total = 0;
        for (unsigned w = 0; w < nwaves; w++) { uint32_t c = s_counts[w]; s_counts[w] = total; total += c; }
        s_counts[MAX_WAVES] = total ? atomicAdd(queue, total) : 0u;
    }
    __syncthreads();
    uint32_t at = 0;                       // position in the queue (>= 1) of this lane's entry, 0 if none
    if (pred) {
        at = s_counts[MAX_WAVES] + s_counts[wave] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        queue[at] = value;
    }
    return at;
}

__device__ inline unsigned long long wave_sum_u64(unsigned long long v)
{
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
    return v;
}
