// kernel_raycast_retry.h -- k_raycast_retry: the strict lane-per-ray loop (intersect_mesh_strict) for the rays the fast walks hand over; ALL = every ray (LITERAL_LANE).
// One of the kernel families of libchroma_hip.so; included by chroma_hip.hip (one translation unit: the families share
// device helpers and launch-time constants, and are included in dependency order).
#pragma once

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
