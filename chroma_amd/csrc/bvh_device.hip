// bvh_device.hip -- the recursive-grid BVH builder ON THE DEVICE (gfx950), and the direction sort of a photon set.
//
// Replaces make_recursive_grid_bvh (chroma/bvh/grid.py:11-95) with the kernels it drives in the reference --
// make_leaves (chroma/cuda/bvh.cu:149-203), the Morton sort of the leaves (grid.py:26-28), make_parents_detailed
// (bvh.cu:270-308) layer by layer, copy_and_offset (bvh.cu:365-384; concatenate_layers, chroma/gpu/bvh.py:239-267) and
// collapse_child (bvh.cu:530-543; collapse_chains, gpu/bvh.py:114-130).  The reference keeps the node array on the
// device and loops over the layers on the host; so does this file.  What is different from the reference's kernels:
//   * the group boundaries of a layer (grid.py:37-76: np.unique on the shifted codes, a Python loop over over-long
//     runs) are found on the device: ONE pass over the sorted codes histograms the highest differing bit of every
//     neighbouring pair, which gives the number of distinct codes for every shift at once (the reference sorts and
//     uniques once per trial shift); run starts, the cuts of runs longer than 15 and the parents' first children come
//     from two device scans;
//   * everything after the float32 quantisation is integer arithmetic, and the quantisation is the same three IEEE
//     operations as in the host builder (csrc/bvh_build.cpp; compiled with -ffp-contract=off and correctly rounded
//     division), so the node array is BIT-IDENTICAL to chroma_bvh_build's and to the NumPy restatement
//     (chroma_amd/bvh/grid.py): tests/test_gpu_bvh.py compares them.
// The sort is rocPRIM's radix sort through hipCUB (a plain library sort; stable, so equal codes keep triangle order).
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <vector>

#include "../../include/chroma_hip.h"
#include "bvh_result.h"
#include "ctx_access.h"
#include "device_common.h"

using chroma_host::Node;
using chroma_host::BvhResult;

namespace {

#define DEV_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            chroma_internal_set_error((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                                               \
        }                                                                                                 \
    } while (0)

const int MAX_CHILD = 15;          // 2^(32-28) - 1, chroma/bvh/grid.py:6

// every device buffer of one build; freed together whatever happens
struct Arena {
    chroma_ctx *ctx = nullptr;       // (blocks come from and go back to the context's pool -- chroma_malloc / chroma_free: a repeated call
                                     //  allocates nothing, and out of memory gives parked blocks back and tries again)
    std::vector<void *> ptrs;
    void drop(void *p) { if (ctx) chroma_free(ctx, p); else hipFree(p); }
    ~Arena() { for (void *p : ptrs) if (p) drop(p); }
    template <class T> hipError_t get(T **out, size_t count)
    {
        void *p = nullptr;
        const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
        hipError_t e = ctx ? (chroma_malloc(ctx, bytes, &p) == CHROMA_OK ? hipSuccess : hipErrorOutOfMemory) : hipMalloc(&p, bytes);
        if (e == hipSuccess) ptrs.push_back(p);
        *out = (T *)p;
        return e;
    }
    void release(void *p)
    {
        for (auto &q : ptrs) if (q == p) { drop(q); q = nullptr; }
    }
};

__device__ inline uint64_t spread3_16(uint32_t input)          // cuda/bvh.cu:42-52
{
    uint64_t x = input;
    x = (x | (x << 16)) & 0x00000000FF0000FFull;
    x = (x | (x << 8)) & 0x000000F00F00F00Full;
    x = (x | (x << 4)) & 0x00000C30C30C30C3ull;
    x = (x | (x << 2)) & 0x0000249249249249ull;
    return x;
}
__device__ inline uint32_t quantize(float v, float origin, float scale)     // cuda/bvh.cu:65-69: truncate
{
    const float d = v - origin;
    const float q = d / scale;
    return (uint32_t)q;
}

// make_leaves (cuda/bvh.cu:149-203): the padded, quantised box of triangle i and the Morton code of its centroid
__device__ inline uint4 leaf_of(const float *vertices, const uint32_t *triangles, uint32_t i, float ox, float oy, float oz, float ws,
                                uint64_t *code)
{
    const float *a = vertices + 3 * (size_t)triangles[3 * (size_t)i];
    const float *b = vertices + 3 * (size_t)triangles[3 * (size_t)i + 1];
    const float *c = vertices + 3 * (size_t)triangles[3 * (size_t)i + 2];
    const float org[3] = {ox, oy, oz};
    uint32_t ql[3], qu[3], qc[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float lower = fminf(fminf(a[k], b[k]), c[k]);
        const float upper = fmaxf(fmaxf(a[k], b[k]), c[k]);
        const float s1 = a[k] + b[k];
        const float s2 = s1 + c[k];
        const float cen = s2 / 3.0f;
        ql[k] = quantize(lower, org[k], ws);
        if (ql[k] > 0) ql[k]--;
        qu[k] = quantize(upper, org[k], ws) + 1;
        qc[k] = quantize(cen, org[k], ws);
    }
    if (code) *code = spread3_16(qc[0]) | (spread3_16(qc[1]) << 1) | (spread3_16(qc[2]) << 2);
    return make_uint4(ql[0] | (qu[0] << 16), ql[1] | (qu[1] << 16), ql[2] | (qu[2] << 16), i);
}

__global__ void k_bvh_check_indices(const uint32_t *triangles, size_t nindices, uint32_t nvertices, uint32_t *bad)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nindices; i += (size_t)gridDim.x * blockDim.x)
        if (triangles[i] >= nvertices) *bad = 1u;
}

__global__ void k_bvh_leaf_codes(const float *vertices, const uint32_t *triangles, uint32_t n, float ox, float oy, float oz, float ws,
                                 uint64_t *codes, uint32_t *ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint64_t code;
    leaf_of(vertices, triangles, i, ox, oy, oz, ws, &code);
    codes[i] = code;
    ids[i] = i;
}

__global__ void k_bvh_gather_leaves(const float *vertices, const uint32_t *triangles, uint32_t n, float ox, float oy, float oz, float ws,
                                    const uint32_t *sorted_ids, uint4 *leaves)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    leaves[i] = leaf_of(vertices, triangles, sorted_ids[i], ox, oy, oz, ws, nullptr);
}

// For every neighbouring pair of the sorted codes: the highest bit in which they differ.  hist[d] = number of pairs whose
// highest differing bit is d, so the number of distinct values of (code >> s) is 1 + sum of hist[d] over d >= s -- for
// every trial shift of grid.py:37-42 at once.
__global__ __launch_bounds__(256) void k_bvh_diff_histogram(const uint64_t *codes, uint32_t n, unsigned long long *hist /* [64] */)
{
    __shared__ uint32_t s_hist[64];
    if (threadIdx.x < 64) s_hist[threadIdx.x] = 0u;
    __syncthreads();
    // (n < 2^28: the index cannot wrap)
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x + 1u; i < n; i += gridDim.x * blockDim.x) {
        const uint64_t x = codes[i] ^ codes[i - 1];
        if (x) atomicAdd(&s_hist[63 - __clzll((long long)x)], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 64 && s_hist[threadIdx.x]) atomicAdd(&hist[threadIdx.x], (unsigned long long)s_hist[threadIdx.x]);
}

// run_key[i] = i where a run of equal shifted codes starts, else 0: an inclusive max-scan turns it into the start of
// the run every element belongs to
__global__ void k_bvh_run_keys(const uint64_t *codes, uint32_t n, int shift, uint32_t *run_key)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    run_key[i] = (i > 0 && (codes[i] >> shift) != (codes[i - 1] >> shift)) ? i : 0u;
}
// a parent starts at the first element of a run and after every MAX_CHILD members of it (grid.py:51-76)
__global__ void k_bvh_parent_flags(const uint32_t *run_start, uint32_t n, uint32_t *flag)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    flag[i] = ((i - run_start[i]) % (uint32_t)MAX_CHILD == 0u) ? 1u : 0u;
}
__global__ void k_bvh_first_children(const uint32_t *flag, const uint32_t *pos, uint32_t n, uint32_t *first_child)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (flag[i]) first_child[pos[i]] = i;
}
// make_parents_detailed (cuda/bvh.cu:270-308): union of the children's boxes, w = nchild << 28 | first child
__global__ void k_bvh_make_parents(const uint4 *children, uint32_t nchildren, const uint32_t *first_child, uint32_t nparents,
                                   const uint64_t *codes, int shift, uint4 *parents, uint64_t *parent_codes)
{
    const uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nparents) return;
    const uint32_t f = first_child[p], e = (p + 1 < nparents) ? first_child[p + 1] : nchildren;
    uint32_t l[3] = {0xFFFFu, 0xFFFFu, 0xFFFFu}, u[3] = {0u, 0u, 0u};
    for (uint32_t c = f; c < e; c++) {
        const uint4 ch = children[c];
        const uint32_t w[3] = {ch.x, ch.y, ch.z};
#pragma unroll
        for (int k = 0; k < 3; k++) { l[k] = min(l[k], w[k] & 0xFFFFu); u[k] = max(u[k], w[k] >> 16); }
    }
    parents[p] = make_uint4(l[0] | (u[0] << 16), l[1] | (u[1] << 16), l[2] | (u[2] << 16), ((e - f) << CHROMA_CHILD_BITS) | f);
    parent_codes[p] = codes[f] >> shift;          // (every member of a run has the same shifted code)
}
// copy_and_offset (cuda/bvh.cu:365-384)
__global__ void k_bvh_copy_offset(const uint4 *src, uint32_t n, uint4 *dst, uint32_t offset)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint4 v = src[i];
    v.w += offset;
    dst[i] = v;
}
// collapse_child (cuda/bvh.cu:530-543): a node with ONE child becomes that child
__global__ void k_bvh_collapse(uint4 *nodes, uint32_t lo, uint32_t hi)
{
    const uint32_t i = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= hi) return;
    const uint32_t w = nodes[i].w;
    if ((w >> CHROMA_CHILD_BITS) == 1u) nodes[i] = nodes[w & ~CHROMA_NCHILD_MASK];
}

struct MaxOp { __device__ __host__ uint32_t operator()(uint32_t a, uint32_t b) const { return a > b ? a : b; } };

inline unsigned blocks_for(size_t n, unsigned block = 256) { return (unsigned)((n + block - 1) / block); }

struct Lap {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = getenv("CHROMA_BVH_VERBOSE") != nullptr;
    hipStream_t s;
    explicit Lap(hipStream_t st) : s(st) {}
    void lap(const char *what)
    {
        if (!on) return;
        hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[bvh device] %-28s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

}  // namespace

extern "C" {

// The device counterpart of chroma_bvh_build: same arguments (host arrays in, a handle out that chroma_bvh_fetch /
// _data / _free serve), same node array bit for bit.
int chroma_bvh_build_device(chroma_ctx *ctx, const float *vertices, uint32_t nvertices, const uint32_t *triangles, uint32_t ntriangles,
                            const float world_origin[3], float world_scale, int32_t target_degree,
                            void **handle, uint64_t *nnodes, uint32_t *nlayers)
{
    if (!ctx || !vertices || !triangles || !handle || ntriangles == 0 || ntriangles >= (1u << CHROMA_CHILD_BITS) || target_degree < 1)
        return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_bvh_build_device: bad argument");
    hipStream_t stream = chroma_internal_stream(ctx);
    DEV_TRY(hipSetDevice(chroma_internal_device(ctx)));
    Lap lap(stream);
    Arena arena;
    arena.ctx = ctx;
    const uint32_t n = ntriangles;
    const float ox = world_origin[0], oy = world_origin[1], oz = world_origin[2], ws = world_scale;

    float *d_vertices; uint32_t *d_triangles;
    DEV_TRY(arena.get(&d_vertices, 3 * (size_t)nvertices));
    DEV_TRY(arena.get(&d_triangles, 3 * (size_t)n));
    DEV_TRY(hipMemcpyAsync(d_vertices, vertices, 3 * (size_t)nvertices * sizeof(float), hipMemcpyHostToDevice, stream));
    DEV_TRY(hipMemcpyAsync(d_triangles, triangles, 3 * (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice, stream));
    unsigned long long *d_hist; uint32_t *d_word;
    DEV_TRY(arena.get(&d_hist, 64));
    DEV_TRY(arena.get(&d_word, 4));
    DEV_TRY(hipMemsetAsync(d_word, 0, 16, stream));
    hipLaunchKernelGGL(k_bvh_check_indices, dim3(std::min(blocks_for(3 * (size_t)n), 65535u)), dim3(256), 0, stream, d_triangles, 3 * (size_t)n, nvertices, d_word);
    uint32_t h_word[4] = {0, 0, 0, 0};
    DEV_TRY(hipMemcpyAsync(h_word, d_word, 4, hipMemcpyDeviceToHost, stream));
    DEV_TRY(hipStreamSynchronize(stream));
    if (h_word[0]) return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_bvh_build_device: a triangle names a vertex outside the mesh");
    lap.lap("upload + validate");

    // make_leaves: Morton codes, then the stable sort (grid.py:26-28)
    uint64_t *d_codes, *d_codes_sorted; uint32_t *d_ids, *d_ids_sorted;
    DEV_TRY(arena.get(&d_codes, n)); DEV_TRY(arena.get(&d_codes_sorted, n));
    DEV_TRY(arena.get(&d_ids, n)); DEV_TRY(arena.get(&d_ids_sorted, n));
    hipLaunchKernelGGL(k_bvh_leaf_codes, dim3(blocks_for(n)), dim3(256), 0, stream, d_vertices, d_triangles, n, ox, oy, oz, ws, d_codes, d_ids);
    size_t tmp_bytes = 0;
    DEV_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_codes, d_codes_sorted, d_ids, d_ids_sorted, (int)n, 0, 48, stream));
    // the scans below reuse this scratch area: size it for the largest request
    size_t scan_bytes = 0, sum_bytes = 0;
    {
        uint32_t *nul = nullptr;
        DEV_TRY(hipcub::DeviceScan::InclusiveScan(nullptr, scan_bytes, nul, nul, MaxOp(), (int)n, stream));
        DEV_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, sum_bytes, nul, nul, (int)n, stream));
    }
    tmp_bytes = std::max(tmp_bytes, std::max(scan_bytes, sum_bytes));
    uint8_t *d_tmp;
    DEV_TRY(arena.get(&d_tmp, tmp_bytes));
    {
        size_t b = tmp_bytes;
        DEV_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, b, d_codes, d_codes_sorted, d_ids, d_ids_sorted, (int)n, 0, 48, stream));
    }
    lap.lap("leaf codes + morton sort");
    std::vector<uint4 *> layers;            // leaves first
    std::vector<uint32_t> layer_size;
    uint4 *d_leaves;
    DEV_TRY(arena.get(&d_leaves, n));
    hipLaunchKernelGGL(k_bvh_gather_leaves, dim3(blocks_for(n)), dim3(256), 0, stream, d_vertices, d_triangles, n, ox, oy, oz, ws, d_ids_sorted, d_leaves);
    DEV_TRY(hipGetLastError());
    DEV_TRY(hipStreamSynchronize(stream));
    arena.release(d_vertices); arena.release(d_triangles); arena.release(d_codes); arena.release(d_ids);
    layers.push_back(d_leaves); layer_size.push_back(n);
    lap.lap("gather leaves");

    // the parent layers.  d_codes_sorted holds the codes of the top layer; the parents' codes go to d_next_codes.
    uint64_t *d_cur_codes = d_codes_sorted, *d_next_codes;
    uint32_t *d_run, *d_flag, *d_pos, *d_first;
    DEV_TRY(arena.get(&d_next_codes, n));
    DEV_TRY(arena.get(&d_run, n)); DEV_TRY(arena.get(&d_flag, n)); DEV_TRY(arena.get(&d_pos, n)); DEV_TRY(arena.get(&d_first, n));
    d_ids = d_ids_sorted;                    // (kept only because the arena owns it)
    while (layer_size.back() > 1) {
        const uint32_t nn = layer_size.back();
        const uint4 *top = layers.back();
        // grid.py:37-42: shift the codes until the mean group size reaches target_degree
        DEV_TRY(hipMemsetAsync(d_hist, 0, 64 * sizeof(unsigned long long), stream));
        hipLaunchKernelGGL(k_bvh_diff_histogram, dim3(std::min(blocks_for(nn), 4096u)), dim3(256), 0, stream, d_cur_codes, nn, d_hist);
        unsigned long long hist[64];
        DEV_TRY(hipMemcpyAsync(hist, d_hist, sizeof hist, hipMemcpyDeviceToHost, stream));
        DEV_TRY(hipStreamSynchronize(stream));
        unsigned long long above[65];
        above[64] = 0;
        for (int d = 63; d >= 0; d--) above[d] = above[d + 1] + hist[d];
        int shift = 0;
        unsigned long long nunique = 1 + above[0];
        while ((double)nn / (double)nunique < (double)target_degree && nunique > 1) {
            shift++;
            nunique = 1 + above[shift];
        }
        // grid.py:45-76: one parent per run of equal codes, runs cut at MAX_CHILD
        hipLaunchKernelGGL(k_bvh_run_keys, dim3(blocks_for(nn)), dim3(256), 0, stream, d_cur_codes, nn, shift, d_run);
        { size_t b = tmp_bytes; DEV_TRY(hipcub::DeviceScan::InclusiveScan(d_tmp, b, d_run, d_run, MaxOp(), (int)nn, stream)); }
        hipLaunchKernelGGL(k_bvh_parent_flags, dim3(blocks_for(nn)), dim3(256), 0, stream, d_run, nn, d_flag);
        { size_t b = tmp_bytes; DEV_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, b, d_flag, d_pos, (int)nn, stream)); }
        uint32_t last[2];
        DEV_TRY(hipMemcpyAsync(&last[0], d_pos + (nn - 1), 4, hipMemcpyDeviceToHost, stream));
        DEV_TRY(hipMemcpyAsync(&last[1], d_flag + (nn - 1), 4, hipMemcpyDeviceToHost, stream));
        DEV_TRY(hipStreamSynchronize(stream));
        const uint32_t np = last[0] + last[1];
        if (np == 0 || np >= nn + (nn == 1)) return chroma_internal_set_error(CHROMA_ERR_INTERNAL, "chroma_bvh_build_device: a layer of %u nodes got %u parents", nn, np);
        hipLaunchKernelGGL(k_bvh_first_children, dim3(blocks_for(nn)), dim3(256), 0, stream, d_flag, d_pos, nn, d_first);
        uint4 *d_parents;
        DEV_TRY(arena.get(&d_parents, np));
        hipLaunchKernelGGL(k_bvh_make_parents, dim3(blocks_for(np)), dim3(256), 0, stream, top, nn, d_first, np, d_cur_codes, shift, d_parents, d_next_codes);
        DEV_TRY(hipGetLastError());
        std::swap(d_cur_codes, d_next_codes);
        layers.push_back(d_parents); layer_size.push_back(np);
    }
    lap.lap("parent layers");

    // concatenate_layers (gpu/bvh.py:239-267): root first; child index += start of the next layer; leaves keep triangle ids
    BvhResult *res = new BvhResult;
    const size_t nl = layers.size();
    res->layer_bounds.assign(nl + 1, 0);
    for (size_t l = 0; l < nl; l++) res->layer_bounds[l + 1] = res->layer_bounds[l] + layer_size[nl - 1 - l];
    const uint64_t total = res->layer_bounds[nl];
    if (total >= (1ull << CHROMA_CHILD_BITS)) { delete res; return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_bvh_build_device: %llu nodes do not fit 28-bit child indices", (unsigned long long)total); }
    uint4 *d_nodes;
    if (arena.get(&d_nodes, total) != hipSuccess) { delete res; return chroma_internal_set_error(CHROMA_ERR_INTERNAL, "chroma_bvh_build_device: out of device memory"); }
    for (size_t l = 0; l < nl; l++) {
        const uint32_t cnt = layer_size[nl - 1 - l];
        const uint32_t offset = (l + 1 < nl) ? (uint32_t)res->layer_bounds[l + 1] : 0u;
        hipLaunchKernelGGL(k_bvh_copy_offset, dim3(blocks_for(cnt)), dim3(256), 0, stream, layers[nl - 1 - l], cnt, d_nodes + res->layer_bounds[l], offset);
    }
    // collapse_chains (gpu/bvh.py:114-130): bottom-up over the inner layers
    for (size_t l = nl - 1; l-- > 0;) {
        const uint32_t lo = (uint32_t)res->layer_bounds[l], hi = (uint32_t)res->layer_bounds[l + 1];
        hipLaunchKernelGGL(k_bvh_collapse, dim3(blocks_for(hi - lo)), dim3(256), 0, stream, d_nodes, lo, hi);
    }
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { delete res; return chroma_internal_set_error((int)e, "chroma_bvh_build_device: %s", hipGetErrorString(e)); }
    res->nodes.resize(total);
    { const int rc_ = chroma_internal_dtoh(ctx, res->nodes.data(), d_nodes, total * sizeof(uint4)); if (rc_ != CHROMA_OK) { delete res; return rc_; } }
    lap.lap("concatenate + collapse + download");
    *handle = res;
    if (nnodes) *nnodes = total;
    if (nlayers) *nlayers = (uint32_t)nl;
    return CHROMA_OK;
}

}  // extern "C"

// ---- tools.argsort_direction (chroma/tools.py:175-193) + the reordering it is used for, on the device -------------
// The reference's own benchmark sorts its photons by a Morton code of (theta, phi) before it starts the clock
// (chroma/benchmark.py:80-82: "organize photons in such a way that they enhance the cache benefits of the GPU");
// chroma_photons_sort_direction is that step for a photon set that lives on the device: 32-bit codes (16 bits of
// theta interleaved with 16 bits of phi, the reference's formula with the numeric contract's acos / atan2), a stable
// radix sort of (code, slot), and every array of the set gathered through it.  Slot i afterwards holds the photon
// of rank i; random streams are keyed by slot (id_base + i) as before.
namespace {
__global__ void k_direction_codes(const float *dir, uint32_t n, uint32_t *codes, uint32_t *ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = dir[3 * (size_t)i], y = dir[3 * (size_t)i + 1], z = dir[3 * (size_t)i + 2];
    const float maxint = 65535.0f;
    const uint32_t theta = (uint32_t)(cm_acosf(cm_fmaxf(-1.0f, cm_fminf(1.0f, z))) / CM_PI_F * maxint);
    const uint32_t phi = (uint32_t)((cm_atan2f(y, x) / CM_PI_F / 2.0f + 0.5f) * maxint);
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < 16; b++) m |= ((theta & (1u << b)) << b) | ((phi & (1u << b)) << (b + 1));
    codes[i] = m;
    ids[i] = i;
}
template <int WORDS>
__global__ void k_gather_words(const uint32_t *src, const uint32_t *order, uint32_t n, uint32_t *dst)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const size_t j = order[i];
#pragma unroll
    for (int w = 0; w < WORDS; w++) dst[(size_t)WORDS * i + w] = src[(size_t)WORDS * j + w];
}
}  // namespace

extern "C" int chroma_photons_sort_direction(chroma_ctx *ctx, const chroma_photon_arrays *photons, uint64_t nphotons)
{
    if (!ctx || !photons || !photons->dir) return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_photons_sort_direction: bad argument");
    if (nphotons >= 0x7fffffffull) return chroma_internal_set_error(CHROMA_ERR_INVALID, "at most 2^31-2 photons per call");
    if (nphotons < 2) return CHROMA_OK;
    hipStream_t stream = chroma_internal_stream(ctx);
    DEV_TRY(hipSetDevice(chroma_internal_device(ctx)));
    const uint32_t n = (uint32_t)nphotons;
    Arena arena;
    arena.ctx = ctx;
    uint32_t *d_codes, *d_codes_sorted, *d_ids, *d_order, *d_buf;
    DEV_TRY(arena.get(&d_codes, n)); DEV_TRY(arena.get(&d_codes_sorted, n));
    DEV_TRY(arena.get(&d_ids, n)); DEV_TRY(arena.get(&d_order, n));
    hipLaunchKernelGGL(k_direction_codes, dim3(blocks_for(n)), dim3(256), 0, stream, photons->dir, n, d_codes, d_ids);
    size_t tmp_bytes = 0;
    DEV_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_codes, d_codes_sorted, d_ids, d_order, (int)n, 0, 32, stream));
    uint8_t *d_tmp;
    DEV_TRY(arena.get(&d_tmp, tmp_bytes));
    DEV_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_codes, d_codes_sorted, d_ids, d_order, (int)n, 0, 32, stream));
    DEV_TRY(hipStreamSynchronize(stream));
    arena.release(d_codes); arena.release(d_codes_sorted); arena.release(d_ids); arena.release(d_tmp);
    DEV_TRY(arena.get(&d_buf, 3 * (size_t)n));
    float *f3[3] = {photons->pos, photons->dir, photons->pol};
    for (float *a : f3) {
        if (!a) continue;
        hipLaunchKernelGGL((k_gather_words<3>), dim3(blocks_for(n)), dim3(256), 0, stream, (const uint32_t *)a, d_order, n, d_buf);
        DEV_TRY(hipMemcpyAsync(a, d_buf, 3 * (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
    }
    void *w1[7] = {photons->wavelengths, photons->t, photons->flags, photons->last_hit_triangles, photons->weights, photons->evidx, photons->rng_counters};
    for (void *a : w1) {
        if (!a) continue;
        hipLaunchKernelGGL((k_gather_words<1>), dim3(blocks_for(n)), dim3(256), 0, stream, (const uint32_t *)a, d_order, n, d_buf);
        DEV_TRY(hipMemcpyAsync(a, d_buf, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
    }
    DEV_TRY(hipGetLastError());
    DEV_TRY(hipStreamSynchronize(stream));
    return CHROMA_OK;
}

// ---- flat hits in (event, channel) order ---------------------------------------------------------------------------------
// What a caller does next with a batch's flat hits (chroma/sim.py:118-123, chroma/gpu/photon.py:96-105) is to split them by event
// and by channel -- the reference with one boolean mask over all hits per event and per channel.  Ordered here, on the device, a
// split is a slice: a 64-bit key (evidx, channel) per hit, a stable radix sort of (key, slot), every array gathered through it.
// The order of flat hits is unspecified in the reference (its compaction goes through an atomic); this one is a valid instance.
namespace {
__global__ void k_hit_keys(const uint32_t *evidx, const int32_t *channels, uint32_t n, unsigned long long *keys, uint32_t *ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    keys[i] = ((unsigned long long)(evidx ? evidx[i] : 0u) << 32) | (uint32_t)channels[i];
    ids[i] = i;
}
}  // namespace

extern "C" int chroma_hits_sort(chroma_ctx *ctx, const chroma_photon_arrays *hits, int32_t *d_channels, uint64_t nhits)
{
    if (!ctx || !hits || !d_channels) return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_hits_sort: bad argument");
    if (nhits >= 0x7fffffffull) return chroma_internal_set_error(CHROMA_ERR_INVALID, "at most 2^31-2 hits per call");
    if (nhits < 2) return CHROMA_OK;
    hipStream_t stream = chroma_internal_stream(ctx);
    DEV_TRY(hipSetDevice(chroma_internal_device(ctx)));
    const uint32_t n = (uint32_t)nhits;
    Arena arena;
    arena.ctx = ctx;
    unsigned long long *d_keys, *d_keys_sorted;
    uint32_t *d_ids, *d_order, *d_buf;
    DEV_TRY(arena.get(&d_keys, n)); DEV_TRY(arena.get(&d_keys_sorted, n));
    DEV_TRY(arena.get(&d_ids, n)); DEV_TRY(arena.get(&d_order, n));
    hipLaunchKernelGGL(k_hit_keys, dim3(blocks_for(n)), dim3(256), 0, stream, (const uint32_t *)hits->evidx, (const int32_t *)d_channels, n, d_keys, d_ids);
    size_t tmp_bytes = 0;
    DEV_TRY(hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, d_keys, d_keys_sorted, d_ids, d_order, (int)n, 0, 64, stream));
    uint8_t *d_tmp;
    DEV_TRY(arena.get(&d_tmp, tmp_bytes));
    DEV_TRY(hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp_bytes, d_keys, d_keys_sorted, d_ids, d_order, (int)n, 0, 64, stream));
    DEV_TRY(arena.get(&d_buf, 3 * (size_t)n));
    float *f3[3] = {hits->pos, hits->dir, hits->pol};
    for (float *a : f3) {
        if (!a) continue;
        hipLaunchKernelGGL((k_gather_words<3>), dim3(blocks_for(n)), dim3(256), 0, stream, (const uint32_t *)a, d_order, n, d_buf);
        DEV_TRY(hipMemcpyAsync(a, d_buf, 3 * (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
    }
    void *w1[8] = {hits->wavelengths, hits->t, hits->flags, hits->last_hit_triangles, hits->weights, hits->evidx, hits->rng_counters, d_channels};
    for (void *a : w1) {
        if (!a) continue;
        hipLaunchKernelGGL((k_gather_words<1>), dim3(blocks_for(n)), dim3(256), 0, stream, (const uint32_t *)a, d_order, n, d_buf);
        DEV_TRY(hipMemcpyAsync(a, d_buf, (size_t)n * 4, hipMemcpyDeviceToDevice, stream));
    }
    DEV_TRY(hipGetLastError());
    DEV_TRY(hipStreamSynchronize(stream));
    return CHROMA_OK;
}

// ---- the order in which chroma_propagate takes up the photons of a POINT-LIKE source that arrive in no particular order ----
// (chroma_hip.hip: propagate_order).  A 16-bit cell of the direction on an octahedral map of the sphere (no arc functions; a
// heuristic that steers speed only -- results are stored by photon id and drawn from per-photon streams), Morton-interleaved
// so that neighbouring cells are neighbours in the order; a stable two-pass radix sort of (cell, photon).  The scratch
// arrays come from the context's pool, so a second call allocates nothing.
namespace {
__global__ void k_order_codes(const float *dir, uint32_t n, uint32_t *codes, uint32_t *ids)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = dir[3 * (size_t)i], y = dir[3 * (size_t)i + 1], z = dir[3 * (size_t)i + 2];
    const float s = fabsf(x) + fabsf(y) + fabsf(z);
    float u = s > 0.0f ? x / s : 0.0f, v = s > 0.0f ? y / s : 0.0f;
    if (z < 0.0f) { const float uu = (1.0f - fabsf(v)) * (u < 0.0f ? -1.0f : 1.0f), vv = (1.0f - fabsf(u)) * (v < 0.0f ? -1.0f : 1.0f); u = uu; v = vv; }
    const uint32_t iu = (uint32_t)fminf(255.0f, fmaxf(0.0f, (u * 0.5f + 0.5f) * 256.0f)), iv = (uint32_t)fminf(255.0f, fmaxf(0.0f, (v * 0.5f + 0.5f) * 256.0f));
    uint32_t m = 0;
#pragma unroll
    for (int b = 0; b < 8; b++) m |= ((iu & (1u << b)) << b) | ((iv & (1u << b)) << (b + 1));
    codes[i] = m;
    ids[i] = i;
}
}  // namespace

extern "C" int chroma_internal_direction_order(chroma_ctx *ctx, const float *d_dir, uint32_t n, uint32_t *d_order)
{
    hipStream_t stream = chroma_internal_stream(ctx);
    void *codes = nullptr, *sorted = nullptr, *ids = nullptr, *tmp = nullptr;
    int rc = chroma_malloc(ctx, (size_t)n * 4, &codes);
    if (rc == CHROMA_OK) rc = chroma_malloc(ctx, (size_t)n * 4, &sorted);
    if (rc == CHROMA_OK) rc = chroma_malloc(ctx, (size_t)n * 4, &ids);
    size_t tmp_bytes = 0;
    hipError_t e = hipSuccess;
    if (rc == CHROMA_OK) {
        e = hipcub::DeviceRadixSort::SortPairs(nullptr, tmp_bytes, (uint32_t *)codes, (uint32_t *)sorted, (uint32_t *)ids, d_order, (int)n, 0, 16, stream);
        if (e == hipSuccess) rc = chroma_malloc(ctx, tmp_bytes, &tmp);
    }
    if (rc == CHROMA_OK && e == hipSuccess) {
        hipLaunchKernelGGL(k_order_codes, dim3(blocks_for(n)), dim3(256), 0, stream, d_dir, n, (uint32_t *)codes, (uint32_t *)ids);
        e = hipcub::DeviceRadixSort::SortPairs(tmp, tmp_bytes, (uint32_t *)codes, (uint32_t *)sorted, (uint32_t *)ids, d_order, (int)n, 0, 16, stream);
        if (e == hipSuccess) e = hipGetLastError();
    }
    chroma_free(ctx, codes); chroma_free(ctx, sorted); chroma_free(ctx, ids); chroma_free(ctx, tmp);      // (parked behind the stream's work)
    if (rc != CHROMA_OK) return rc;
    if (e != hipSuccess) return chroma_internal_set_error((int)e, "direction order: %s", hipGetErrorString(e));
    return CHROMA_OK;
}
