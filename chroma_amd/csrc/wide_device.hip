// wide_device.hip -- the derived 8-wide traversal tree built ON THE DEVICE (gfx950).
//
// The tree is the one csrc/wide_build.cpp's "levels" topology defines (see levels_topology there for the rules and for why
// they leave nothing to the schedule): one binary tree of surface-area-heuristic splits over the reference's leaf boxes, cut
// into eight-wide nodes at the least total area, the wide nodes numbered breadth first.  The host twin builds it subtree by
// subtree; here every step is a pass over an array, one LEVEL of the tree at a time:
//
//   top-down, per level of the binary tree (segments = the sets of that level, in triangle-array order):
//     k_classify         a set of one triangle becomes a leaf; the others go to one of four lists by size
//     k_split_tiny       2..8 triangles: ONE THREAD per set (registers, a sorting network)
//     k_split_small      9..32 triangles: ONE WAVE per set -- ranks by counting (readlane), boxes moved to sorted order with
//                        ds_permute, prefix / suffix unions by lane scans, the cost in doubles, the first minimum by a wave
//                        reduction, the set written back sorted on the winning axis
//     k_split_wave       33..2048: one wave per set -- centroid bounds, 3 x 32 bins filled with LDS atomics, the same scans
//                        over the bins, a stable partition by ballots
//     k_large_*          larger sets: chunks of 4096 triangles, one block each -- bounds and bins reduced in LDS and merged with
//                        global atomics, one wave per set for the decision, a device scan over the chunks' left counts, a
//                        stable scatter
//     children are numbered by a device scan over the level (hipCUB), so a level's sets stay in array order
//   bottom-up, per level: k_dp -- boxes and the table D(n, k) of the least-area collapse (build_subtree_dp, wide_build.cpp)
//   top-down over the WIDE levels: k_emit_count / k_emit_write -- the entries of every wide node of a level, the next
//     level's nodes numbered by a device scan.
//
// Areas are integers (< 2^34) held in doubles and costs are two products and a sum of doubles, compiled without contraction;
// the table D is float sums in a fixed order: the result is BIT-IDENTICAL to the host twin's (tests/test_gpu_wide.py), and
// does not depend on how the atomics of a run fall (they are min / max / add on integers, or hand out places in lists
// whose order nothing depends on).
// Measured at C3 (170 M triangles): see profiles/r03/wide_device_build.txt.
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <chrono>
#include <string>
#include <vector>

#include "../../include/chroma_hip.h"
#include "ctx_access.h"
#include "wide_build.h"

using chroma_host::WideTree;

namespace {

#define DEV_TRY(expr)                                                                                     \
    do {                                                                                                  \
        hipError_t e_ = (expr);                                                                           \
        if (e_ != hipSuccess) {                                                                           \
            chroma_internal_set_error((int)e_, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return (int)e_;                                                                               \
        }                                                                                                 \
    } while (0)

struct Arena {          // every device buffer of one build; freed together whatever happens
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
    void release(void *p) { for (auto &q : ptrs) if (q == p && p) { drop(q); q = nullptr; } }
};

inline unsigned blocks_for(size_t n, unsigned block = 256) { return (unsigned)((n + block - 1) / block); }

struct Lap {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = getenv("CHROMA_TIMING") != nullptr;
    hipStream_t s;
    explicit Lap(hipStream_t st) : s(st) {}
    void lap(const char *what)
    {
        if (!on) return;
        hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        fprintf(stderr, "[wide device] %-34s %.3f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

constexpr uint32_t LEAFBIT = 0x80000000u;       // binary node: w = LEAFBIT | position of the triangle; else w = left child (right = w + 1)
constexpr int NBINS = 32;                        // SAH_BINS of wide_build.cpp
constexpr uint32_t SWEEP_MAX = 32;               // SWEEP_MAX of wide_build.cpp
constexpr uint32_t TINY_MAX = 8;                 // sets up to this size are swept by ONE THREAD each (k_split_tiny)
constexpr uint32_t WAVE_MAX = 2048;              // sets up to this size are split by one wave
constexpr uint32_t CHUNK = 4096;                 // triangles per block of the large-set kernels
constexpr int BIN_WORDS = 7;                     // lo[3], hi[3], count
constexpr int BINS_WORDS = 3 * NBINS * BIN_WORDS;
constexpr uint32_t WIDE_LEAF = chroma_host::WIDE_LEAF, WIDE_EMPTY = chroma_host::WIDE_EMPTY;

// a triangle in the builder: x, y, z = lo | hi << 16 of its leaf box (the reference node's own words), w = triangle
__device__ inline uint32_t axis_word(const uint4 &p, int a) { return a == 0 ? p.x : a == 1 ? p.y : p.z; }
__device__ inline uint32_t cent2(uint32_t w) { return (w & 0xFFFFu) + (w >> 16); }          // doubled centroid
__device__ inline uint32_t bin_of(uint32_t c2, uint32_t cmin, uint32_t ext) { return ((c2 - cmin) * (uint32_t)NBINS) / (ext + 1u); }   // < 2^22: fits

struct BoxC { uint32_t lo[3], hi[3], c; };
__device__ inline BoxC boxc_identity() { BoxC b; b.lo[0] = b.lo[1] = b.lo[2] = 0xFFFFFFFFu; b.hi[0] = b.hi[1] = b.hi[2] = 0u; b.c = 0u; return b; }
__device__ inline BoxC boxc_of(const uint4 &p) { BoxC b; b.lo[0] = p.x & 0xFFFFu; b.lo[1] = p.y & 0xFFFFu; b.lo[2] = p.z & 0xFFFFu; b.hi[0] = p.x >> 16; b.hi[1] = p.y >> 16; b.hi[2] = p.z >> 16; b.c = 1u; return b; }
__device__ inline double boxc_area(const BoxC &b)
{
    const double dx = (double)(b.hi[0] - b.lo[0]), dy = (double)(b.hi[1] - b.lo[1]), dz = (double)(b.hi[2] - b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}
// Lane k (< 32) holds part k of a row of at most 32 parts (lanes from 32 on: the identity).  The cost of cutting the row
// behind part k: area(parts 0..k) * count + area(parts k+1..) * count, as best_split / split_sweep of wide_build.cpp form it.
__device__ inline bool cut_cost(const BoxC &mine, unsigned lane, double &cost, uint32_t &nleft)
{
    BoxC l = mine, r = mine;
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        BoxC o;
#pragma unroll
        for (int a = 0; a < 3; a++) { o.lo[a] = __shfl_up(l.lo[a], off); o.hi[a] = __shfl_up(l.hi[a], off); }
        o.c = __shfl_up(l.c, off);
        if ((int)lane >= off) { for (int a = 0; a < 3; a++) { l.lo[a] = min(l.lo[a], o.lo[a]); l.hi[a] = max(l.hi[a], o.hi[a]); } l.c += o.c; }
    }
#pragma unroll
    for (int off = 1; off < 32; off <<= 1) {
        BoxC o;
#pragma unroll
        for (int a = 0; a < 3; a++) { o.lo[a] = __shfl_down(r.lo[a], off); o.hi[a] = __shfl_down(r.hi[a], off); }
        o.c = __shfl_down(r.c, off);
        if ((int)lane + off < 64) { for (int a = 0; a < 3; a++) { r.lo[a] = min(r.lo[a], o.lo[a]); r.hi[a] = max(r.hi[a], o.hi[a]); } r.c += o.c; }
    }
    double ra = boxc_area(r);
    uint32_t rc = r.c;
    ra = __shfl_down(ra, 1);
    rc = __shfl_down(rc, 1);
    nleft = l.c;
    if (lane >= 31u || l.c == 0u || rc == 0u) return false;
    cost = boxc_area(l) * (double)l.c + ra * (double)rc;
    return true;
}
// the least (cost, index) of the wave, in every lane
__device__ inline void wave_argmin(double &cost, uint32_t &idx)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double oc = __shfl_xor(cost, off);
        const uint32_t oi = __shfl_xor(idx, off);
        if (oc < cost || (oc == cost && oi < idx)) { cost = oc; idx = oi; }
    }
}
__device__ inline uint32_t wave_min_u32(uint32_t v) { for (int off = 32; off > 0; off >>= 1) v = min(v, (uint32_t)__shfl_xor(v, off)); return v; }
__device__ inline uint32_t wave_max_u32(uint32_t v) { for (int off = 32; off > 0; off >>= 1) v = max(v, (uint32_t)__shfl_xor(v, off)); return v; }

// segment of a level: x = first triangle, y = count, z = its binary node
__device__ inline void make_children(const uint4 &seg, uint32_t nl, uint32_t r, uint4 *bin, uint4 *next, uint32_t next_base)
{
    const uint32_t node = next_base + 2u * r;
    next[2u * r] = make_uint4(seg.x, nl, node, 0u);
    next[2u * r + 1u] = make_uint4(seg.x + nl, seg.y - nl, node + 1u, 0u);
    bin[seg.z] = make_uint4(0u, 0u, 0u, node);
}

// ---- the triangles of the builder, from the reference's leaves ------------------------------------------------------------
__global__ void k_prim_flags(const uint32_t *leaf_node, uint32_t ntriangles, uint32_t *flag)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < ntriangles) flag[t] = leaf_node[t] != 0xFFFFFFFFu;
}
__global__ void k_make_prims(const uint4 *nodes, const uint32_t *leaf_node, const uint32_t *pos, uint32_t ntriangles, uint4 *prims, int tight)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntriangles) return;
    const uint32_t ln = leaf_node[t];
    if (ln == 0xFFFFFFFFu) return;
    const uint4 nd = nodes[ln];
    // (wide_build.h: the reference's padding quantum below the triangle is not needed here)
    prims[pos[t]] = make_uint4(wide_tight_bound_word(nd.x, tight), wide_tight_bound_word(nd.y, tight), wide_tight_bound_word(nd.z, tight), t);
}
__global__ void k_last_sum(const uint32_t *a, const uint32_t *b, uint32_t n, uint32_t *out) { if (threadIdx.x == 0 && blockIdx.x == 0) *out = n ? a[n - 1] + b[n - 1] : 0u; }

// ---- the reference's test order (reference_test_order of wide_build.cpp) ---------------------------------------------------
// The host code sweeps the node array layer by layer -- or, when the array is not stored that way (a chroma tree with collapsed
// single-child chains is not: a collapsed node's children sit two layers down), sequentially.  Here the two recurrences are
// solved by PASSES over the whole array until nothing changes (children follow their parents, so the depth of the tree bounds
// the number of passes): leaves(node) = sum of leaves(children) rises to its value from below; base(child) -- the rank of the
// first leaf tested under it -- is written once its parent's is known, with its final value.
constexpr uint32_t NCHILD_SHIFT = 28, CHILD_MASK = 0x0FFFFFFFu, UNREACHED = 0xFFFFFFFFu;
__global__ void k_ref_count_leaves(const uint4 *nodes, uint32_t nnodes, uint32_t *leaves, uint32_t *flags /* [0] changed, [1] bad */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnodes) return;
    const uint32_t w = nodes[i].w, k = w >> NCHILD_SHIFT, c = w & CHILD_MASK;
    uint32_t s = 1u;
    if (k) {
        if ((uint64_t)c + k > nnodes || c <= i) { flags[1] = 1u; return; }
        s = 0u;
        for (uint32_t j = 0; j < k; j++) s += leaves[c + j];
    }
    if (leaves[i] != s) { leaves[i] = s; flags[0] = 1u; }
}
// leaves of a range take their ranks in order, then the inner children -- last first -- take the ranks of their subtrees
// (a triangle under several leaves keeps its least rank: one 64-bit minimum of rank << 32 | leaf node)
__global__ void k_ref_assign(const uint4 *nodes, uint32_t nnodes, const uint32_t *leaves, uint32_t *base, uint32_t ntriangles,
                             unsigned long long *rank_leaf, uint32_t *flags /* [0] changed, [1] bad */)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nnodes) return;
    const uint32_t w = nodes[i].w, k = w >> NCHILD_SHIFT, c = w & CHILD_MASK;
    if (k == 0u) return;
    uint32_t run = base[i];
    if (run == UNREACHED) return;
    for (uint32_t j = 0; j < k; j++) {
        const uint32_t cw = nodes[c + j].w;
        if ((cw >> NCHILD_SHIFT) == 0u) {
            const uint32_t t = cw & CHILD_MASK;
            if (t >= ntriangles) { flags[1] = 1u; continue; }
            atomicMin(rank_leaf + t, (unsigned long long)run << 32 | (unsigned long long)(c + j));
            run++;
        }
    }
    for (uint32_t j = k; j-- > 0u;) {
        const uint32_t cw = nodes[c + j].w;
        if ((cw >> NCHILD_SHIFT) != 0u) {
            if (base[c + j] != run) { base[c + j] = run; flags[0] = 1u; }
            run += leaves[c + j];
        }
    }
}
__global__ void k_ref_unpack(const unsigned long long *rank_leaf, uint32_t ntriangles, uint32_t *rank, uint32_t *leaf_node)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ntriangles) return;
    const unsigned long long v = rank_leaf[t];
    rank[t] = v == ~0ull ? 0xFFFFFFFFu : (uint32_t)(v >> 32);
    leaf_node[t] = v == ~0ull ? 0xFFFFFFFFu : (uint32_t)v;
}

// ---- a level, top-down ---------------------------------------------------------------------------------------------------
__global__ void k_classify(const uint4 *segs, uint32_t nseg, const uint4 *in, uint4 *bin, uint32_t *dev_to_tri, uint32_t *flag,
                           uint32_t *list_small, uint32_t *list_wave, uint32_t *list_large, uint32_t *list_tiny, uint32_t *counters)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    int cls = -1;
    if (i < nseg) {
        const uint4 seg = segs[i];
        if (seg.y == 1u) {
            const uint4 p = in[seg.x];
            bin[seg.z] = make_uint4(p.x, p.y, p.z, LEAFBIT | seg.x);
            dev_to_tri[seg.x] = p.w;
            flag[i] = 0u;
        } else {
            flag[i] = 1u;
            cls = seg.y <= TINY_MAX ? 3 : seg.y <= SWEEP_MAX ? 0 : seg.y <= WAVE_MAX ? 1 : 2;
        }
    }
    const unsigned lane = threadIdx.x & 63u;
#pragma unroll
    for (int c = 0; c < 4; c++) {
        const unsigned long long m = __ballot(cls == c);
        if (!m) continue;
        uint32_t base = 0;
        const int leader = __ffsll((long long)m) - 1;
        if ((int)lane == leader) base = atomicAdd(counters + (c == 3 ? 5 : c), (uint32_t)__popcll(m));
        base = __shfl(base, leader);
        if (cls == c) (c == 0 ? list_small : c == 1 ? list_wave : c == 2 ? list_large : list_tiny)[base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull))] = i;
    }
}

// 2..TINY_MAX triangles: the same exact sweep, ONE THREAD per set.  Most sets of the lower levels are this small (half of all
// inner nodes of a binary tree have two or three triangles below them), and a wave per set leaves 56 to 62 of its lanes idle:
// k_split_small alone was half of the builder's time at C3.  Here the set lives in registers (every loop unrolled to eight
// with guards): the keys, packed with the index, go through a 19-exchange sorting network -- which is the stable order --,
// the triangles are picked in that order by select chains, prefix and suffix unions run sequentially.
__device__ inline void cswap(uint32_t &a, uint32_t &b) { const uint32_t lo = min(a, b), hi = max(a, b); a = lo; b = hi; }
__global__ __launch_bounds__(256) void k_split_tiny(const uint32_t *list, uint32_t nlist, const uint4 *segs, const uint32_t *rank,
                                                    const uint4 *in, uint4 *out, uint4 *bin, uint4 *next, uint32_t next_base)
{
    const uint32_t w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nlist) return;
    const uint32_t s = list[w];
    const uint4 seg = segs[s];
    const uint32_t first = seg.x, n = seg.y;
    uint4 p[8];
#pragma unroll
    for (int i = 0; i < 8; i++) p[i] = (uint32_t)i < n ? in[first + i] : make_uint4(0u, 0u, 0u, 0u);
    double best = __builtin_huge_val();
    uint32_t bidx = 0xFFFFFFFFu, orders[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        uint32_t k[8];          // doubled centroid << 3 | index: sorting these IS the stable sort by centroid
#pragma unroll
        for (int i = 0; i < 8; i++) k[i] = (uint32_t)i < n ? (cent2(axis_word(p[i], a)) << 3 | (uint32_t)i) : 0xFFFFFFFFu;
        cswap(k[0], k[1]); cswap(k[2], k[3]); cswap(k[4], k[5]); cswap(k[6], k[7]);
        cswap(k[0], k[2]); cswap(k[1], k[3]); cswap(k[4], k[6]); cswap(k[5], k[7]);
        cswap(k[1], k[2]); cswap(k[5], k[6]); cswap(k[0], k[4]); cswap(k[3], k[7]);
        cswap(k[1], k[5]); cswap(k[2], k[6]);
        cswap(k[1], k[4]); cswap(k[3], k[6]);
        cswap(k[2], k[4]); cswap(k[3], k[5]);
        cswap(k[3], k[4]);
        uint32_t order = 0;     // nibble r = index of the triangle of rank r
        BoxC q[8];
#pragma unroll
        for (int r = 0; r < 8; r++) {
            const uint32_t idx = k[r] & 7u;
            order |= idx << (4 * r);
            uint4 t = p[0];
#pragma unroll
            for (int i = 1; i < 8; i++) if (idx == (uint32_t)i) t = p[i];
            q[r] = (uint32_t)r < n ? boxc_of(t) : boxc_identity();
        }
        orders[a] = order;
        double la[8], ra[8];
        BoxC run = boxc_identity();
#pragma unroll
        for (int r = 0; r < 8; r++) {
            for (int c = 0; c < 3; c++) { run.lo[c] = min(run.lo[c], q[r].lo[c]); run.hi[c] = max(run.hi[c], q[r].hi[c]); }
            la[r] = boxc_area(run);
        }
        run = boxc_identity();
#pragma unroll
        for (int r = 7; r >= 1; r--) {
            for (int c = 0; c < 3; c++) { run.lo[c] = min(run.lo[c], q[r].lo[c]); run.hi[c] = max(run.hi[c], q[r].hi[c]); }
            ra[r] = boxc_area(run);
        }
#pragma unroll
        for (int c = 0; c < 7; c++) {
            if ((uint32_t)c + 1u < n) {
                const double cost = la[c] * (double)(c + 1) + ra[c + 1] * (double)(n - (uint32_t)c - 1u);
                if (cost < best) { best = cost; bidx = (uint32_t)a * 32u + (uint32_t)c; }
            }
        }
    }
    const uint32_t axis = bidx >> 5, nl = (bidx & 31u) + 1u;
    const uint32_t order = axis == 0u ? orders[0] : axis == 1u ? orders[1] : orders[2];
#pragma unroll
    for (int r = 0; r < 8; r++) {
        if ((uint32_t)r < n) {
            const uint32_t idx = (order >> (4 * r)) & 7u;
            uint4 t = p[0];
#pragma unroll
            for (int i = 1; i < 8; i++) if (idx == (uint32_t)i) t = p[i];
            out[first + r] = t;
        }
    }
    make_children(seg, nl, rank[s], bin, next, next_base);
}

// 9..32 triangles (and any set up to 32 the tiny kernel does not take): the exact sweep of split_sweep (wide_build.cpp), one wave per set
__global__ __launch_bounds__(256) void k_split_small(const uint32_t *list, uint32_t nlist, const uint4 *segs, const uint32_t *rank,
                                                     const uint4 *in, uint4 *out, uint4 *bin, uint4 *next, uint32_t next_base)
{
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t w = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (w >= nlist) return;
    const uint32_t s = list[w];
    uint4 seg = segs[s];
    seg.x = __builtin_amdgcn_readfirstlane(seg.x); seg.y = __builtin_amdgcn_readfirstlane(seg.y); seg.z = __builtin_amdgcn_readfirstlane(seg.z);
    const uint32_t first = seg.x, n = seg.y;
    const bool valid = lane < n;
    const uint4 p = valid ? in[first + lane] : make_uint4(0u, 0u, 0u, 0u);
    double best = __builtin_huge_val();
    uint32_t bidx = 0xFFFFFFFFu;
    uint32_t ranks[3];
#pragma unroll
    for (int a = 0; a < 3; a++) {
        const uint32_t key = valid ? cent2(axis_word(p, a)) : 0xFFFFFFFFu;
        uint32_t r = 0;
        for (uint32_t j = 0; j < n; j++) {          // stable rank: smaller keys, and equal keys that stand before
            const uint32_t kj = (uint32_t)__builtin_amdgcn_readlane((int)key, (int)j);
            r += (kj < key || (kj == key && j < lane)) ? 1u : 0u;
        }
        if (!valid) r = lane;
        ranks[a] = r;
        uint4 q;            // the triangle of rank `lane`
        q.x = (uint32_t)__builtin_amdgcn_ds_permute((int)(r << 2), (int)p.x);
        q.y = (uint32_t)__builtin_amdgcn_ds_permute((int)(r << 2), (int)p.y);
        q.z = (uint32_t)__builtin_amdgcn_ds_permute((int)(r << 2), (int)p.z);
        const BoxC mine = valid ? boxc_of(q) : boxc_identity();
        double cost; uint32_t nleft;
        if (cut_cost(mine, lane, cost, nleft)) {
            const uint32_t idx = (uint32_t)a * 32u + lane;
            if (cost < best) { best = cost; bidx = idx; }          // (a lane's candidates come in rising index)
        }
    }
    wave_argmin(best, bidx);
    const uint32_t axis = bidx >> 5, nl = (bidx & 31u) + 1u;
    const uint32_t r = axis == 0u ? ranks[0] : axis == 1u ? ranks[1] : ranks[2];
    if (valid) out[first + r] = p;
    if (lane == 0) make_children(seg, nl, rank[s], bin, next, next_base);
}

// how a set larger than 32 is cut: x = axis (3: by position -- every centroid coincides), y = last bin of the left part, z = size of the left part
__device__ inline bool goes_left(const uint4 &p, uint32_t index_in_set, const uint4 &split, const uint32_t *cmin, const uint32_t *ext)
{
    if (split.x == 3u) return index_in_set < split.z;
    const int a = (int)split.x;
    return bin_of(cent2(axis_word(p, a)), cmin[a], ext[a]) <= split.y;
}
// the decision from the bins of a set: lanes 0..31 read bin `lane` of each axis through `load`
template <class Load>
__device__ inline uint4 decide_split(Load load, const uint32_t *cmin, const uint32_t *cmax, uint32_t n, unsigned lane)
{
    double best = __builtin_huge_val();
    uint32_t bidx = 0xFFFFFFFFu, bleft = 0u;
#pragma unroll
    for (int a = 0; a < 3; a++) {
        if (cmax[a] == cmin[a]) continue;           // (wave-uniform)
        const BoxC mine = lane < 32u ? load(a, lane) : boxc_identity();
        double cost; uint32_t nleft;
        if (cut_cost(mine, lane, cost, nleft) && cost < best) { best = cost; bidx = (uint32_t)a * 32u + lane; bleft = nleft; }
    }
    wave_argmin(best, bidx);
    if (bidx == 0xFFFFFFFFu) return make_uint4(3u, 0u, n / 2u, 0u);
    const uint32_t nl = (uint32_t)__builtin_amdgcn_readlane((int)bleft, __builtin_amdgcn_readfirstlane((int)(bidx & 31u)));
    return make_uint4(bidx >> 5, bidx & 31u, nl, 0u);
}

// 33..WAVE_MAX triangles: split_stable's binned branch (wide_build.cpp), one wave per set
__global__ __launch_bounds__(256) void k_split_wave(const uint32_t *list, uint32_t nlist, const uint4 *segs, const uint32_t *rank,
                                                    const uint4 *in, uint4 *out, uint4 *bin, uint4 *next, uint32_t next_base)
{
    __shared__ uint32_t s_bins[4][BINS_WORDS];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const uint32_t w = blockIdx.x * 4u + wave;
    const bool active = w < nlist;
    uint4 seg = make_uint4(0u, 0u, 0u, 0u);
    uint32_t s = 0;
    if (active) { s = list[w]; seg = segs[s]; }
    seg.x = __builtin_amdgcn_readfirstlane(seg.x); seg.y = __builtin_amdgcn_readfirstlane(seg.y); seg.z = __builtin_amdgcn_readfirstlane(seg.z);
    const uint32_t first = seg.x, n = seg.y;
    uint32_t *b = s_bins[wave];
    uint32_t cmin[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, cmax[3] = {0u, 0u, 0u}, ext[3];
    for (uint32_t i = lane; i < n; i += 64u) {
        const uint4 p = in[first + i];
#pragma unroll
        for (int a = 0; a < 3; a++) { const uint32_t c = cent2(axis_word(p, a)); cmin[a] = min(cmin[a], c); cmax[a] = max(cmax[a], c); }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) { cmin[a] = wave_min_u32(cmin[a]); cmax[a] = wave_max_u32(cmax[a]); ext[a] = cmax[a] - cmin[a]; }
    for (uint32_t i = lane; i < (uint32_t)BINS_WORDS; i += 64u) b[i] = (i % BIN_WORDS) < 3u ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    for (uint32_t i = lane; i < n; i += 64u) {
        const uint4 p = in[first + i];
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (!ext[a]) continue;
            uint32_t *e = b + ((uint32_t)a * NBINS + bin_of(cent2(axis_word(p, a)), cmin[a], ext[a])) * BIN_WORDS;
            atomicMin(e + 0, p.x & 0xFFFFu); atomicMin(e + 1, p.y & 0xFFFFu); atomicMin(e + 2, p.z & 0xFFFFu);
            atomicMax(e + 3, p.x >> 16); atomicMax(e + 4, p.y >> 16); atomicMax(e + 5, p.z >> 16);
            atomicAdd(e + 6, 1u);
        }
    }
    __syncthreads();
    if (!active) return;
    const uint4 split = decide_split([&](int a, unsigned k) {
        const uint32_t *e = b + ((uint32_t)a * NBINS + k) * BIN_WORDS;
        BoxC v; v.lo[0] = e[0]; v.lo[1] = e[1]; v.lo[2] = e[2]; v.hi[0] = e[3]; v.hi[1] = e[4]; v.hi[2] = e[5]; v.c = e[6];
        return v;
    }, cmin, cmax, n, lane);
    const uint32_t nl = split.z;
    uint32_t run_l = 0, run_r = 0;
    for (uint32_t base = 0; base < n; base += 64u) {
        const uint32_t i = base + lane;
        const bool ok = i < n;
        const uint4 p = ok ? in[first + i] : make_uint4(0u, 0u, 0u, 0u);
        const bool left = ok && goes_left(p, i, split, cmin, ext);
        const unsigned long long ml = __ballot(left), mr = __ballot(ok && !left);
        const unsigned long long below = (1ull << lane) - 1ull;
        if (left) out[first + run_l + (uint32_t)__popcll(ml & below)] = p;
        else if (ok) out[first + nl + run_r + (uint32_t)__popcll(mr & below)] = p;
        run_l += (uint32_t)__popcll(ml); run_r += (uint32_t)__popcll(mr);
    }
    if (lane == 0) make_children(seg, nl, rank[s], bin, next, next_base);
}

// ---- sets larger than WAVE_MAX: chunks of CHUNK triangles ------------------------------------------------------------------
__global__ void k_large_nchunks(const uint32_t *list, uint32_t nlarge, const uint4 *segs, uint32_t *nch)
{
    const uint32_t l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l < nlarge) nch[l] = (segs[list[l]].y + CHUNK - 1u) / CHUNK;
}
__global__ void k_large_init(uint32_t *lcb, uint32_t *lbins, uint32_t nlarge)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (size_t)nlarge * 6u) lcb[i] = (i % 6u) < 3u ? 0xFFFFFFFFu : 0u;
    if (i < (size_t)nlarge * BINS_WORDS) lbins[i] = (i % BIN_WORDS) < 3u ? 0xFFFFFFFFu : 0u;
}
__global__ void k_large_chunkmap(const uint32_t *ch0, uint32_t nlarge, uint32_t nchunks, uint32_t *chunk_l)
{
    const uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    uint32_t lo = 0, hi = nlarge;              // the last l with ch0[l] <= c
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) / 2u; if (ch0[mid] <= c) lo = mid; else hi = mid; }
    chunk_l[c] = lo;
}
struct ChunkView { uint4 seg; uint32_t l, begin, end; };           // [begin, end): this chunk's part of the set, as indices in the set
__device__ inline ChunkView chunk_view(uint32_t c, const uint32_t *chunk_l, const uint32_t *ch0, const uint32_t *list, const uint4 *segs)
{
    ChunkView v;
    v.l = chunk_l[c];
    v.seg = segs[list[v.l]];
    v.begin = (c - ch0[v.l]) * CHUNK;
    v.end = min(v.seg.y, v.begin + CHUNK);
    return v;
}
__global__ __launch_bounds__(256) void k_large_bounds(const uint32_t *chunk_l, const uint32_t *ch0, const uint32_t *list, const uint4 *segs,
                                                      const uint4 *in, uint32_t *lcb)
{
    __shared__ uint32_t s_cb[6];
    const ChunkView v = chunk_view(blockIdx.x, chunk_l, ch0, list, segs);
    if (threadIdx.x < 6) s_cb[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    uint32_t cmin[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, cmax[3] = {0u, 0u, 0u};
    for (uint32_t i = v.begin + threadIdx.x; i < v.end; i += 256u) {
        const uint4 p = in[v.seg.x + i];
#pragma unroll
        for (int a = 0; a < 3; a++) { const uint32_t c = cent2(axis_word(p, a)); cmin[a] = min(cmin[a], c); cmax[a] = max(cmax[a], c); }
    }
#pragma unroll
    for (int a = 0; a < 3; a++) { cmin[a] = wave_min_u32(cmin[a]); cmax[a] = wave_max_u32(cmax[a]); }
    if ((threadIdx.x & 63u) == 0u) { for (int a = 0; a < 3; a++) { atomicMin(&s_cb[a], cmin[a]); atomicMax(&s_cb[3 + a], cmax[a]); } }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(lcb + (size_t)v.l * 6u + threadIdx.x, s_cb[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(lcb + (size_t)v.l * 6u + threadIdx.x, s_cb[threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_large_bins(const uint32_t *chunk_l, const uint32_t *ch0, const uint32_t *list, const uint4 *segs,
                                                    const uint4 *in, const uint32_t *lcb, uint32_t *lbins)
{
    // (one set of bins PER WAVE: near the top of the tree the triangles of a chunk fall into a few bins, and 256 threads
    //  queueing on the same LDS words was most of this kernel's time)
    __shared__ uint32_t s_b[4][BINS_WORDS];
    const ChunkView v = chunk_view(blockIdx.x, chunk_l, ch0, list, segs);
    uint32_t cmin[3], ext[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { cmin[a] = lcb[(size_t)v.l * 6u + a]; ext[a] = lcb[(size_t)v.l * 6u + 3 + a] - cmin[a]; }
    for (uint32_t i = threadIdx.x; i < 4u * (uint32_t)BINS_WORDS; i += 256u) (&s_b[0][0])[i] = ((i % BINS_WORDS) % BIN_WORDS) < 3u ? 0xFFFFFFFFu : 0u;
    __syncthreads();
    uint32_t *b = s_b[threadIdx.x >> 6];
    const unsigned lane = threadIdx.x & 63u;
    for (uint32_t base = v.begin; base < v.end; base += 256u) {          // (every lane goes round: the wave reductions below need them all)
        const uint32_t i = base + threadIdx.x;
        const bool ok = i < v.end;
        const uint4 p = ok ? in[v.seg.x + i] : make_uint4(0xFFFFu, 0xFFFFu, 0xFFFFu, 0u);      // (lo 0xFFFF, hi 0: the identity)
        const unsigned long long m = __ballot(ok);
        if (m == 0ull) continue;                        // (a wave past the end of the chunk)
#pragma unroll
        for (int a = 0; a < 3; a++) {
            if (!ext[a]) continue;
            const uint32_t k = ok ? bin_of(cent2(axis_word(p, a)), cmin[a], ext[a]) : 0xFFFFFFFFu;
            const uint32_t k0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);        // (lane 0 of a round always holds a triangle)
            uint32_t *e = b + ((uint32_t)a * NBINS + (ok ? k : 0u)) * BIN_WORDS;
            if (__ballot(ok && k != k0) == 0ull) {
                // the whole wave in ONE bin (the usual case a few levels down, where a chunk is a small region): reduce in
                // registers, one lane touches the bin
                const uint32_t lx = wave_min_u32(p.x & 0xFFFFu), ly = wave_min_u32(p.y & 0xFFFFu), lz = wave_min_u32(p.z & 0xFFFFu);
                const uint32_t hx = wave_max_u32(p.x >> 16), hy = wave_max_u32(p.y >> 16), hz = wave_max_u32(p.z >> 16);
                if (lane == 0) {
                    atomicMin(e + 0, lx); atomicMin(e + 1, ly); atomicMin(e + 2, lz);
                    atomicMax(e + 3, hx); atomicMax(e + 4, hy); atomicMax(e + 5, hz);
                    atomicAdd(e + 6, (uint32_t)__popcll(m));
                }
            } else if (ok) {
                atomicMin(e + 0, p.x & 0xFFFFu); atomicMin(e + 1, p.y & 0xFFFFu); atomicMin(e + 2, p.z & 0xFFFFu);
                atomicMax(e + 3, p.x >> 16); atomicMax(e + 4, p.y >> 16); atomicMax(e + 5, p.z >> 16);
                atomicAdd(e + 6, 1u);
            }
        }
    }
    __syncthreads();
    uint32_t *g = lbins + (size_t)v.l * BINS_WORDS;
    for (uint32_t i = threadIdx.x; i < (uint32_t)BINS_WORDS; i += 256u) {
        const uint32_t kind = i % BIN_WORDS;
        uint32_t count = 0, val = kind < 3u ? 0xFFFFFFFFu : 0u;
#pragma unroll
        for (int w = 0; w < 4; w++) {
            count += s_b[w][i - kind + 6u];
            val = kind < 3u ? min(val, s_b[w][i]) : kind < 6u ? max(val, s_b[w][i]) : val + s_b[w][i];
        }
        if (count == 0u) continue;                      // an empty bin of this chunk
        if (kind < 3u) atomicMin(g + i, val); else if (kind < 6u) atomicMax(g + i, val); else atomicAdd(g + i, val);
    }
}
__global__ __launch_bounds__(256) void k_large_decide(const uint32_t *list, uint32_t nlarge, const uint4 *segs, const uint32_t *rank,
                                                      const uint32_t *lcb, const uint32_t *lbins, uint4 *lsplit, uint4 *bin, uint4 *next, uint32_t next_base)
{
    const unsigned lane = threadIdx.x & 63u;
    const uint32_t l = blockIdx.x * 4u + (threadIdx.x >> 6);
    if (l >= nlarge) return;
    const uint32_t s = list[l];
    uint4 seg = segs[s];
    seg.x = __builtin_amdgcn_readfirstlane(seg.x); seg.y = __builtin_amdgcn_readfirstlane(seg.y); seg.z = __builtin_amdgcn_readfirstlane(seg.z);
    uint32_t cmin[3], cmax[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { cmin[a] = __builtin_amdgcn_readfirstlane(lcb[(size_t)l * 6u + a]); cmax[a] = __builtin_amdgcn_readfirstlane(lcb[(size_t)l * 6u + 3 + a]); }
    const uint32_t *g = lbins + (size_t)l * BINS_WORDS;
    const uint4 split = decide_split([&](int a, unsigned k) {
        const uint32_t *e = g + ((uint32_t)a * NBINS + k) * BIN_WORDS;
        BoxC v; v.lo[0] = e[0]; v.lo[1] = e[1]; v.lo[2] = e[2]; v.hi[0] = e[3]; v.hi[1] = e[4]; v.hi[2] = e[5]; v.c = e[6];
        return v;
    }, cmin, cmax, seg.y, lane);
    if (lane == 0) {
        lsplit[l] = split;
        make_children(seg, split.z, rank[s], bin, next, next_base);
    }
}
__global__ __launch_bounds__(256) void k_large_count(const uint32_t *chunk_l, const uint32_t *ch0, const uint32_t *list, const uint4 *segs,
                                                     const uint4 *in, const uint32_t *lcb, const uint4 *lsplit, uint32_t *chunk_nl)
{
    __shared__ uint32_t s_n;
    const ChunkView v = chunk_view(blockIdx.x, chunk_l, ch0, list, segs);
    const uint4 split = lsplit[v.l];
    uint32_t cmin[3], ext[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { cmin[a] = lcb[(size_t)v.l * 6u + a]; ext[a] = lcb[(size_t)v.l * 6u + 3 + a] - cmin[a]; }
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    uint32_t c = 0;
    for (uint32_t i = v.begin + threadIdx.x; i < v.end; i += 256u) c += goes_left(in[v.seg.x + i], i, split, cmin, ext) ? 1u : 0u;
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63u) == 0u) atomicAdd(&s_n, c);
    __syncthreads();
    if (threadIdx.x == 0) chunk_nl[blockIdx.x] = s_n;
}
__global__ __launch_bounds__(256) void k_large_scatter(const uint32_t *chunk_l, const uint32_t *ch0, const uint32_t *list, const uint4 *segs,
                                                       const uint4 *in, uint4 *out, const uint32_t *lcb, const uint4 *lsplit, const uint32_t *chunk_off)
{
    __shared__ uint32_t s_l[4], s_r[4];
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const ChunkView v = chunk_view(blockIdx.x, chunk_l, ch0, list, segs);
    const uint4 split = lsplit[v.l];
    uint32_t cmin[3], ext[3];
#pragma unroll
    for (int a = 0; a < 3; a++) { cmin[a] = lcb[(size_t)v.l * 6u + a]; ext[a] = lcb[(size_t)v.l * 6u + 3 + a] - cmin[a]; }
    const uint32_t lefts_before = chunk_off[blockIdx.x] - chunk_off[ch0[v.l]];
    uint32_t run_l = v.seg.x + lefts_before, run_r = v.seg.x + split.z + (v.begin - lefts_before);
    for (uint32_t base = v.begin; base < v.end; base += 256u) {
        const uint32_t i = base + threadIdx.x;
        const bool ok = i < v.end;
        const uint4 p = ok ? in[v.seg.x + i] : make_uint4(0u, 0u, 0u, 0u);
        const bool left = ok && goes_left(p, i, split, cmin, ext);
        const unsigned long long ml = __ballot(left), mr = __ballot(ok && !left);
        if (lane == 0) { s_l[wave] = (uint32_t)__popcll(ml); s_r[wave] = (uint32_t)__popcll(mr); }
        __syncthreads();
        uint32_t wl = 0, wr = 0, tl = 0, tr = 0;
#pragma unroll
        for (unsigned k = 0; k < 4; k++) { if (k < wave) { wl += s_l[k]; wr += s_r[k]; } tl += s_l[k]; tr += s_r[k]; }
        const unsigned long long below = (1ull << lane) - 1ull;
        if (left) out[run_l + wl + (uint32_t)__popcll(ml & below)] = p;
        else if (ok) out[run_r + wr + (uint32_t)__popcll(mr & below)] = p;
        run_l += tl; run_r += tr;
        __syncthreads();
    }
}

// ---- bottom-up: boxes and the least-area table (dp_fill_node of wide_build.cpp) ------------------------------------------
// share word of a node: nibble k-1 = entries given to the left child when the node's children share k entries (bit 3: with
// k entries the node is a wide node of its own; nibble 0: the share of its own eight)
__global__ void k_dp(uint4 *bin, float *cost, uint32_t *share, uint32_t lo, uint32_t cnt)
{
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const uint32_t n = lo + i;
    const uint4 b = bin[n];
    if (b.w & LEAFBIT) return;                           // D(leaf, k) = 0: the table starts zeroed
    const uint4 l = bin[b.w], r = bin[b.w + 1u];
    uint4 me;
    me.x = min(l.x & 0xFFFFu, r.x & 0xFFFFu) | max(l.x >> 16, r.x >> 16) << 16;
    me.y = min(l.y & 0xFFFFu, r.y & 0xFFFFu) | max(l.y >> 16, r.y >> 16) << 16;
    me.z = min(l.z & 0xFFFFu, r.z & 0xFFFFu) | max(l.z >> 16, r.z >> 16) << 16;
    me.w = b.w;
    bin[n] = me;
    float cl[8], cr[8];
    {
        const float4 *pl = (const float4 *)(cost + (size_t)b.w * 8u), *pr = (const float4 *)(cost + (size_t)(b.w + 1u) * 8u);
        const float4 a0 = pl[0], a1 = pl[1], b0 = pr[0], b1 = pr[1];
        cl[0] = a0.x; cl[1] = a0.y; cl[2] = a0.z; cl[3] = a0.w; cl[4] = a1.x; cl[5] = a1.y; cl[6] = a1.z; cl[7] = a1.w;
        cr[0] = b0.x; cr[1] = b0.y; cr[2] = b0.z; cr[3] = b0.w; cr[4] = b1.x; cr[5] = b1.y; cr[6] = b1.z; cr[7] = b1.w;
    }
    float sh[9]; uint32_t shj[9];
#pragma unroll
    for (int k = 2; k <= 8; k++) {
        float best = 0.0f; int bj = 0;
#pragma unroll
        for (int j = 1; j < k; j++) {
            const float c = cl[j - 1] + cr[k - j - 1];
            if (bj == 0 || c < best) { best = c; bj = j; }
        }
        sh[k] = best; shj[k] = (uint32_t)bj;
    }
    const double dx = (double)((me.x >> 16) - (me.x & 0xFFFFu)), dy = (double)((me.y >> 16) - (me.y & 0xFFFFu)), dz = (double)((me.z >> 16) - (me.z & 0xFFFFu));
    const float own = (float)(dx * dy + dy * dz + dz * dx) + sh[8];
    float cn[8];
    cn[0] = own;
    uint32_t word = shj[8] | 8u;
#pragma unroll
    for (int k = 2; k <= 8; k++) {
        if (own <= sh[k]) { cn[k - 1] = own; word |= (shj[8] | 8u) << (4 * (k - 1)); }
        else { cn[k - 1] = sh[k]; word |= shj[k] << (4 * (k - 1)); }
    }
    float4 *pn = (float4 *)(cost + (size_t)n * 8u);
    pn[0] = make_float4(cn[0], cn[1], cn[2], cn[3]);
    pn[1] = make_float4(cn[4], cn[5], cn[6], cn[7]);
    share[n] = word;
}

// ---- the wide nodes, breadth first (emit_breadth_first of wide_build.cpp) --------------------------------------------------
__global__ void k_emit_count(const uint4 *bin, const uint32_t *share, const uint32_t *level, uint32_t cnt, uint32_t *items, uint32_t *info, uint32_t *ninner)
{
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= cnt) return;
    const uint32_t n = level[k];
    uint32_t sn[8], sk[8], it[8];
    int sp = 0, ni = 0;
    {
        const uint4 b = bin[n];
        const uint32_t j = share[n] & 7u;
        sn[sp] = b.w + 1u; sk[sp++] = 8u - j;
        sn[sp] = b.w; sk[sp++] = j;
    }
    uint32_t innermask = 0;
    while (sp) {
        --sp;
        const uint32_t m = sn[sp], kk = sk[sp];
        const uint4 bm = bin[m];
        const bool leaf = (bm.w & LEAFBIT) != 0u;
        const uint32_t nib = leaf ? 0u : (share[m] >> (4u * (kk - 1u))) & 15u;
        if (leaf || kk == 1u || (nib & 8u)) { if (!leaf) innermask |= 1u << ni; it[ni++] = m; }
        else {
            const uint32_t jj = nib & 7u;
            sn[sp] = bm.w + 1u; sk[sp++] = kk - jj;
            sn[sp] = bm.w; sk[sp++] = jj;
        }
    }
    for (int i = 0; i < 8; i++) items[(size_t)k * 8u + i] = i < ni ? it[i] : 0xFFFFFFFFu;
    info[k] = innermask;
    ninner[k] = (uint32_t)__popc(innermask);
}
__global__ void k_emit_write(const uint4 *bin, const uint32_t *items, const uint32_t *info, const uint32_t *first_child, uint32_t cnt,
                             uint32_t child_base, uint4 *wnodes, uint32_t *next)
{
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t k = t >> 3, i = t & 7u;
    if (k >= cnt) return;
    const uint32_t item = items[(size_t)k * 8u + i];
    uint4 o = make_uint4(0x0000FFFFu, 0x0000FFFFu, 0x0000FFFFu, WIDE_EMPTY);
    if (item != 0xFFFFFFFFu) {
        const uint4 c = bin[item];
        o.x = c.x; o.y = c.y; o.z = c.z;
        if (c.w & LEAFBIT) o.w = WIDE_LEAF | (c.w & ~LEAFBIT);
        else {
            const uint32_t child = first_child[k] + (uint32_t)__popc(info[k] & ((1u << i) - 1u));
            o.w = child_base + child;
            next[child] = item;
        }
    }
    wnodes[(size_t)k * 8u + i] = o;
}

}  // namespace

extern "C" {

// The wide tree of a reference-format BVH (host array), built on the device of `ctx`.  Same handle as chroma_wide_build:
// chroma_wide_data / chroma_wide_free serve both.  The tree is the "levels" topology of wide_build.cpp, bit for bit.
int chroma_wide_build_device(chroma_ctx *ctx, const uint32_t *nodes, uint64_t nnodes, uint32_t ntriangles, void **handle,
                             uint64_t *nwide, uint64_t *nrecords, uint32_t *depth)
{
    if (!ctx || !nodes || !handle || nnodes == 0) return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_wide_build_device: bad argument");
    hipStream_t stream = chroma_internal_stream(ctx);
    DEV_TRY(hipSetDevice(chroma_internal_device(ctx)));
    Lap lap(stream);
    WideTree *t = new WideTree;
    struct Guard { WideTree *t; ~Guard() { delete t; } } guard{t};
    if (nnodes >= 0xFFFFFFFFull) return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_wide_build_device: too many nodes");
    Arena arena;
    arena.ctx = ctx;
    uint4 *d_nodes; uint32_t *d_leaf_node, *d_flag, *d_rank, *d_counters;
    DEV_TRY(arena.get(&d_nodes, (size_t)nnodes));
    DEV_TRY(arena.get(&d_leaf_node, ntriangles));
    DEV_TRY(arena.get(&d_flag, ntriangles)); DEV_TRY(arena.get(&d_rank, ntriangles));
    DEV_TRY(arena.get(&d_counters, 8));
    { const int rc_ = chroma_internal_htod(ctx, d_nodes, nodes, (size_t)nnodes * 16u); if (rc_ != CHROMA_OK) return rc_; }
    lap.lap("nodes upload");

    // ---- the reference's test order: leaves under every node, then ranks (passes until nothing changes)
    std::vector<uint32_t> leaf_node;              // (host copy: only for a tree of one triangle)
    {
        const uint32_t nn = (uint32_t)nnodes;
        uint32_t *d_leaves, *d_base; unsigned long long *d_rank_leaf;
        DEV_TRY(arena.get(&d_leaves, nn)); DEV_TRY(arena.get(&d_base, nn)); DEV_TRY(arena.get(&d_rank_leaf, ntriangles));
        DEV_TRY(hipMemsetAsync(d_leaves, 0, (size_t)nn * 4u, stream));
        DEV_TRY(hipMemsetAsync(d_base, 0xFF, (size_t)nn * 4u, stream));
        DEV_TRY(hipMemsetAsync(d_base, 0, 4, stream));
        DEV_TRY(hipMemsetAsync(d_rank_leaf, 0xFF, (size_t)ntriangles * 8u, stream));
        for (int phase = 0; phase < 2; phase++) {
            uint32_t h[2] = {1u, 0u};
            for (int it = 0; h[0] && !h[1]; it++) {
                if (it > 4096) return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_wide_build_device: the reference tree does not settle");
                DEV_TRY(hipMemsetAsync(d_counters, 0, 2 * sizeof(uint32_t), stream));
                for (int k = 0; k < 4; k++) {                    // (four passes per question)
                    if (phase == 0) hipLaunchKernelGGL(k_ref_count_leaves, dim3(blocks_for(nn)), dim3(256), 0, stream, d_nodes, nn, d_leaves, d_counters);
                    else hipLaunchKernelGGL(k_ref_assign, dim3(blocks_for(nn)), dim3(256), 0, stream, d_nodes, nn, d_leaves, d_base, ntriangles, d_rank_leaf, d_counters);
                }
                DEV_TRY(hipMemcpyAsync(h, d_counters, sizeof h, hipMemcpyDeviceToHost, stream));
                DEV_TRY(hipStreamSynchronize(stream));
            }
            if (h[1]) return chroma_internal_set_error(CHROMA_ERR_INVALID, phase == 0 ? "chroma_wide_build_device: wide tree: bad child range"
                                                                                        : "chroma_wide_build_device: wide tree: leaf references a triangle outside the mesh");
        }
        if (ntriangles) hipLaunchKernelGGL(k_ref_unpack, dim3(blocks_for(ntriangles)), dim3(256), 0, stream, d_rank_leaf, ntriangles, d_rank, d_leaf_node);
        DEV_TRY(hipGetLastError());
        DEV_TRY(hipStreamSynchronize(stream));
        t->rank.resize(ntriangles);
        { const int rc_ = chroma_internal_dtoh(ctx, t->rank.data(), d_rank, (size_t)ntriangles * 4u); if (rc_ != CHROMA_OK) return rc_; }
        arena.release(d_leaves); arena.release(d_base); arena.release(d_rank_leaf);
        lap.lap("reference test order");
    }
    size_t tmp_bytes = 0;
    { uint32_t *nul = nullptr; DEV_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, nul, nul, (int)std::max<uint32_t>(ntriangles, 1u), stream)); }
    uint8_t *d_tmp;
    DEV_TRY(arena.get(&d_tmp, tmp_bytes));
    uint32_t np = 0;
    if (ntriangles) {
        hipLaunchKernelGGL(k_prim_flags, dim3(blocks_for(ntriangles)), dim3(256), 0, stream, d_leaf_node, ntriangles, d_flag);
        { size_t b = tmp_bytes; DEV_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, b, d_flag, d_rank, (int)ntriangles, stream)); }
        hipLaunchKernelGGL(k_last_sum, dim3(1), dim3(64), 0, stream, d_rank, d_flag, ntriangles, d_counters);
        DEV_TRY(hipMemcpyAsync(&np, d_counters, 4, hipMemcpyDeviceToHost, stream));
        DEV_TRY(hipStreamSynchronize(stream));
    }
    if (np > 0x3FFFFFFFu) return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_wide_build_device: too many triangles");
    if (np < 2) {
        // nothing to split: an empty node, or one leaf entry
        t->wnodes.assign(32, 0);
        for (int i = 0; i < 8; i++) { uint32_t *o = t->wnodes.data() + 4 * i; o[0] = o[1] = o[2] = 0x0000FFFFu; o[3] = WIDE_EMPTY; }
        if (np == 1) {
            if (leaf_node.empty()) {
                leaf_node.resize(ntriangles);
                DEV_TRY(hipMemcpy(leaf_node.data(), d_leaf_node, (size_t)ntriangles * 4u, hipMemcpyDeviceToHost));
            }
            uint32_t tri = 0;
            while (leaf_node[tri] == 0xFFFFFFFFu) tri++;          // the one triangle under a reachable leaf
            memcpy(t->wnodes.data(), nodes + 4 * (size_t)leaf_node[tri], 12);
            t->wnodes[3] = WIDE_LEAF | 0u;
            t->dev_to_tri.assign(1, tri);
        }
        t->nwide = 1; t->depth = 1;
        chroma_host::finish_wide_tree(*t, ntriangles);
        guard.t = nullptr;
        *handle = t;
        if (nwide) *nwide = t->nwide;
        if (nrecords) *nrecords = t->dev_to_tri.size();
        if (depth) *depth = t->depth;
        return CHROMA_OK;
    }
    { std::vector<uint32_t>().swap(leaf_node); }
    uint4 *d_prims_a, *d_prims_b;
    DEV_TRY(arena.get(&d_prims_a, np)); DEV_TRY(arena.get(&d_prims_b, np));
    hipLaunchKernelGGL(k_make_prims, dim3(blocks_for(ntriangles)), dim3(256), 0, stream, d_nodes, d_leaf_node, d_rank, ntriangles, d_prims_a, wide_tight_leaves());
    DEV_TRY(hipGetLastError());
    DEV_TRY(hipStreamSynchronize(stream));
    arena.release(d_nodes); arena.release(d_leaf_node); arena.release(d_flag); arena.release(d_rank); arena.release(d_tmp);
    lap.lap("upload + triangles");

    // ---- the binary tree, level by level
    const size_t nb = 2 * (size_t)np - 1;
    uint4 *d_bin, *d_segs_a, *d_segs_b, *d_lsplit;
    uint32_t *d_dev_to_tri, *d_list[4], *d_nch, *d_ch0, *d_chunk_l, *d_chunk_nl, *d_chunk_off, *d_lcb, *d_lbins;
    DEV_TRY(arena.get(&d_bin, nb));
    DEV_TRY(arena.get(&d_segs_a, np)); DEV_TRY(arena.get(&d_segs_b, np));
    DEV_TRY(arena.get(&d_flag, np)); DEV_TRY(arena.get(&d_rank, np));
    DEV_TRY(arena.get(&d_dev_to_tri, np));
    for (int c = 0; c < 4; c++) DEV_TRY(arena.get(&d_list[c], c == 3 ? (size_t)np / 2 + 1 : c == 0 ? (size_t)np / (TINY_MAX + 1) + 1 : c == 1 ? (size_t)np / (SWEEP_MAX + 1) + 1 : (size_t)np / (WAVE_MAX + 1) + 1));
    const size_t max_large = (size_t)np / (WAVE_MAX + 1) + 1, max_chunks = (size_t)np / CHUNK + max_large + 1;
    DEV_TRY(arena.get(&d_nch, max_large)); DEV_TRY(arena.get(&d_ch0, max_large));
    DEV_TRY(arena.get(&d_lcb, max_large * 6)); DEV_TRY(arena.get(&d_lbins, max_large * BINS_WORDS)); DEV_TRY(arena.get(&d_lsplit, max_large));
    DEV_TRY(arena.get(&d_chunk_l, max_chunks)); DEV_TRY(arena.get(&d_chunk_nl, max_chunks)); DEV_TRY(arena.get(&d_chunk_off, max_chunks));
    { uint32_t *nul = nullptr; DEV_TRY(hipcub::DeviceScan::ExclusiveSum(nullptr, tmp_bytes, nul, nul, (int)np, stream)); }
    DEV_TRY(arena.get(&d_tmp, tmp_bytes));
    {
        const uint4 root = make_uint4(0u, np, 0u, 0u);
        DEV_TRY(hipMemcpyAsync(d_segs_a, &root, sizeof root, hipMemcpyHostToDevice, stream));
    }
    std::vector<uint32_t> level_base, level_count;
    uint32_t nseg = 1, base = 0;
    for (;;) {
        if (level_base.size() > 4096) return chroma_internal_set_error(CHROMA_ERR_INTERNAL, "chroma_wide_build_device: the binary tree does not end");
        DEV_TRY(hipMemsetAsync(d_counters, 0, 8 * sizeof(uint32_t), stream));
        hipLaunchKernelGGL(k_classify, dim3(blocks_for(nseg)), dim3(256), 0, stream, d_segs_a, nseg, d_prims_a, d_bin, d_dev_to_tri, d_flag,
                           d_list[0], d_list[1], d_list[2], d_list[3], d_counters);
        { size_t b = tmp_bytes; DEV_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, b, d_flag, d_rank, (int)nseg, stream)); }
        hipLaunchKernelGGL(k_last_sum, dim3(1), dim3(64), 0, stream, d_rank, d_flag, nseg, d_counters + 3);
        uint32_t h[6];                  // [0] small, [1] wave, [2] large, [3] sets to split, [4] chunks (below), [5] tiny
        DEV_TRY(hipMemcpyAsync(h, d_counters, sizeof h, hipMemcpyDeviceToHost, stream));
        DEV_TRY(hipStreamSynchronize(stream));
        level_base.push_back(base); level_count.push_back(nseg);
        const uint32_t nsmall = h[0], nwave = h[1], nlarge = h[2], nonleaf = h[3], ntiny = h[5];
        if (ntiny + nsmall + nwave + nlarge != nonleaf) return chroma_internal_set_error(CHROMA_ERR_INTERNAL, "chroma_wide_build_device: level %zu: lists of %u + %u + %u + %u sets, %u to split", level_base.size() - 1, ntiny, nsmall, nwave, nlarge, nonleaf);
        if (nonleaf == 0) break;
        const uint32_t next_base = base + nseg;
        if (ntiny) hipLaunchKernelGGL(k_split_tiny, dim3(blocks_for(ntiny)), dim3(256), 0, stream, d_list[3], ntiny, d_segs_a, d_rank, d_prims_a, d_prims_b, d_bin, d_segs_b, next_base);
        if (nsmall) hipLaunchKernelGGL(k_split_small, dim3(blocks_for(nsmall, 4)), dim3(256), 0, stream, d_list[0], nsmall, d_segs_a, d_rank, d_prims_a, d_prims_b, d_bin, d_segs_b, next_base);
        if (nwave) hipLaunchKernelGGL(k_split_wave, dim3(blocks_for(nwave, 4)), dim3(256), 0, stream, d_list[1], nwave, d_segs_a, d_rank, d_prims_a, d_prims_b, d_bin, d_segs_b, next_base);
        if (nlarge) {
            hipLaunchKernelGGL(k_large_nchunks, dim3(blocks_for(nlarge)), dim3(256), 0, stream, d_list[2], nlarge, d_segs_a, d_nch);
            { size_t b = tmp_bytes; DEV_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, b, d_nch, d_ch0, (int)nlarge, stream)); }
            hipLaunchKernelGGL(k_last_sum, dim3(1), dim3(64), 0, stream, d_ch0, d_nch, nlarge, d_counters + 4);
            uint32_t nchunks = 0;
            DEV_TRY(hipMemcpyAsync(&nchunks, d_counters + 4, 4, hipMemcpyDeviceToHost, stream));
            DEV_TRY(hipStreamSynchronize(stream));
            if (nchunks == 0 || nchunks > max_chunks) return chroma_internal_set_error(CHROMA_ERR_INTERNAL, "chroma_wide_build_device: %u chunks", nchunks);
            hipLaunchKernelGGL(k_large_init, dim3(blocks_for((size_t)nlarge * BINS_WORDS)), dim3(256), 0, stream, d_lcb, d_lbins, nlarge);
            hipLaunchKernelGGL(k_large_chunkmap, dim3(blocks_for(nchunks)), dim3(256), 0, stream, d_ch0, nlarge, nchunks, d_chunk_l);
            hipLaunchKernelGGL(k_large_bounds, dim3(nchunks), dim3(256), 0, stream, d_chunk_l, d_ch0, d_list[2], d_segs_a, d_prims_a, d_lcb);
            hipLaunchKernelGGL(k_large_bins, dim3(nchunks), dim3(256), 0, stream, d_chunk_l, d_ch0, d_list[2], d_segs_a, d_prims_a, d_lcb, d_lbins);
            hipLaunchKernelGGL(k_large_decide, dim3(blocks_for(nlarge, 4)), dim3(256), 0, stream, d_list[2], nlarge, d_segs_a, d_rank, d_lcb, d_lbins, d_lsplit, d_bin, d_segs_b, next_base);
            hipLaunchKernelGGL(k_large_count, dim3(nchunks), dim3(256), 0, stream, d_chunk_l, d_ch0, d_list[2], d_segs_a, d_prims_a, d_lcb, d_lsplit, d_chunk_nl);
            { size_t b = tmp_bytes; DEV_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, b, d_chunk_nl, d_chunk_off, (int)nchunks, stream)); }
            hipLaunchKernelGGL(k_large_scatter, dim3(nchunks), dim3(256), 0, stream, d_chunk_l, d_ch0, d_list[2], d_segs_a, d_prims_a, d_prims_b, d_lcb, d_lsplit, d_chunk_off);
        }
        DEV_TRY(hipGetLastError());
        std::swap(d_prims_a, d_prims_b);
        std::swap(d_segs_a, d_segs_b);
        base = next_base;
        nseg = 2u * nonleaf;
    }
    if ((size_t)base + nseg != nb) return chroma_internal_set_error(CHROMA_ERR_INTERNAL, "chroma_wide_build_device: %zu binary nodes made, %zu expected", (size_t)base + nseg, nb);
    t->dev_to_tri.resize(np);
    { const int rc_ = chroma_internal_dtoh(ctx, t->dev_to_tri.data(), d_dev_to_tri, (size_t)np * 4u); if (rc_ != CHROMA_OK) return rc_; }
    arena.release(d_prims_a); arena.release(d_prims_b); arena.release(d_segs_a); arena.release(d_segs_b); arena.release(d_flag); arena.release(d_rank);
    arena.release(d_dev_to_tri);
    for (int c = 0; c < 4; c++) arena.release(d_list[c]);
    arena.release(d_nch); arena.release(d_ch0); arena.release(d_lcb); arena.release(d_lbins); arena.release(d_lsplit);
    arena.release(d_chunk_l); arena.release(d_chunk_nl); arena.release(d_chunk_off);
    if (lap.on) fprintf(stderr, "[wide device] %zu levels of the binary tree\n", level_base.size());
    lap.lap("binary tree");

    // ---- boxes and the least-area table, bottom-up
    float *d_cost; uint32_t *d_share;
    DEV_TRY(arena.get(&d_cost, nb * 8)); DEV_TRY(arena.get(&d_share, nb));
    DEV_TRY(hipMemsetAsync(d_cost, 0, nb * 8 * sizeof(float), stream));
    for (size_t l = level_base.size(); l-- > 0;)
        hipLaunchKernelGGL(k_dp, dim3(blocks_for(level_count[l])), dim3(256), 0, stream, d_bin, d_cost, d_share, level_base[l], level_count[l]);
    DEV_TRY(hipGetLastError());
    DEV_TRY(hipStreamSynchronize(stream));
    arena.release(d_cost);
    lap.lap("boxes + least-area table");

    // ---- wide nodes, breadth first
    uint32_t *d_level, *d_next, *d_items, *d_info, *d_ninner, *d_first;
    uint32_t cnt = 1;
    size_t cap = 1;                       // capacity of the per-level arrays
    DEV_TRY(arena.get(&d_level, 1));
    { const uint32_t root = 0; DEV_TRY(hipMemcpyAsync(d_level, &root, 4, hipMemcpyHostToDevice, stream)); }
    size_t wbase = 0;
    uint32_t wdepth = 0;
    t->wnodes.reserve(((size_t)np / 3 + 1024) * 32);          // (address space only: a node per ~5.6 triangles at C3)
    d_items = d_info = d_ninner = d_first = nullptr;
    while (cnt) {
        if (wbase + cnt > 0x7FFFFFFFull) return chroma_internal_set_error(CHROMA_ERR_INVALID, "chroma_wide_build_device: more than 2^31 wide nodes");
        if (!d_items || cnt > cap) {
            arena.release(d_items); arena.release(d_info); arena.release(d_ninner); arena.release(d_first);
            cap = cnt;
            DEV_TRY(arena.get(&d_items, cap * 8)); DEV_TRY(arena.get(&d_info, cap)); DEV_TRY(arena.get(&d_ninner, cap)); DEV_TRY(arena.get(&d_first, cap));
        }
        hipLaunchKernelGGL(k_emit_count, dim3(blocks_for(cnt)), dim3(256), 0, stream, d_bin, d_share, d_level, cnt, d_items, d_info, d_ninner);
        { size_t b = tmp_bytes; DEV_TRY(hipcub::DeviceScan::ExclusiveSum(d_tmp, b, d_ninner, d_first, (int)cnt, stream)); }
        hipLaunchKernelGGL(k_last_sum, dim3(1), dim3(64), 0, stream, d_first, d_ninner, cnt, d_counters);
        uint32_t nnext = 0;
        DEV_TRY(hipMemcpyAsync(&nnext, d_counters, 4, hipMemcpyDeviceToHost, stream));
        DEV_TRY(hipStreamSynchronize(stream));
        uint4 *d_w;
        DEV_TRY(arena.get(&d_w, (size_t)cnt * 8));
        DEV_TRY(arena.get(&d_next, nnext));
        hipLaunchKernelGGL(k_emit_write, dim3(blocks_for((size_t)cnt * 8)), dim3(256), 0, stream, d_bin, d_items, d_info, d_first, cnt,
                           (uint32_t)(wbase + cnt), d_w, d_next);
        DEV_TRY(hipGetLastError());
        t->wnodes.resize((wbase + cnt) * 32);
        { const int rc_ = chroma_internal_dtoh(ctx, t->wnodes.data() + wbase * 32, d_w, (size_t)cnt * 128u); if (rc_ != CHROMA_OK) return rc_; }
        arena.release(d_w); arena.release(d_level);
        d_level = d_next;
        wbase += cnt;
        cnt = nnext;
        wdepth++;
    }
    t->nwide = wbase;
    t->depth = wdepth;
    lap.lap("wide nodes + download");
    chroma_host::finish_wide_tree(*t, ntriangles, false);
    lap.lap("record map (host)");
    guard.t = nullptr;
    *handle = t;
    if (nwide) *nwide = t->nwide;
    if (nrecords) *nrecords = t->dev_to_tri.size();
    if (depth) *depth = t->depth;
    return CHROMA_OK;
}

}  // extern "C"
