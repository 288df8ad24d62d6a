// bvh_build.cpp -- host-side recursive-grid BVH builder (multi-threaded C++).
//
// Replaces the GPU-assisted builder of the reference: chroma/bvh/grid.py:11-95 together with
// the kernels make_leaves (chroma/cuda/bvh.cu:149-203), make_parents_detailed (:270-308),
// copy_and_offset (:365-384), collapse_child (:530-543) and their host drivers
// (chroma/gpu/bvh.py:18-130,239-267).  Everything after the float32 quantisation of the
// triangle boxes is integer arithmetic, so the result is fully determined by the mesh; the
// NumPy restatement in chroma_amd/bvh/grid.py must produce identical nodes (tests/test_bvh.py).
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <algorithm>
#include <thread>
#include <vector>
#include <chrono>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/chroma_hip.h"
#include "host_utils.h"
#include "bvh_result.h"

namespace {
struct Timer {
    std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
    bool on = getenv("CHROMA_BVH_VERBOSE") != nullptr;
    void lap(const char *what) {
        auto t1 = std::chrono::steady_clock::now();
        if (on) fprintf(stderr, "[bvh] %-28s %.2f s\n", what, std::chrono::duration<double>(t1 - t0).count());
        t0 = t1;
    }
};

using chroma_host::Node;
using chroma_host::BvhResult;
const int MAX_CHILD = 15;   // 2^(32-28) - 1, chroma/bvh/grid.py:6

inline uint64_t spread3_16(uint32_t input)   // cuda/bvh.cu:42-52
{
    uint64_t x = input;
    x = (x | (x << 16)) & 0x00000000FF0000FFull;
    x = (x | (x << 8)) & 0x000000F00F00F00Full;
    x = (x | (x << 4)) & 0x00000C30C30C30C3ull;
    x = (x | (x << 2)) & 0x0000249249249249ull;
    return x;
}

inline uint32_t quantize(float v, float origin, float scale)   // cuda/bvh.cu:65-69: truncate
{
    float d = v - origin;              // -ffp-contract=off: subtraction and division round separately
    float q = d / scale;
    return (uint32_t)q;
}

using chroma_host::parallel_for;

// stable sort of (Morton code, triangle id) records: three 16-bit LSD passes on all cores
struct MortonRec { uint64_t code; uint32_t id; uint32_t pad; };
void sort_by_morton(std::vector<MortonRec> &recs)
{
    size_t n = recs.size();
    std::vector<MortonRec> tmp(n);
    MortonRec *in = recs.data(), *out = tmp.data();
    for (int pass = 0; pass < 3; pass++) {
        int shift = 16 * pass;
        bool skipped = chroma_host::radix_pass16(n, in, out, [shift](const MortonRec &r) { return (uint32_t)((r.code >> shift) & 0xFFFFu); });
        if (!skipped) std::swap(in, out);
    }
    if (in != recs.data()) memcpy(recs.data(), in, n * sizeof(MortonRec));
}

size_t count_unique_sorted_shifted(const std::vector<uint64_t> &m, int shift)
{
    size_t n = m.size();
    if (n == 0) return 0;
    std::vector<size_t> partial(64, 0);
    unsigned nt = std::max(1u, std::min(chroma_host::hw_threads(), 32u));
    if (n < (1u << 16)) nt = 1;
    std::vector<std::thread> th;
    size_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        size_t lo = std::max((size_t)1, std::min(n, (size_t)t * chunk)), hi = std::min(n, (size_t)(t + 1) * chunk);
        if (lo >= hi) continue;
        th.emplace_back([&m, &partial, t, lo, hi, shift] {
            size_t c = 0;
            for (size_t i = lo; i < hi; i++) c += ((m[i] >> shift) != (m[i - 1] >> shift));
            partial[t] = c;
        });
    }
    for (auto &t : th) t.join();
    size_t total = 1;
    for (size_t c : partial) total += c;
    return total;
}

}  // namespace

extern "C" {

int chroma_bvh_build(const float *vertices, uint32_t nvertices, const uint32_t *triangles, uint32_t ntriangles,
                     const float world_origin[3], float world_scale, int32_t target_degree,
                     void **handle, uint64_t *nnodes, uint32_t *nlayers)
{
    if (!vertices || !triangles || !handle || ntriangles == 0 || ntriangles >= (1u << CHROMA_CHILD_BITS) || target_degree < 1)
        return CHROMA_ERR_INVALID;
    {
        std::vector<int> bad(1, 0);
        chroma_host::parallel_for((size_t)ntriangles * 3, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) if (triangles[i] >= nvertices) { bad[0] = 1; return; }
        });
        if (bad[0]) return CHROMA_ERR_INVALID;
    }

    Timer timer;
    timer.lap("validate");
    const float ox = world_origin[0], oy = world_origin[1], oz = world_origin[2];
    size_t n = ntriangles;
    std::vector<uint64_t> morton(n);
    std::vector<MortonRec> recs(n);

    timer.lap("allocate");
    // make_leaves (cuda/bvh.cu:149-203), split in two: Morton codes now, the boxes once the
    // leaves' final order is known (saves one 16-byte-per-triangle array)
    const float org[3] = {ox, oy, oz};
    auto leaf_of = [&](size_t i, uint64_t *code) -> Node {
        const float *a = vertices + 3 * (size_t)triangles[3 * i];
        const float *b = vertices + 3 * (size_t)triangles[3 * i + 1];
        const float *c = vertices + 3 * (size_t)triangles[3 * i + 2];
        uint32_t ql[3], qu[3], qc[3];
        for (int k = 0; k < 3; k++) {
            float lower = fminf(fminf(a[k], b[k]), c[k]);
            float upper = fmaxf(fmaxf(a[k], b[k]), c[k]);
            float s1 = a[k] + b[k];        // compiled with -ffp-contract=off: three separate roundings
            float s2 = s1 + c[k];
            float cen = s2 / 3.0f;
            ql[k] = quantize(lower, org[k], world_scale);
            if (ql[k] > 0) ql[k]--;
            qu[k] = quantize(upper, org[k], world_scale) + 1;
            qc[k] = quantize(cen, org[k], world_scale);
        }
        if (code) *code = spread3_16(qc[0]) | (spread3_16(qc[1]) << 1) | (spread3_16(qc[2]) << 2);
        return Node{ql[0] | (qu[0] << 16), ql[1] | (qu[1] << 16), ql[2] | (qu[2] << 16), (uint32_t)i};
    };
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            leaf_of(i, &recs[i].code);
            recs[i].id = (uint32_t)i;
            recs[i].pad = 0;
        }
    });

    timer.lap("make_leaves");
    // Morton order (grid.py:26-28; stable, so equal codes keep triangle order)
    sort_by_morton(recs);
    timer.lap("morton sort");
    std::vector<Node> sorted(n);
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) { sorted[i] = leaf_of(recs[i].id, nullptr); morton[i] = recs[i].code; }
    });
    std::vector<MortonRec>().swap(recs);
    timer.lap("morton sort + gather");

    std::vector<std::vector<Node>> layers;   // leaves first here, reversed at the end
    layers.push_back(std::move(sorted));

    while (layers.back().size() > 1) {
        const std::vector<Node> &top = layers.back();
        size_t nn = top.size();
        // grid.py:37-42: shift until the mean group size reaches target_degree
        int shift = 0;
        size_t nunique = count_unique_sorted_shifted(morton, 0);
        while ((double)nn / (double)nunique < (double)target_degree && nunique > 1) {
            shift++;
            nunique = count_unique_sorted_shifted(morton, shift);
        }
        // grid.py:45-76: one parent per run of equal codes, runs cut at MAX_CHILD
        std::vector<uint32_t> first_child;
        std::vector<uint64_t> parent_morton;
        first_child.reserve(nunique + nunique / 8);
        parent_morton.reserve(nunique + nunique / 8);
        size_t run_start = 0;
        for (size_t i = 1; i <= nn; i++) {
            if (i == nn || (morton[i] >> shift) != (morton[i - 1] >> shift)) {
                for (size_t f = run_start; f < i; f += MAX_CHILD) {
                    first_child.push_back((uint32_t)f);
                    parent_morton.push_back(morton[run_start] >> shift);
                }
                run_start = i;
            }
        }
        size_t np = first_child.size();
        std::vector<Node> parents(np);
        // make_parents_detailed (cuda/bvh.cu:270-308)
        parallel_for(np, [&](size_t lo, size_t hi) {
            for (size_t p = lo; p < hi; p++) {
                size_t f = first_child[p];
                size_t e = (p + 1 < np) ? first_child[p + 1] : nn;
                uint32_t l[3] = {0xFFFFu, 0xFFFFu, 0xFFFFu}, u[3] = {0, 0, 0};
                for (size_t c = f; c < e; c++) {
                    const Node &ch = top[c];
                    const uint32_t w[3] = {ch.x, ch.y, ch.z};
                    for (int k = 0; k < 3; k++) {
                        l[k] = std::min(l[k], w[k] & 0xFFFFu);
                        u[k] = std::max(u[k], w[k] >> 16);
                    }
                }
                parents[p] = Node{l[0] | (u[0] << 16), l[1] | (u[1] << 16), l[2] | (u[2] << 16),
                                  ((uint32_t)(e - f) << CHROMA_CHILD_BITS) | (uint32_t)f};
            }
        });
        morton.swap(parent_morton);
        layers.push_back(std::move(parents));
    }

    timer.lap("parent layers");
    // concatenate_layers (gpu/bvh.py:239-267): root first; child index += start of next layer
    BvhResult *res = new BvhResult;
    size_t nl = layers.size();
    res->layer_bounds.assign(nl + 1, 0);
    for (size_t l = 0; l < nl; l++) res->layer_bounds[l + 1] = res->layer_bounds[l] + layers[nl - 1 - l].size();
    uint64_t total = res->layer_bounds[nl];
    if (total >= (1ull << CHROMA_CHILD_BITS)) { delete res; return CHROMA_ERR_INVALID; }
    res->nodes.resize(total);
    for (size_t l = 0; l < nl; l++) {
        std::vector<Node> &src = layers[nl - 1 - l];
        uint64_t base = res->layer_bounds[l];
        uint32_t offset = (l + 1 < nl) ? (uint32_t)res->layer_bounds[l + 1] : 0u;   // leaves keep triangle ids
        Node *dst = res->nodes.data() + base;
        parallel_for(src.size(), [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) { dst[i] = src[i]; dst[i].w += offset; }
        });
        std::vector<Node>().swap(src);
    }
    // collapse_chains (gpu/bvh.py:114-130, cuda/bvh.cu:530-543): bottom-up over the inner layers
    for (size_t l = nl - 1; l-- > 0;) {
        Node *nodes = res->nodes.data();
        uint64_t lo0 = res->layer_bounds[l], hi0 = res->layer_bounds[l + 1];
        parallel_for((size_t)(hi0 - lo0), [&](size_t lo, size_t hi) {
            for (size_t i = lo0 + lo; i < lo0 + hi; i++) {
                uint32_t w = nodes[i].w;
                if ((w >> CHROMA_CHILD_BITS) == 1) nodes[i] = nodes[w & ~CHROMA_NCHILD_MASK];
            }
        });
    }
    timer.lap("concatenate + collapse");
    *handle = res;
    if (nnodes) *nnodes = total;
    if (nlayers) *nlayers = (uint32_t)nl;
    return CHROMA_OK;
}

int chroma_bvh_fetch(void *handle, uint32_t *nodes_out, uint64_t *layer_bounds_out)
{
    if (!handle) return CHROMA_ERR_INVALID;
    BvhResult *res = (BvhResult *)handle;
    if (nodes_out) memcpy(nodes_out, res->nodes.data(), res->nodes.size() * sizeof(Node));
    if (layer_bounds_out) memcpy(layer_bounds_out, res->layer_bounds.data(), res->layer_bounds.size() * sizeof(uint64_t));
    return CHROMA_OK;
}

int chroma_bvh_data(void *handle, const uint32_t **nodes, const uint64_t **layer_bounds)
{
    if (!handle) return CHROMA_ERR_INVALID;
    BvhResult *res = (BvhResult *)handle;
    if (nodes) *nodes = (const uint32_t *)res->nodes.data();
    if (layer_bounds) *layer_bounds = res->layer_bounds.data();
    return CHROMA_OK;
}

int chroma_bvh_free(void *handle)
{
    delete (BvhResult *)handle;
    return CHROMA_OK;
}

}  // extern "C"
