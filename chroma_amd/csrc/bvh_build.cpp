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

#include "../../include/chroma_hip.h"

namespace {

struct Node { uint32_t x, y, z, w; };
const int MAX_CHILD = 15;   // 2^(32-28) - 1, chroma/bvh/grid.py:6

struct BvhResult {
    std::vector<Node> nodes;
    std::vector<uint64_t> layer_bounds;   // nlayers + 1 entries
};

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
    volatile float d = v - origin;     // keep the two roundings separate (no contraction)
    volatile float q = d / scale;
    return (uint32_t)q;
}

template <class F>
void parallel_for(size_t n, F f)
{
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
    if (n < (1u << 16)) nt = 1;
    if (nt == 1) { f(0, n); return; }
    std::vector<std::thread> th;
    size_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; t++) {
        size_t lo = std::min(n, (size_t)t * chunk), hi = std::min(n, lo + chunk);
        if (lo < hi) th.emplace_back([=] { f(lo, hi); });
    }
    for (auto &t : th) t.join();
}

// stable LSD radix sort of 48-bit keys with a 32-bit payload (16 bits per pass)
void radix_sort_48(std::vector<uint64_t> &keys, std::vector<uint32_t> &vals)
{
    size_t n = keys.size();
    std::vector<uint64_t> k2(n);
    std::vector<uint32_t> v2(n);
    for (int pass = 0; pass < 3; pass++) {
        int shift = 16 * pass;
        std::vector<size_t> count(65536 + 1, 0);
        for (size_t i = 0; i < n; i++) count[((keys[i] >> shift) & 0xFFFF) + 1]++;
        for (size_t b = 0; b < 65536; b++) count[b + 1] += count[b];
        for (size_t i = 0; i < n; i++) {
            size_t dst = count[(keys[i] >> shift) & 0xFFFF]++;
            k2[dst] = keys[i];
            v2[dst] = vals[i];
        }
        keys.swap(k2);
        vals.swap(v2);
    }
}

size_t count_unique_sorted_shifted(const std::vector<uint64_t> &m, int shift)
{
    size_t n = m.size();
    if (n == 0) return 0;
    std::vector<size_t> partial(64, 0);
    unsigned nt = std::max(1u, std::min(std::thread::hardware_concurrency(), 32u));
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
    for (size_t i = 0; i < (size_t)ntriangles * 3; i += 1)
        if (triangles[i] >= nvertices) return CHROMA_ERR_INVALID;

    const float ox = world_origin[0], oy = world_origin[1], oz = world_origin[2];
    size_t n = ntriangles;
    std::vector<Node> leaves(n);
    std::vector<uint64_t> morton(n);
    std::vector<uint32_t> order(n);

    // make_leaves (cuda/bvh.cu:149-203)
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            const float *a = vertices + 3 * (size_t)triangles[3 * i];
            const float *b = vertices + 3 * (size_t)triangles[3 * i + 1];
            const float *c = vertices + 3 * (size_t)triangles[3 * i + 2];
            float lower[3], upper[3], cen[3];
            for (int k = 0; k < 3; k++) {
                lower[k] = fminf(fminf(a[k], b[k]), c[k]);
                upper[k] = fmaxf(fmaxf(a[k], b[k]), c[k]);
                volatile float s1 = a[k] + b[k];
                volatile float s2 = s1 + c[k];
                volatile float s3 = s2 / 3.0f;
                cen[k] = s3;
            }
            const float org[3] = {ox, oy, oz};
            uint32_t ql[3], qu[3], qc[3];
            for (int k = 0; k < 3; k++) {
                ql[k] = quantize(lower[k], org[k], world_scale);
                if (ql[k] > 0) ql[k]--;
                qu[k] = quantize(upper[k], org[k], world_scale) + 1;
                qc[k] = quantize(cen[k], org[k], world_scale);
            }
            morton[i] = spread3_16(qc[0]) | (spread3_16(qc[1]) << 1) | (spread3_16(qc[2]) << 2);
            leaves[i] = Node{ql[0] | (qu[0] << 16), ql[1] | (qu[1] << 16), ql[2] | (qu[2] << 16), (uint32_t)i};
            order[i] = (uint32_t)i;
        }
    });

    // Morton order (grid.py:26-28; stable, so equal codes keep triangle order)
    radix_sort_48(morton, order);
    std::vector<Node> sorted(n);
    parallel_for(n, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) sorted[i] = leaves[order[i]]; });
    leaves.clear(); leaves.shrink_to_fit();
    order.clear(); order.shrink_to_fit();

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

int chroma_bvh_free(void *handle)
{
    delete (BvhResult *)handle;
    return CHROMA_OK;
}

}  // extern "C"
