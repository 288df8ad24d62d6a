// mesh_utils.cpp -- host-side helpers for flattening large meshes (multi-threaded C++).
//
// chroma_dedupe_vertices does what Mesh.remove_duplicate_vertices does in the reference
// (chroma/geometry.py:58-67, called from Geometry.flatten, :367): merge identical vertices,
// leave the survivors in lexicographic (x, y, z) order and remap the triangle indices.  The
// reference does it with np.unique on a structured view (minutes for the 29k-PMT geometry);
// here it is an LSD radix sort of the three float keys on all host cores.
#include <stdint.h>
#include <string.h>
#include <vector>

#include "../../include/chroma_hip.h"
#include "host_utils.h"

using namespace chroma_host;

namespace {
// monotone map float -> uint32 (so unsigned order == float order), -0.0 treated as +0.0
inline uint32_t sortable(float f)
{
    uint32_t b;
    memcpy(&b, &f, 4);
    if (b == 0x80000000u) b = 0;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
}  // namespace

extern "C" int chroma_dedupe_vertices(const float *vertices, uint64_t nvertices, uint32_t *triangle_indices,
                                      uint64_t nindices, float *unique_out, uint64_t *nunique)
{
    if (!vertices || !unique_out || !nunique || nvertices >= 0xFFFFFFFFull) return CHROMA_ERR_INVALID;
    size_t n = (size_t)nvertices;
    if (n == 0) { *nunique = 0; return CHROMA_OK; }
    struct Rec { uint32_t k[3]; uint32_t id; };
    std::vector<Rec> a(n), b(n);
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            for (int k = 0; k < 3; k++) a[i].k[k] = sortable(vertices[3 * i + k]);
            a[i].id = (uint32_t)i;
        }
    });
    Rec *in = a.data(), *out = b.data();
    // least significant first: z, y, x; two 16-bit digits each
    for (int comp = 2; comp >= 0; comp--)
        for (int half = 0; half < 2; half++) {
            int shift = 16 * half;
            bool skipped = radix_pass16(n, in, out, [comp, shift](const Rec &r) { return (r.k[comp] >> shift) & 0xFFFFu; });
            if (!skipped) std::swap(in, out);
        }
    // `in` is now sorted; mark group starts
    std::vector<uint32_t> group(n);
    std::vector<uint8_t> first(n);
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            if (i == 0) { first[i] = 1; continue; }
            const uint32_t *p = in[i].k, *q = in[i - 1].k;
            first[i] = (p[0] != q[0] || p[1] != q[1] || p[2] != q[2]);
        }
    });
    uint32_t g = 0;
    for (size_t i = 0; i < n; i++) {      // serial scan (memory bound, ~n bytes)
        g += first[i];
        group[i] = g - 1;
    }
    size_t nu = g;
    std::vector<uint32_t> inverse(n);
    parallel_for(n, [&](size_t lo, size_t hi) {
        for (size_t i = lo; i < hi; i++) {
            uint32_t id = in[i].id;
            inverse[id] = group[i];
            if (first[i]) memcpy(unique_out + 3 * (size_t)group[i], vertices + 3 * (size_t)id, 12);
        }
    });
    if (triangle_indices) {
        for (uint64_t i = 0; i < nindices; i++)
            if (triangle_indices[i] >= n) return CHROMA_ERR_INVALID;
        parallel_for((size_t)nindices, [&](size_t lo, size_t hi) {
            for (size_t i = lo; i < hi; i++) triangle_indices[i] = inverse[triangle_indices[i]];
        });
    }
    *nunique = nu;
    return CHROMA_OK;
}
