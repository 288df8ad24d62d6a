// wide_build.h -- the derived 8-wide traversal tree (see wide_build.cpp).
#pragma once
#include <stdlib.h>
#include <stdint.h>
#include <stddef.h>
#include <string>
#include <vector>
#include "uninit_vector.h"

namespace chroma_host {

enum : uint32_t { WIDE_K = 8, WIDE_LEAF = 0x80000000u, WIDE_EMPTY = 0xFFFFFFFFu };
// how the hierarchy above the reference's leaf boxes is chosen
enum { WIDE_TOPOLOGY_COLLAPSE = 0,     // the reference tree with children pulled up until a node has eight
       WIDE_TOPOLOGY_SAH = 1,          // rebuilt with surface-area-heuristic splits, binary tree collapsed to wide nodes at the least total area (rounds 2-3)
       WIDE_TOPOLOGY_SAH_GREEDY = 2,   // the same splits, a wide node = a set split greedily until it has eight parts (round 1)
       WIDE_TOPOLOGY_PLOC = 3,       // bottom-up: parallel locally-ordered clustering of the Morton-ordered leaves (Meister & Bittner 2018), then the
                                       // same least-area collapse; breadth-first node order.  Every step is a data-parallel pass over an array -- the
                                       // shape a DEVICE builder would take -- but the tree is 3.5 % slower to walk at C3 (profiles/r03/ab_tree_ploc.txt): not the default
       WIDE_TOPOLOGY_LEVELS = 4 };     // the SAH splits of (1) with nothing left to the schedule: one binary tree over all triangles, stable
                                       // partitions, the least-area collapse over the whole tree, breadth-first node order -- what the device
                                       // builder (csrc/wide_device.hip) makes, bit for bit.  THE DEFAULT.

typedef uninit_vector<uint32_t> WordBuffer;       // (4 GB of wide nodes at C3: see uninit_vector.h)

struct WideTree {
    WordBuffer wnodes;       // nwide * 8 entries of 4 words: x, y, z boxes, w = child (see above)
    std::vector<uint32_t> tri_to_dev;   // [ntriangles] device record index of a triangle
    std::vector<uint32_t> dev_to_tri;   // [nrecords >= ntriangles] triangle of a device record
    std::vector<uint32_t> rank;         // [ntriangles] position in the reference's test order (0xFFFFFFFF: under no leaf)
    size_t nwide = 0;
    uint32_t depth = 0;                 // levels of wide nodes
    uint32_t stack_need = 0;            // most entries the nearest-first walk can hold at once
};

// nodes: reference-format BVH (4 words per node, root first, children of a node contiguous and
// stored after every node of their parent's layer).  Returns 0, or -1 with `err` set.
int build_wide_tree(const uint32_t *nodes, size_t nnodes, uint32_t ntriangles, WideTree &out, std::string &err,
                    int topology = WIDE_TOPOLOGY_SAH);
// The pieces of build_wide_tree that the device builder (csrc/wide_device.hip) shares with it:
// the reference's test order -- `rank` of every triangle and the reachable leaf that holds it (0xFFFFFFFF: none) --
int reference_test_order(const uint32_t *nodes, size_t nnodes, uint32_t ntriangles, std::vector<uint32_t> &rank,
                         std::vector<uint32_t> &leaf_node, std::string &err, size_t *nlayers = nullptr, bool *layered = nullptr);
// and, once wnodes / nwide / depth / dev_to_tri are made, the stack need and the record maps.
void finish_wide_tree(WideTree &out, uint32_t ntriangles, bool with_stack_need = true);
// most entries the nearest-first walk over this tree can hold at once (children follow their parents)
uint32_t wide_stack_need(const uint32_t *wnodes, size_t nwide);
// Index checks of a wide tree and its record maps (see wide_build.cpp).  Returns 0, or -1 with `err` set.
int validate_wide_tree(const uint32_t *wnodes, size_t nwide, const uint32_t *tri_to_dev, uint32_t ntriangles,
                       const uint32_t *dev_to_tri, size_t nrecords, std::string &err);
// CHROMA_TREE=collapse|greedy|sah|ploc|levels (default levels: the one topology the device builder makes too)
int wide_topology_from_env();
// search radius of the PLOC nearest-neighbour step (clusters to either side in Morton order)
enum { PLOC_RADIUS = 16 };

}  // namespace chroma_host

// ---- the leaf boxes the derived tree is built over ---------------------------------------------------------------------------
// The reference pads a leaf's LOWER bound by a whole quantum (cuda/bvh.cu:181: ql = trunc((min - origin) / scale), then ql-- when
// ql > 0) while its upper bound is the plain ceiling (trunc + 1).  A triangle lies inside [trunc(min), trunc(max) + 1] by
// construction, so the derived tree could do without that quantum: CHROMA_TIGHT_LEAVES=1 moves the lower bounds up by one
// (a stored 0 stays: it may be an unpadded 0) -- 8.27 -> 7.56 triangle tests per ray step at C3, step -2.2 %
// (profiles/r04/ab_tight_leaves.txt).  OFF by default: the padding also covers Moeller-Trumbore results of grazing rays that
// land up to a quantum off their triangle yet inside the reference's leaf box -- hits the reference keeps -- and without it the
// aimed-ray sweep of C3 finds one ray in 3.6e5 on which the default walk then differs from the reference INSIDE the leaf box.
// The default walk's deviation class stays "hits outside the reference's own leaf box" (DESIGN.md section 4.1).
#ifdef __HIPCC__
__host__ __device__
#endif
inline uint32_t wide_tight_bound_word(uint32_t w, int tight)
{
    const uint32_t lo = w & 0xFFFFu;
    return (tight && lo > 0u && lo + 1u <= (w >> 16)) ? w + 1u : w;
}
inline int wide_tight_leaves()
{
    const char *e = getenv("CHROMA_TIGHT_LEAVES");
    return e && e[0] == '1';
}
