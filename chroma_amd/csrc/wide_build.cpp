// wide_build.cpp -- derive the 8-wide traversal tree from a reference-format BVH (host, threaded).
//
// The reference walks a tree of 16-byte nodes whose inner nodes have ~3 children
// (chroma/cuda/mesh.h:42-118, chroma/bvh/grid.py:11-95): a ray makes ~60 dependent fetches of
// 48-byte child ranges.  On MI355X a fetch moves a 128-byte L2 line whatever it needs, so the
// device walks a DERIVED tree instead: every wide node is one aligned 128-byte line holding eight
// child entries in the reference's own node format (x, y, z = lo16 | hi16 << 16 in the same
// fixed-point frame), with w = index of the child's wide node, or 0x80000000 | device triangle
// index.  A wide node is a reference node with its largest children replaced by THEIR children
// until eight entries are reached, so every box in it is a box of the reference tree and the
// tree stays conservative.
//
// What makes the result the reference's, bit for bit, although the visiting order differs: the
// nearest hit of a conservative tree is the minimum over all triangles of the Moeller-Trumbore
// distance; only exact ties depend on the order of the tests, and the reference keeps the first
// one tested (mesh.h:96-101, strict '<').  Its test order is a fixed total order on the leaves --
// in a range, leaves are tested as they come and inner children are walked afterwards, last
// pushed first (mesh.h:68-110) -- so every triangle gets that position as its `rank` and the wide
// walk breaks ties by rank.
#include "wide_build.h"
#include "host_utils.h"
#include <string.h>
#include <atomic>

namespace chroma_host {

static const uint32_t NCHILD_SHIFT = 28, CHILD_MASK = 0x0FFFFFFFu;

static inline double box_area(const uint32_t *n)
{
    double dx = (double)(n[0] >> 16) - (double)(n[0] & 0xFFFFu);
    double dy = (double)(n[1] >> 16) - (double)(n[1] & 0xFFFFu);
    double dz = (double)(n[2] >> 16) - (double)(n[2] & 0xFFFFu);
    return dx * dy + dy * dz + dz * dx;
}

// an entry of a wide node under construction: one reference node, or (synthetic) the tail
// [first, first+count) of a reference child range that has more than WIDE_K members
struct Entry { uint32_t first, count; };      // count == 0: the single reference node `first`

struct Item { uint32_t first, count; };       // the reference child range a wide node is made from

static int expand_item(const uint32_t *nodes, Item it, Entry *e)
{
    int n = 0;
    if (it.count > WIDE_K) {
        for (uint32_t j = 0; j < WIDE_K - 1; j++) e[n++] = Entry{it.first + j, 0};
        e[n++] = Entry{it.first + (WIDE_K - 1), it.count - (WIDE_K - 1)};
        return n;
    }
    for (uint32_t j = 0; j < it.count; j++) e[n++] = Entry{it.first + j, 0};
    for (;;) {
        int pick = -1;
        double best = -1.0;
        for (int i = 0; i < n; i++) {
            if (e[i].count) continue;
            const uint32_t *nd = nodes + 4 * (size_t)e[i].first;
            uint32_t k = nd[3] >> NCHILD_SHIFT;
            if (k == 0 || n - 1 + (int)k > WIDE_K) continue;
            double a = box_area(nd);
            if (a > best) { best = a; pick = i; }
        }
        if (pick < 0) break;
        const uint32_t *nd = nodes + 4 * (size_t)e[pick].first;
        uint32_t k = nd[3] >> NCHILD_SHIFT, c = nd[3] & CHILD_MASK;
        e[pick] = Entry{c, 0};
        for (uint32_t j = 1; j < k; j++) e[n++] = Entry{c + j, 0};
    }
    return n;
}

int build_wide_tree(const uint32_t *nodes, size_t nnodes, uint32_t ntriangles, WideTree &out, std::string &err)
{
    out = WideTree();
    if (!nodes || nnodes == 0) { err = "wide tree: no nodes"; return -1; }

    // ---- layers of the reference tree (root first, children behind their parents' layer); a tree
    // that is not stored that way (only child > parent is guaranteed) is swept sequentially instead
    std::vector<size_t> layer_start;
    bool layered = true;
    {
        size_t lo = 0, hi = 1;
        while (lo < hi && layered) {
            layer_start.push_back(lo);
            size_t next_hi = hi;
            for (size_t i = lo; i < hi; i++) {
                uint32_t w = nodes[4 * i + 3], k = w >> NCHILD_SHIFT, c = w & CHILD_MASK;
                if (k == 0) continue;
                if ((size_t)c + k > nnodes || c <= i) { err = "wide tree: bad child range"; return -1; }
                if (c < hi) { layered = false; break; }
                next_hi = std::max(next_hi, (size_t)c + k);
            }
            lo = hi; hi = next_hi;
        }
        layer_start.push_back(lo);
        if (!layered) {
            for (size_t i = 0; i < nnodes; i++) {
                uint32_t w = nodes[4 * i + 3], k = w >> NCHILD_SHIFT, c = w & CHILD_MASK;
                if (k && ((size_t)c + k > nnodes || c <= i)) { err = "wide tree: bad child range"; return -1; }
            }
            layer_start.assign({0, nnodes});
        }
    }
    const size_t nlayers = layer_start.size() - 1;
    const size_t nreach = layer_start[nlayers];

    // ---- reference test order: leaves under each node, then a rank for every triangle
    // (the array also holds nodes no walk reaches -- e.g. leaves whose content was moved up into a
    //  single-child parent, bvh.cu:530-543 -- which must not hand out ranks: base == UNREACHED)
    const uint32_t UNREACHED = 0xFFFFFFFFu;
    std::vector<uint32_t> leaves(nreach, 0), base(nreach, UNREACHED);
    base[0] = 0;
    out.rank.assign(ntriangles, 0xFFFFFFFFu);
    std::atomic<int> bad(0);
    auto count_leaves = [&](size_t i) {
        uint32_t w = nodes[4 * i + 3], k = w >> NCHILD_SHIFT, c = w & CHILD_MASK;
        if (k == 0) { leaves[i] = 1; return; }
        uint32_t s = 0;
        for (uint32_t j = 0; j < k; j++) s += leaves[c + j];
        leaves[i] = s;
    };
    auto assign_ranks = [&](size_t i) {
        uint32_t w = nodes[4 * i + 3], k = w >> NCHILD_SHIFT, c = w & CHILD_MASK;
        if (k == 0 || base[i] == UNREACHED) return;
        uint32_t run = base[i];
        for (uint32_t j = 0; j < k; j++) {                 // leaves of the range, in order
            uint32_t cw = nodes[4 * (size_t)(c + j) + 3];
            if ((cw >> NCHILD_SHIFT) == 0) {
                uint32_t t = cw & CHILD_MASK;
                if (t >= ntriangles) { bad = 1; continue; }
                if (out.rank[t] == 0xFFFFFFFFu) out.rank[t] = run;      // (a duplicate leaf keeps one of its ranks)
                run++;
            }
        }
        for (uint32_t j = k; j-- > 0;) {                   // then the inner children, last first
            uint32_t cw = nodes[4 * (size_t)(c + j) + 3];
            if ((cw >> NCHILD_SHIFT) != 0) { base[c + j] = run; run += leaves[c + j]; }
        }
    };
    if (layered) {
        for (size_t l = nlayers; l-- > 0;) {
            size_t lo = layer_start[l], hi = layer_start[l + 1];
            parallel_for(hi - lo, [&](size_t a, size_t b) { for (size_t i = lo + a; i < lo + b; i++) count_leaves(i); });
        }
        for (size_t l = 0; l < nlayers; l++) {
            size_t lo = layer_start[l], hi = layer_start[l + 1];
            parallel_for(hi - lo, [&](size_t a, size_t b) { for (size_t i = lo + a; i < lo + b; i++) assign_ranks(i); });
        }
    } else {
        for (size_t i = nnodes; i-- > 0;) count_leaves(i);
        for (size_t i = 0; i < nnodes; i++) assign_ranks(i);
    }
    if (bad) { err = "wide tree: leaf references a triangle outside the mesh"; return -1; }
    { std::vector<uint32_t>().swap(base); std::vector<uint32_t>().swap(leaves); }

    // ---- collapse, one level of wide nodes at a time
    out.tri_to_dev.assign(ntriangles, 0xFFFFFFFFu);
    out.dev_to_tri.clear();
    out.dev_to_tri.reserve(ntriangles);
    std::vector<Item> level, next;
    {
        uint32_t w = nodes[3];
        level.push_back(Item{w & CHILD_MASK, w >> NCHILD_SHIFT});      // a leaf root gives an empty node
    }
    size_t nwide_done = 0;          // wide nodes of earlier levels
    uint32_t depth = 0;
    while (!level.empty()) {
        const size_t n = level.size();
        const size_t first_index = nwide_done;                         // this level occupies [first_index, first_index + n)
        if (first_index + n > 0x7FFFFFFFull) { err = "wide tree: more than 2^31 nodes"; return -1; }
        out.wnodes.resize((first_index + n) * 32);
        std::vector<Entry> ent(n * WIDE_K);
        std::vector<uint8_t> nent(n);
        std::vector<uint32_t> ninner(n + 1, 0), nleaf(n + 1, 0);
        parallel_for(n, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                Entry *e = ent.data() + i * WIDE_K;
                int m = expand_item(nodes, level[i], e);
                nent[i] = (uint8_t)m;
                uint32_t ni = 0, nl = 0;
                for (int j = 0; j < m; j++) {
                    if (e[j].count) ni++;
                    else if ((nodes[4 * (size_t)e[j].first + 3] >> NCHILD_SHIFT) == 0) nl++;
                    else ni++;
                }
                ninner[i + 1] = ni; nleaf[i + 1] = nl;
            }
        }, 1u << 12);
        for (size_t i = 0; i < n; i++) { ninner[i + 1] += ninner[i]; nleaf[i + 1] += nleaf[i]; }
        const size_t child_index0 = first_index + n;                   // the next level starts here
        const size_t dev0 = out.dev_to_tri.size();
        if (child_index0 + ninner[n] > 0x7FFFFFFFull || dev0 + nleaf[n] > 0x7FFFFFFFull) { err = "wide tree: index overflow"; return -1; }
        next.assign(ninner[n], Item{0, 0});
        out.dev_to_tri.resize(dev0 + nleaf[n]);
        parallel_for(n, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                const Entry *e = ent.data() + i * WIDE_K;
                uint32_t *wn = out.wnodes.data() + (first_index + i) * 32;
                uint32_t ki = ninner[i], kl = nleaf[i];
                for (int j = 0; j < WIDE_K; j++) {
                    uint32_t *o = wn + 4 * j;
                    if (j >= nent[i]) { o[0] = o[1] = o[2] = 0x0000FFFFu; o[3] = WIDE_EMPTY; continue; }   // inverted box
                    if (e[j].count) {                                   // tail of an over-wide range
                        uint32_t lo[3] = {0xFFFF, 0xFFFF, 0xFFFF}, hi[3] = {0, 0, 0};
                        for (uint32_t c = 0; c < e[j].count; c++) {
                            const uint32_t *nd = nodes + 4 * (size_t)(e[j].first + c);
                            for (int ax = 0; ax < 3; ax++) { lo[ax] = std::min(lo[ax], nd[ax] & 0xFFFFu); hi[ax] = std::max(hi[ax], nd[ax] >> 16); }
                        }
                        for (int ax = 0; ax < 3; ax++) o[ax] = lo[ax] | hi[ax] << 16;
                        o[3] = (uint32_t)(child_index0 + ki);
                        next[ki++] = Item{e[j].first, e[j].count};
                        continue;
                    }
                    const uint32_t *nd = nodes + 4 * (size_t)e[j].first;
                    o[0] = nd[0]; o[1] = nd[1]; o[2] = nd[2];
                    uint32_t k = nd[3] >> NCHILD_SHIFT, c = nd[3] & CHILD_MASK;
                    if (k == 0) {
                        uint32_t dev = (uint32_t)(dev0 + kl++);
                        out.dev_to_tri[dev] = c;
                        o[3] = WIDE_LEAF | dev;
                    } else {
                        o[3] = (uint32_t)(child_index0 + ki);
                        next[ki++] = Item{c, k};
                    }
                }
            }
        }, 1u << 12);
        nwide_done += n;
        level.swap(next);
        depth++;
    }
    out.nwide = nwide_done;
    out.depth = depth;
    // worst case of the walk's stack: at a node, every inner child but the one walked next is
    // pushed, then the same below -- whichever child is walked, so the maximum over children
    {
        std::vector<uint16_t> need(out.nwide, 0);
        for (size_t i = out.nwide; i-- > 0;) {
            const uint32_t *wn = out.wnodes.data() + i * 32;
            uint32_t inner = 0, below = 0;
            for (int j = 0; j < WIDE_K; j++) {
                uint32_t w = wn[4 * j + 3];
                if (w == WIDE_EMPTY || (w & WIDE_LEAF)) continue;
                inner++;
                below = std::max<uint32_t>(below, need[w]);
            }
            need[i] = (uint16_t)std::min<uint32_t>(0xFFFF, inner ? inner - 1 + below : 0);
        }
        out.stack_need = out.nwide ? need[0] : 0;
    }
    // device index of every triangle (a triangle under several leaves keeps the first; triangles
    // under no leaf go to the end so that every triangle has a record)
    for (size_t d = 0; d < out.dev_to_tri.size(); d++) {
        uint32_t t = out.dev_to_tri[d];
        if (out.tri_to_dev[t] == 0xFFFFFFFFu) out.tri_to_dev[t] = (uint32_t)d;
    }
    for (uint32_t t = 0; t < ntriangles; t++)
        if (out.tri_to_dev[t] == 0xFFFFFFFFu) { out.tri_to_dev[t] = (uint32_t)out.dev_to_tri.size(); out.dev_to_tri.push_back(t); }
    return 0;
}

}  // namespace chroma_host

// ---- C ABI (include/chroma_hip.h) ---------------------------------------------------------------
#include "../../include/chroma_hip.h"
extern "C" {
int chroma_wide_build(const uint32_t *nodes, uint64_t nnodes, uint32_t ntriangles, void **handle,
                      uint64_t *nwide, uint64_t *nrecords, uint32_t *depth)
{
    if (!nodes || !handle) return CHROMA_ERR_INVALID;
    chroma_host::WideTree *t = new chroma_host::WideTree;
    std::string err;
    if (chroma_host::build_wide_tree(nodes, (size_t)nnodes, ntriangles, *t, err) != 0) { delete t; return CHROMA_ERR_INVALID; }
    *handle = t;
    if (nwide) *nwide = t->nwide;
    if (nrecords) *nrecords = t->dev_to_tri.size();
    if (depth) *depth = t->depth;
    return CHROMA_OK;
}
int chroma_wide_data(void *handle, const uint32_t **wnodes, const uint32_t **tri_to_record,
                     const uint32_t **record_to_tri, const uint32_t **rank)
{
    if (!handle) return CHROMA_ERR_INVALID;
    chroma_host::WideTree *t = (chroma_host::WideTree *)handle;
    if (wnodes) *wnodes = t->wnodes.data();
    if (tri_to_record) *tri_to_record = t->tri_to_dev.data();
    if (record_to_tri) *record_to_tri = t->dev_to_tri.data();
    if (rank) *rank = t->rank.data();
    return CHROMA_OK;
}
int chroma_wide_free(void *handle)
{
    delete (chroma_host::WideTree *)handle;
    return CHROMA_OK;
}
}
