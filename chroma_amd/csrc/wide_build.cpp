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
#include <stdlib.h>
#include <atomic>
#include <chrono>
#include <stdio.h>
#include <thread>

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


// ---- topology by surface-area heuristic ------------------------------------------------------------
// The reference's tree is a fixed Morton-grid hierarchy (chroma/bvh/grid.py:11-95); only its LEAF
// boxes matter for the result (see the header).  This builder keeps those leaf boxes -- one per
// triangle, quantised and padded by the reference's own rule (bvh.cu:149-203) -- and puts a new
// hierarchy on top of them: SAH splits (32 bins on 3 axes; an exact sweep for sets of up to 32), a wide node being a set that is
// split in two, then its larger parts again, until it has eight parts.  Inner boxes are unions of
// leaf boxes on the same 16-bit grid, so the tree stays conservative.  A ray visits ~20 % fewer nodes
// and tests ~12 % fewer triangles than in the collapsed reference tree.
// Parallel in two stages: sets larger than `task_size` are split by all threads together; the
// others are independent tasks.  Node order (top part, then tasks in creation order) and triangle
// record order (the final position in the partitioned array) do not depend on thread timing.
struct Prim { uint16_t lo[3], hi[3]; uint32_t tri; };

struct IBox {
    uint32_t lo[3], hi[3];
    IBox() { clear(); }
    void clear() { for (int a = 0; a < 3; a++) { lo[a] = 0xFFFFFFFFu; hi[a] = 0; } }
    void add(const Prim &p) { for (int a = 0; a < 3; a++) { lo[a] = std::min<uint32_t>(lo[a], p.lo[a]); hi[a] = std::max<uint32_t>(hi[a], p.hi[a]); } }
    void add(const IBox &b) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], b.lo[a]); hi[a] = std::max(hi[a], b.hi[a]); } }
    bool empty() const { return lo[0] > hi[0]; }
    double area() const
    {
        if (empty()) return 0.0;
        double dx = (double)(hi[0] - lo[0]), dy = (double)(hi[1] - lo[1]), dz = (double)(hi[2] - lo[2]);
        return dx * dy + dy * dz + dz * dx;
    }
};

#ifndef SAH_BINS_N
#define SAH_BINS_N 32
#endif
static const int SAH_BINS = SAH_BINS_N;
static const size_t SWEEP_MAX = 32;      // sets up to this size are split by an exact sweep
struct Bins {
    IBox box[3][SAH_BINS];
    uint32_t count[3][SAH_BINS];
    IBox cbox;      // bounds of the doubled centroids lo+hi
    void clear() { for (int a = 0; a < 3; a++) for (int k = 0; k < SAH_BINS; k++) { box[a][k].clear(); count[a][k] = 0; } }
};
static inline uint32_t cent2(const Prim &p, int a) { return (uint32_t)p.lo[a] + (uint32_t)p.hi[a]; }
static inline int bin_of(uint32_t c2, uint32_t cmin, uint32_t ext) { return (int)(((uint64_t)(c2 - cmin) * SAH_BINS) / ((uint64_t)ext + 1)); }

static void centroid_bounds(const Prim *p, size_t n, IBox &cb)
{
    cb.clear();
    for (size_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) { uint32_t c = cent2(p[i], a); cb.lo[a] = std::min(cb.lo[a], c); cb.hi[a] = std::max(cb.hi[a], c); }
}
static void fill_bins(const Prim *p, size_t n, const IBox &cb, Bins &b)
{
    for (size_t i = 0; i < n; i++)
        for (int a = 0; a < 3; a++) {
            uint32_t ext = cb.hi[a] - cb.lo[a];
            if (!ext) continue;
            int k = bin_of(cent2(p[i], a), cb.lo[a], ext);
            b.box[a][k].add(p[i]);
            b.count[a][k]++;
        }
}
// best (axis, bin) of the filled bins; false when every centroid coincides
static bool best_split(const Bins &b, const IBox &cb, int &axis, int &bin)
{
    double best = -1.0;
    axis = -1;
    for (int a = 0; a < 3; a++) {
        if (cb.hi[a] == cb.lo[a]) continue;
        double ra[SAH_BINS]; uint64_t rc[SAH_BINS];
        IBox r; r.clear(); uint64_t c = 0;
        for (int k = SAH_BINS - 1; k > 0; k--) { r.add(b.box[a][k]); c += b.count[a][k]; ra[k] = r.area(); rc[k] = c; }
        IBox l; l.clear(); c = 0;
        for (int k = 0; k < SAH_BINS - 1; k++) {
            l.add(b.box[a][k]); c += b.count[a][k];
            if (c == 0 || rc[k + 1] == 0) continue;
            double cost = l.area() * (double)c + ra[k + 1] * (double)rc[k + 1];
            if (best < 0.0 || cost < best) { best = cost; axis = a; bin = k; }
        }
    }
    return axis >= 0;
}

// exact SAH sweep for a small set: sort by centroid on each axis, try every split position
static size_t split_sweep(Prim *p, size_t count)
{
    double best = -1.0;
    int best_axis = -1;
    size_t best_nl = count / 2;
    Prim tmp[SWEEP_MAX];
    double ra[SWEEP_MAX];
    for (int a = 0; a < 3; a++) {
        memcpy(tmp, p, count * sizeof(Prim));
        std::stable_sort(tmp, tmp + count, [a](const Prim &x, const Prim &y) { return cent2(x, a) < cent2(y, a); });
        IBox r;
        for (size_t k = count; k-- > 1;) { r.add(tmp[k]); ra[k] = r.area(); }
        IBox l;
        for (size_t k = 0; k + 1 < count; k++) {
            l.add(tmp[k]);
            double cost = l.area() * (double)(k + 1) + ra[k + 1] * (double)(count - k - 1);
            if (best < 0.0 || cost < best) { best = cost; best_axis = a; best_nl = k + 1; }
        }
    }
    if (best_axis < 0) return count / 2;
    std::stable_sort(p, p + count, [best_axis](const Prim &x, const Prim &y) { return cent2(x, best_axis) < cent2(y, best_axis); });
    return best_nl;
}

// split [first, first+count) in two non-empty parts, single thread; returns the size of the left part
static size_t split_serial(Prim *prims, size_t first, size_t count)
{
    Prim *p = prims + first;
    if (count <= SWEEP_MAX) return split_sweep(p, count);
    IBox cb; centroid_bounds(p, count, cb);
    Bins b; b.clear();
    fill_bins(p, count, cb, b);
    int axis, bin;
    if (!best_split(b, cb, axis, bin)) return count / 2;
    uint32_t cmin = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
    Prim *mid = std::partition(p, p + count, [&](const Prim &q) { return bin_of(cent2(q, axis), cmin, ext) <= bin; });
    size_t nl = (size_t)(mid - p);
    return (nl == 0 || nl == count) ? count / 2 : nl;
}
// the same with all threads (for the few very large sets at the top)
static size_t split_parallel(Prim *prims, size_t first, size_t count, std::vector<Prim> &tmp)
{
    Prim *p = prims + first;
    const unsigned nt = hw_threads();
    const size_t chunk = (count + nt - 1) / nt;
    std::vector<IBox> cbs(nt);
    parallel_for(nt, [&](size_t a, size_t b) { for (size_t t = a; t < b; t++) { size_t lo = std::min(count, t * chunk), hi = std::min(count, lo + chunk); centroid_bounds(p + lo, hi - lo, cbs[t]); } }, 1);
    IBox cb; cb.clear();
    for (auto &c : cbs) if (!c.empty()) cb.add(c);
    std::vector<Bins> bins(nt);
    parallel_for(nt, [&](size_t a, size_t b) { for (size_t t = a; t < b; t++) { size_t lo = std::min(count, t * chunk), hi = std::min(count, lo + chunk); bins[t].clear(); fill_bins(p + lo, hi - lo, cb, bins[t]); } }, 1);
    Bins all; all.clear();
    for (auto &bt : bins) for (int a = 0; a < 3; a++) for (int k = 0; k < SAH_BINS; k++) { if (!bt.box[a][k].empty()) all.box[a][k].add(bt.box[a][k]); all.count[a][k] += bt.count[a][k]; }
    int axis, bin;
    if (!best_split(all, cb, axis, bin)) return count / 2;
    uint32_t cmin = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
    // stable two-way scatter through tmp
    std::vector<size_t> nleft(nt + 1, 0);
    parallel_for(nt, [&](size_t a, size_t b) { for (size_t t = a; t < b; t++) { size_t lo = std::min(count, t * chunk), hi = std::min(count, lo + chunk), c = 0; for (size_t i = lo; i < hi; i++) c += bin_of(cent2(p[i], axis), cmin, ext) <= bin; nleft[t + 1] = c; } }, 1);
    for (unsigned t = 0; t < nt; t++) nleft[t + 1] += nleft[t];
    const size_t nl = nleft[nt];
    if (nl == 0 || nl == count) return count / 2;
    if (tmp.size() < count) tmp.resize(count);
    parallel_for(nt, [&](size_t a, size_t b) {
        for (size_t t = a; t < b; t++) {
            size_t lo = std::min(count, t * chunk), hi = std::min(count, lo + chunk);
            size_t l = nleft[t], r = nl + (lo - nleft[t]);
            for (size_t i = lo; i < hi; i++) { if (bin_of(cent2(p[i], axis), cmin, ext) <= bin) tmp[l++] = p[i]; else tmp[r++] = p[i]; }
        }
    }, 1);
    parallel_for(count, [&](size_t a, size_t b) { memcpy(p + a, tmp.data() + a, (b - a) * sizeof(Prim)); });
    return nl;
}

struct Segment { size_t first, count; IBox box; };

static IBox segment_box(const Prim *prims, size_t first, size_t count)
{
    IBox b; b.clear();
    if (count < (1u << 20)) {
        for (size_t i = first; i < first + count; i++) b.add(prims[i]);
        return b;
    }
    const unsigned nt = hw_threads();
    const size_t chunk = (count + nt - 1) / nt;
    std::vector<IBox> part(nt);
    parallel_for(nt, [&](size_t a, size_t e) {
        for (size_t t = a; t < e; t++) {
            size_t lo = std::min(count, t * chunk), hi = std::min(count, lo + chunk);
            for (size_t i = first + lo; i < first + hi; i++) part[t].add(prims[i]);
        }
    }, 1);
    for (auto &p : part) if (!p.empty()) b.add(p);
    return b;
}

// One subtree, single thread.  Nodes go to `nodes` (32 words each) with indices local to it; a
// leaf entry holds WIDE_LEAF | position of the triangle in the prim array.
static uint32_t build_subtree(Prim *prims, size_t first, size_t count, std::vector<uint32_t> &nodes, uint32_t depth, uint32_t &max_depth)
{
    const uint32_t me = (uint32_t)(nodes.size() / 32);
    nodes.resize(nodes.size() + 32);
    max_depth = std::max(max_depth, depth + 1);
    Segment seg[WIDE_K];
    int n = 1;
    seg[0] = Segment{first, count, IBox()};
    while (n < (int)WIDE_K) {
        int pick = -1; double best = -1.0;
        for (int i = 0; i < n; i++) {
            if (seg[i].count < 2) continue;
            if (seg[i].box.empty()) seg[i].box = segment_box(prims, seg[i].first, seg[i].count);
            // sets that can become leaves of this very node are split first, then the largest
            double a = seg[i].box.area() + (seg[i].count <= (size_t)(WIDE_K - n + 1) ? 1e300 : 0.0);
            if (a > best) { best = a; pick = i; }
        }
        if (pick < 0) break;
        Segment s = seg[pick];
        size_t nl = split_serial(prims, s.first, s.count);
        seg[pick] = Segment{s.first, nl, IBox()};
        seg[n++] = Segment{s.first + nl, s.count - nl, IBox()};
    }
    for (int i = 0; i < (int)WIDE_K; i++) {
        uint32_t word[4];
        if (i >= n || seg[i].count == 0) { word[0] = word[1] = word[2] = 0x0000FFFFu; word[3] = WIDE_EMPTY; }
        else {
            if (seg[i].box.empty()) seg[i].box = segment_box(prims, seg[i].first, seg[i].count);
            for (int a = 0; a < 3; a++) word[a] = seg[i].box.lo[a] | seg[i].box.hi[a] << 16;
            if (seg[i].count == 1) word[3] = WIDE_LEAF | (uint32_t)seg[i].first;
            else word[3] = build_subtree(prims, seg[i].first, seg[i].count, nodes, depth + 1, max_depth);
        }
        memcpy(nodes.data() + (size_t)me * 32 + 4 * i, word, 16);       // (nodes may have been reallocated by the recursion)
    }
    return me;
}

// ---- the same sets as a BINARY tree first, then collapsed to eight-wide nodes at the least total area ----
// The greedy rule above ("split the largest part until there are eight") fills a node with whatever sizes
// the binary SAH splits happen to give: near the bottom of the tree it leaves many wide nodes with two or
// three triangles (C3: 71 % of all entries used), and every such node is a visit -- one dependent fetch
// and a pass of the walk's bookkeeping -- that tests few boxes.  Here a task's triangles are first split
// all the way down to single triangles (the same binned / swept SAH splits), and the binary tree is then
// cut into wide nodes by the dynamic programme of Ylitie, Karras & Laine ("Efficient incoherent ray
// traversal on GPUs through compressed wide BVHs", HPG 2017, section 4.1, with one triangle per leaf
// entry): D(n, k) = least sum of wide-node areas below binary node n when n may use k entries of its
// parent -- as ONE entry it becomes a wide node of its own, area(n) + best distribution of eight entries
// over its two children; with k >= 2 entries it may instead dissolve into the parent, its children sharing
// the k entries.  The sum of the areas is the SAH estimate of the number of node visits per ray.
struct BinNode { uint16_t lo[3], hi[3]; uint32_t left, right; };     // leaf: left = 0xFFFFFFFF, right = prim position
static const uint32_t BIN_LEAF = 0xFFFFFFFFu;

static inline double bin_area(const BinNode &b)
{
    double dx = (double)(b.hi[0] - b.lo[0]), dy = (double)(b.hi[1] - b.lo[1]), dz = (double)(b.hi[2] - b.lo[2]);
    return dx * dy + dy * dz + dz * dx;
}

struct DpScratch {
    std::vector<BinNode> bin;
    std::vector<float> cost;          // [node][8]: D(n, k), k = 1..8
    std::vector<uint8_t> left_share;  // [node][8]: entries given to the left child when n's children share k entries (k >= 2; [..][0] for its own eight)
    std::vector<uint8_t> own_node;    // [node][8]: D(n, k) is reached by n being a wide node of its own
};

static void emit_entries(const DpScratch &d, uint32_t n, int k, uint32_t *items, int &nitems)
{
    const BinNode &b = d.bin[n];
    if (b.left == BIN_LEAF || k == 1 || d.own_node[(size_t)n * 8 + (k - 1)]) { items[nitems++] = n; return; }
    const int j = d.left_share[(size_t)n * 8 + (k - 1)];
    emit_entries(d, b.left, j, items, nitems);
    emit_entries(d, b.right, k - j, items, nitems);
}

static uint32_t emit_wide(const DpScratch &d, uint32_t n, std::vector<uint32_t> &nodes, uint32_t depth, uint32_t &max_depth)
{
    const uint32_t me = (uint32_t)(nodes.size() / 32);
    nodes.resize(nodes.size() + 32);
    max_depth = std::max(max_depth, depth + 1);
    uint32_t items[WIDE_K];
    int nitems = 0;
    const BinNode &b = d.bin[n];
    const int j = d.left_share[(size_t)n * 8];
    emit_entries(d, b.left, j, items, nitems);
    emit_entries(d, b.right, (int)WIDE_K - j, items, nitems);
    for (int i = 0; i < (int)WIDE_K; i++) {
        uint32_t word[4];
        if (i >= nitems) { word[0] = word[1] = word[2] = 0x0000FFFFu; word[3] = WIDE_EMPTY; }
        else {
            const BinNode &c = d.bin[items[i]];
            for (int a = 0; a < 3; a++) word[a] = (uint32_t)c.lo[a] | (uint32_t)c.hi[a] << 16;
            word[3] = (c.left == BIN_LEAF) ? (WIDE_LEAF | c.right) : emit_wide(d, items[i], nodes, depth + 1, max_depth);
        }
        memcpy(nodes.data() + (size_t)me * 32 + 4 * i, word, 16);
    }
    return me;
}

static uint32_t build_subtree_dp(Prim *prims, size_t first, size_t count, std::vector<uint32_t> &nodes, uint32_t &max_depth, DpScratch &d)
{
    // binary tree, parents before children (an explicit stack: degenerate sets fall back to halving, so the
    // depth is bounded, but a task holds up to a million triangles)
    d.bin.clear();
    d.bin.reserve(2 * count);
    struct Todo { size_t first, count; uint32_t node; };
    std::vector<Todo> todo;
    d.bin.push_back(BinNode());
    todo.push_back(Todo{first, count, 0});
    while (!todo.empty()) {
        Todo t = todo.back(); todo.pop_back();
        if (t.count == 1) {
            BinNode &b = d.bin[t.node];
            const Prim &p = prims[t.first];
            for (int a = 0; a < 3; a++) { b.lo[a] = p.lo[a]; b.hi[a] = p.hi[a]; }
            b.left = BIN_LEAF; b.right = (uint32_t)t.first;
            continue;
        }
        const size_t nl = split_serial(prims, t.first, t.count);
        const uint32_t l = (uint32_t)d.bin.size();
        d.bin.push_back(BinNode()); d.bin.push_back(BinNode());
        d.bin[t.node].left = l; d.bin[t.node].right = l + 1;
        todo.push_back(Todo{t.first + nl, t.count - nl, l + 1});
        todo.push_back(Todo{t.first, nl, l});
    }
    const size_t nb = d.bin.size();
    d.cost.assign(nb * 8, 0.0f);
    d.left_share.assign(nb * 8, 0);
    d.own_node.assign(nb * 8, 0);
    // children have larger indices than their parents: one backward sweep fills boxes and the table
    for (size_t n = nb; n-- > 0;) {
        BinNode &b = d.bin[n];
        if (b.left == BIN_LEAF) continue;                 // D(leaf, k) = 0
        const BinNode &l = d.bin[b.left], &r = d.bin[b.right];
        for (int a = 0; a < 3; a++) { b.lo[a] = std::min(l.lo[a], r.lo[a]); b.hi[a] = std::max(l.hi[a], r.hi[a]); }
        const float *cl = &d.cost[(size_t)b.left * 8], *cr = &d.cost[(size_t)b.right * 8];
        float *cn = &d.cost[n * 8];
        float share[9]; uint8_t share_j[9];
        for (int k = 2; k <= 8; k++) {
            float best = 0.0f; int bj = 0;
            for (int j = 1; j < k; j++) {
                const float c = cl[j - 1] + cr[k - j - 1];
                if (bj == 0 || c < best) { best = c; bj = j; }
            }
            share[k] = best; share_j[k] = (uint8_t)bj;
        }
        const float own = (float)bin_area(b) + share[8];
        cn[0] = own;
        d.left_share[n * 8] = share_j[8];
        d.own_node[n * 8] = 1;
        for (int k = 2; k <= 8; k++) {
            if (own <= share[k]) { cn[k - 1] = own; d.own_node[n * 8 + (k - 1)] = 1; d.left_share[n * 8 + (k - 1)] = share_j[8]; }
            else { cn[k - 1] = share[k]; d.left_share[n * 8 + (k - 1)] = share_j[k]; }
        }
    }
    return emit_wide(d, 0, nodes, 0, max_depth);
}

// one primitive per triangle that hangs under a reachable leaf, in triangle order (threaded: count, place)
static void make_prims(const uint32_t *ref, uint32_t ntriangles, const std::vector<uint32_t> &leaf_node, std::vector<Prim> &prims)
{
    const int tight = wide_tight_leaves();
    {
        const unsigned nt = hw_threads();
        const size_t chunk = ((size_t)ntriangles + nt - 1) / nt;
        std::vector<size_t> start(nt + 1, 0);
        parallel_for(nt, [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k++) {
                size_t lo = std::min<size_t>(ntriangles, k * chunk), hi = std::min<size_t>(ntriangles, lo + chunk), c = 0;
                for (size_t t = lo; t < hi; t++) c += leaf_node[t] != 0xFFFFFFFFu;
                start[k + 1] = c;
            }
        }, 1);
        for (unsigned k = 0; k < nt; k++) start[k + 1] += start[k];
        prims.resize(start[nt]);
        parallel_for(nt, [&](size_t a, size_t b) {
            for (size_t k = a; k < b; k++) {
                size_t lo = std::min<size_t>(ntriangles, k * chunk), hi = std::min<size_t>(ntriangles, lo + chunk), o = start[k];
                for (size_t t = lo; t < hi; t++) {
                    if (leaf_node[t] == 0xFFFFFFFFu) continue;
                    const uint32_t *nd = ref + 4 * (size_t)leaf_node[t];
                    Prim p;
                    for (int ax = 0; ax < 3; ax++) {
                        const uint32_t w = wide_tight_bound_word(nd[ax], tight);      // (wide_build.h: the reference's padding quantum)
                        p.lo[ax] = (uint16_t)(w & 0xFFFFu); p.hi[ax] = (uint16_t)(w >> 16);
                    }
                    p.tri = (uint32_t)t;
                    prims[o++] = p;
                }
            }
        }, 1);
    }
}

static int sah_topology(const uint32_t *ref, uint32_t ntriangles, const std::vector<uint32_t> &leaf_node, WideTree &out, std::string &err, bool collapse_dp)
{
    std::vector<Prim> prims;
    make_prims(ref, ntriangles, leaf_node, prims);
    const size_t np = prims.size();
    if (np > 0x7FFFFFFFull) { err = "wide tree: too many triangles"; return -1; }
    const size_t task_size = std::max<size_t>(1u << 16, np / (4 * (size_t)hw_threads()));

    // top part: sets larger than task_size, split by all threads together
    struct Task { size_t first, count; uint32_t parent, slot; };
    std::vector<uint32_t> top;                 // nodes of the top part
    std::vector<Task> tasks;
    std::vector<Prim> tmp;
    uint32_t top_depth = 0;
    struct Work { size_t first, count; uint32_t node, depth; };
    std::vector<Work> work;
    top.resize(32);
    if (np == 0) {
        for (int i = 0; i < (int)WIDE_K; i++) { uint32_t *o = top.data() + 4 * i; o[0] = o[1] = o[2] = 0x0000FFFFu; o[3] = WIDE_EMPTY; }
    } else {
        work.push_back(Work{0, np, 0, 0});
    }
    for (size_t wi = 0; wi < work.size(); wi++) {          // breadth first: parents before children
        Work w = work[wi];
        top_depth = std::max(top_depth, w.depth + 1);
        Segment seg[WIDE_K];
        int n = 1;
        seg[0] = Segment{w.first, w.count, IBox()};
        while (n < (int)WIDE_K) {
            int pick = -1; double best = -1.0;
            for (int i = 0; i < n; i++) {
                if (seg[i].count < 2) continue;
                if (seg[i].box.empty()) seg[i].box = segment_box(prims.data(), seg[i].first, seg[i].count);
                double a = seg[i].box.area() + (seg[i].count <= (size_t)(WIDE_K - n + 1) ? 1e300 : 0.0);
                if (a > best) { best = a; pick = i; }
            }
            if (pick < 0) break;
            Segment s = seg[pick];
            size_t nl = (s.count > task_size) ? split_parallel(prims.data(), s.first, s.count, tmp) : split_serial(prims.data(), s.first, s.count);
            seg[pick] = Segment{s.first, nl, IBox()};
            seg[n++] = Segment{s.first + nl, s.count - nl, IBox()};
        }
        for (int i = 0; i < (int)WIDE_K; i++) {
            uint32_t word[4];
            if (i >= n || seg[i].count == 0) { word[0] = word[1] = word[2] = 0x0000FFFFu; word[3] = WIDE_EMPTY; }
            else {
                if (seg[i].box.empty()) seg[i].box = segment_box(prims.data(), seg[i].first, seg[i].count);
                for (int a = 0; a < 3; a++) word[a] = seg[i].box.lo[a] | seg[i].box.hi[a] << 16;
                if (seg[i].count == 1) word[3] = WIDE_LEAF | (uint32_t)seg[i].first;
                else if (seg[i].count > task_size) {
                    uint32_t child = (uint32_t)(top.size() / 32);
                    top.resize(top.size() + 32);
                    work.push_back(Work{seg[i].first, seg[i].count, child, w.depth + 1});
                    word[3] = child;
                } else {
                    tasks.push_back(Task{seg[i].first, seg[i].count, w.node, (uint32_t)i});
                    word[3] = WIDE_EMPTY;            // patched below
                }
            }
            memcpy(top.data() + (size_t)w.node * 32 + 4 * i, word, 16);
        }
    }
    { std::vector<Prim>().swap(tmp); }

    // independent subtrees
    const size_t ntasks = tasks.size();
    std::vector<std::vector<uint32_t>> sub(ntasks);
    std::vector<uint32_t> sub_depth(ntasks, 0);
    {
        std::atomic<size_t> next(0);
        unsigned nt = std::min<size_t>(hw_threads(), std::max<size_t>(1, ntasks));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&] {
                DpScratch scratch;
                for (;;) {
                    size_t k = next.fetch_add(1);
                    if (k >= ntasks) break;
                    sub[k].reserve((tasks[k].count / 3 + 8) * 32);
                    if (collapse_dp) build_subtree_dp(prims.data(), tasks[k].first, tasks[k].count, sub[k], sub_depth[k], scratch);
                    else build_subtree(prims.data(), tasks[k].first, tasks[k].count, sub[k], 0, sub_depth[k]);
                }
            });
        for (auto &t : th) t.join();
    }
    // concatenate: top part, then the tasks in creation order
    size_t total = top.size() / 32;
    std::vector<size_t> offset(ntasks);
    uint32_t depth = top_depth;
    for (size_t k = 0; k < ntasks; k++) { offset[k] = total; total += sub[k].size() / 32; depth = std::max(depth, top_depth + sub_depth[k]); }
    if (total > 0x7FFFFFFFull) { err = "wide tree: more than 2^31 nodes"; return -1; }
    for (size_t k = 0; k < ntasks; k++) top[(size_t)tasks[k].parent * 32 + 4 * tasks[k].slot + 3] = (uint32_t)offset[k];
    out.wnodes.resize(total * 32);
    memcpy(out.wnodes.data(), top.data(), top.size() * 4);
    parallel_for(ntasks, [&](size_t a, size_t b) {
        for (size_t k = a; k < b; k++) {
            uint32_t *dst = out.wnodes.data() + offset[k] * 32;
            const std::vector<uint32_t> &src = sub[k];
            memcpy(dst, src.data(), src.size() * 4);
            for (size_t i = 3; i < src.size(); i += 4)
                if (dst[i] != WIDE_EMPTY && !(dst[i] & WIDE_LEAF)) dst[i] += (uint32_t)offset[k];
            std::vector<uint32_t>().swap(sub[k]);
        }
    }, 1);
    out.nwide = total;
    out.depth = depth;
    out.dev_to_tri.resize(np);
    parallel_for(np, [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) out.dev_to_tri[i] = prims[i].tri; });
    return 0;
}

// ---- PLOC: the hierarchy built bottom-up ------------------------------------------------------------------
// Parallel locally-ordered clustering (Meister & Bittner, "Parallel Locally-Ordered Clustering for Bounding Volume
// Hierarchy Construction", TVCG 2018): the clusters -- at first the reference's leaves in their Morton order, which is
// the order of the reference tree's last layer -- each look PLOC_RADIUS places to either side for the neighbour whose
// union with them has the least area (ties: the lower index); two clusters that chose each other merge into a binary
// node that takes the place of the lower one; repeat until one cluster is left.  Every step is a data-parallel pass
// over an array (areas are integers, node numbers are prefix counts: a device version would give the same tree bit for
// bit), which is why this topology was built: as the host twin of a device builder.  MEASURED at C3 before writing that
// builder (profiles/r03/ab_tree_ploc.txt): the tree has 3 % fewer node entries per ray but 6 % more triangle tests, and a
// step takes 3.5 % longer than with the top-down SAH tree (radius 32: 3.3 %) -- so it stays an option
// (CHROMA_TREE=ploc), and a device builder has to be the binned top-down one.  The binary tree is then cut into
// eight-wide nodes by the same least-area dynamic programme as the SAH topology, and the wide nodes are numbered
// breadth first (a level's nodes in the order of their parents' entries).
struct PlocCluster { uint16_t lo[3], hi[3]; uint32_t node; };

static inline uint64_t ploc_union_area(const PlocCluster &a, const PlocCluster &b)
{
    uint64_t d[3];
    for (int k = 0; k < 3; k++) d[k] = (uint64_t)(std::max(a.hi[k], b.hi[k]) - std::min(a.lo[k], b.lo[k]));
    return d[0] * d[1] + d[1] * d[2] + d[2] * d[0];
}

// D(n, k) of a binary node whose children's rows are filled (see build_subtree_dp)
static inline void dp_fill_node(DpScratch &d, size_t n)
{
    const BinNode &b = d.bin[n];
    const float *cl = &d.cost[(size_t)b.left * 8], *cr = &d.cost[(size_t)b.right * 8];
    float *cn = &d.cost[n * 8];
    float share[9]; uint8_t share_j[9];
    for (int k = 2; k <= 8; k++) {
        float best = 0.0f; int bj = 0;
        for (int j = 1; j < k; j++) {
            const float c = cl[j - 1] + cr[k - j - 1];
            if (bj == 0 || c < best) { best = c; bj = j; }
        }
        share[k] = best; share_j[k] = (uint8_t)bj;
    }
    const float own = (float)bin_area(b) + share[8];
    cn[0] = own;
    d.left_share[n * 8] = share_j[8];
    d.own_node[n * 8] = 1;
    for (int k = 2; k <= 8; k++) {
        if (own <= share[k]) { cn[k - 1] = own; d.own_node[n * 8 + (k - 1)] = 1; d.left_share[n * 8 + (k - 1)] = share_j[8]; }
        else { cn[k - 1] = share[k]; d.left_share[n * 8 + (k - 1)] = share_j[k]; }
    }
}

// The binary tree of `d` (boxes and D(n, k) filled) cut into eight-wide nodes at the least total area, the wide nodes numbered
// BREADTH FIRST: a level's nodes in the order of their parents' entries.  Every level is a data-parallel pass.
static int emit_breadth_first(const DpScratch &d, uint32_t root, WideTree &out, std::string &err)
{
    out.wnodes.clear();
    std::vector<uint32_t> level(1, root), next;
    size_t base = 0;
    uint32_t depth = 0;
    if (d.bin[root].left == BIN_LEAF) {                          // a single triangle: one node, one leaf entry
        out.wnodes.assign(32, 0);
        for (int i = 0; i < (int)WIDE_K; i++) { uint32_t *o = out.wnodes.data() + 4 * i; o[0] = o[1] = o[2] = 0x0000FFFFu; o[3] = WIDE_EMPTY; }
        const BinNode &b = d.bin[root];
        for (int a = 0; a < 3; a++) out.wnodes[a] = (uint32_t)b.lo[a] | (uint32_t)b.hi[a] << 16;
        out.wnodes[3] = WIDE_LEAF | b.right;
        out.nwide = 1; out.depth = 1;
        return 0;
    }
    while (!level.empty()) {
        const size_t cnt = level.size();
        if (base + cnt > 0x7FFFFFFFull) { err = "wide tree: more than 2^31 nodes"; return -1; }
        out.wnodes.resize((base + cnt) * 32);
        // entries of every node of the level, and how many inner children each has
        std::vector<uint32_t> items(cnt * WIDE_K);
        std::vector<uint8_t> nitems(cnt);
        std::vector<uint32_t> first_child(cnt + 1, 0);
        parallel_for(cnt, [&](size_t a, size_t e) {
            for (size_t k = a; k < e; k++) {
                const BinNode &b = d.bin[level[k]];
                uint32_t *it = items.data() + k * WIDE_K;
                int ni = 0;
                const int j = d.left_share[(size_t)level[k] * 8];
                emit_entries(d, b.left, j, it, ni);
                emit_entries(d, b.right, (int)WIDE_K - j, it, ni);
                nitems[k] = (uint8_t)ni;
                uint32_t inner = 0;
                for (int i = 0; i < ni; i++) inner += d.bin[it[i]].left != BIN_LEAF;
                first_child[k + 1] = inner;
            }
        }, 1u << 12);
        for (size_t k = 0; k < cnt; k++) first_child[k + 1] += first_child[k];
        next.assign(first_child[cnt], 0);
        parallel_for(cnt, [&](size_t a, size_t e) {
            for (size_t k = a; k < e; k++) {
                uint32_t *wn = out.wnodes.data() + (base + k) * 32;
                const uint32_t *it = items.data() + k * WIDE_K;
                uint32_t child = first_child[k];
                for (int i = 0; i < (int)WIDE_K; i++) {
                    uint32_t *o = wn + 4 * i;
                    if (i >= nitems[k]) { o[0] = o[1] = o[2] = 0x0000FFFFu; o[3] = WIDE_EMPTY; continue; }
                    const BinNode &c = d.bin[it[i]];
                    for (int ax = 0; ax < 3; ax++) o[ax] = (uint32_t)c.lo[ax] | (uint32_t)c.hi[ax] << 16;
                    if (c.left == BIN_LEAF) o[3] = WIDE_LEAF | c.right;
                    else { o[3] = (uint32_t)(base + cnt + child); next[child++] = it[i]; }
                }
            }
        }, 1u << 12);
        base += cnt;
        level.swap(next);
        depth++;
    }
    out.nwide = base;
    out.depth = depth;
    return 0;
}

static int ploc_topology(const uint32_t *ref, size_t leaf_lo, size_t leaf_hi, uint32_t ntriangles, WideTree &out, std::string &err)
{
    const size_t n = leaf_hi - leaf_lo;
    if (n == 0 || n > 0x3FFFFFFFull) { err = "wide tree: no leaves, or too many"; return -1; }
    DpScratch d;
    d.bin.resize(2 * n - 1);
    out.dev_to_tri.resize(n);
    std::vector<PlocCluster> cur(n), nxt;
    for (size_t i = 0; i < n; i++) {
        const uint32_t *nd = ref + 4 * (leaf_lo + i);
        if ((nd[3] >> NCHILD_SHIFT) != 0 || (nd[3] & CHILD_MASK) >= ntriangles) { err = "wide tree: the last layer of the reference tree is not a layer of leaves"; return -1; }
        PlocCluster c;
        for (int a = 0; a < 3; a++) { const uint32_t w = wide_tight_bound_word(nd[a], wide_tight_leaves()); c.lo[a] = (uint16_t)(w & 0xFFFFu); c.hi[a] = (uint16_t)(w >> 16); }
        c.node = (uint32_t)i;
        cur[i] = c;
        BinNode &b = d.bin[i];
        for (int a = 0; a < 3; a++) { b.lo[a] = c.lo[a]; b.hi[a] = c.hi[a]; }
        b.left = BIN_LEAF; b.right = (uint32_t)i;
        out.dev_to_tri[i] = nd[3] & CHILD_MASK;
    }
    d.cost.assign((2 * n - 1) * 8, 0.0f);
    d.left_share.assign((2 * n - 1) * 8, 0);
    d.own_node.assign((2 * n - 1) * 8, 0);
    size_t nb = n;                       // binary nodes so far: the leaves
    const size_t radius = getenv("CHROMA_PLOC_RADIUS") ? (size_t)std::max(1, atoi(getenv("CHROMA_PLOC_RADIUS"))) : (size_t)PLOC_RADIUS;
    std::vector<uint32_t> nn;
    while (cur.size() > 1) {
        const size_t m = cur.size();
        nn.resize(m);
        parallel_for(m, [&](size_t a, size_t b) {
            for (size_t i = a; i < b; i++) {
                const size_t lo = i > radius ? i - radius : 0, hi = std::min(m - 1, i + radius);
                uint64_t best = ~0ull; size_t bj = i;
                for (size_t j = lo; j <= hi; j++) {
                    if (j == i) continue;
                    const uint64_t ar = ploc_union_area(cur[i], cur[j]);
                    if (ar < best) { best = ar; bj = j; }
                }
                nn[i] = (uint32_t)bj;
            }
        }, 1u << 12);
        nxt.clear();
        nxt.reserve(m);
        for (size_t i = 0; i < m; i++) {
            const size_t j = nn[i];
            const bool mutual = nn[j] == i;
            if (mutual && j < i) continue;                       // merged into the lower one
            if (!mutual) { nxt.push_back(cur[i]); continue; }
            PlocCluster c;
            for (int a = 0; a < 3; a++) { c.lo[a] = std::min(cur[i].lo[a], cur[j].lo[a]); c.hi[a] = std::max(cur[i].hi[a], cur[j].hi[a]); }
            c.node = (uint32_t)nb;
            BinNode &b = d.bin[nb];
            for (int a = 0; a < 3; a++) { b.lo[a] = c.lo[a]; b.hi[a] = c.hi[a]; }
            b.left = cur[i].node; b.right = cur[j].node;
            dp_fill_node(d, nb);                                 // (the children were made in earlier passes)
            nb++;
            nxt.push_back(c);
        }
        if (nxt.size() == m) { err = "wide tree: clustering made no progress"; return -1; }      // (cannot happen: the least pair is mutual)
        cur.swap(nxt);
    }
    return emit_breadth_first(d, cur[0].node, out, err);
}

// ---- "levels": the same SAH splits with NOTHING left to the schedule -- the host twin of the device builder ------------------
// sah_topology's tree depends on the machine a little: sets above a size that follows from the thread count become wide nodes by
// the greedy rule, and std::partition leaves the two halves of a set in an order of its own.  The device builder
// (csrc/wide_device.hip) cannot follow either, so the algorithm is pinned down here, once, for both:
//   * a set of n triangles (in the order it has reached) is split in two, always: n <= 32 by the exact sweep -- stable sort
//     by doubled centroid on each axis, least l.area * nl + r.area * nr, first minimum in the order (axis, position), the set
//     left sorted on the winning axis; n > 32 by 32 bins on each axis that has extent, first minimum in the order (axis, bin),
//     a STABLE partition; centroids that all coincide: the set is halved where it stands;
//   * that all the way down to single triangles: ONE binary tree over all triangles;
//   * boxes and the least-area table D(n, k) bottom-up over the whole tree, wide nodes emitted breadth first
//     (emit_breadth_first): no top part, no tasks -- the threads here only share out work whose result is fixed.
// Areas are products of 16-bit integers in doubles, costs products of those with counts: IEEE arithmetic without contraction
// gives the device the same numbers.  Binary nodes are numbered in preorder here (a set of n triangles owns 2n-1 consecutive
// numbers, so subtrees can be built by different threads) and by level on the device; the wide tree does not depend on it.
static size_t split_stable(Prim *p, size_t count, std::vector<Prim> &tmp)
{
    if (count <= SWEEP_MAX) return split_sweep(p, count);
    IBox cb; centroid_bounds(p, count, cb);
    Bins b; b.clear();
    fill_bins(p, count, cb, b);
    int axis, bin;
    if (!best_split(b, cb, axis, bin)) return count / 2;
    const uint32_t cmin = cb.lo[axis], ext = cb.hi[axis] - cb.lo[axis];
    if (tmp.size() < count) tmp.resize(count);
    size_t nl = 0, nr = 0;
    for (size_t i = 0; i < count; i++) { if (bin_of(cent2(p[i], axis), cmin, ext) <= bin) p[nl++] = p[i]; else tmp[nr++] = p[i]; }
    memcpy(p + nl, tmp.data(), nr * sizeof(Prim));
    return (nl == 0 || nl == count) ? count / 2 : nl;
}

static inline void bin_from_children(DpScratch &d, size_t n)
{
    BinNode &b = d.bin[n];
    const BinNode &l = d.bin[b.left], &r = d.bin[b.right];
    for (int a = 0; a < 3; a++) { b.lo[a] = std::min(l.lo[a], r.lo[a]); b.hi[a] = std::max(l.hi[a], r.hi[a]); }
    dp_fill_node(d, n);
}

static int levels_topology(const uint32_t *ref, uint32_t ntriangles, const std::vector<uint32_t> &leaf_node, WideTree &out, std::string &err)
{
    std::vector<Prim> prims;
    make_prims(ref, ntriangles, leaf_node, prims);
    const size_t np = prims.size();
    if (np > 0x3FFFFFFFull) { err = "wide tree: too many triangles"; return -1; }
    if (np == 0) {
        out.wnodes.assign(32, 0);
        for (int i = 0; i < (int)WIDE_K; i++) { uint32_t *o = out.wnodes.data() + 4 * i; o[0] = o[1] = o[2] = 0x0000FFFFu; o[3] = WIDE_EMPTY; }
        out.nwide = 1; out.depth = 1;
        return 0;
    }
    DpScratch d;
    const size_t nb = 2 * np - 1;
    d.bin.resize(nb);
    d.cost.assign(nb * 8, 0.0f);
    d.left_share.assign(nb * 8, 0);
    d.own_node.assign(nb * 8, 0);
    struct Job { size_t first, count; uint32_t node; };
    const size_t task_size = std::max<size_t>(1u << 16, np / (4 * (size_t)hw_threads()));      // (scheduling only)
    std::vector<Job> stack(1, Job{0, np, 0}), tasks;
    std::vector<uint32_t> top_nodes;
    {
        std::vector<Prim> tmp;
        while (!stack.empty()) {
            Job j = stack.back(); stack.pop_back();
            if (j.count <= task_size) { tasks.push_back(j); continue; }
            const size_t nl = split_parallel(prims.data(), j.first, j.count, tmp);        // (binned, stable: the same split as split_stable)
            BinNode &b = d.bin[j.node];
            b.left = j.node + 1; b.right = j.node + (uint32_t)(2 * nl);
            top_nodes.push_back(j.node);
            stack.push_back(Job{j.first + nl, j.count - nl, b.right});
            stack.push_back(Job{j.first, nl, b.left});
        }
    }
    {
        std::atomic<size_t> next(0);
        const size_t ntasks = tasks.size();
        unsigned nt = (unsigned)std::min<size_t>(hw_threads(), std::max<size_t>(1, ntasks));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([&] {
                std::vector<Prim> tmp;
                std::vector<Job> todo;
                for (;;) {
                    const size_t k = next.fetch_add(1);
                    if (k >= ntasks) break;
                    todo.assign(1, tasks[k]);
                    while (!todo.empty()) {
                        Job j = todo.back(); todo.pop_back();
                        BinNode &b = d.bin[j.node];
                        if (j.count == 1) {
                            const Prim &q = prims[j.first];
                            for (int a = 0; a < 3; a++) { b.lo[a] = q.lo[a]; b.hi[a] = q.hi[a]; }
                            b.left = BIN_LEAF; b.right = (uint32_t)j.first;
                            continue;
                        }
                        const size_t nl = split_stable(prims.data() + j.first, j.count, tmp);
                        b.left = j.node + 1; b.right = j.node + (uint32_t)(2 * nl);
                        todo.push_back(Job{j.first + nl, j.count - nl, b.right});
                        todo.push_back(Job{j.first, nl, b.left});
                    }
                    // children carry larger numbers than their parents: one backward sweep over the task's own numbers
                    for (size_t n = (size_t)tasks[k].node + 2 * tasks[k].count - 1; n-- > (size_t)tasks[k].node;)
                        if (d.bin[n].left != BIN_LEAF) bin_from_children(d, n);
                }
            });
        for (auto &t : th) t.join();
    }
    std::sort(top_nodes.begin(), top_nodes.end());
    for (size_t i = top_nodes.size(); i-- > 0;) bin_from_children(d, top_nodes[i]);
    out.dev_to_tri.resize(np);
    parallel_for(np, [&](size_t a, size_t b) { for (size_t i = a; i < b; i++) out.dev_to_tri[i] = prims[i].tri; });
    { std::vector<Prim>().swap(prims); }
    return emit_breadth_first(d, 0, out, err);
}

// stack need and the record maps of a tree whose wnodes / dev_to_tri are made
void finish_wide_tree(WideTree &out, uint32_t ntriangles, bool with_stack_need)
{
    out.tri_to_dev.assign(ntriangles, 0xFFFFFFFFu);
    // worst case of the walk's stack: at a node, every inner child but the one walked next is
    // pushed, then the same below -- whichever child is walked, so the maximum over children
    // (chroma_geometry_create works it out on the device for the tree it uploads: the device builder skips it here)
    out.stack_need = with_stack_need ? wide_stack_need(out.wnodes.data(), out.nwide) : 0u;
    // device index of every triangle (a triangle under several leaves keeps the first; triangles
    // under no leaf go to the end so that every triangle has a record)
    uint32_t *t2d = out.tri_to_dev.data();
    const uint32_t *d2t = out.dev_to_tri.data();
    parallel_for(out.dev_to_tri.size(), [&](size_t a, size_t b) {
        for (size_t d = a; d < b; d++) {
            uint32_t *slot = t2d + d2t[d];
            uint32_t cur = __atomic_load_n(slot, __ATOMIC_RELAXED);
            while ((uint32_t)d < cur && !__atomic_compare_exchange_n(slot, &cur, (uint32_t)d, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        }
    });
    std::atomic<int> unmapped(0);
    parallel_for(ntriangles, [&](size_t a, size_t b) { for (size_t t = a; t < b; t++) if (t2d[t] == 0xFFFFFFFFu) { unmapped = 1; break; } });
    if (unmapped)
        for (uint32_t t = 0; t < ntriangles; t++)
            if (out.tri_to_dev[t] == 0xFFFFFFFFu) { out.tri_to_dev[t] = (uint32_t)out.dev_to_tri.size(); out.dev_to_tri.push_back(t); }
}

// The reference's test order (see the header): `rank` of every triangle, and the reachable leaf that holds it.
int reference_test_order(const uint32_t *nodes, size_t nnodes, uint32_t ntriangles, std::vector<uint32_t> &rank,
                         std::vector<uint32_t> &leaf_node, std::string &err, size_t *nlayers_out, bool *layered_out)
{
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
    rank.assign(ntriangles, 0xFFFFFFFFu);
    leaf_node.assign(ntriangles, 0xFFFFFFFFu);      // the reachable leaf that holds a triangle
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
                if (rank[t] == 0xFFFFFFFFu) { rank[t] = run; leaf_node[t] = c + j; }   // (a duplicate leaf keeps one of its ranks)
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

    if (nlayers_out) *nlayers_out = nlayers;
    if (layered_out) *layered_out = layered;
    return 0;
}

int build_wide_tree(const uint32_t *nodes, size_t nnodes, uint32_t ntriangles, WideTree &out, std::string &err, int topology)
{
    const bool timing = getenv("CHROMA_TIMING") != nullptr;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char *what) {
        if (!timing) return;
        auto now = std::chrono::steady_clock::now();
        fprintf(stderr, "[build_wide_tree] %-28s %.2f s\n", what, std::chrono::duration<double>(now - t_phase).count());
        t_phase = now;
    };
    out = WideTree();
    std::vector<uint32_t> leaf_node;
    size_t nlayers = 0; bool layered = true;
    if (reference_test_order(nodes, nnodes, ntriangles, out.rank, leaf_node, err, &nlayers, &layered) != 0) return -1;
    phase("reference test order");
    out.dev_to_tri.clear();
    // (PLOC starts from the reference tree's leaf layer: its last ntriangles nodes, one leaf per triangle in Morton order --
    //  a tree that does not end that way keeps the top-down builder)
    if (topology == WIDE_TOPOLOGY_PLOC) {
        bool leaf_layer = nnodes >= ntriangles && ntriangles > 0;
        if (leaf_layer) {
            std::atomic<int> not_leaf(0);
            parallel_for(ntriangles, [&](size_t a, size_t b) {
                for (size_t i = a; i < b; i++) {
                    const uint32_t w = nodes[4 * (nnodes - ntriangles + i) + 3];
                    if ((w >> NCHILD_SHIFT) != 0 || (w & CHILD_MASK) >= ntriangles || leaf_node[w & CHILD_MASK] == 0xFFFFFFFFu) { not_leaf = 1; return; }
                }
            });
            leaf_layer = !not_leaf;
        }
        if (!leaf_layer) topology = WIDE_TOPOLOGY_SAH;
    }
    if (timing) fprintf(stderr, "[build_wide_tree] topology %d, layered %d, %zu layers\n", topology, (int)layered, nlayers);
    if (topology == WIDE_TOPOLOGY_PLOC) {
        if (ploc_topology(nodes, nnodes - ntriangles, nnodes, ntriangles, out, err) != 0) return -1;
    } else
    if (topology == WIDE_TOPOLOGY_LEVELS) {
        if (levels_topology(nodes, ntriangles, leaf_node, out, err) != 0) return -1;
    } else
    if (topology == WIDE_TOPOLOGY_SAH || topology == WIDE_TOPOLOGY_SAH_GREEDY) {
        if (sah_topology(nodes, ntriangles, leaf_node, out, err, topology == WIDE_TOPOLOGY_SAH) != 0) return -1;
    } else {
    // ---- collapse, one level of wide nodes at a time
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
    }

    phase("topology");
    finish_wide_tree(out, ntriangles);
    phase("stack need + record map");
    return 0;
}

uint32_t wide_stack_need(const uint32_t *wnodes, size_t nwide)
{
    std::vector<uint16_t> need(nwide, 0);
    for (size_t i = nwide; i-- > 0;) {
        const uint32_t *wn = wnodes + i * 32;
        uint32_t inner = 0, below = 0;
        for (int j = 0; j < (int)WIDE_K; j++) {
            uint32_t w = wn[4 * j + 3];
            if (w == WIDE_EMPTY || (w & WIDE_LEAF)) continue;
            inner++;
            if ((size_t)w < nwide && (size_t)w > i) below = std::max<uint32_t>(below, need[w]);
        }
        need[i] = (uint16_t)std::min<uint32_t>(0xFFFF, inner ? inner - 1 + below : 0);
    }
    return nwide ? need[0] : 0;
}

// What the device walks index with, checked before anything is uploaded: every inner child word names a
// LATER wide node (so a walk terminates and the stack-need sweep above is right), every leaf word a
// record inside the record table, every record a triangle of the mesh, every triangle a record that
// names it back.  The buffers on the device are sized from exactly these counts (nwide * 128 bytes of
// nodes, nrecords * 48 bytes of triangle records -- nrecords, not ntriangles: a triangle listed by
// several leaves has several records), so a tree that passes cannot make a kernel read past them.
int validate_wide_tree(const uint32_t *wnodes, size_t nwide, const uint32_t *tri_to_dev, uint32_t ntriangles,
                       const uint32_t *dev_to_tri, size_t nrecords, std::string &err)
{
    if (!wnodes || nwide == 0) { err = "wide tree: no nodes"; return -1; }
    if (nwide > 0x7FFFFFFFull || nrecords > 0x7FFFFFFFull) { err = "wide tree: more than 2^31 nodes or records"; return -1; }
    if (nrecords < ntriangles) { err = "wide tree: fewer triangle records than triangles"; return -1; }
    std::atomic<size_t> bad(SIZE_MAX);
    auto note = [&](size_t i) { size_t cur = bad.load(); while (i < cur && !bad.compare_exchange_weak(cur, i)) {} };
    parallel_for(nwide, [&](size_t a, size_t b) {
        for (size_t i = a; i < b; i++) {
            const uint32_t *wn = wnodes + i * 32;
            for (int j = 0; j < (int)WIDE_K; j++) {
                const uint32_t w = wn[4 * j + 3];
                // k_raycast_quad does not look at the child word of an entry whose box fails the slab test, and relies on
                // an EMPTY entry carrying the inverted box (lo = 0xFFFF, hi = 0 on every axis) to fail it: an empty word
                // with a real box would be taken for leaf record 0x7FFFFFFF
                if (w == WIDE_EMPTY) {
                    if (wn[4 * j] != 0x0000FFFFu || wn[4 * j + 1] != 0x0000FFFFu || wn[4 * j + 2] != 0x0000FFFFu) { note(i); break; }
                    continue;
                }
                bool ok = (w & WIDE_LEAF) ? (size_t)(w & ~WIDE_LEAF) < nrecords : ((size_t)w < nwide && (size_t)w > i);
                for (int ax = 0; ax < 3; ax++) ok = ok && (wn[4 * j + ax] & 0xFFFFu) <= (wn[4 * j + ax] >> 16);
                if (!ok) { note(i); break; }
            }
        }
    }, 1u << 14);
    if (bad != SIZE_MAX) {
        char buf[160];
        snprintf(buf, sizeof buf, "wide tree: node %zu has a child word outside the tree (%zu nodes, %zu records)", (size_t)bad, nwide, nrecords);
        err = buf;
        return -1;
    }
    parallel_for(nrecords, [&](size_t a, size_t b) { for (size_t k = a; k < b; k++) if (dev_to_tri[k] >= ntriangles) { note(k); break; } }, 1u << 16);
    if (bad != SIZE_MAX) { err = "wide tree: a triangle record names a triangle outside the mesh"; return -1; }
    parallel_for(ntriangles, [&](size_t a, size_t b) {
        for (size_t t = a; t < b; t++) if (tri_to_dev[t] >= nrecords || dev_to_tri[tri_to_dev[t]] != t) { note(t); break; }
    }, 1u << 16);
    if (bad != SIZE_MAX) { err = "wide tree: a triangle has no record that names it"; return -1; }
    return 0;
}

int wide_topology_from_env()
{
    const char *e = getenv("CHROMA_TREE");
    if (e && !strcmp(e, "collapse")) return WIDE_TOPOLOGY_COLLAPSE;
    if (e && !strcmp(e, "greedy")) return WIDE_TOPOLOGY_SAH_GREEDY;
    if (e && !strcmp(e, "ploc")) return WIDE_TOPOLOGY_PLOC;
    if (e && !strcmp(e, "sah")) return WIDE_TOPOLOGY_SAH;
    return WIDE_TOPOLOGY_LEVELS;
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
    if (chroma_host::build_wide_tree(nodes, (size_t)nnodes, ntriangles, *t, err, chroma_host::wide_topology_from_env()) != 0) { delete t; return CHROMA_ERR_INVALID; }
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
int chroma_wide_validate(const uint32_t *wnodes, uint64_t nwide, const uint32_t *tri_to_record, uint32_t ntriangles,
                         const uint32_t *record_to_tri, uint64_t nrecords)
{
    if (!wnodes || !tri_to_record || !record_to_tri) return CHROMA_ERR_INVALID;
    std::string err;
    return chroma_host::validate_wide_tree(wnodes, (size_t)nwide, tri_to_record, ntriangles, record_to_tri, (size_t)nrecords, err) == 0
               ? CHROMA_OK : CHROMA_ERR_INVALID;
}
int chroma_wide_free(void *handle)
{
    delete (chroma_host::WideTree *)handle;
    return CHROMA_OK;
}
}
