// sah_proto.cpp -- CPU experiment: visit counts of a freshly built binned-SAH BVH collapsed to K-wide
// nodes, against the reference tree.  Not part of the product.
// usage: sah_proto DIR nrays maxleaf
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <algorithm>
#include <random>
#include <string>
#include <numeric>
struct V3 { float x, y, z; };
static std::vector<V3> verts; static std::vector<uint32_t> tris;
template <class T> static std::vector<T> slurp(const std::string &p) {
    FILE *f = fopen(p.c_str(), "rb"); if (!f) { perror(p.c_str()); exit(1); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<T> v(n / sizeof(T)); if (fread(v.data(), 1, n, f) != (size_t)n) exit(1); fclose(f); return v;
}
static inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static bool tri_hit(V3 o, V3 d, uint32_t t, float &dist) {
    V3 v0 = verts[tris[3 * t]], v1 = verts[tris[3 * t + 1]], v2 = verts[tris[3 * t + 2]];
    V3 e1 = sub(v1, v0), e2 = sub(v2, v0), h = cross(d, e2); float a = dot(e1, h);
    if (a > -1.19e-7f && a < 1.19e-7f) return false;
    float f = 1.0f / a; V3 s = sub(o, v0); float u = f * dot(s, h);
    if (u < -1e-6 || u > 1 + 1e-6) return false;
    V3 q = cross(s, e1); float v = f * dot(d, q);
    if (v < -1e-6 || u + v > 1 + 1e-6) return false;
    float tt = f * dot(e2, q); if (tt > 1e-6) { dist = tt; return true; } return false;
}
struct Box { float lo[3], hi[3];
    void clear() { for (int k = 0; k < 3; k++) { lo[k] = INFINITY; hi[k] = -INFINITY; } }
    void add(const Box &b) { for (int k = 0; k < 3; k++) { lo[k] = std::min(lo[k], b.lo[k]); hi[k] = std::max(hi[k], b.hi[k]); } }
    double area() const { double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2]; return dx * dy + dy * dz + dz * dx; } };
static float box_t(V3 o, V3 id, const Box &b) {
    float oo[3] = {o.x, o.y, o.z}, ii[3] = {id.x, id.y, id.z}; float tmin = 0, tmax = INFINITY;
    for (int k = 0; k < 3; k++) { float t0 = (b.lo[k] - oo[k]) * ii[k], t1 = (b.hi[k] - oo[k]) * ii[k];
        tmin = std::max(tmin, std::min(t0, t1)); tmax = std::min(tmax, std::max(t0, t1)); }
    return tmin > tmax ? -1.f : tmin;
}
struct N2 { Box b; int left, right, first, count; };   // leaf: left == -1
static std::vector<N2> n2; static std::vector<uint32_t> order; static std::vector<Box> tb; static std::vector<V3> tc;
static int MAXLEAF = 1;
static int build(int first, int count) {
    N2 n; n.b.clear(); Box cb; cb.clear();
    for (int i = first; i < first + count; i++) { n.b.add(tb[order[i]]); V3 c = tc[order[i]]; Box p; p.lo[0] = p.hi[0] = c.x; p.lo[1] = p.hi[1] = c.y; p.lo[2] = p.hi[2] = c.z; cb.add(p); }
    n.left = n.right = -1; n.first = first; n.count = count;
    int id = n2.size(); n2.push_back(n);
    if (count <= 1) return id;
    // binned SAH
    const int NB = 16; double bestc = INFINITY; int besta = -1, bestb = -1;
    for (int a = 0; a < 3; a++) {
        float ext = cb.hi[a] - cb.lo[a]; if (!(ext > 0)) continue;
        Box bb[NB]; int bc[NB] = {0}; for (auto &x : bb) x.clear();
        for (int i = first; i < first + count; i++) { float c = a == 0 ? tc[order[i]].x : a == 1 ? tc[order[i]].y : tc[order[i]].z; int k = std::min(NB - 1, (int)((c - cb.lo[a]) / ext * NB)); bb[k].add(tb[order[i]]); bc[k]++; }
        Box r; double ra[NB]; int rc[NB]; r.clear(); int c = 0;
        for (int k = NB - 1; k > 0; k--) { r.add(bb[k]); c += bc[k]; ra[k] = r.area(); rc[k] = c; }
        Box l; l.clear(); c = 0;
        for (int k = 0; k < NB - 1; k++) { l.add(bb[k]); c += bc[k]; if (c == 0 || rc[k + 1] == 0) continue; double cost = l.area() * c + ra[k + 1] * rc[k + 1]; if (cost < bestc) { bestc = cost; besta = a; bestb = k; } }
    }
    int mid;
    if (besta < 0) { if (count <= MAXLEAF) return id; mid = first + count / 2; }
    else {
        if (count <= MAXLEAF && bestc >= n.b.area() * count) return id;   // leaf is cheaper
        float ext = cb.hi[besta] - cb.lo[besta];
        auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) { float c = besta == 0 ? tc[t].x : besta == 1 ? tc[t].y : tc[t].z; int k = std::min(NB - 1, (int)((c - cb.lo[besta]) / ext * NB)); return k <= bestb; });
        mid = it - order.begin();
        if (mid == first || mid == first + count) mid = first + count / 2;
    }
    int l = build(first, mid - first); int r = build(mid, first + count - mid);
    n2[id].left = l; n2[id].right = r; return id;
}
struct W { int n; int child[16]; };   // child >= 0: n2 index (inner or leaf)
static std::vector<W> wide; static std::vector<int> wide_of;
static int collapse(int root, int K) {
    W w; w.n = 0; w.child[w.n++] = n2[root].left; w.child[w.n++] = n2[root].right;
    for (;;) { int pick = -1; double best = -1; if (w.n >= K) break;
        for (int i = 0; i < w.n; i++) { const N2 &c = n2[w.child[i]]; if (c.left < 0) continue; double a = c.b.area(); if (a > best) { best = a; pick = i; } }
        if (pick < 0) break; int c = w.child[pick]; w.child[pick] = n2[c].left; w.child[w.n++] = n2[c].right; }
    int id = wide.size(); wide.push_back(w); wide_of[root] = id;
    for (int i = 0; i < w.n; i++) if (n2[w.child[i]].left >= 0) collapse(w.child[i], K);
    return id;
}
struct Stat { double nodes = 0, tris = 0, leaves = 0, maxsp = 0; };
static int walk(V3 o, V3 d, int last, float &md, Stat &st) {
    V3 id = {1 / d.x, 1 / d.y, 1 / d.z}; md = -1; int best = -1;
    struct E { int n; float t; }; std::vector<E> stack; stack.push_back({0, 0.f});
    while (!stack.empty()) {
        E e = stack.back(); stack.pop_back(); if (md >= 0 && e.t > md) continue;
        const W &w = wide[wide_of[e.n]]; st.nodes++;
        E hits[16]; int nh = 0;
        for (int i = 0; i < w.n; i++) { const N2 &c = n2[w.child[i]]; float t = box_t(o, id, c.b); if (t < 0 || (md >= 0 && t > md)) continue;
            if (c.left < 0) { st.leaves++; for (int j = c.first; j < c.first + c.count; j++) { if ((int)order[j] == last) continue; st.tris++; float dd; if (tri_hit(o, d, order[j], dd) && (best < 0 || dd < md)) { best = order[j]; md = dd; } } }
            else hits[nh++] = {w.child[i], t}; }
        std::sort(hits, hits + nh, [](const E &a, const E &b) { return a.t > b.t; });
        for (int i = 0; i < nh; i++) stack.push_back(hits[i]);
        st.maxsp = std::max(st.maxsp, (double)stack.size());
    }
    return best;
}
int main(int argc, char **argv) {
    std::string dir = argv[1]; int nr = atoi(argv[2]); MAXLEAF = atoi(argv[3]);
    verts = slurp<V3>(dir + "/vertices.bin"); tris = slurp<uint32_t>(dir + "/triangles.bin");
    size_t nt = tris.size() / 3; tb.resize(nt); tc.resize(nt); order.resize(nt); std::iota(order.begin(), order.end(), 0);
    for (size_t t = 0; t < nt; t++) { Box b; b.clear(); for (int k = 0; k < 3; k++) { V3 v = verts[tris[3 * t + k]]; Box p; p.lo[0] = p.hi[0] = v.x; p.lo[1] = p.hi[1] = v.y; p.lo[2] = p.hi[2] = v.z; b.add(p); } tb[t] = b; tc[t] = {(b.lo[0] + b.hi[0]) / 2, (b.lo[1] + b.hi[1]) / 2, (b.lo[2] + b.hi[2]) / 2}; }
    n2.reserve(2 * nt); build(0, nt); printf("bvh2 nodes %zu\n", n2.size());
    std::mt19937 g(1); std::uniform_real_distribution<float> U(0, 1);
    auto iso = [&]() { float th = 6.2831853f * U(g), u = 2 * U(g) - 1, c = sqrtf(1 - u * u); return V3{c * cosf(th), c * sinf(th), u}; };
    for (int K : {4, 8}) {
        wide.clear(); wide_of.assign(n2.size(), -1); collapse(0, K);
        double fill = 0; for (auto &w : wide) fill += w.n; printf("K=%d wide nodes %zu fill %.2f\n", K, wide.size(), fill / wide.size());
        g.seed(1);
        std::vector<V3> O, D; std::vector<int> L;
        for (int i = 0; i < nr; i++) { O.push_back({0, 0, 0}); D.push_back(iso()); L.push_back(-1); }
        for (int gen = 0; gen < 2; gen++) {
            Stat s; std::vector<V3> O2, D2; std::vector<int> L2; int nh = 0;
            for (size_t i = 0; i < O.size(); i++) { float dd; int h = walk(O[i], D[i], L[i], dd, s); if (h >= 0) { nh++; V3 p = {O[i].x + D[i].x * dd, O[i].y + D[i].y * dd, O[i].z + D[i].z * dd}; V3 nd = iso(); if (dot(nd, D[i]) > 0) nd = {-nd.x, -nd.y, -nd.z}; O2.push_back(p); D2.push_back(nd); L2.push_back(h); } }
            printf("  gen %d: wide nodes %.1f leaves %.2f tris %.2f maxsp %.0f hits %d\n", gen, s.nodes / O.size(), s.leaves / O.size(), s.tris / O.size(), s.maxsp, nh);
            O = O2; D = D2; L = L2;
        }
    }
    return 0;
}
