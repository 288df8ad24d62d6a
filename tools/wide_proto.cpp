// wide_proto.cpp -- CPU experiment: how many node / triangle fetches does a ray need when the
// reference BVH is collapsed into K-wide nodes walked nearest-first?  Not part of the product.
// usage: wide_proto DIR nrays   (DIR holds nodes.bin vertices.bin triangles.bin meta.txt)
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cmath>
#include <vector>
#include <unordered_map>
#include <algorithm>
#include <random>
#include <string>
struct U4 { uint32_t x, y, z, w; };
struct V3 { float x, y, z; };
static std::vector<U4> nodes; static std::vector<V3> verts; static std::vector<uint32_t> tris;
static float wo[3], ws;
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
static float box_t(V3 o, V3 id, U4 n, int grow) {
    float lo[3] = {wo[0] + ((float)(n.x & 0xFFFF) - grow) * ws, wo[1] + ((float)(n.y & 0xFFFF) - grow) * ws, wo[2] + ((float)(n.z & 0xFFFF) - grow) * ws};
    float hi[3] = {wo[0] + ((float)(n.x >> 16) + grow) * ws, wo[1] + ((float)(n.y >> 16) + grow) * ws, wo[2] + ((float)(n.z >> 16) + grow) * ws};
    float oo[3] = {o.x, o.y, o.z}, ii[3] = {id.x, id.y, id.z};
    float tmin = 0, tmax = INFINITY;
    for (int k = 0; k < 3; k++) {
        float t0 = (lo[k] - oo[k]) * ii[k], t1 = (hi[k] - oo[k]) * ii[k];
        tmin = std::max(tmin, std::min(t0, t1)); tmax = std::min(tmax, std::max(t0, t1));
    }
    return tmin > tmax ? -1.f : tmin;
}
struct Stat { double nodes = 0, tris = 0, maxsp = 0, stale = 0; int n = 0; };
// reference walk
static int ref_walk(V3 o, V3 d, int last, float &md, Stat &st) {
    V3 id = {1 / d.x, 1 / d.y, 1 / d.z}; md = -1; int best = -1;
    std::vector<uint32_t> stack; U4 root = nodes[0];
    if (box_t(o, id, root, 0) < 0) return -1;
    stack.push_back(root.w);
    while (!stack.empty()) {
        uint32_t w = stack.back(); stack.pop_back();
        uint32_t c = w & 0x0FFFFFFF, nc = w >> 28;
        for (uint32_t i = c; i < c + nc; i++) {
            U4 n = nodes[i]; st.nodes++;
            float t = box_t(o, id, n, 0);
            if (t < 0 || (md >= 0 && t > md)) continue;
            if ((n.w >> 28) == 0) { uint32_t tr = n.w & 0x0FFFFFFF; if ((int)tr == last) continue; st.tris++; float dd; if (tri_hit(o, d, tr, dd) && (best < 0 || dd < md)) { best = tr; md = dd; } }
            else { stack.push_back(n.w); st.maxsp = std::max(st.maxsp, (double)stack.size()); }
        }
    }
    return best;
}
// lazily collapsed wide nodes
struct Wide { int n; uint32_t ref[16]; };
static std::unordered_map<uint32_t, Wide> memo;
static double area(U4 n) { double dx = (double)(n.x >> 16) - (n.x & 0xFFFF), dy = (double)(n.y >> 16) - (n.y & 0xFFFF), dz = (double)(n.z >> 16) - (n.z & 0xFFFF); return dx * dy + dy * dz + dz * dx; }
static const Wide &wide_of(uint32_t refidx, int K) {
    auto it = memo.find(refidx); if (it != memo.end()) return it->second;
    Wide w; w.n = 0; U4 r = nodes[refidx]; uint32_t c = r.w & 0x0FFFFFFF, nc = r.w >> 28;
    for (uint32_t i = c; i < c + nc; i++) w.ref[w.n++] = i;
    for (;;) {
        int pick = -1; double best = -1;
        for (int i = 0; i < w.n; i++) { U4 n = nodes[w.ref[i]]; int k = n.w >> 28; if (k == 0 || w.n - 1 + k > K) continue; double a = area(n); if (a > best) { best = a; pick = i; } }
        if (pick < 0) break;
        U4 n = nodes[w.ref[pick]]; uint32_t cc = n.w & 0x0FFFFFFF, k = n.w >> 28;
        w.ref[pick] = cc; for (uint32_t j = 1; j < k; j++) w.ref[w.n++] = cc + j;
    }
    return memo.emplace(refidx, w).first->second;
}
static double fill_sum = 0, fill_n = 0;
static int wide_walk(V3 o, V3 d, int last, float &md, Stat &st, int K, bool sortall, int defer) {
    V3 id = {1 / d.x, 1 / d.y, 1 / d.z}; md = -1; int best = -1;
    struct E { uint32_t ref; float t; };
    std::vector<E> stack; std::vector<uint32_t> pend;
    if (box_t(o, id, nodes[0], 1) < 0) return -1;
    stack.push_back({0, 0.f});
    auto flush = [&]() { for (uint32_t tr : pend) { st.tris++; float dd; if (tri_hit(o, d, tr, dd) && (best < 0 || dd < md)) { best = tr; md = dd; } } pend.clear(); };
    while (true) {
        if (stack.empty()) { if (pend.empty()) break; flush(); continue; }
        E e = stack.back(); stack.pop_back();
        if (md >= 0 && e.t > md) { continue; }
        const Wide &w = wide_of(e.ref, K); st.nodes++; fill_sum += w.n; fill_n++;
        E hits[16]; int nh = 0;
        for (int i = 0; i < w.n; i++) {
            U4 n = nodes[w.ref[i]]; float t = box_t(o, id, n, 1);
            if (t < 0 || (md >= 0 && t > md)) continue;
            if ((n.w >> 28) == 0) { uint32_t tr = n.w & 0x0FFFFFFF; if ((int)tr != last) pend.push_back(tr); }
            else hits[nh++] = {w.ref[i], t};
        }
        if ((int)pend.size() > defer) flush();
        if (sortall) std::sort(hits, hits + nh, [](const E &a, const E &b) { return a.t > b.t; });
        else if (nh > 1) { int m = 0; for (int i = 1; i < nh; i++) if (hits[i].t < hits[m].t) m = i; std::swap(hits[m], hits[nh - 1]); }
        for (int i = 0; i < nh; i++) stack.push_back(hits[i]);
        st.maxsp = std::max(st.maxsp, (double)stack.size());
    }
    return best;
}
int main(int argc, char **argv) {
    std::string dir = argv[1]; int nr = atoi(argv[2]);
    nodes = slurp<U4>(dir + "/nodes.bin"); verts = slurp<V3>(dir + "/vertices.bin"); tris = slurp<uint32_t>(dir + "/triangles.bin");
    FILE *f = fopen((dir + "/meta.txt").c_str(), "r"); if (fscanf(f, "%f %f %f %f", &wo[0], &wo[1], &wo[2], &ws) != 4) return 1; fclose(f);
    printf("nodes %zu tris %zu\n", nodes.size(), tris.size() / 3);
    std::mt19937 g(1); std::uniform_real_distribution<float> U(0, 1);
    auto iso = [&]() { float th = 6.2831853f * U(g), u = 2 * U(g) - 1, c = sqrtf(1 - u * u); return V3{c * cosf(th), c * sinf(th), u}; };
    std::vector<V3> O, D; std::vector<int> L;
    for (int i = 0; i < nr; i++) { O.push_back({0, 0, 0}); D.push_back(iso()); L.push_back(-1); }
    for (int gen = 0; gen < 2; gen++) {
        printf("== generation %d (%zu rays)\n", gen, O.size());
        Stat s0; std::vector<int> h0(O.size()); std::vector<float> d0(O.size());
        for (size_t i = 0; i < O.size(); i++) h0[i] = ref_walk(O[i], D[i], L[i], d0[i], s0);
        printf("  reference: nodes %.1f (%.0f B) tris %.2f maxsp %.0f\n", s0.nodes / O.size(), 16 * s0.nodes / O.size(), s0.tris / O.size(), s0.maxsp);
        int Ks[] = {4, 6, 8, 12};
        for (int K : Ks) for (int so = 0; so < 2; so++) for (int defer : {0, 8}) {
            memo.clear(); fill_sum = fill_n = 0; Stat s; int bad = 0;
            for (size_t i = 0; i < O.size(); i++) { float dd; int h = wide_walk(O[i], D[i], L[i], dd, s, K, so, defer); if (h != h0[i] || (h >= 0 && dd != d0[i])) bad++; }
            printf("  wide K=%2d sortall=%d defer=%d: nodes %.1f fill %.2f tris %.2f maxsp %.0f  bytes(16/child) %.0f  mismatches %d\n", K, so, defer, s.nodes / O.size(), fill_sum / fill_n, s.tris / O.size(), s.maxsp,
                   s.nodes / O.size() * 16 * K + 48 * s.tris / O.size(), bad);
        }
        // secondary rays: from the hit point, random direction in the hemisphere back toward the origin side
        std::vector<V3> O2, D2; std::vector<int> L2;
        for (size_t i = 0; i < O.size(); i++) if (h0[i] >= 0) { V3 p = {O[i].x + D[i].x * d0[i], O[i].y + D[i].y * d0[i], O[i].z + D[i].z * d0[i]}; V3 nd = iso(); if (dot(nd, D[i]) > 0) nd = {-nd.x, -nd.y, -nd.z}; O2.push_back(p); D2.push_back(nd); L2.push_back(h0[i]); }
        O = O2; D = D2; L = L2;
    }
    return 0;
}
