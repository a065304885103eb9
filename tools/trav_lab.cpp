// tools/trav_lab.cpp — developer sandbox (not product, not oracle): counts BVH node/triangle
// work of candidate traversal strategies on a logged ray set, on the CPU.
//   trav_lab nodes.bin tris88.bin rays7.bin
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
struct Node { float bMin[3], bMax[3]; int childL, childR, primStart, primEnd; };
struct Tri { float V0[3], E1[3], E2[3]; };
static std::vector<Node> N; static std::vector<Tri> T;
template <class X> static std::vector<X> slurp(const char* p) { FILE* f = fopen(p, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<X> v(n / sizeof(X)); fread(v.data(), 1, n, f); fclose(f); return v; }
static inline void cross(const float* a, const float* b, float* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = -(a[0] * b[2] - a[2] * b[0]); o[2] = a[0] * b[1] - a[1] * b[0]; }
static inline float dot(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static bool tri(int i, const float* o, const float* d, float tmax, float& t) {
    const Tri& r = T[i]; float Tv[3] = {o[0] - r.V0[0], o[1] - r.V0[1], o[2] - r.V0[2]}, P[3], Q[3];
    cross(d, r.E2, P); cross(Tv, r.E1, Q); float det = dot(P, r.E1); if (det < 1e-4f) return false;
    float inv = 1.f / det; t = dot(Q, r.E2) * inv; if (t < 0 || t > tmax) return false;
    float u = dot(P, Tv); if (u < 0 || u > det) return false; float v = dot(Q, d); if (v < 0 || u + v > det) return false; return true;
}
struct RayS { float o[3], d[3], invD[3], L; bool degen; };
static bool box(const float* mn, const float* mx, const RayS& r, float& tn, float& tf) {
    tn = 0; tf = 1e30f;
    for (int a = 0; a < 3; a++) { float x1 = (mn[a] - r.o[a]) * r.invD[a], x2 = (mx[a] - r.o[a]) * r.invD[a]; tn = std::max(tn, std::min(x1, x2)); tf = std::min(tf, std::max(x1, x2)); }
    tf *= 1.00000024f; return r.degen || tn <= tf;
}
struct Stat { double nodes = 0, tris = 0, leaves = 0, maxsp = 0; };
// S0: reference order, reference (weak) cull
static void s0(const RayS& r, float tmax, Stat& st, float& bt, int& bp) {
    bt = tmax; bp = -1; int stack[256], sp = 0; stack[sp++] = 0;
    while (sp) { const Node& n = N[stack[--sp]]; st.nodes++; float tn, tf; if (!box(n.bMin, n.bMax, r, tn, tf)) continue; if (!r.degen && tn > bt * 1.00000024f) continue;
        if (n.primStart != -1) { st.leaves++; for (int i = n.primStart; i <= n.primEnd; i++) { st.tris++; float t; if (tri(i, r.o, r.d, bt, t)) { bt = t; bp = i; } } }
        if (n.childR > 0) stack[sp++] = n.childR; if (n.childL > 0) stack[sp++] = n.childL; }
}
// S1: near-first, cull against closest (slack), children tested at the parent (wide-node emulation)
static void s1(const RayS& r, float tmax, Stat& st, float& bt, int& bp, bool recheckPop, bool shadowAny, float stopBelow) {
    bt = tmax; bp = -1; const float k = 1.0078125f / r.L;
    struct E { int n; float tn; } stack[256]; int sp = 0; int cur = 0;
    if (N[0].primStart != -1) { for (int i = N[0].primStart; i <= N[0].primEnd; i++) { st.tris++; float t; if (tri(i, r.o, r.d, bt, t)) { bt = t; bp = i; } } return; }
    for (;;) {
        const Node& n = N[cur]; st.nodes++;
        int c[2] = {n.childL, n.childR}; float tn[2], tf; bool ok[2];
        for (int j = 0; j < 2; j++) { ok[j] = box(N[c[j]].bMin, N[c[j]].bMax, r, tn[j], tf) && (r.degen || tn[j] <= bt * k); }
        for (int j = 0; j < 2; j++) if (ok[j] && N[c[j]].primStart != -1) {
            if (r.degen || tn[j] <= bt * k) { st.leaves++; for (int i = N[c[j]].primStart; i <= N[c[j]].primEnd; i++) { st.tris++; float t; if (tri(i, r.o, r.d, bt, t) && (t < bt || i > bp)) { bt = t; bp = i; } } }
            ok[j] = false; }
        if (shadowAny && bp >= 0 && bt < stopBelow) return;
        if (ok[0] && ok[1]) { int nr = tn[0] <= tn[1] ? 0 : 1; stack[sp++] = {c[1 - nr], tn[1 - nr]}; if (sp > st.maxsp) st.maxsp = sp; cur = c[nr]; }
        else if (ok[0]) cur = c[0]; else if (ok[1]) cur = c[1];
        else { for (;;) { if (!sp) return; E e = stack[--sp]; if (recheckPop && !r.degen && e.tn > bt * k) continue; cur = e.n; break; } }
    }
}
int main(int argc, char** argv) {
    N = slurp<Node>(argv[1]); auto tf = slurp<float>(argv[2]); auto rays = slurp<float>(argv[3]);
    size_t nt = tf.size() / 88; T.resize(nt);
    for (size_t i = 0; i < nt; i++) { memcpy(T[i].V0, &tf[i * 88], 12); memcpy(T[i].E1, &tf[i * 88 + 39], 12); memcpy(T[i].E2, &tf[i * 88 + 42], 12); }
    size_t nr = rays.size() / 7;
    // tree stats: "big" nodes (box diagonal > 30)
    int big = 0, inter = 0; for (auto& n : N) { float d = 0; for (int a = 0; a < 3; a++) d += (n.bMax[a] - n.bMin[a]) * (n.bMax[a] - n.bMin[a]); if (sqrtf(d) > 30) big++; if (n.primStart == -1) inter++; }
    printf("nodes %zu (interior %d), tris %zu, rays %zu, nodes with diagonal > 30: %d\n", N.size(), inter, nt, nr, big);
    Stat a, b, c, d; int mism = 0; long shadow = 0;
    for (size_t i = 0; i < nr; i++) {
        RayS r; memcpy(r.o, &rays[i * 7], 12); memcpy(r.d, &rays[i * 7 + 3], 12); float tmax = rays[i * 7 + 6];
        float inv[3] = {1.f / r.d[0], 1.f / r.d[1], 1.f / r.d[2]}; r.L = sqrtf(inv[0] * inv[0] + inv[1] * inv[1] + inv[2] * inv[2]); for (int k = 0; k < 3; k++) r.invD[k] = inv[k] / r.L; r.degen = !(r.L < INFINITY);
        float t0, t1, t2; int p0, p1, p2;
        s0(r, tmax, a, t0, p0); s1(r, tmax, b, t1, p1, false, false, 0); s1(r, tmax, c, t2, p2, true, false, 0);
        if (p0 != p1 || t0 != t1 || p0 != p2) mism++;
        bool isShadow = tmax < 999998.f; if (isShadow) shadow++;
        float t3; int p3; s1(r, tmax, d, t3, p3, true, isShadow, tmax - 1.0f - 3e-4f);
    }
    printf("mismatches vs reference order: %d; shadow rays %ld\n", mism, shadow);
    printf("S0 reference        : nodes/ray %.1f leaves/ray %.1f tris/ray %.1f\n", a.nodes / nr, a.leaves / nr, a.tris / nr);
    printf("S1 wide near-first  : wide/ray %.1f leaves/ray %.1f tris/ray %.1f maxsp %.0f\n", b.nodes / nr, b.leaves / nr, b.tris / nr, b.maxsp);
    printf("S1 + recheck on pop : wide/ray %.1f leaves/ray %.1f tris/ray %.1f\n", c.nodes / nr, c.leaves / nr, c.tris / nr);
    printf("S1 + any-hit shadows: wide/ray %.1f leaves/ray %.1f tris/ray %.1f\n", d.nodes / nr, d.leaves / nr, d.tris / nr);
    return 0;
}
