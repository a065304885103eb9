// tools/trav_lab2.cpp — developer sandbox: does re-clustering the reference's leaves (A) or
// triangles (B) under a binned-SAH tree cut traversal work?  Counts only; CPU.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#include <cfloat>
struct Node { float bMin[3], bMax[3]; int childL, childR, primStart, primEnd; };
struct Tri { float V0[3], E1[3], E2[3]; };
static std::vector<Node> N; static std::vector<Tri> T;
template <class X> static std::vector<X> slurp(const char* p) { FILE* f = fopen(p, "rb"); fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET); std::vector<X> v(n / sizeof(X)); if (fread(v.data(), 1, n, f) != (size_t)n) exit(1); fclose(f); return v; }
static inline void cross(const float* a, const float* b, float* o) { o[0] = a[1] * b[2] - a[2] * b[1]; o[1] = -(a[0] * b[2] - a[2] * b[0]); o[2] = a[0] * b[1] - a[1] * b[0]; }
static inline float dot(const float* a, const float* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static bool tri(int i, const float* o, const float* d, float tmax, float& t) {
    const Tri& r = T[i]; float Tv[3] = {o[0] - r.V0[0], o[1] - r.V0[1], o[2] - r.V0[2]}, P[3], Q[3];
    cross(d, r.E2, P); cross(Tv, r.E1, Q); float det = dot(P, r.E1); if (det < 1e-4f) return false;
    float inv = 1.f / det; t = dot(Q, r.E2) * inv; if (t < 0 || t > tmax) return false;
    float u = dot(P, Tv); if (u < 0 || u > det) return false; float v = dot(Q, d); if (v < 0 || u + v > det) return false; return true;
}
struct RayS { float o[3], d[3], invD[3], L; bool degen; };
static bool box(const float* mn, const float* mx, const RayS& r, float& tn) {
    tn = 0; float tf = 1e30f;
    for (int a = 0; a < 3; a++) { float x1 = (mn[a] - r.o[a]) * r.invD[a], x2 = (mx[a] - r.o[a]) * r.invD[a]; tn = std::max(tn, std::min(x1, x2)); tf = std::min(tf, std::max(x1, x2)); }
    tf *= 1.00000024f; return r.degen || tn <= tf;
}
// ---- generic items + binned SAH ----
struct Item { float mn[3], mx[3]; int a, b; };          // payload: [a,b] prim range (A) or single tri a (B)
struct BN { float mn[3], mx[3]; int l, r, first, count; };   // count>0 leaf over items[first..first+count)
static std::vector<Item> items; static std::vector<BN> B;
static float area(const float* mn, const float* mx) { float d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]}; return 2 * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]); }
static int build(int lo, int hi, int maxLeaf, int depth, int& maxd) {
    int me = B.size(); B.emplace_back(); if (depth > maxd) maxd = depth;
    float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}, cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (int i = lo; i < hi; i++) for (int a = 0; a < 3; a++) { mn[a] = std::min(mn[a], items[i].mn[a]); mx[a] = std::max(mx[a], items[i].mx[a]); float c = 0.5f * (items[i].mn[a] + items[i].mx[a]); cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c); }
    memcpy(B[me].mn, mn, 12); memcpy(B[me].mx, mx, 12);
    int n = hi - lo;
    if (n <= maxLeaf) { B[me].first = lo; B[me].count = n; B[me].l = B[me].r = -1; return me; }
    const int NB = 32; float best = FLT_MAX; int bax = -1, bsp = -1;
    for (int ax = 0; ax < 3; ax++) {
        float ext = cmx[ax] - cmn[ax]; if (!(ext > 0)) continue;
        int cnt[NB] = {0}; float bmn[NB][3], bmx[NB][3]; for (int k = 0; k < NB; k++) for (int a = 0; a < 3; a++) { bmn[k][a] = FLT_MAX; bmx[k][a] = -FLT_MAX; }
        for (int i = lo; i < hi; i++) { float c = 0.5f * (items[i].mn[ax] + items[i].mx[ax]); int k = std::min(NB - 1, (int)((c - cmn[ax]) / ext * NB)); cnt[k]++; for (int a = 0; a < 3; a++) { bmn[k][a] = std::min(bmn[k][a], items[i].mn[a]); bmx[k][a] = std::max(bmx[k][a], items[i].mx[a]); } }
        float ra[NB]; int rc[NB]; float am[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, aM[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}; int c = 0;
        for (int k = NB - 1; k > 0; k--) { c += cnt[k]; for (int a = 0; a < 3; a++) { am[a] = std::min(am[a], bmn[k][a]); aM[a] = std::max(aM[a], bmx[k][a]); } ra[k] = c ? area(am, aM) : 0; rc[k] = c; }
        float lm[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lM[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}; c = 0;
        for (int k = 0; k < NB - 1; k++) { c += cnt[k]; for (int a = 0; a < 3; a++) { lm[a] = std::min(lm[a], bmn[k][a]); lM[a] = std::max(lM[a], bmx[k][a]); } if (!c || !rc[k + 1]) continue; float cost = area(lm, lM) * c + ra[k + 1] * rc[k + 1]; if (cost < best) { best = cost; bax = ax; bsp = k; } }
    }
    int mid;
    if (bax < 0) mid = (lo + hi) / 2;
    else { float ext = cmx[bax] - cmn[bax]; auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const Item& x) { float c = 0.5f * (x.mn[bax] + x.mx[bax]); int k = std::min(NB - 1, (int)((c - cmn[bax]) / ext * NB)); return k <= bsp; }); mid = it - items.begin(); if (mid == lo || mid == hi) mid = (lo + hi) / 2; }
    int l = build(lo, mid, maxLeaf, depth + 1, maxd), r = build(mid, hi, maxLeaf, depth + 1, maxd); B[me].l = l; B[me].r = r; B[me].count = 0; return me;
}
struct Stat { double nodes = 0, tris = 0, leaves = 0, boxes = 0; int maxsp = 0; };
static std::vector<int> leafOfTri;   // B: reference leaf id per tri
static void trav(const RayS& r, float tmax, Stat& st, float& bt, int& bp, bool modeB, bool shadowAny, float stopBelow) {
    bt = tmax; bp = -1; const float k = 1.0078125f / r.L; int stack[128], sp = 0, cur = 0;
    auto leaf = [&](const BN& n) { st.leaves++; for (int q = n.first; q < n.first + n.count; q++) { const Item& it = items[q];
            if (!modeB) { for (int i = it.a; i <= it.b; i++) { st.tris++; float t; if (tri(i, r.o, r.d, bt, t) && (t < bt || i > bp)) { bt = t; bp = i; } } }
            else { st.tris++; float t; int i = it.a; if (tri(i, r.o, r.d, bt, t) && (t < bt || i > bp)) { float tn; if (box(N[leafOfTri[i]].bMin, N[leafOfTri[i]].bMax, r, tn)) { bt = t; bp = i; } } } } };
    if (B[0].count) { leaf(B[0]); return; }
    for (;;) { const BN& n = B[cur]; st.nodes++; int c[2] = {n.l, n.r}; float tn[2]; bool ok[2];
        for (int j = 0; j < 2; j++) { st.boxes++; ok[j] = box(B[c[j]].mn, B[c[j]].mx, r, tn[j]) && (r.degen || tn[j] <= bt * k); }
        for (int j = 0; j < 2; j++) if (ok[j] && B[c[j]].count) { if (r.degen || tn[j] <= bt * k) leaf(B[c[j]]); ok[j] = false; }
        if (shadowAny && bp >= 0 && bt < stopBelow) return;
        if (ok[0] && ok[1]) { int nr = tn[0] <= tn[1] ? 0 : 1; stack[sp++] = c[1 - nr]; st.maxsp = std::max(st.maxsp, sp); cur = c[nr]; } else if (ok[0]) cur = c[0]; else if (ok[1]) cur = c[1]; else { if (!sp) return; cur = stack[--sp]; } }
}
static void refcast(const RayS& r, float tmax, float& bt, int& bp) { bt = tmax; bp = -1; int stack[256], sp = 0; stack[sp++] = 0;
    while (sp) { const Node& n = N[stack[--sp]]; float tn; if (!box(n.bMin, n.bMax, r, tn)) continue; if (!r.degen && tn > bt * 1.00000024f) continue;
        if (n.primStart != -1) for (int i = n.primStart; i <= n.primEnd; i++) { float t; if (tri(i, r.o, r.d, bt, t)) { bt = t; bp = i; } }
        if (n.childR > 0) stack[sp++] = n.childR; if (n.childL > 0) stack[sp++] = n.childL; } }
int main(int argc, char** argv) {
    N = slurp<Node>(argv[1]); auto tf = slurp<float>(argv[2]); auto rays = slurp<float>(argv[3]);
    size_t nt = tf.size() / 88; T.resize(nt); std::vector<float> V1(nt * 3), V2(nt * 3);
    for (size_t i = 0; i < nt; i++) { memcpy(T[i].V0, &tf[i * 88], 12); memcpy(T[i].E1, &tf[i * 88 + 39], 12); memcpy(T[i].E2, &tf[i * 88 + 42], 12); memcpy(&V1[i * 3], &tf[i * 88 + 3], 12); memcpy(&V2[i * 3], &tf[i * 88 + 6], 12); }
    size_t nr = rays.size() / 7; leafOfTri.resize(nt);
    for (size_t i = 0; i < N.size(); i++) if (N[i].primStart != -1) for (int k = N[i].primStart; k <= N[i].primEnd; k++) leafOfTri[k] = i;
    for (int mode = 0; mode < 2; mode++) for (int maxLeaf : {1, 2, 4}) {
        if (mode == 1 && maxLeaf == 1) continue;
        items.clear(); B.clear();
        if (mode == 0) { for (auto& n : N) if (n.primStart != -1) { Item it; memcpy(it.mn, n.bMin, 12); memcpy(it.mx, n.bMax, 12); it.a = n.primStart; it.b = n.primEnd; items.push_back(it); } }
        else for (size_t i = 0; i < nt; i++) { Item it; for (int a = 0; a < 3; a++) { it.mn[a] = std::min(T[i].V0[a], std::min(V1[i * 3 + a], V2[i * 3 + a])); it.mx[a] = std::max(T[i].V0[a], std::max(V1[i * 3 + a], V2[i * 3 + a])); } it.a = it.b = i; items.push_back(it); }
        int maxd = 0; build(0, items.size(), maxLeaf, 0, maxd);
        Stat s, s2; int mism = 0;
        for (size_t i = 0; i < nr; i++) { RayS r; memcpy(r.o, &rays[i * 7], 12); memcpy(r.d, &rays[i * 7 + 3], 12); float tmax = rays[i * 7 + 6];
            float inv[3] = {1.f / r.d[0], 1.f / r.d[1], 1.f / r.d[2]}; r.L = sqrtf(inv[0] * inv[0] + inv[1] * inv[1] + inv[2] * inv[2]); for (int k = 0; k < 3; k++) r.invD[k] = inv[k] / r.L; r.degen = !(r.L < INFINITY);
            float t0, t1; int p0, p1; refcast(r, tmax, t0, p0); trav(r, tmax, s, t1, p1, mode == 1, false, 0); if (p0 != p1 || t0 != t1) mism++;
            bool sh = tmax < 999998.f; float t2; int p2; trav(r, tmax, s2, t2, p2, mode == 1, sh, tmax - 1.0f - 3e-4f); }
        printf("%s maxLeaf %d: nodes %zu depth %d | wide/ray %.1f leaves/ray %.1f tris/ray %.1f maxsp %d mism %d | any-hit shadows: wide %.1f tris %.1f\n", mode ? "B(tris)  " : "A(leaves)", maxLeaf, B.size(), maxd, s.nodes / nr, s.leaves / nr, s.tris / nr, s.maxsp, mism, s2.nodes / nr, s2.tris / nr);
    }
    return 0;
}
