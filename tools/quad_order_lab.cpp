// quad_order_lab.cpp — host-only experiment: how much does the ORDER in which wf_trace visits the hit children of a 4-wide node matter?
// Builds a config scene and its quantised 4-wide tree exactly as the upload does (host/accel_build.cpp), shoots a representative ray
// set (camera rays, cosine-weighted bounce rays from their hits, shadow rays towards the light) and walks the tree one ray at a time
// — same slab test on the dequantised boxes, far side cut at the closest hit, nearest child next, the others pushed far to near —
// under different ordering policies, counting node steps and leaf visits per ray:
//   0 exact     children sorted by entry distance (what wf_trace does: five compare-exchanges per node step)
//   1 axis      children stored sorted along the axis of their largest centroid spread; visited in that order or the reverse by the
//               sign of the ray direction on that axis (no per-node sort: one sign test)
//   2 nearest   the nearest child exactly (a 4-way minimum), the others in slot order
//   3 slot      slot order, no ordering at all
//   g++ -std=c++17 -O2 -I include tools/quad_order_lab.cpp pathtrace-on-cuda_amd/build/{accel_build,bvh_build,scenes,pt_host,obj_loader}.o -pthread -o /tmp/quad_order_lab
//   /tmp/quad_order_lab [kind=1] [lat_lon=187] [pixels=60000]
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "../include/pt_api.h"
#include "../pathtrace-on-cuda_amd/host/accel_build.h"

struct V { float x, y, z; };
static V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
static V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static V operator*(V a, float s) { return {a.x * s, a.y * s, a.z * s}; }
static float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
static V norm(V a) { return a * (1.f / std::sqrt(dot(a, a))); }

static const PtAccel* A;
struct Ray { V o, d; float tmax; bool any; };
struct Cnt { double nodes = 0, leaves = 0, tris = 0, rays = 0; };

static bool tri_hit(int q, const Ray& r, float& best, V* nrm)
{
    const float* t = &A->tri[(size_t)q * 12];
    const V v0{t[0], t[1], t[2]}, e1{t[4], t[5], t[6]}, e2{t[8], t[9], t[10]};
    const V T = r.o - v0, P = cross(r.d, e2), Q = cross(T, e1);
    const float det = dot(P, e1);
    if (det < 1e-4f) return false;
    const float inv = 1.f / det, tt = dot(Q, e2) * inv;
    if (tt < 0.f || tt > best) return false;
    const float u = dot(P, T), v = dot(Q, r.d);
    if (u < 0.f || u > det || v < 0.f || u + v > det) return false;
    best = tt;
    if (nrm) *nrm = norm(cross(e1, e2));
    return true;
}

// per-node storage order for policy 1: a permutation of the four slots, sorted along the axis of largest centroid spread
struct AxisOrder { uint8_t perm[4]; uint8_t axis; };
static std::vector<AxisOrder> g_axis;

static void child_box(const uint32_t* d, int k, float* lo, float* hi)
{
    float org[3]; memcpy(org, d, 12);
    for (int a = 0; a < 3; a++) {
        float sc; memcpy(&sc, &d[a == 0 ? 3 : 13 + a], 4);
        lo[a] = org[a] + sc * (float)((d[8 + a] >> (8 * k)) & 0xff);
        hi[a] = org[a] + sc * (float)((d[11 + a] >> (8 * k)) & 0xff);
    }
}

static void build_axis_orders()
{
    g_axis.resize((size_t)A->n_quad);
    for (int n = 0; n < A->n_quad; n++) {
        const uint32_t* d = &A->quad[(size_t)n * 16];
        float c[4][3]; bool have[4];
        float mn[3] = {1e30f, 1e30f, 1e30f}, mx[3] = {-1e30f, -1e30f, -1e30f};
        for (int k = 0; k < 4; k++) {
            have[k] = (int32_t)d[4 + k] != ~0;
            float lo[3], hi[3]; child_box(d, k, lo, hi);
            for (int a = 0; a < 3; a++) { c[k][a] = 0.5f * (lo[a] + hi[a]); if (have[k]) { mn[a] = std::min(mn[a], c[k][a]); mx[a] = std::max(mx[a], c[k][a]); } }
        }
        int ax = 0;
        for (int a = 1; a < 3; a++) if (mx[a] - mn[a] > mx[ax] - mn[ax]) ax = a;
        AxisOrder& o = g_axis[(size_t)n];
        o.axis = (uint8_t)ax;
        int idx[4] = {0, 1, 2, 3};
        std::stable_sort(idx, idx + 4, [&](int a, int b) { if (have[a] != have[b]) return have[a]; return c[a][ax] < c[b][ax]; });
        for (int k = 0; k < 4; k++) o.perm[k] = (uint8_t)idx[k];
    }
}

// returns primitive hit or -1; counts into c
static int trace(const Ray& r, int policy, Cnt& c, float& tHit, V* nrm)
{
    const V inv{1.f / r.d.x, 1.f / r.d.y, 1.f / r.d.z};
    float best = r.tmax; int prim = -1;
    int stack[128]; int sp = 0; int cur = 0;
    c.rays++;
    for (;;) {
        if (cur >= 0) {
            c.nodes++;
            const uint32_t* d = &A->quad[(size_t)cur * 16];
            float tn[4]; bool hit[4];
            for (int k = 0; k < 4; k++) {
                hit[k] = false; tn[k] = 1e30f;
                if ((int32_t)d[4 + k] == ~0) continue;
                float lo[3], hi[3]; child_box(d, k, lo, hi);
                float t0 = 0.f, t1 = best;
                const float o[3] = {r.o.x, r.o.y, r.o.z}, iv[3] = {inv.x, inv.y, inv.z};
                for (int a = 0; a < 3; a++) {
                    float a0 = (lo[a] - o[a]) * iv[a], a1 = (hi[a] - o[a]) * iv[a];
                    if (a0 > a1) std::swap(a0, a1);
                    t0 = std::max(t0, a0); t1 = std::min(t1, a1);
                }
                if (t0 <= t1 * 1.00001f + 1e-6f) { hit[k] = true; tn[k] = t0; }
            }
            int order[4]; int nh = 0;
            if (policy == 0) {
                int idx[4] = {0, 1, 2, 3};
                std::stable_sort(idx, idx + 4, [&](int a, int b) { return tn[a] < tn[b]; });
                for (int k = 0; k < 4; k++) if (hit[idx[k]]) order[nh++] = idx[k];
            } else if (policy == 1) {
                const AxisOrder& ao = g_axis[(size_t)cur];
                const float dc = ao.axis == 0 ? r.d.x : (ao.axis == 1 ? r.d.y : r.d.z);
                for (int k = 0; k < 4; k++) { const int s = ao.perm[dc >= 0.f ? k : 3 - k]; if (hit[s]) order[nh++] = s; }
            } else if (policy == 2) {
                int m = -1;
                for (int k = 0; k < 4; k++) if (hit[k] && (m < 0 || tn[k] < tn[m])) m = k;
                if (m >= 0) { order[nh++] = m; for (int k = 0; k < 4; k++) if (hit[k] && k != m) order[nh++] = k; }
            } else {
                for (int k = 0; k < 4; k++) if (hit[k]) order[nh++] = k;
            }
            if (nh == 0) { if (sp == 0) break; cur = stack[--sp]; continue; }
            for (int k = nh - 1; k >= 1; k--) stack[sp++] = (int32_t)d[4 + order[k]];
            cur = (int32_t)d[4 + order[0]];
        } else {
            c.leaves++;
            const int code = ~cur, first = code >> 3, cnt = code & 7;
            bool stop = false;
            for (int k = 0; k < cnt; k++) {
                c.tris++;
                if (tri_hit(first + k, r, best, nrm)) { int p; memcpy(&p, &A->tri[(size_t)(first + k) * 12 + 3], 4); prim = p; if (r.any) stop = true; }
            }
            if (stop || sp == 0) break;
            cur = stack[--sp];
        }
    }
    tHit = best;
    return prim;
}

int main(int argc, char** argv)
{
    const int kind = argc > 1 ? atoi(argv[1]) : 1, ll = argc > 2 ? atoi(argv[2]) : 187, npix = argc > 3 ? atoi(argv[3]) : 60000;
    const int n = pt_scene_gen(kind, ll, nullptr, 0);
    std::vector<PtPrimitive> prims((size_t)n);
    pt_scene_gen(kind, ll, prims.data(), n);
    PtFlatBVH* bvh = nullptr;
    if (pt_bvh_build_sah(prims.data(), n, &bvh)) { printf("bvh build failed\n"); return 1; }
    PtAccel acc;
    pt_build_accel(pt_bvh_nodes(bvh), pt_bvh_num_nodes(bvh), pt_bvh_tris(bvh), pt_bvh_num_tris(bvh), acc);
    A = &acc;
    build_axis_orders();
    printf("scene kind %d lat_lon %d: %d tris, %d quad nodes, depth %d\n", kind, ll, n, acc.n_quad, acc.quad_depth);

    // ray set: camera rays of the reference's camera, then two generations of bounce + shadow rays
    std::mt19937 rng(12345);
    std::uniform_real_distribution<float> U(0.f, 1.f);
    std::vector<Ray> rays;
    const V cam{0.f, 20.f, 60.f};
    const float th = std::tan(0.5f * 45.f * 3.14159265f / 180.f), aspect = 16.f / 9.f;
    std::vector<Ray> gen;
    for (int i = 0; i < npix; i++) {
        const float sx = (2.f * U(rng) - 1.f) * th * aspect, sy = (2.f * U(rng) - 1.f) * th;
        gen.push_back({cam, norm(V{sx, sy, -1.f}), 1e30f, false});
    }
    for (int g = 0; g < 3; g++) {
        std::vector<Ray> next;
        for (const Ray& r : gen) {
            rays.push_back(r);
            Cnt dummy; float t; V nrm{0, 1, 0};
            const int prim = trace(r, 0, dummy, t, &nrm);
            if (prim < 0) continue;
            const V p = r.o + r.d * t;
            if (dot(nrm, r.d) > 0.f) nrm = nrm * -1.f;
            // shadow ray to a point on the light quad (any-hit semantics approximated: stop at the first hit found)
            const V lp{-5.f + 10.f * U(rng), 39.98f, -5.f + 10.f * U(rng)};
            const V tl = lp - p; const float dl = std::sqrt(dot(tl, tl));
            if (dl > 1e-3f) rays.push_back({p + nrm * 1e-3f, tl * (1.f / dl), dl - 2e-3f, true});
            // cosine-weighted bounce
            const float u1 = U(rng), u2 = U(rng), rr = std::sqrt(u1), ph = 6.2831853f * u2;
            V tng = std::fabs(nrm.x) > 0.5f ? V{0, 1, 0} : V{1, 0, 0};
            const V bt = norm(cross(nrm, tng)); tng = cross(bt, nrm);
            const V d = norm(tng * (rr * std::cos(ph)) + bt * (rr * std::sin(ph)) + nrm * std::sqrt(1.f - u1));
            next.push_back({p + nrm * 1e-3f, d, 1e30f, false});
        }
        gen.swap(next);
    }
    printf("%zu rays (camera, 2 bounce generations, shadow)\n", rays.size());
    // ---- ray classes by the "core box" (round 3, queue ordering): the AABB of the small triangles; a ray whose segment misses it can only
    // meet the few big triangles: is it really short, and how many rays are there?
    {
        const PtTriangle* T = pt_bvh_tris(bvh);
        const int nT = pt_bvh_num_tris(bvh);
        float smn[3] = {1e30f, 1e30f, 1e30f}, smx[3] = {-1e30f, -1e30f, -1e30f};
        for (int i = 0; i < nT; i++) for (int a = 0; a < 3; a++) { const float v[3] = {T[i].V0[a], T[i].V1[a], T[i].V2[a]}; for (float x : v) { smn[a] = std::min(smn[a], x); smx[a] = std::max(smx[a], x); } }
        const float sdiag = std::sqrt((smx[0]-smn[0])*(smx[0]-smn[0]) + (smx[1]-smn[1])*(smx[1]-smn[1]) + (smx[2]-smn[2])*(smx[2]-smn[2]));
        float cmn[3] = {1e30f, 1e30f, 1e30f}, cmx[3] = {-1e30f, -1e30f, -1e30f}; int nSmall = 0;
        for (int i = 0; i < nT; i++) {
            float mn[3], mx[3]; float d2 = 0.f;
            for (int a = 0; a < 3; a++) { mn[a] = std::min(T[i].V0[a], std::min(T[i].V1[a], T[i].V2[a])); mx[a] = std::max(T[i].V0[a], std::max(T[i].V1[a], T[i].V2[a])); d2 += (mx[a]-mn[a])*(mx[a]-mn[a]); }
            if (std::sqrt(d2) * 8.f < sdiag) { nSmall++; for (int a = 0; a < 3; a++) { cmn[a] = std::min(cmn[a], mn[a]); cmx[a] = std::max(cmx[a], mx[a]); } }
        }
        printf("core box: %d of %d triangles are small; [%.1f %.1f %.1f] - [%.1f %.1f %.1f] in scene [%.1f %.1f %.1f] - [%.1f %.1f %.1f]\n", nSmall, nT, cmn[0], cmn[1], cmn[2], cmx[0], cmx[1], cmx[2], smn[0], smn[1], smn[2], smx[0], smx[1], smx[2]);
        double cnt[2][2] = {{0,0},{0,0}}, sum[2][2] = {{0,0},{0,0}}, mxs[2][2] = {{0,0},{0,0}}, ge24[2][2] = {{0,0},{0,0}};
        for (const Ray& r : rays) {
            float t0 = 0.f, t1 = r.tmax;
            const float o[3] = {r.o.x, r.o.y, r.o.z}, d[3] = {r.d.x, r.d.y, r.d.z};
            for (int a = 0; a < 3; a++) { const float iv = 1.f / d[a]; float a0 = (cmn[a] - o[a]) * iv, a1 = (cmx[a] - o[a]) * iv; if (a0 > a1) std::swap(a0, a1); t0 = std::fmax(t0, a0); t1 = std::fmin(t1, a1); }
            const int miss = !(t0 <= t1);
            Cnt c; float t; trace(r, 0, c, t, nullptr);
            const int k = r.any ? 1 : 0;
            const double st = c.nodes + c.leaves;
            cnt[k][miss]++; sum[k][miss] += st; mxs[k][miss] = std::max(mxs[k][miss], st); ge24[k][miss] += st >= 24;
        }
        for (int k = 0; k < 2; k++) for (int m = 0; m < 2; m++)
            printf("  %-7s rays %-22s: %8.0f (%.1f %%)  trips/ray mean %5.2f  max %3.0f  share with >= 24 trips %.4f\n", k ? "shadow" : "path", m ? "missing the core box" : "through the core box", cnt[k][m], 100.0 * cnt[k][m] / rays.size(), sum[k][m] / std::max(1.0, cnt[k][m]), mxs[k][m], ge24[k][m] / std::max(1.0, cnt[k][m]));
    }
    static const char* name[4] = {"exact sort", "axis order", "nearest + slot order", "slot order"};
    Cnt base;
    for (int pol = 0; pol < 4; pol++) {
        Cnt c;
        for (const Ray& r : rays) { float t; trace(r, pol, c, t, nullptr); }
        if (pol == 0) base = c;
        printf("%-22s node steps/ray %7.3f (%+6.1f %%)   leaf visits/ray %6.3f (%+6.1f %%)   tri tests/ray %6.3f\n", name[pol], c.nodes / c.rays,
               100.0 * (c.nodes / base.nodes - 1.0), c.leaves / c.rays, 100.0 * (c.leaves / base.leaves - 1.0), c.tris / c.rays);
    }
    pt_bvh_free(bvh);
    return 0;
}
