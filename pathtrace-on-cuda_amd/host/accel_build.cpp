// accel_build.cpp — the device-side acceleration structure, built at upload time.
//
// The boundary hands over the reference's own flattened tree (CudaBVHNode[] + CudaPrims,
// built exactly as srcs/bvh.cpp does).  That tree is what defines the RESULT — a triangle is
// accepted by the reference iff its LEAF box passes intersectionAABB (CudaUtil.cuh:65-88)
// and Triangle::hit passes — but it is a poor tree to TRAVERSE: the split axis is chosen
// round-robin, so a third of the splits cut a surface patch along its thin axis and leave
// two children that overlap completely (25.8 node records + 8.4 triangle tests per ray on the
// 69,576-triangle scene, with a heavy tail).  So the kernels traverse a second tree:
//
//   * a binned-SAH BVH over the individual triangles (longest useful axis, 32 bins, leaves of
//     <= 2 triangles), emitted as the same 64-byte two-child records (pt_device.h);
//   * its boxes are padded by 2^-16 relative (+ tiny absolute) so that, with the slab test's own
//     slack, no triangle Triangle::hit can accept is ever culled;
//   * every triangle record carries its index in the reference's order (the tie rule and the
//     shading lookup use it) and the id of its reference leaf; when Triangle::hit accepts a
//     triangle, the reference's slab arithmetic is run on that reference leaf box, and the hit
//     only counts if the box passes — which reproduces the reference's acceptance exactly.
//
// 16.2 node records + 4.1 triangle tests per ray on the same rays (tools/trav_lab2.cpp).
//
// The queue-driven traversal kernel (csrc/pt_wavefront.hip) walks a 4-WIDE collapse of that tree:
// a ray's traversal is a chain of dependent fetches, and halving the depth of the tree shortens
// the chain.  A 4-wide node is still one 64-byte record because its child boxes are quantised to
// 8 bits per coordinate against the node's own origin and power-of-two scale (rounded outward, on
// top of the padding above).  That is safe for the same reason the padding is: the boxes only
// steer the search; acceptance is decided by Triangle::hit and the reference's leaf box.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/pt_api.h"
#include "accel_build.h"

namespace {

struct Item { float mn[3], mx[3]; int tri; };
struct BN { float mn[3], mx[3]; int l, r, first, count; };

struct Builder {
    std::vector<Item> items;
    std::vector<BN> nodes;
    int maxLeaf = 2;
    int maxDepth = 0;

    static float area(const float* mn, const float* mx)
    {
        const float d[3] = {mx[0] - mn[0], mx[1] - mn[1], mx[2] - mn[2]};
        return 2.f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
    static int ceil_log2(int n) { int k = 0; while ((1 << k) < n) k++; return k; }

    int build(int lo, int hi, int depth)
    {
        const int me = (int)nodes.size();
        nodes.emplace_back();
        if (depth > maxDepth) maxDepth = depth;
        float mn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, mx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        float cmn[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, cmx[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        for (int i = lo; i < hi; i++)
            for (int a = 0; a < 3; a++) {
                mn[a] = std::min(mn[a], items[i].mn[a]); mx[a] = std::max(mx[a], items[i].mx[a]);
                const float c = 0.5f * (items[i].mn[a] + items[i].mx[a]);
                cmn[a] = std::min(cmn[a], c); cmx[a] = std::max(cmx[a], c);
            }
        memcpy(nodes[me].mn, mn, 12); memcpy(nodes[me].mx, mx, 12);
        const int n = hi - lo;
        if (n <= maxLeaf) { nodes[me].first = lo; nodes[me].count = n; nodes[me].l = nodes[me].r = -1; return me; }
        int mid = -1;
        // keep the tree within the traversal stack: when the remaining depth budget only just
        // covers a balanced subtree, split at the median
        const bool forceMedian = depth + ceil_log2((n + maxLeaf - 1) / maxLeaf) >= kAccelMaxDepth - 1;
        if (!forceMedian) {
            const int NB = 32;
            float best = FLT_MAX; int bax = -1, bsp = -1;
            for (int ax = 0; ax < 3; ax++) {
                const float ext = cmx[ax] - cmn[ax];
                if (!(ext > 0.f)) continue;
                int cnt[NB] = {0}; float bmn[NB][3], bmx[NB][3];
                for (int k = 0; k < NB; k++) for (int a = 0; a < 3; a++) { bmn[k][a] = FLT_MAX; bmx[k][a] = -FLT_MAX; }
                for (int i = lo; i < hi; i++) {
                    const float c = 0.5f * (items[i].mn[ax] + items[i].mx[ax]);
                    const int k = std::min(NB - 1, (int)((c - cmn[ax]) / ext * NB));
                    cnt[k]++;
                    for (int a = 0; a < 3; a++) { bmn[k][a] = std::min(bmn[k][a], items[i].mn[a]); bmx[k][a] = std::max(bmx[k][a], items[i].mx[a]); }
                }
                float ra[NB]; int rc[NB];
                float am[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, aM[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}; int c = 0;
                for (int k = NB - 1; k > 0; k--) {
                    c += cnt[k];
                    for (int a = 0; a < 3; a++) { am[a] = std::min(am[a], bmn[k][a]); aM[a] = std::max(aM[a], bmx[k][a]); }
                    ra[k] = c ? area(am, aM) : 0.f; rc[k] = c;
                }
                float lm[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, lM[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX}; c = 0;
                for (int k = 0; k < NB - 1; k++) {
                    c += cnt[k];
                    for (int a = 0; a < 3; a++) { lm[a] = std::min(lm[a], bmn[k][a]); lM[a] = std::max(lM[a], bmx[k][a]); }
                    if (!c || !rc[k + 1]) continue;
                    const float cost = area(lm, lM) * c + ra[k + 1] * rc[k + 1];
                    if (cost < best) { best = cost; bax = ax; bsp = k; }
                }
            }
            if (bax >= 0) {
                const float ext = cmx[bax] - cmn[bax], c0 = cmn[bax];
                auto it = std::partition(items.begin() + lo, items.begin() + hi, [&](const Item& x) {
                    const float c = 0.5f * (x.mn[bax] + x.mx[bax]);
                    return std::min(NB - 1, (int)((c - c0) / ext * NB)) <= bsp;
                });
                mid = (int)(it - items.begin());
                if (mid == lo || mid == hi) mid = -1;
            }
        }
        if (mid < 0) {
            // median along the widest centroid axis (also the fallback when all centroids coincide)
            int ax = 0;
            if (cmx[1] - cmn[1] > cmx[ax] - cmn[ax]) ax = 1;
            if (cmx[2] - cmn[2] > cmx[ax] - cmn[ax]) ax = 2;
            mid = (lo + hi) / 2;
            std::nth_element(items.begin() + lo, items.begin() + mid, items.begin() + hi, [ax](const Item& a, const Item& b) {
                return (a.mn[ax] + a.mx[ax]) < (b.mn[ax] + b.mx[ax]);
            });
        }
        const int l = build(lo, mid, depth + 1);
        const int r = build(mid, hi, depth + 1);
        nodes[me].l = l; nodes[me].r = r; nodes[me].count = 0;
        return me;
    }
};

inline float as_float(int32_t i) { float f; memcpy(&f, &i, 4); return f; }
inline float pad_lo(float v) { return v - (std::fabs(v) * 1.52587890625e-5f + 1e-30f); }
inline float pad_hi(float v) { return v + (std::fabs(v) * 1.52587890625e-5f + 1e-30f); }

}  // namespace

void pt_build_accel(const PtBVHNode* rnodes, int n_rnodes, const PtTriangle* tris, int n_tris, PtAccel& out)
{
    // reference leaf of every triangle + the leaf boxes (exact copies)
    std::vector<int> leafOf((size_t)n_tris, -1);
    std::vector<int> leafIndex((size_t)n_rnodes, -1);
    out.leafbox.clear();
    int n_leaves = 0;
    for (int i = 0; i < n_rnodes; i++) {
        const PtBVHNode& n = rnodes[i];
        if (n.primStart == -1 || n.primEnd == -1) continue;
        leafIndex[(size_t)i] = n_leaves;
        for (int k = n.primStart; k <= n.primEnd; k++) leafOf[(size_t)k] = n_leaves;
        const float rec[8] = {n.bMin[0], n.bMin[1], n.bMin[2], n.bMax[0], n.bMax[1], n.bMax[2], 0.f, 0.f};
        out.leafbox.insert(out.leafbox.end(), rec, rec + 8);
        n_leaves++;
    }

    Builder b;
    if (const char* e = getenv("PTAMD_LEAF")) { const int v = atoi(e); if (v >= 1 && v <= 7) b.maxLeaf = v; }   // tuning only
    b.items.resize((size_t)n_tris);
    for (int i = 0; i < n_tris; i++) {
        const PtTriangle& t = tris[i];
        Item& it = b.items[(size_t)i];
        for (int a = 0; a < 3; a++) {
            it.mn[a] = std::min(t.V0[a], std::min(t.V1[a], t.V2[a]));
            it.mx[a] = std::max(t.V0[a], std::max(t.V1[a], t.V2[a]));
        }
        it.tri = i;
    }
    b.nodes.reserve((size_t)n_tris * 2);
    b.build(0, n_tris, 0);
    out.depth = b.maxDepth;

    // triangle test records in tree order
    out.tri.resize((size_t)n_tris * 12);
    for (int q = 0; q < n_tris; q++) {
        const int i = b.items[(size_t)q].tri;
        const PtTriangle& t = tris[i];
        float* a = &out.tri[(size_t)q * 12];
        a[0] = t.V0[0]; a[1] = t.V0[1]; a[2] = t.V0[2]; a[3] = as_float(i);
        a[4] = t.E1[0]; a[5] = t.E1[1]; a[6] = t.E1[2]; a[7] = as_float(leafOf[(size_t)i]);
        a[8] = t.E2[0]; a[9] = t.E2[1]; a[10] = t.E2[2]; a[11] = 0.f;
    }

    // pair records (128 B = one cache line each): record q interleaves the test records of triangles q and q+1 component by
    // component, so that wf_trace's 2-wide arithmetic reads its operand pairs straight from consecutive registers, and carries
    // the two reference leaf boxes inline — the box a candidate hit needs is then in the line the test has just pulled in
    out.tripair.resize((size_t)n_tris * 32);
    for (int q = 0; q < n_tris; q++) {
        const float* a = &out.tri[(size_t)q * 12];
        const float* c = &out.tri[(size_t)(q + 1 < n_tris ? q + 1 : q) * 12];
        float* r = &out.tripair[(size_t)q * 32];
        static const int src[9] = {0, 1, 2, 4, 5, 6, 8, 9, 10};      // V0.xyz E1.xyz E2.xyz
        for (int k = 0; k < 9; k++) { r[2 * k] = a[src[k]]; r[2 * k + 1] = c[src[k]]; }
        r[18] = a[3]; r[19] = c[3];      // prim
        int la, lc; memcpy(&la, &a[7], 4); memcpy(&lc, &c[7], 4);
        memcpy(&r[20], &out.leafbox[(size_t)la * 8], 24);      // bMin bMax of triangle q's reference leaf
        memcpy(&r[26], &out.leafbox[(size_t)lc * 8], 24);      // ... of triangle q+1's
    }

    // wide records: one per interior node, in depth-first order
    std::vector<int> widx(b.nodes.size(), -1);
    int n_wide = 0;
    {
        std::vector<int> st; st.push_back(0);
        while (!st.empty()) {
            const int i = st.back(); st.pop_back();
            const BN& n = b.nodes[(size_t)i];
            if (n.count > 0) continue;
            widx[(size_t)i] = n_wide++;
            st.push_back(n.r); st.push_back(n.l);
        }
    }
    auto ref_of = [&](int c) -> int32_t {
        const BN& n = b.nodes[(size_t)c];
        if (n.count > 0) return ~((n.first << 3) | n.count);
        return widx[(size_t)c];
    };
    auto put_box = [&](float* r, const BN& n) {
        r[0] = pad_lo(n.mn[0]); r[1] = pad_lo(n.mn[1]); r[2] = pad_lo(n.mn[2]);
        r[3] = pad_hi(n.mx[0]); r[4] = pad_hi(n.mx[1]); r[5] = pad_hi(n.mx[2]);
    };
    if (n_wide == 0) {
        // a single leaf: one record whose L side is that leaf and whose R side is "no child"
        out.wide.assign(16, 0.f);
        put_box(&out.wide[0], b.nodes[0]); put_box(&out.wide[6], b.nodes[0]);
        out.wide[12] = as_float(ref_of(0)); out.wide[13] = as_float(~0);
        n_wide = 1;
    } else {
        out.wide.assign((size_t)n_wide * 16, 0.f);
        for (size_t i = 0; i < b.nodes.size(); i++) {
            if (widx[i] < 0) continue;
            const BN& n = b.nodes[i];
            float* r = &out.wide[(size_t)widx[i] * 16];
            put_box(r, b.nodes[(size_t)n.l]); put_box(r + 6, b.nodes[(size_t)n.r]);
            r[12] = as_float(ref_of(n.l)); r[13] = as_float(ref_of(n.r));
        }
    }
    out.n_wide = n_wide;
    out.n_leaves = n_leaves;

    // ---- 4-wide quantised collapse ----
    // absolute pad on top of the relative one: Moeller-Trumbore's own rounding is relative to the ray's
    // coordinates, not to a box face that happens to lie near a coordinate plane
    float maxAbs = 0.f;
    for (const BN& n : b.nodes) for (int a = 0; a < 3; a++) maxAbs = std::max(maxAbs, std::max(std::fabs(n.mn[a]), std::fabs(n.mx[a])));
    const float absPad = maxAbs * 9.5367431640625e-7f;      // 2^-20
    struct Q4 { uint32_t d[16]; };
    std::vector<Q4> quad;
    int quadDepth = 0;
    struct Emit {
        Builder& b; std::vector<Q4>& quad; float absPad; int& maxDepth;
        int run(int bnode, int depth)
        {
            if (depth > maxDepth) maxDepth = depth;
            int ch[4]; int nc = 0;
            ch[nc++] = b.nodes[(size_t)bnode].l; ch[nc++] = b.nodes[(size_t)bnode].r;
            while (nc < 4) {
                int pick = -1; float pa = -1.f;
                for (int k = 0; k < nc; k++) {
                    const BN& c = b.nodes[(size_t)ch[k]];
                    if (c.count > 0) continue;
                    const float ar = Builder::area(c.mn, c.mx);
                    if (ar > pa) { pa = ar; pick = k; }
                }
                if (pick < 0) break;
                const BN& c = b.nodes[(size_t)ch[pick]];
                const int l = c.l, r = c.r;
                ch[pick] = l; ch[nc++] = r;
            }
            const int me = (int)quad.size();
            quad.emplace_back();
            float lo[4][3], hi[4][3];
            for (int k = 0; k < nc; k++) {
                const BN& c = b.nodes[(size_t)ch[k]];
                for (int a = 0; a < 3; a++) { lo[k][a] = pad_lo(c.mn[a]) - absPad; hi[k][a] = pad_hi(c.mx[a]) + absPad; }
            }
            float org[3]; int e[3];
            uint32_t qlo[3] = {0, 0, 0}, qhi[3] = {0, 0, 0};
            for (int a = 0; a < 3; a++) {
                float mn = lo[0][a], mx = hi[0][a];
                for (int k = 1; k < nc; k++) { mn = std::min(mn, lo[k][a]); mx = std::max(mx, hi[k][a]); }
                org[a] = mn;
                int ex = -100;
                if (mx > mn) { int t; std::frexp((mx - mn) / 255.f, &t); ex = std::max(t - 1, -100); }
                for (;;) {
                    const float sc = std::ldexp(1.f, ex);
                    bool ok = true;
                    uint32_t wl = 0, wh = 0;
                    for (int k = 0; k < 4 && ok; k++) {
                        int ql, qh;
                        if (k < nc) {
                            ql = (int)std::floor((lo[k][a] - mn) / sc);
                            if (ql < 0) ql = 0;
                            while (ql > 0 && mn + sc * (float)ql > lo[k][a]) ql--;
                            qh = (int)std::ceil((hi[k][a] - mn) / sc);
                            if (qh < ql) qh = ql;
                            while (qh <= 255 && mn + sc * (float)qh < hi[k][a]) qh++;
                            if (qh > 255 || ql > 255) { ok = false; break; }
                        } else { ql = 255; qh = 0; }          // no child: an inverted box
                        wl |= (uint32_t)ql << (8 * k); wh |= (uint32_t)qh << (8 * k);
                    }
                    if (ok) { qlo[a] = wl; qhi[a] = wh; e[a] = ex; break; }
                    ex++;
                }
            }
            uint32_t refs[4];
            for (int k = 0; k < 4; k++) {
                if (k >= nc) { refs[k] = (uint32_t)~0; continue; }
                const BN& c = b.nodes[(size_t)ch[k]];
                if (c.count > 0) refs[k] = (uint32_t)~((c.first << 3) | c.count);
                else refs[k] = (uint32_t)run(ch[k], depth + 1);
            }
            Q4& q = quad[(size_t)me];
            memcpy(&q.d[0], org, 12);
            { const float sx = std::ldexp(1.f, e[0]), sy = std::ldexp(1.f, e[1]), sz = std::ldexp(1.f, e[2]);      // the scales as floats: one multiply per axis in the kernel
              memcpy(&q.d[3], &sx, 4); memcpy(&q.d[14], &sy, 4); memcpy(&q.d[15], &sz, 4); }
            for (int k = 0; k < 4; k++) q.d[4 + k] = refs[k];
            q.d[8] = qlo[0]; q.d[9] = qlo[1]; q.d[10] = qlo[2];
            q.d[11] = qhi[0]; q.d[12] = qhi[1]; q.d[13] = qhi[2];
            return me;
        }
    };
    if (b.nodes[0].count > 0) {
        // a single leaf: one node whose first child is that leaf
        quad.emplace_back();
        Q4& q = quad[0];
        memset(&q, 0, sizeof(q));
        float org[3]; uint32_t qh[3];
        int e[3];
        for (int a = 0; a < 3; a++) {
            const float mn = pad_lo(b.nodes[0].mn[a]) - absPad, mx = pad_hi(b.nodes[0].mx[a]) + absPad;
            org[a] = mn;
            int ex = -100;
            if (mx > mn) { int t; std::frexp((mx - mn) / 255.f, &t); ex = std::max(t - 1, -100); }
            for (;;) { const float sc = std::ldexp(1.f, ex); int q1 = (int)std::ceil((mx - mn) / sc); while (q1 <= 255 && mn + sc * (float)q1 < mx) q1++; if (q1 <= 255) { qh[a] = (uint32_t)q1; e[a] = ex; break; } ex++; }
        }
        memcpy(&q.d[0], org, 12);
        { const float sx = std::ldexp(1.f, e[0]), sy = std::ldexp(1.f, e[1]), sz = std::ldexp(1.f, e[2]);
          memcpy(&q.d[3], &sx, 4); memcpy(&q.d[14], &sy, 4); memcpy(&q.d[15], &sz, 4); }
        q.d[4] = (uint32_t)~((b.nodes[0].first << 3) | b.nodes[0].count);
        q.d[5] = q.d[6] = q.d[7] = (uint32_t)~0;
        for (int a = 0; a < 3; a++) { q.d[8 + a] = 0u | (255u << 8) | (255u << 16) | (255u << 24); q.d[11 + a] = qh[a]; }
    } else {
        Emit em{b, quad, absPad, quadDepth};
        em.run(0, 0);
    }
    // Renumber: the first kQuadTopBfs nodes in breadth-first order — every ray walks them, and wf_trace keeps a prefix of
    // them in LDS (any node index below its prefix length is an LDS read) — the rest keep their depth-first order
    // (a subtree stays contiguous in memory).
    {
        const size_t n = quad.size();
        static const int kBfs = getenv("PTAMD_BFS") ? atoi(getenv("PTAMD_BFS")) : kQuadTopBfs;      // tuning / A-B only
        std::vector<int> order; order.reserve(n);
        std::vector<char> placed(n, 0);
        order.push_back(0); placed[0] = 1;
        for (size_t head = 0; head < order.size() && order.size() < (size_t)kBfs; head++) {
            const Q4& q = quad[(size_t)order[head]];
            for (int k = 0; k < 4 && order.size() < (size_t)kBfs; k++) {
                const int r = (int)q.d[4 + k];
                if (r >= 0 && !placed[(size_t)r]) { placed[(size_t)r] = 1; order.push_back(r); }
            }
        }
        for (size_t i = 0; i < n; i++) if (!placed[i]) order.push_back((int)i);
        std::vector<int> newIdx(n);
        for (size_t i = 0; i < n; i++) newIdx[(size_t)order[i]] = (int)i;
        std::vector<Q4> re(n);
        for (size_t i = 0; i < n; i++) {
            Q4 q = quad[(size_t)order[i]];
            for (int k = 0; k < 4; k++) if ((int)q.d[4 + k] >= 0) q.d[4 + k] = (uint32_t)newIdx[(size_t)q.d[4 + k]];
            re[i] = q;
        }
        quad.swap(re);
    }
    out.quad.resize(quad.size() * 16);
    memcpy(out.quad.data(), quad.data(), quad.size() * sizeof(Q4));
    out.n_quad = (int)quad.size();
    out.quad_depth = quadDepth;
}
