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
}
