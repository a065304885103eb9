// bvh_build.cpp — host acceleration-structure build + flatten (C-ABI pt_bvh_build_sah).
//
// Produces exactly what the reference's SAHBVH::GenBVHTree (srcs/bvh.cpp:426-511) followed
// by LoadFromBVH (srcs/CudaPrimitive.cu:8-145) and the Triangle::Copy loop
// (srcs/pathtracer.cu:164-166) produce — the same node array, the same primitive order,
// the same floats — without the reference's two pointer trees:
//
//   * one index array is partitioned in place; a node's primitive list is a range of it.
//     std::sort is applied to the same sequence of indices with a comparator that returns
//     the same booleans as the reference's (centroid[axis] descending; the keys are the
//     reference's `(p1+p2+p3)*0.333333f`, computed once per primitive instead of inside
//     every comparison), so even the unstable-sort tie order is the reference's
//     (same libstdc++ introsort on the same input).
//   * the split is the reference's cumulative-triangle-area cost
//     `CSA[i-1]*i + (total-CSA[i-1])*(n-i)`, first minimum wins (bvh.cpp:467-477).
//   * the sort+split recursion runs first and in parallel across subtrees (siblings are independent;
//     every std::sort call still sees exactly the sequence the sequential build would give it), the
//     emission pass that follows is serial and only reads the recorded split positions;
//   * nodes are emitted directly in the flattened order: LoadFromBVH pops Child[1] before
//     Child[0], so the array is a pre-order in which childL = Child[1] = index+1.
//     Boxes are filled in on the way back up (leaf: over the three vertices of each
//     primitive, bvh.cpp:400-406; interior: union of the children, bvh.cpp:505-506).
#include <algorithm>
#include <atomic>
#include <cmath>
#include <future>
#include <thread>
#include <unordered_map>
#include <cstring>
#include <limits>
#include <vector>

#include "../../include/pt_api.h"

void pt_set_error(const char* fmt, ...);

struct PtFlatBVH {
    std::vector<PtBVHNode> nodes;
    std::vector<PtTriangle> tris;
    int max_depth = -1;
};

namespace {

struct Builder {
    const PtPrimitive* prims;
    int n;
    std::vector<unsigned> idx;          // working index array (the reference's Cluster::primitives, concatenated)
    std::vector<float> key[3];          // centroid per axis
    std::vector<float> area;            // |cross(v2-v1, v3-v1)| (not halved, bvh.cpp:460)
    PtFlatBVH* out;
    std::unordered_map<unsigned long long, int> splitOf;   // (lo,hi) -> split, filled by sort_split
    std::vector<PtTriangle> flat;       // flattened triangle per primitive id (filled in parallel)

    static inline float min2(float x, float y) { return (y < x) ? y : x; }   // glm::min
    static inline float max2(float x, float y) { return (x < y) ? y : x; }   // glm::max

    void prepare()
    {
        idx.resize((size_t)n);
        for (int a = 0; a < 3; a++) key[a].resize((size_t)n);
        area.resize((size_t)n);
        for (int i = 0; i < n; i++) {
            idx[(size_t)i] = (unsigned)i;
            const PtVec3 &p1 = prims[i].v1.Position, &p2 = prims[i].v2.Position, &p3 = prims[i].v3.Position;
            key[0][(size_t)i] = ((p1.x + p2.x) + p3.x) * 0.333333f;      // BVH::GetCentroid, bvh.cpp:100-103
            key[1][(size_t)i] = ((p1.y + p2.y) + p3.y) * 0.333333f;
            key[2][(size_t)i] = ((p1.z + p2.z) + p3.z) * 0.333333f;
            const float ax = p2.x - p1.x, ay = p2.y - p1.y, az = p2.z - p1.z;
            const float bx = p3.x - p1.x, by = p3.y - p1.y, bz = p3.z - p1.z;
            const float cx = ay * bz - by * az, cy = az * bx - bz * ax, cz = ax * by - bx * ay;   // glm::cross
            const float tx = cx * cx, ty = cy * cy, tz = cz * cz;                                   // glm::dot
            area[(size_t)i] = std::sqrt(tx + ty + tz);                                              // glm::length
        }
    }

    typedef std::vector<std::pair<unsigned long long, int>> SplitList;
    static unsigned long long key_of(int lo, int hi) { return ((unsigned long long)(unsigned)lo << 32) | (unsigned)hi; }

    // Phase 1: sort + split of idx[lo,hi) and, recursively, of its two halves (bvh.cpp:439-494).
    // Large subtrees near the root are handed to other threads.
    void sort_split(int lo, int hi, int axis, int depth, SplitList& rec, std::vector<float>& csa)
    {
        const int cnt = hi - lo;
        if (cnt <= 4) return;                                             // stopNumber, bvh.h:128 / bvh.cpp:441
        const float* k = key[axis].data();
        std::sort(idx.begin() + lo, idx.begin() + hi, [k](const unsigned A, const unsigned B) { return k[A] > k[B]; });   // bvh.cpp:451-454
        if ((int)csa.size() < cnt) csa.resize((size_t)cnt);
        for (int i = 0; i < cnt; i++) {
            const float a = area[idx[(size_t)(lo + i)]];
            csa[(size_t)i] = (i > 0) ? csa[(size_t)(i - 1)] + a : a;
        }
        float minValue = std::numeric_limits<float>::max();
        int split = 0;
        for (int i = 1; i < cnt; i++) {
            const float fi = csa[(size_t)(i - 1)] * i + (csa[(size_t)(cnt - 1)] - csa[(size_t)(i - 1)]) * (cnt - i);
            if (fi < minValue) { minValue = fi; split = i; }
        }
        rec.emplace_back(key_of(lo, hi), split);
        const int next = (axis + 1) % 3;
        if (depth < 4 && cnt > 16384 && std::thread::hardware_concurrency() > 1) {
            SplitList other;
            auto fut = std::async(std::launch::async, [&, this]() { std::vector<float> c2; sort_split(lo + split, hi, next, depth + 1, other, c2); });
            sort_split(lo, lo + split, next, depth + 1, rec, csa);
            fut.get();
            rec.insert(rec.end(), other.begin(), other.end());
        } else {
            sort_split(lo + split, hi, next, depth + 1, rec, csa);
            sort_split(lo, lo + split, next, depth + 1, rec, csa);
        }
    }

    void flatten_all()
    {
        flat.resize((size_t)n);
        const unsigned nt = std::max(1u, std::min(8u, std::thread::hardware_concurrency()));
        std::vector<std::thread> th;
        for (unsigned t = 0; t < nt; t++)
            th.emplace_back([this, t, nt]() { for (int i = (int)t; i < n; i += (int)nt) flat[(size_t)i] = flatten_tri(prims[i]); });
        for (auto& x : th) x.join();
    }

    // Phase 2: emits the subtree over idx[lo,hi) (already ordered by phase 1) and returns its node index.
    int build(int lo, int hi, int axis, int depth)
    {
        const int me = (int)out->nodes.size();
        out->nodes.emplace_back();
        if (depth > out->max_depth) out->max_depth = depth;
        const int cnt = hi - lo;
        if (cnt <= 4) {                                                   // stopNumber, bvh.h:128 / bvh.cpp:441
            float mn[3] = {std::numeric_limits<float>::max(), std::numeric_limits<float>::max(), std::numeric_limits<float>::max()};
            float mx[3] = {std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest(), std::numeric_limits<float>::lowest()};
            const int first = (int)out->tris.size();
            for (int k = lo; k < hi; k++) {
                const PtPrimitive& p = prims[idx[(size_t)k]];
                const float* a = &p.v1.Position.x; const float* b = &p.v2.Position.x; const float* c = &p.v3.Position.x;
                for (int d = 0; d < 3; d++) {
                    mn[d] = min2(mn[d], min2(a[d], min2(b[d], c[d])));
                    mx[d] = max2(mx[d], max2(a[d], max2(b[d], c[d])));
                }
                out->tris.push_back(flat[idx[(size_t)k]]);
            }
            PtBVHNode& nd = out->nodes[(size_t)me];
            memcpy(nd.bMin, mn, 12); memcpy(nd.bMax, mx, 12);
            nd.childL = nd.childR = -1;
            nd.primStart = first; nd.primEnd = first + cnt - 1;
            return me;
        }
        const int split = splitOf.at(key_of(lo, hi));
        // Children[0] = idx[lo, lo+split), Children[1] = the rest.  The flattened array holds
        // Child[1]'s subtree first (childL), then Child[0]'s (childR).
        const int next = (axis + 1) % 3;
        const int l = build(lo + split, hi, next, depth + 1);
        const int r = build(lo, lo + split, next, depth + 1);
        PtBVHNode& nd = out->nodes[(size_t)me];
        const PtBVHNode &c0 = out->nodes[(size_t)r], &c1 = out->nodes[(size_t)l];
        for (int d = 0; d < 3; d++) {
            nd.bMin[d] = min2(c0.bMin[d], c1.bMin[d]);                    // glm::min(Child[0], Child[1]), bvh.cpp:505
            nd.bMax[d] = max2(c0.bMax[d], c1.bMax[d]);
        }
        nd.childL = l; nd.childR = r;
        nd.primStart = nd.primEnd = -1;
        return me;
    }

    // the `Triangle` LoadFromBVH builds (CudaPrimitive.cu:55-108) after Triangle::Copy (CudaPrimitive.cuh:171-215)
    static inline void nrm(const PtVec3& v, float* o)
    {
        const float len = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);     // Normalize = v / length, CudaVector.cuh:284
        o[0] = v.x / len; o[1] = v.y / len; o[2] = v.z / len;
    }
    static inline void mat(const PtMaterialOnCPU& m, PtMaterial& o)
    {
        memcpy(o.emittance, &m.emittance, 12); memcpy(o.albedo, &m.albedo, 12); memcpy(o.specular, &m.specular, 12);
        o.opacity = m.opacity; o.roughness = m.roughness; o.metallic = m.metallic;
    }
    static PtTriangle flatten_tri(const PtPrimitive& p)
    {
        PtTriangle t;
        memcpy(t.V0, &p.v1.Position, 12); memcpy(t.V1, &p.v2.Position, 12); memcpy(t.V2, &p.v3.Position, 12);
        nrm(p.v1.Tangent, t.T0); nrm(p.v2.Tangent, t.T1); nrm(p.v3.Tangent, t.T2);
        nrm(p.v1.Bitangent, t.B0); nrm(p.v2.Bitangent, t.B1); nrm(p.v3.Bitangent, t.B2);
        nrm(p.v1.Normal, t.N0); nrm(p.v2.Normal, t.N1); nrm(p.v3.Normal, t.N2);
        for (int d = 0; d < 3; d++) { t.E1[d] = t.V1[d] - t.V0[d]; t.E2[d] = t.V2[d] - t.V0[d]; }
        // cross with the reference's component forms (CudaVector.cuh:109-113)
        const float cx = t.E1[1] * t.E2[2] - t.E1[2] * t.E2[1];
        const float cy = -(t.E1[0] * t.E2[2] - t.E1[2] * t.E2[0]);
        const float cz = t.E1[0] * t.E2[1] - t.E1[1] * t.E2[0];
        const float len = std::sqrt(cx * cx + cy * cy + cz * cz);
        t.normal[0] = cx / len; t.normal[1] = cy / len; t.normal[2] = cz / len;
        t.area = len * 0.5f;
        t.u0 = p.v1.u; t.v0 = p.v1.v; t.u1 = p.v2.u; t.v1 = p.v2.v; t.u2 = p.v3.u; t.v2 = p.v3.v;
        mat(p.v1.mat, t.mat0); mat(p.v2.mat, t.mat1); mat(p.v3.mat, t.mat2);
        return t;
    }
};

}  // namespace

extern "C" {

int pt_bvh_build_sah(const PtPrimitive* prims, int32_t n_prims, PtFlatBVH** out)
{
    if (!out) { pt_set_error("pt_bvh_build_sah: out is NULL"); return PT_ERR_INVALID; }
    *out = nullptr;
    if (!prims || n_prims < 1) { pt_set_error("pt_bvh_build_sah: no primitives"); return PT_ERR_INVALID; }
    if (n_prims > (1 << 28)) { pt_set_error("pt_bvh_build_sah: too many primitives"); return PT_ERR_UNSUPPORTED; }
    PtFlatBVH* b = new PtFlatBVH();
    b->nodes.reserve((size_t)n_prims);
    b->tris.reserve((size_t)n_prims);
    Builder bl;
    bl.prims = prims; bl.n = n_prims; bl.out = b;
    bl.prepare();
    {
        Builder::SplitList rec; std::vector<float> csa;
        std::thread fl([&bl]() { bl.flatten_all(); });           // independent of the sorting
        bl.sort_split(0, n_prims, 0, 0, rec, csa);
        fl.join();
        bl.splitOf.reserve(rec.size() * 2);
        for (auto& kv : rec) bl.splitOf.emplace(kv.first, kv.second);
    }
    bl.build(0, n_prims, 0, 0);
    *out = b;
    return PT_OK;
}
void pt_bvh_free(PtFlatBVH* b) { delete b; }
int32_t pt_bvh_num_nodes(const PtFlatBVH* b) { return b ? (int32_t)b->nodes.size() : 0; }
int32_t pt_bvh_num_tris(const PtFlatBVH* b) { return b ? (int32_t)b->tris.size() : 0; }
int32_t pt_bvh_max_depth(const PtFlatBVH* b) { return b ? b->max_depth : -1; }
const PtBVHNode* pt_bvh_nodes(const PtFlatBVH* b) { return b ? b->nodes.data() : nullptr; }
const PtTriangle* pt_bvh_tris(const PtFlatBVH* b) { return b ? b->tris.data() : nullptr; }

}  // extern "C"
