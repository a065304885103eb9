// check_quad.cpp — host-only consistency check of the 4-wide quantised tree (host/accel_build.cpp):
// every triangle is reachable exactly once, every child box (origin + 2^e * q) contains the
// triangles below it, and the node/depth statistics are printed.
//   g++ -std=c++17 -O2 -I include tools/check_quad.cpp pathtrace-on-cuda_amd/build/{accel_build,bvh_build,scenes,pt_host,obj_loader}.o -pthread -o /tmp/check_quad
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../include/pt_api.h"
#include "../pathtrace-on-cuda_amd/host/accel_build.h"

static const PtAccel* A;
static const PtTriangle* T;
static std::vector<int> seen;
static long nodes = 0, leaves = 0, children = 0; static int maxDepth = 0; static int bad = 0;

static void triBox(int q, float* mn, float* mx)
{
    const float* r = &A->tri[(size_t)q * 12];
    int prim; memcpy(&prim, &r[3], 4);
    const PtTriangle& t = T[prim];
    for (int a = 0; a < 3; a++) { mn[a] = std::fmin(t.V0[a], std::fmin(t.V1[a], t.V2[a])); mx[a] = std::fmax(t.V0[a], std::fmax(t.V1[a], t.V2[a])); }
    seen[(size_t)prim]++;
}
// returns the bounds of everything below `ref`
static void walk(int32_t ref, int depth, float* mn, float* mx)
{
    for (int a = 0; a < 3; a++) { mn[a] = 1e30f; mx[a] = -1e30f; }
    if (ref < 0) {
        const int code = ~ref, first = code >> 3, cnt = code & 7;
        leaves++;
        for (int k = 0; k < cnt; k++) { float a[3], b[3]; triBox(first + k, a, b); for (int c = 0; c < 3; c++) { mn[c] = std::fmin(mn[c], a[c]); mx[c] = std::fmax(mx[c], b[c]); } }
        return;
    }
    nodes++; if (depth > maxDepth) maxDepth = depth;
    const uint32_t* d = &A->quad[(size_t)ref * 16];
    float org[3]; memcpy(org, d, 12);
    for (int k = 0; k < 4; k++) {
        const int32_t cr = (int32_t)d[4 + k];
        if (cr == ~0) continue;
        children++;
        float cmn[3], cmx[3];
        walk(cr, depth + 1, cmn, cmx);
        for (int a = 0; a < 3; a++) {
            float sc; memcpy(&sc, &d[a == 0 ? 3 : 13 + a], 4);      // per-axis scale 2^e, stored as a float (d3, d14, d15)
            const float lo = org[a] + sc * (float)((d[8 + a] >> (8 * k)) & 0xff), hi = org[a] + sc * (float)((d[11 + a] >> (8 * k)) & 0xff);
            if (!(lo <= cmn[a] && hi >= cmx[a])) { if (bad++ < 10) printf("BAD box node %d child %d axis %d: [%g,%g] vs [%g,%g]\n", ref, k, a, lo, hi, cmn[a], cmx[a]); }
            mn[a] = std::fmin(mn[a], cmn[a]); mx[a] = std::fmax(mx[a], cmx[a]);
        }
    }
}

int main(int argc, char** argv)
{
    const int kind = argc > 1 ? atoi(argv[1]) : 1, ll = argc > 2 ? atoi(argv[2]) : 187;
    const int n = pt_scene_gen(kind, ll, nullptr, 0);
    std::vector<PtPrimitive> prims((size_t)n);
    pt_scene_gen(kind, ll, prims.data(), n);
    PtFlatBVH* bvh = nullptr;
    if (pt_bvh_build_sah(prims.data(), n, &bvh)) { printf("build failed\n"); return 1; }
    PtAccel acc;
    T = pt_bvh_tris(bvh);
    pt_build_accel(pt_bvh_nodes(bvh), pt_bvh_num_nodes(bvh), T, pt_bvh_num_tris(bvh), acc);
    A = &acc;
    seen.assign((size_t)pt_bvh_num_tris(bvh), 0);
    float mn[3], mx[3];
    walk(0, 0, mn, mx);
    int miss = 0; for (int v : seen) if (v != 1) miss++;
    printf("tris %d  binary nodes %d (depth %d)  quad nodes %d (depth %d, walked %ld, %.2f children/node)  leaves %ld  tris not seen exactly once %d  bad boxes %d\n",
           pt_bvh_num_tris(bvh), acc.n_wide, acc.depth, acc.n_quad, maxDepth, nodes, (double)children / nodes, leaves, miss, bad);
    return (miss || bad) ? 1 : 0;
}
