// accel_build.h — device acceleration structure built at upload (see accel_build.cpp).
#pragma once
#include <cstdint>
#include <vector>
#include "../../include/pt_api.h"

constexpr int kQuadTopBfs = 1024;       // leading quad nodes numbered breadth-first (wf_trace stages a prefix of them in LDS)
constexpr int kAccelMaxDepth = 32;       // == ptd::kStackDepth; the builder never exceeds it

struct PtAccel {
    std::vector<float> wide;             // n_wide x 16 floats (two child boxes + two refs)
    std::vector<float> tri;              // n_tris x 12 floats, tree order: (V0,prim) (E1,refLeaf) (E2,0)
    std::vector<float> tripair;          // n_tris x 32 floats: record q = triangles q and q+1 (the last: q twice) interleaved for wf_trace's 2-wide test + their two reference leaf boxes (csrc/pt_device.h)
    std::vector<float> leafbox;          // n_leaves x 8 floats: the reference's leaf boxes, verbatim
    std::vector<uint32_t> quad;          // n_quad x 16 dwords: the 4-wide quantised tree (layout: csrc/pt_device.h)
    int n_wide = 0, n_leaves = 0, depth = 0;
    int n_quad = 0, quad_depth = 0;
};

void pt_build_accel(const PtBVHNode* ref_nodes, int n_ref_nodes, const PtTriangle* tris, int n_tris, PtAccel& out);
