// pt_device.h — HBM data layout of an uploaded scene and the kernel argument blocks.
// Shared by the host upload code (pt_api.hip) and the kernels (pt_kernels.hip).
//
// Everything is 16-byte records read with one global_load_dwordx4 per float4:
//
//  wide node (64 B, 4 x float4) — one per interior node of the traversal tree (a binned-SAH
//      BVH over the triangles built at upload, host/accel_build.cpp; the reference's own tree
//      only supplies the leaf boxes that decide acceptance), holding BOTH children's boxes so
//      one dependent fetch decides two box tests:
//        q0 = Lmin.x Lmin.y Lmin.z Lmax.x
//        q1 = Lmax.y Lmax.z Rmin.x Rmin.y
//        q2 = Rmin.z Rmax.x Rmax.y Rmax.z
//        q3 = refL refR (int bits) | unused | unused
//      ref >= 0 : index of the child's own wide node
//      ref <  0 : leaf, ~ref = (primStart << 3) | primCount   (primCount 0 = "no child")
//  quad node (64 B, 16 dwords) — the 4-wide collapse of the same tree, walked by wf_trace.  Child boxes are
//      quantised to 8 bits per coordinate against the node's origin and per-axis power-of-two scale,
//      rounded outward (host/accel_build.cpp):
//        d0..2  origin.xyz (float)      d3, d14, d15  per-axis scale 2^ex, 2^ey, 2^ez (floats)
//        d4..7  child refs (same encoding as above; ~0 = no child, its box is inverted)
//        d8..10 lo.x lo.y lo.z          d11..13 hi.x hi.y hi.z   (byte k of each dword = child k)
//      child box = origin + scale * q
//  tri test record (48 B, 3 x float4):  (V0,prim) (E1,refLeaf) (E2,0) — all a triangle test reads;
//      prim = index in the reference's order (tie rule, shading), refLeaf = its reference leaf
//  tri pair record (128 B = one cache line, 8 x float4), one per triangle q: V0 E1 E2 of triangles q and q+1 interleaved
//      component by component (V0x[q] V0x[q+1] V0y[q] V0y[q+1] ...: floats 0..17), prim[q] prim[q+1] (18, 19), then the reference
//      leaf boxes of the two triangles inline (bMin bMax: floats 20..25 and 26..31) — wf_trace tests the two triangles of a leaf
//      with 2-wide arithmetic reading its operand pairs from consecutive registers, and the box a candidate hit needs comes from
//      the line the test has just pulled in (an L1 hit instead of a second trip to L2 / HBM)
//  reference leaf box (32 B): bMin bMax — read only when Triangle::hit accepts (exact acceptance)
//  surface record (192 B, 12 x float4), indexed by primitive in the REFERENCE's order — everything
//      shading needs about a hit triangle in ONE fetch level (no index chasing: the shade kernel
//      is bound by the depth of its dependent-load chain, not by bytes):
//        floats  0.. 8  V0 E1 E2          (u,v are recomputed with the traversal's own operations)
//        floats  9..35  N0 N1 N2 T0 T1 T2 B0 B1 B2
//        floats 36..47  mat0: emittance albedo specular opacity roughness metallic
//      (Triangle::hit copies mat0 only, CudaPrimitive.cuh:149-154)
//  light (64 B, 4 x float4): V0 V1 V2 normal area (13 f)            — srcs/pathtracer.cu:164-174
//  sphere (64 B, 4 x float4): center rad | material (12 f)
#pragma once
#include <stdint.h>

namespace ptd {

constexpr int kTile = 8;                 // tiles are 8x8 pixels = one wavefront
constexpr int kTilePixels = 64;
constexpr int kStackDepth = 32;          // per-lane traversal stack entries (LDS)
constexpr int kWavesPerBlock = 4;
constexpr int kBlockThreads = 64 * kWavesPerBlock;

// Diagnostic counter buffer (PtScene::d_counters, PTAMD_TSTAT): kStatWords 64-bit words of counters, launch timeline and histograms,
// then — for one chosen launch of wf_trace (PTAMD_TDUMP) — 8 words per wave (kStatWaves) and a per-trip log of every kStatLogEvery-th wave
constexpr int kStatWords = 8 + 2700 * 3 + 32 + 2700 + 64 + 32 + 8;
constexpr int kStatWaves = 8192, kStatLogEvery = 112, kStatLogWaves = 64, kStatLogTrips = 1024;
constexpr int kStatStripes = 64;      // MODE 2: the launch timeline is kept in 64 copies (workgroup % 64), reduced on the host — one word per launch was 7,168 atomics on one address
constexpr size_t kStatStripeOff = (size_t)kStatWords * 8 + (size_t)kStatWaves * 64 + (size_t)kStatLogWaves * kStatLogTrips * 4;      // bytes
constexpr size_t kStatBytes = kStatStripeOff + (size_t)2700 * kStatStripes * 3 * 8;

struct DevScene {
    const float4* nodes;      // traversal tree (SAH over triangles), 4 x float4 per record
    const uint4* quad;        // its 4-wide quantised collapse, 4 x uint4 per record (wf_trace)
    const float4* tri;        // triangle test records in TREE order: (V0,prim) (E1,refLeaf) (E2,0)
    const float4* tripair;    // pair records for wf_trace, 8 x float4 per triangle q: triangles q and q+1 interleaved (x0 x1 y0 y1 ...), prims, the two reference leaf boxes
    const float4* leafbox;    // the reference's leaf boxes: 2 x float4 per reference leaf
    const float4* surf;       // surface records, 12 x float4 per primitive (reference order)
    const float4* lights;
    const float4* spheres;
    int32_t n_nodes, n_tris, n_lights, n_spheres;
    int32_t nee_prune;        // 1: every emittance in the scene is finite, >= 0 and <= 1e8, so dead NEE terms need no shadow ray (pt_stream.h: bounce)
    int32_t n_quad;           // records in `quad`; the first min(n_quad, 1024) are numbered breadth-first
    int32_t quad_depth;       // levels of the 4-wide tree (a walk with a per-lane stack needs 3 * quad_depth + 2 entries)
    const float* core;        // 6 floats, lo.xyz hi.xyz: the box around the scene's SMALL triangles (nullptr: none) — a ray whose segment misses it can
                              // only meet the few big ones and is short; such rays are queued last (pt_stream.h: ray_is_short). Scheduling only.
};

struct DevCamera {
    float pos[3], forward[3], up[3], right[3];
    float tan_half_fovx, tan_half_fovy;  // tan(CameraFovX*0.5f), tan(CameraFovY*0.5f): per-launch constants of GetPixelDirection
    int32_t W, H;
};

struct DevParams {
    int32_t passes, spp_per_pass, max_bounce, rr_bounce;
    float rr_floor;
    int32_t max_refract, first_pass;
    int32_t rank, world;
    int32_t tiles_x, tiles_y, n_tiles_total, n_tiles_local;
    int32_t n_units;                     // n_tiles_local * passes (of this launch / cohort)
    int32_t unit_base;                   // first global unit of this cohort (wavefront pipeline)
};

}  // namespace ptd
