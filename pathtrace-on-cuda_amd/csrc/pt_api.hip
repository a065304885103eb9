// pt_api.hip — device half of the C-ABI (include/pt_api.h): scene upload into the HBM
// layout of pt_device.h, render launch sequence, tile gather helpers, parity hooks.
//
// Replaces the body of PathTracer::Render (srcs/pathtracer.cu:124-259): instead of five
// cudaMallocManaged regions filled element by element from the host and a device vtable
// plant, the scene is repacked once on the host into 16-byte records and copied with one
// hipMemcpy per array; instead of NUM_MULTI_SAMPLE synchronous launches there is one
// persistent launch over all (tile, pass) units on the caller's stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <map>
#include <string>
#include <vector>

#include "../../include/pt_api.h"
#include "pt_device.h"
#include "../host/accel_build.h"

extern "C" {
hipError_t ptk_render_units(const ptd::DevScene*, const ptd::DevCamera*, const ptd::DevParams*, float*, unsigned int*, void*, int, int, hipStream_t);
hipError_t ptk_sum_passes(const float*, int, long long, float*, hipStream_t);
hipError_t ptk_untile(const float*, int, int, int, int, int, long long, float*, hipStream_t);
hipError_t ptk_dbg_raycast(const ptd::DevScene*, const float*, int, float*, int*, hipStream_t);
hipError_t ptk_dbg_bxdf(int, const float*, int, float*, hipStream_t);
hipError_t ptk_dbg_rng(unsigned long long, int, uint32_t*, float*, hipStream_t);
hipError_t ptk_dbg_math(const float*, int, float*, hipStream_t);
hipError_t ptk_dbg_sincos(const float*, int, float*, hipStream_t);
hipError_t ptk_dbg_ray_setup(const float*, int, float*, hipStream_t);
hipError_t ptk_dbg_pixel_dir(const ptd::DevCamera*, const int*, int, float*, hipStream_t);
hipError_t ptk_dbg_nee(const ptd::DevScene*, const float*, int, float*, hipStream_t);
size_t ptk_wf_work_bytes(size_t nUnits, int traceBlocks);
int ptk_wf_cohorts(size_t nUnits);
const float* ptk_wf_staging(void* work);
int ptk_wf_stack_capacity(void);
hipError_t ptk_wf_render(int, const ptd::DevScene*, const ptd::DevCamera*, const ptd::DevParams*, void*, int, uint32_t*, hipStream_t, hipStream_t*,
                         hipEvent_t, hipEvent_t, hipEvent_t, hipEvent_t*, int*, hipEvent_t*, int, int*, int, int, void*, int);
}

void pt_set_error(const char* fmt, ...);   // pt_host.cpp

#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            pt_set_error("HIP error %d at %s:%d '%s': %s", (int)e_, __FILE__, __LINE__, #expr, hipGetErrorString(e_)); \
            return PT_ERR_DEVICE;                                                           \
        }                                                                                   \
    } while (0)

static const size_t kCounterBytes = ptd::kStatBytes;   // 8 work counters + diagnostic launch timeline (3 x u64 per wf_trace launch) + wave-lifetime histogram + rays per launch

struct PtScene {
    int device = 0;
    ptd::DevScene dev{};
    void* d_nodes = nullptr; void* d_quad = nullptr; void* d_tri = nullptr; void* d_tripair = nullptr; void* d_leafbox = nullptr;
    void* d_surf = nullptr;
    void* d_lights = nullptr; void* d_spheres = nullptr; void* d_core = nullptr;
    unsigned int* d_unit_counter = nullptr;
    void* d_counters = nullptr;
    int64_t bytes = 0;
    int n_lights = 0;
    int max_depth = 0;
    int num_cus = 256;
    bool count_next = false;
    int mode = 1;            // 1 = wavefront pipeline (default), 0 = one-kernel state machine
    uint32_t* h_poll = nullptr;   // pinned, for the pipeline's live-stream count
    int last_iters = 0;
    int shade_rounds = 1;        // wf_shade: 1 = a stream may start its next sample in the step its path ends, 0 = one bounce per step, -1 = by live-stream count (PTAMD_TRS)
    int early_below = 2500000;   // renders of at most this many streams (pixels x passes of one call) run wf_shade's early phase beside the draining wf_trace (0 = never; pt_set_early_shade)
    int drain_below = 80000;     // hand the last streams of a render to wf_drain once this few are live (0 = never; PTAMD_DRAIN, pt_set_drain_threshold):
                                 // the last ~200 of ~1,100 bounce iterations serve < 5 % of the streams at the latency of the longest ray each
                                 // (~200 us); wf_drain runs those streams to their end in one launch, spread over every SIMD.  40,000-120,000 is flat:
                                 // +5...7 % for an 8-way rank, +3 % 4-way, +1 % on one GPU (r03_b31.log, r03_b32.log, r03_b33.log)
    // optional per-launch timing of the traversal kernel (pt_enable_trace_timing)
    std::vector<hipEvent_t> trace_ev;
    int trace_ev_used[4] = {0, 0, 0, 0};     // per cohort
    int trace_ev_per = 0;                    // event pairs per cohort in the last render
    hipStream_t xstreams[3] = {nullptr, nullptr, nullptr};   // extra streams for concurrent cohorts
    hipEvent_t ev_fork = nullptr, ev_join[3] = {nullptr, nullptr, nullptr};
    // ring of HIP event pairs, one pair per render_units launch (pt_render_timings)
    static constexpr int kEvRing = 64;
    hipEvent_t ev[kEvRing][2] = {};
    int ev_count = 0;        // launches recorded since the last pt_render_timings(reset)
};

static int upload(void** dptr, const void* h, size_t bytes, int64_t& total)
{
    size_t alloc = bytes ? bytes : 16;
    HIPCHK(hipMalloc(dptr, alloc));
    if (bytes) HIPCHK(hipMemcpy(*dptr, h, bytes, hipMemcpyHostToDevice));
    total += (int64_t)alloc;
    return PT_OK;
}

static inline float as_float(int32_t i) { float f; memcpy(&f, &i, 4); return f; }

template <class F>
static int with_buffers(int device, const void* in, size_t in_bytes, void* out, size_t out_bytes, void* out2, size_t out2_bytes, F launch)
{
    HIPCHK(hipSetDevice(device));
    void *d_in = nullptr, *d_out = nullptr, *d_out2 = nullptr;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc(&d_in, in_bytes ? in_bytes : 16));
        HIPCHK(hipMalloc(&d_out, out_bytes ? out_bytes : 16));
        HIPCHK(hipMalloc(&d_out2, out2_bytes ? out2_bytes : 16));
        if (in_bytes) HIPCHK(hipMemcpy(d_in, in, in_bytes, hipMemcpyHostToDevice));
        HIPCHK(launch(d_in, d_out, d_out2));
        HIPCHK(hipDeviceSynchronize());
        if (out_bytes) HIPCHK(hipMemcpy(out, d_out, out_bytes, hipMemcpyDeviceToHost));
        if (out2_bytes) HIPCHK(hipMemcpy(out2, d_out2, out2_bytes, hipMemcpyDeviceToHost));
        return PT_OK;
    };
    const int rc = body();
    (void)hipFree(d_in); (void)hipFree(d_out); (void)hipFree(d_out2);      // on every path (hipFree(nullptr) is a no-op)
    return rc;
}


extern "C" {

int pt_scene_create(const PtBVHNode* nodes, int32_t n_nodes, const PtTriangle* tris, int32_t n_tris,
                    const PtSphere* spheres, int32_t n_spheres, int32_t device, PtScene** out)
{
    if (!out) { pt_set_error("pt_scene_create: out is NULL"); return PT_ERR_INVALID; }
    *out = nullptr;
    if (!nodes || n_nodes < 1 || !tris || n_tris < 1 || n_spheres < 0 || (n_spheres > 0 && !spheres)) {
        pt_set_error("pt_scene_create: empty or NULL scene arrays (n_nodes=%d n_tris=%d n_spheres=%d)", n_nodes, n_tris, n_spheres);
        return PT_ERR_INVALID;
    }
    // ---- validate the flattened tree and measure its depth (host check before any kernel sees it) ----
    std::vector<int> depth((size_t)n_nodes, -1);
    std::vector<int> widx((size_t)n_nodes, -1);
    int n_wide = 0, max_depth = 0;
    {
        std::vector<int> st; st.push_back(0); depth[0] = 0;
        std::vector<char> seen((size_t)n_nodes, 0);
        while (!st.empty()) {
            int i = st.back(); st.pop_back();
            if (seen[(size_t)i]) { pt_set_error("pt_scene_create: node %d reachable twice", i); return PT_ERR_INVALID; }
            seen[(size_t)i] = 1;
            const PtBVHNode& n = nodes[i];
            if (depth[i] > max_depth) max_depth = depth[i];
            const bool leaf = (n.primStart != -1 && n.primEnd != -1);
            if (leaf) {
                if (n.primStart < 0 || n.primEnd < n.primStart || n.primEnd >= n_tris || n.primEnd - n.primStart + 1 > 7) {
                    pt_set_error("pt_scene_create: leaf %d has bad primitive range [%d,%d]", i, n.primStart, n.primEnd);
                    return PT_ERR_INVALID;
                }
                if (n.childL > 0 || n.childR > 0) { pt_set_error("pt_scene_create: leaf %d has children", i); return PT_ERR_INVALID; }
            } else {
                if (n.childL <= 0 || n.childR <= 0 || n.childL >= n_nodes || n.childR >= n_nodes) {
                    pt_set_error("pt_scene_create: interior node %d has bad children (%d,%d)", i, n.childL, n.childR);
                    return PT_ERR_INVALID;
                }
                widx[(size_t)i] = 0;    // numbered below, in index order (= the reference's pre-order)
                depth[n.childL] = depth[i] + 1; depth[n.childR] = depth[i] + 1;
                st.push_back(n.childR); st.push_back(n.childL);
            }
        }
    }
    // every triangle must belong to exactly one reference leaf (its box decides acceptance)
    {
        std::vector<char> covered((size_t)n_tris, 0);
        for (int i = 0; i < n_nodes; i++) {
            const PtBVHNode& n = nodes[i];
            if (widx[(size_t)i] == -1 && depth[(size_t)i] >= 0 && n.primStart != -1 && n.primEnd != -1)
                for (int k = n.primStart; k <= n.primEnd; k++) covered[(size_t)k]++;
        }
        for (int k = 0; k < n_tris; k++)
            if (covered[(size_t)k] != 1) { pt_set_error("pt_scene_create: triangle %d is in %d reference leaves", k, (int)covered[(size_t)k]); return PT_ERR_INVALID; }
    }
    // ---- traversal tree over the triangles (host/accel_build.cpp) ----
    PtAccel accel;
    pt_build_accel(nodes, n_nodes, tris, n_tris, accel);
    if (accel.depth > ptd::kStackDepth) {
        pt_set_error("pt_scene_create: traversal tree depth %d exceeds the traversal stack (%d)", accel.depth, ptd::kStackDepth);
        return PT_ERR_UNSUPPORTED;
    }
    if (3 * accel.quad_depth + 2 > ptk_wf_stack_capacity()) {
        pt_set_error("pt_scene_create: 4-wide traversal tree depth %d needs more than the %d stack entries of the traversal kernel",
                     accel.quad_depth, ptk_wf_stack_capacity());
        return PT_ERR_UNSUPPORTED;
    }
    max_depth = accel.depth;
    n_wide = accel.n_wide;

    // ---- triangles: surface records (reference order), lights ----
    std::vector<float> surf((size_t)n_tris * 48), lights;
    int n_lights = 0;
    for (int i = 0; i < n_tris; i++) {
        const PtTriangle& t = tris[i];
        float* a = &surf[(size_t)i * 48];
        const float* src[12] = {t.V0, t.E1, t.E2, t.N0, t.N1, t.N2, t.T0, t.T1, t.T2, t.B0, t.B1, t.B2};
        for (int k = 0; k < 12; k++) { a[3 * k] = src[k][0]; a[3 * k + 1] = src[k][1]; a[3 * k + 2] = src[k][2]; }
        const PtMaterial& m = t.mat0;           // Triangle::hit copies mat0 only (CudaPrimitive.cuh:149-154)
        const float rec[12] = {m.emittance[0], m.emittance[1], m.emittance[2], m.albedo[0], m.albedo[1], m.albedo[2],
                               m.specular[0], m.specular[1], m.specular[2], m.opacity, m.roughness, m.metallic};
        memcpy(a + 36, rec, sizeof(rec));
        auto len = [](const float* e) { return std::sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]); };
        if (len(t.mat0.emittance) > 0.0001f || len(t.mat1.emittance) > 0.0001f || len(t.mat2.emittance) > 0.0001f) {
            const float rec[16] = {t.V0[0], t.V0[1], t.V0[2], t.V1[0], t.V1[1], t.V1[2], t.V2[0], t.V2[1], t.V2[2],
                                   t.normal[0], t.normal[1], t.normal[2], t.area, 0.f, 0.f, 0.f};
            lights.insert(lights.end(), rec, rec + 16);
            n_lights++;
        }
    }
    // dead-NEE-term pruning (pt_stream.h: bounce) needs every emittance a shadow ray can return to be finite, non-negative and small
    // enough that (weight * brdfcos) * Le cannot overflow where the pruned case assumes it is finite: |wb| < 1e30 and Le <= 1e8 give
    // |wb * Le| < 1e38 < FLT_MAX.  (With a brighter light wb * Le can be inf, inf * 0 is NaN, and the reference adds that NaN to the
    // radiance, include/CudaUtil.cuh:271-272; such scenes keep all their shadow rays.)
    bool emitOk = true;
    auto okE = [](const float* e) { return std::isfinite(e[0]) && std::isfinite(e[1]) && std::isfinite(e[2]) && e[0] >= 0.f && e[1] >= 0.f && e[2] >= 0.f &&
                                           e[0] <= 1e8f && e[1] <= 1e8f && e[2] <= 1e8f; };
    for (int i = 0; i < n_tris; i++) emitOk = emitOk && okE(tris[i].mat0.emittance);
    for (int i = 0; i < n_spheres; i++) emitOk = emitOk && okE(spheres[i].mat.emittance);
    // ---- core box: the AABB of the scene's SMALL triangles (bounding-box diagonal under an eighth of the scene's).  A ray whose
    // segment misses it can only meet the few big triangles, i.e. is short, and wf_shade queues such rays last (pt_stream.h:
    // ray_is_short) so that the traversal kernel's launch tail consists of short rays.  Scheduling only — any box gives the same frame.
    std::vector<float> core;
    {
        float smn[3] = {1e30f, 1e30f, 1e30f}, smx[3] = {-1e30f, -1e30f, -1e30f};
        auto tribox = [&](const PtTriangle& t, float* mn, float* mx) {
            for (int k = 0; k < 3; k++) { mn[k] = std::fmin(t.V0[k], std::fmin(t.V1[k], t.V2[k])); mx[k] = std::fmax(t.V0[k], std::fmax(t.V1[k], t.V2[k])); }
        };
        for (int i = 0; i < n_tris; i++) { float mn[3], mx[3]; tribox(tris[i], mn, mx); for (int k = 0; k < 3; k++) { smn[k] = std::fmin(smn[k], mn[k]); smx[k] = std::fmax(smx[k], mx[k]); } }
        const float sd = std::sqrt((smx[0] - smn[0]) * (smx[0] - smn[0]) + (smx[1] - smn[1]) * (smx[1] - smn[1]) + (smx[2] - smn[2]) * (smx[2] - smn[2]));
        float cmn[3] = {1e30f, 1e30f, 1e30f}, cmx[3] = {-1e30f, -1e30f, -1e30f};
        int nSmall = 0;
        for (int i = 0; i < n_tris; i++) {
            float mn[3], mx[3]; tribox(tris[i], mn, mx);
            const float dd = std::sqrt((mx[0] - mn[0]) * (mx[0] - mn[0]) + (mx[1] - mn[1]) * (mx[1] - mn[1]) + (mx[2] - mn[2]) * (mx[2] - mn[2]));
            if (dd * 8.f < sd) { nSmall++; for (int k = 0; k < 3; k++) { cmn[k] = std::fmin(cmn[k], mn[k]); cmx[k] = std::fmax(cmx[k], mx[k]); } }
        }
        const double sv = (double)(smx[0] - smn[0]) * (smx[1] - smn[1]) * (smx[2] - smn[2]);
        const double cv = nSmall ? (double)(cmx[0] - cmn[0]) * (cmx[1] - cmn[1]) * (cmx[2] - cmn[2]) : 0.0;
        // worth it only if the small triangles are many (they are what makes rays long) and leave a good part of the scene free;
        // PTAMD_CLASS=0 switches the queue order off (A/B)
        if (nSmall >= 64 && std::isfinite(sv) && sv > 0.0 && cv <= 0.6 * sv && !(getenv("PTAMD_CLASS") && atoi(getenv("PTAMD_CLASS")) == 0)) {
            for (int k = 0; k < 3; k++) { const float pad = 0.01f * (cmx[k] - cmn[k]) + 1e-4f * sd; cmn[k] -= pad; cmx[k] += pad; }
            core = {cmn[0], cmn[1], cmn[2], cmx[0], cmx[1], cmx[2]};
        }
    }
    std::vector<float> sph((size_t)n_spheres * 16);
    for (int i = 0; i < n_spheres; i++) {
        const PtSphere& s = spheres[i];
        float* a = &sph[(size_t)i * 16];
        a[0] = s.center[0]; a[1] = s.center[1]; a[2] = s.center[2]; a[3] = s.rad;
        memcpy(a + 4, &s.mat, sizeof(PtMaterial));
    }

    HIPCHK(hipSetDevice(device));
    PtScene* sc = new PtScene();
    sc->device = device;
    sc->n_lights = n_lights;
    sc->max_depth = max_depth;
    int rc;
    if ((rc = upload(&sc->d_nodes, accel.wide.data(), accel.wide.size() * 4, sc->bytes)) ||
        (rc = upload(&sc->d_quad, accel.quad.data(), accel.quad.size() * 4, sc->bytes)) ||
        (rc = upload(&sc->d_tri, accel.tri.data(), accel.tri.size() * 4, sc->bytes)) ||
        (rc = upload(&sc->d_tripair, accel.tripair.data(), accel.tripair.size() * 4, sc->bytes)) ||
        (rc = upload(&sc->d_leafbox, accel.leafbox.data(), accel.leafbox.size() * 4, sc->bytes)) ||
        (rc = upload(&sc->d_surf, surf.data(), surf.size() * 4, sc->bytes)) ||
        (rc = upload(&sc->d_lights, lights.data(), lights.size() * 4, sc->bytes)) ||
        (rc = upload(&sc->d_spheres, sph.data(), sph.size() * 4, sc->bytes)) ||
        (!core.empty() && (rc = upload(&sc->d_core, core.data(), core.size() * 4, sc->bytes)))) {
        pt_scene_destroy(sc);
        return rc;
    }
    // from here on every failure destroys the half-built scene (geometry already uploaded, events, streams)
    auto finish = [&]() -> int {
        HIPCHK(hipMalloc((void**)&sc->d_unit_counter, 64));
        HIPCHK(hipMalloc(&sc->d_counters, kCounterBytes));      // 8 work counters (+ the diagnostic launch timeline of wf_trace)
        HIPCHK(hipMemset(sc->d_counters, 0, kCounterBytes));
        for (int i = 0; i < PtScene::kEvRing; i++) { HIPCHK(hipEventCreate(&sc->ev[i][0])); HIPCHK(hipEventCreate(&sc->ev[i][1])); }
        HIPCHK(hipHostMalloc((void**)&sc->h_poll, 4 * 64, hipHostMallocDefault));
        for (int i = 0; i < 3; i++) { HIPCHK(hipStreamCreateWithFlags(&sc->xstreams[i], hipStreamNonBlocking)); HIPCHK(hipEventCreateWithFlags(&sc->ev_join[i], hipEventDisableTiming)); }
        HIPCHK(hipEventCreateWithFlags(&sc->ev_fork, hipEventDisableTiming));
        hipDeviceProp_t prop;
        HIPCHK(hipGetDeviceProperties(&prop, device));
        sc->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        return PT_OK;
    };
    if ((rc = finish()) != PT_OK) { pt_scene_destroy(sc); return rc; }
    // environment overrides of the per-scene defaults (the same settings have C-ABI setters: pt_set_mode, pt_set_drain_threshold)
    if (const char* m = getenv("PTAMD_MODE")) { const int v = atoi(m); if (v >= 0 && v <= 1) sc->mode = v; }
    if (const char* m = getenv("PTAMD_DRAIN")) sc->drain_below = atoi(m);
    if (const char* m = getenv("PTAMD_EARLY")) sc->early_below = atoi(m) > 0 ? atoi(m) : 0;      // 0 = off
    // shading schedule (pt_set_shade_rounds): one bounce per step pays when wf_shade is bound by its arithmetic rather than by the
    // stream state it moves — measured: scenes whose surface table stays in L2 (+10 % on the Cornell room, 34 triangles) while
    // millions of streams are alive; with the 69,564-triangle bunny it is neutral, and with few streams in flight it loses
    sc->shade_rounds = ((size_t)n_tris * 192 <= ((size_t)2 << 20)) ? -1 : 1;
    if (const char* m = getenv("PTAMD_TR")) { const int v = atoi(m); if (v >= -1 && v <= 1) sc->shade_rounds = v; }
    sc->dev.nodes = (const float4*)sc->d_nodes; sc->dev.quad = (const uint4*)sc->d_quad; sc->dev.tri = (const float4*)sc->d_tri;
    sc->dev.tripair = (const float4*)sc->d_tripair;
    sc->dev.leafbox = (const float4*)sc->d_leafbox; sc->dev.surf = (const float4*)sc->d_surf;
    sc->dev.lights = (const float4*)sc->d_lights; sc->dev.spheres = (const float4*)sc->d_spheres;
    sc->dev.n_quad = accel.n_quad; sc->dev.quad_depth = accel.quad_depth;
    sc->dev.core = (const float*)sc->d_core;      // nullptr: no queue order by ray class
    sc->dev.nee_prune = (emitOk && !(getenv("PTAMD_PRUNE") && atoi(getenv("PTAMD_PRUNE")) == 0)) ? 1 : 0;      // PTAMD_PRUNE=0: A/B only
    sc->dev.n_nodes = n_wide; sc->dev.n_tris = n_tris; sc->dev.n_lights = n_lights; sc->dev.n_spheres = n_spheres;
    *out = sc;
    return PT_OK;
}

void pt_scene_destroy(PtScene* s)
{
    if (!s) return;
    (void)hipSetDevice(s->device);
    void* p[] = {s->d_nodes, s->d_quad, s->d_tri, s->d_tripair, s->d_leafbox, s->d_surf, s->d_lights, s->d_spheres, s->d_core, s->d_unit_counter, s->d_counters};
    for (void* q : p) if (q) (void)hipFree(q);
    for (int i = 0; i < PtScene::kEvRing; i++) for (int j = 0; j < 2; j++) if (s->ev[i][j]) (void)hipEventDestroy(s->ev[i][j]);
    if (s->h_poll) (void)hipHostFree(s->h_poll);
    for (int i = 0; i < 3; i++) { if (s->xstreams[i]) (void)hipStreamDestroy(s->xstreams[i]); if (s->ev_join[i]) (void)hipEventDestroy(s->ev_join[i]); }
    if (s->ev_fork) (void)hipEventDestroy(s->ev_fork);
    for (hipEvent_t e : s->trace_ev) (void)hipEventDestroy(e);
    delete s;
}

int32_t pt_scene_num_lights(const PtScene* s) { return s ? s->n_lights : 0; }
__attribute__((visibility("hidden"))) int ptk_scene_device(const PtScene* s) { return s ? s->device : -1; }      // for pt_comm.hip
int64_t pt_scene_device_bytes(const PtScene* s) { return s ? s->bytes : 0; }

// persistent grid of the traversal kernel: 256 CUs x 7 blocks of 4 waves = 7 waves/SIMD, what its 72 VGPRs allow
// (PTAMD_TB overrides, tuning only; measured: 1536 blocks -3 %, 1024 blocks -22 %)
static const int kTraceBlocks = (getenv("PTAMD_TB") && atoi(getenv("PTAMD_TB")) >= 1) ? (atoi(getenv("PTAMD_TB")) > 16384 ? 16384 : atoi(getenv("PTAMD_TB"))) : 1792;

// ---- geometry of the tile split --------------------------------------------------------
static int fill_params(const PtCamera* cam, const PtParams* prm, ptd::DevParams& d)
{
    if (!cam || !prm) { pt_set_error("NULL camera/params"); return PT_ERR_INVALID; }
    if (cam->W < 2 || cam->H < 2) { pt_set_error("frame %dx%d too small (W-1, H-1 divide, srcs/pathtracer.cu:35-36)", cam->W, cam->H); return PT_ERR_INVALID; }
    if (prm->spp_per_pass > 65535 || prm->max_bounce > 255 || prm->max_refract < 0 || prm->max_refract > 250) {
        pt_set_error("params out of range: spp_per_pass <= 65535, max_bounce <= 255, 0 <= max_refract <= 250");
        return PT_ERR_INVALID;
    }
    if (prm->passes < 1 || prm->spp_per_pass < 1 || prm->max_bounce < 1 || prm->world < 1 || prm->rank < 0 || prm->rank >= prm->world) {
        pt_set_error("bad params: passes=%d spp=%d max_bounce=%d rank=%d world=%d", prm->passes, prm->spp_per_pass, prm->max_bounce, prm->rank, prm->world);
        return PT_ERR_INVALID;
    }
    const long long maxseed = (long long)cam->W * cam->H * (long long)(prm->first_pass + prm->passes);
    if (maxseed > 0x7fffffffLL) { pt_set_error("offset + SampleIDX*W*H overflows int (srcs/pathtracer.cu:71)"); return PT_ERR_INVALID; }
    d.passes = prm->passes; d.spp_per_pass = prm->spp_per_pass; d.max_bounce = prm->max_bounce; d.rr_bounce = prm->rr_bounce;
    d.rr_floor = prm->rr_floor; d.max_refract = prm->max_refract; d.first_pass = prm->first_pass;
    d.rank = prm->rank; d.world = prm->world;
    d.tiles_x = (cam->W + ptd::kTile - 1) / ptd::kTile;
    d.tiles_y = (cam->H + ptd::kTile - 1) / ptd::kTile;
    d.n_tiles_total = d.tiles_x * d.tiles_y;
    d.n_tiles_local = (d.n_tiles_total + prm->world - 1) / prm->world;
    const long long units = (long long)d.n_tiles_local * prm->passes;
    if (units > 0x7fffffffLL) { pt_set_error("too many work units"); return PT_ERR_INVALID; }
    d.n_units = (int)units;
    d.unit_base = 0;
    return PT_OK;
}

// srcs/pathtracer.cu:193-198 and :35-36 — the per-launch camera constants, tan/atan2 as correctly rounded float functions
static void fill_camera(const PtCamera* cam, ptd::DevCamera& c)
{
    memcpy(c.pos, cam->pos, 12); memcpy(c.forward, cam->forward, 12); memcpy(c.up, cam->up, 12); memcpy(c.right, cam->right, 12);
    c.W = cam->W; c.H = cam->H;
    const float fovy = cam->fovy_deg * 0.01745329251994329576923690768489f;                 // glm::radians
    const float fovx = 2.f * (float)std::atan2((double)((float)std::tan((double)(fovy * 0.5f)) * cam->aspect), 1.0);
    c.tan_half_fovx = (float)std::tan((double)(fovx * 0.5f));
    c.tan_half_fovy = (float)std::tan((double)(fovy * 0.5f));
}

int64_t pt_tiles_floats(const PtCamera* cam, const PtParams* prm)
{
    ptd::DevParams d;
    if (fill_params(cam, prm, d)) return -1;
    return (int64_t)d.n_tiles_local * ptd::kTilePixels * 3;
}
int64_t pt_work_bytes(const PtCamera* cam, const PtParams* prm)
{
    ptd::DevParams d;
    if (fill_params(cam, prm, d)) return -1;
    const int64_t mega = (int64_t)d.n_tiles_local * ptd::kTilePixels * 3 * 4 * prm->passes;
    const int64_t wave = (int64_t)ptk_wf_work_bytes((size_t)d.n_units, kTraceBlocks);
    return mega > wave ? mega : wave;
}

int pt_render_tiles(PtScene* s, const PtCamera* cam, const PtParams* prm, float* d_tiles, void* d_work, void* hip_stream)
{
    if (!s || !d_tiles || !d_work) { pt_set_error("pt_render_tiles: NULL argument"); return PT_ERR_INVALID; }
    if (s->n_lights < 1) {
        pt_set_error("scene has no emissive triangle: the reference's `curand(s) %% Nl` is undefined (include/CudaUtil.cuh:235)");
        return PT_ERR_NO_LIGHT;
    }
    ptd::DevParams d;
    int rc = fill_params(cam, prm, d);
    if (rc) return rc;
    ptd::DevCamera c;
    fill_camera(cam, c);

    hipStream_t stream = (hipStream_t)hip_stream;
    HIPCHK(hipSetDevice(s->device));
    const int slot = s->ev_count % PtScene::kEvRing;
    const long long perPass = (long long)d.n_tiles_local * ptd::kTilePixels * 3;
    if (s->mode == 1 && !s->count_next) {
        // queue-driven pipeline (pt_wavefront.hip); polls the live-stream count, so it returns once the render has drained
        int iters = 0;
        const int C = ptk_wf_cohorts((size_t)d.n_units);
        s->trace_ev_per = s->trace_ev.empty() ? 0 : (int)(s->trace_ev.size() / 3) / C;
        for (int k = 0; k < 4; k++) s->trace_ev_used[k] = 0;
        // diagnostic (PTAMD_TSTAT=1): wf_trace counts its trips and the lanes they serve; read with pt_last_counters
        static const bool kTraceStat = getenv("PTAMD_TSTAT") && atoi(getenv("PTAMD_TSTAT")) != 0;
        if (kTraceStat) HIPCHK(hipMemsetAsync(s->d_counters, 0, kCounterBytes, stream));
        HIPCHK(ptk_wf_render(s->device, &s->dev, &c, &d, d_work, kTraceBlocks, s->h_poll, stream, s->xstreams,
                             s->ev[slot][0], s->ev[slot][1], s->ev_fork, s->ev_join, &iters,
                             s->trace_ev.empty() ? nullptr : s->trace_ev.data(), (int)s->trace_ev.size() / 3, s->trace_ev_used, s->drain_below, s->shade_rounds,
                             kTraceStat ? s->d_counters : nullptr, s->early_below));
        s->last_iters = iters;
        s->ev_count++;
        HIPCHK(ptk_sum_passes(ptk_wf_staging(d_work), d.passes, perPass, d_tiles, stream));
        return PT_OK;
    }
    HIPCHK(hipMemsetAsync(s->d_unit_counter, 0, 4, stream));
    if (s->count_next) HIPCHK(hipMemsetAsync(s->d_counters, 0, 64, stream));
    // persistent grid: 4 blocks of 4 waves per CU (16 waves/CU; register- and LDS-feasible), never more blocks than units need
    int blocks = s->num_cus * 4;
    const int need = (d.n_units + ptd::kWavesPerBlock - 1) / ptd::kWavesPerBlock;
    if (blocks > need) blocks = need;
    if (blocks < 1) blocks = 1;
    // events bracket exactly the render_units launch (the dominant kernel), on the launch stream
    HIPCHK(hipEventRecord(s->ev[slot][0], stream));
    HIPCHK(ptk_render_units(&s->dev, &c, &d, (float*)d_work, s->d_unit_counter, s->d_counters, blocks, s->count_next ? 1 : 0, stream));
    HIPCHK(hipEventRecord(s->ev[slot][1], stream));
    s->ev_count++;
    HIPCHK(ptk_sum_passes((const float*)d_work, d.passes, perPass, d_tiles, stream));
    return PT_OK;
}

int pt_untile(const float* d_gathered, const PtCamera* cam, int32_t world, float* d_frame_rgb, void* hip_stream)
{
    if (!d_gathered || !cam || !d_frame_rgb || world < 1) { pt_set_error("pt_untile: bad argument"); return PT_ERR_INVALID; }
    const int tiles_x = (cam->W + ptd::kTile - 1) / ptd::kTile, tiles_y = (cam->H + ptd::kTile - 1) / ptd::kTile;
    const int n_total = tiles_x * tiles_y;
    const long long per_rank = (long long)((n_total + world - 1) / world) * ptd::kTilePixels * 3;
    HIPCHK(ptk_untile(d_gathered, cam->W, cam->H, tiles_x, n_total, world, per_rank, d_frame_rgb, (hipStream_t)hip_stream));
    return PT_OK;
}

int pt_render(PtScene* s, const PtCamera* cam, const PtParams* prm, float* h_accum_rgb)
{
    if (!s || !h_accum_rgb || !prm) { pt_set_error("pt_render: NULL argument"); return PT_ERR_INVALID; }
    PtParams p = *prm; p.rank = 0; p.world = 1;
    const int64_t nt = pt_tiles_floats(cam, &p), wb = pt_work_bytes(cam, &p);
    if (nt < 0 || wb < 0) return PT_ERR_INVALID;
    HIPCHK(hipSetDevice(s->device));
    float *d_tiles = nullptr, *d_frame = nullptr; void* d_work = nullptr;
    auto body = [&]() -> int {
        HIPCHK(hipMalloc((void**)&d_tiles, (size_t)nt * 4));
        HIPCHK(hipMalloc(&d_work, (size_t)wb));
        HIPCHK(hipMalloc((void**)&d_frame, (size_t)cam->W * cam->H * 12));
        int r = pt_render_tiles(s, cam, &p, d_tiles, d_work, nullptr);
        if (!r) r = pt_untile(d_tiles, cam, 1, d_frame, nullptr);
        if (!r) HIPCHK(hipMemcpy(h_accum_rgb, d_frame, (size_t)cam->W * cam->H * 12, hipMemcpyDeviceToHost));
        return r;
    };
    const int rc = body();
    (void)hipFree(d_tiles); (void)hipFree(d_work); (void)hipFree(d_frame);
    return rc;
}

int pt_last_render_ms(PtScene* s, float* ms)
{
    if (!s || !ms || s->ev_count < 1) { pt_set_error("pt_last_render_ms: nothing rendered yet"); return PT_ERR_INVALID; }
    const int slot = (s->ev_count - 1) % PtScene::kEvRing;
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipEventSynchronize(s->ev[slot][1]));
    HIPCHK(hipEventElapsedTime(ms, s->ev[slot][0], s->ev[slot][1]));
    return PT_OK;
}

int pt_render_timings(PtScene* s, float* ms_out, int32_t cap, int32_t reset)
{
    if (!s) { pt_set_error("pt_render_timings: NULL scene"); return PT_ERR_INVALID; }
    int n = s->ev_count < PtScene::kEvRing ? s->ev_count : PtScene::kEvRing;
    if (n > cap) n = cap;
    HIPCHK(hipSetDevice(s->device));
    for (int i = 0; i < n; i++) {
        const int slot = (s->ev_count - n + i) % PtScene::kEvRing;
        HIPCHK(hipEventSynchronize(s->ev[slot][1]));
        HIPCHK(hipEventElapsedTime(&ms_out[i], s->ev[slot][0], s->ev[slot][1]));
    }
    if (reset) s->ev_count = 0;
    return n;
}

int pt_last_counters(PtScene* s, int64_t* out8)
{
    if (!s || !out8) { pt_set_error("pt_last_counters: NULL"); return PT_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out8, s->d_counters, 64, hipMemcpyDeviceToHost));
    return PT_OK;
}
// Record a HIP event pair around each of the first `max_launches` wf_trace launches of every
// following render (0 turns it off); pt_trace_timing then reports their summed duration.
PT_API int pt_enable_trace_timing(PtScene* s, int32_t max_launches)
{
    if (!s || max_launches < 0 || max_launches > (1 << 20)) { pt_set_error("pt_enable_trace_timing: bad argument"); return PT_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    for (hipEvent_t e : s->trace_ev) (void)hipEventDestroy(e);
    s->trace_ev.assign((size_t)max_launches * 3, nullptr);
    for (auto& e : s->trace_ev) HIPCHK(hipEventCreate(&e));
    for (int k = 0; k < 4; k++) s->trace_ev_used[k] = 0;
    return PT_OK;
}
static int kernel_timing(PtScene* s, int first, double* sum_ms, int32_t* launches, double* max_ms)
{
    if (!s || !sum_ms || !launches) { pt_set_error("pt_trace_timing / pt_shade_timing: NULL"); return PT_ERR_INVALID; }
    HIPCHK(hipSetDevice(s->device));
    double sum = 0, mx = 0;
    int total = 0;
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < s->trace_ev_used[c]; i++) {
            const size_t k = ((size_t)c * s->trace_ev_per + i) * 3 + (size_t)first;
            float ms = 0.f;
            HIPCHK(hipEventSynchronize(s->trace_ev[k + 1]));
            HIPCHK(hipEventElapsedTime(&ms, s->trace_ev[k], s->trace_ev[k + 1]));
            sum += ms; if (ms > mx) mx = ms;
            total++;
        }
    *sum_ms = sum; *launches = total; if (max_ms) *max_ms = mx;
    return PT_OK;
}
PT_API int pt_trace_timing(PtScene* s, double* sum_ms, int32_t* launches, double* max_ms) { return kernel_timing(s, 0, sum_ms, launches, max_ms); }
PT_API int pt_shade_timing(PtScene* s, double* sum_ms, int32_t* launches, double* max_ms) { return kernel_timing(s, 1, sum_ms, launches, max_ms); }
PT_API int pt_set_mode(PtScene* s, int32_t mode) { if (!s || mode < 0 || mode > 1) { pt_set_error("pt_set_mode: mode must be 0 or 1"); return PT_ERR_INVALID; } s->mode = mode; return PT_OK; }
PT_API int pt_last_iterations(PtScene* s) { return s ? s->last_iters : -1; }
PT_API int pt_set_drain_threshold(PtScene* s, int32_t live_streams)
{
    if (!s || live_streams < 0) { pt_set_error("pt_set_drain_threshold: bad argument"); return PT_ERR_INVALID; }
    s->drain_below = live_streams;
    return PT_OK;
}
PT_API int pt_set_early_shade(PtScene* s, int32_t live_streams)
{
    if (!s || live_streams < 0) { pt_set_error("pt_set_early_shade: bad argument"); return PT_ERR_INVALID; }
    s->early_below = live_streams;
    return PT_OK;
}
PT_API int pt_set_shade_rounds(PtScene* s, int32_t mode)
{
    if (!s || mode < -1 || mode > 1) { pt_set_error("pt_set_shade_rounds: mode must be -1, 0 or 1"); return PT_ERR_INVALID; }
    s->shade_rounds = mode;
    return PT_OK;
}
// Ask the next pt_render_tiles on this scene to run the counting build of the kernel.
int pt_dbg_trace_timeline(PtScene* s, int64_t* out3n, int32_t n_launches)
{
    if (!s || !out3n || (n_launches < -2700 && (n_launches > -3000 || n_launches < -3005)) || n_launches > 2700) { pt_set_error("pt_dbg_trace_timeline: bad arguments"); return PT_ERR_INVALID; }
    if (n_launches == -3005) {      // PTAMD_TSTAT=2: the raw timeline stripes, 2700 launches x kStatStripes x 3 int64 (maxima of ~start, ~dry, end per stripe of workgroups)
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(out3n, (const char*)s->d_counters + ptd::kStatStripeOff, (size_t)2700 * ptd::kStatStripes * 24, hipMemcpyDeviceToHost));
        return PT_OK;
    }
    if (n_launches == -3003 || n_launches == -3004) {      // PTAMD_TSTAT=2 + PTAMD_TDUMP=launch: 8 x int64 per wave (kStatWaves) / the per-trip log (kStatLogWaves x kStatLogTrips uint32)
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipDeviceSynchronize());
        const size_t off = (size_t)ptd::kStatWords * 8, wb = (size_t)ptd::kStatWaves * 64;
        if (n_launches == -3003) HIPCHK(hipMemcpy(out3n, (const char*)s->d_counters + off, wb, hipMemcpyDeviceToHost));
        else HIPCHK(hipMemcpy(out3n, (const char*)s->d_counters + off + wb, (size_t)ptd::kStatLogWaves * ptd::kStatLogTrips * 4, hipMemcpyDeviceToHost));
        return PT_OK;
    }
    if (n_launches == -3002) {      // shader clocks per section of wf_trace's loop, summed over waves (5 x int64: refill, vote, node step, triangle step, epilogue), PTAMD_TSTAT=1
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(out3n, (const char*)s->d_counters + 64 + 2700 * 24 + 32 * 8 + 2700 * 8 + 64 * 8 + 32 * 8, 5 * 8, hipMemcpyDeviceToHost));
        return PT_OK;
    }
    if (n_launches == -3001) {      // histogram of the stack depth after each node step (32 x int64), PTAMD_TSTAT=1
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(out3n, (const char*)s->d_counters + 64 + 2700 * 24 + 32 * 8 + 2700 * 8 + 64 * 8, 32 * 8, hipMemcpyDeviceToHost));
        return PT_OK;
    }
    if (n_launches == -3000) {      // the histogram of node steps per ray (64 x int64: bins of 4 steps), PTAMD_TSTAT=1
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(out3n, (const char*)s->d_counters + 64 + 2700 * 24 + 32 * 8 + 2700 * 8, 64 * 8, hipMemcpyDeviceToHost));
        return PT_OK;
    }
    if (n_launches < 0) {      // -n: the ray count of each of the first n launches (n x int64)
        HIPCHK(hipSetDevice(s->device));
        HIPCHK(hipDeviceSynchronize());
        HIPCHK(hipMemcpy(out3n, (const char*)s->d_counters + 64 + 2700 * 24 + 32 * 8, (size_t)(-n_launches) * 8, hipMemcpyDeviceToHost));
        return PT_OK;
    }
    HIPCHK(hipSetDevice(s->device));
    HIPCHK(hipDeviceSynchronize());
    static const bool kStriped = getenv("PTAMD_TSTAT") && atoi(getenv("PTAMD_TSTAT")) == 2;
    if (kStriped && n_launches > 0) {
        // the timestamp-only build keeps kStatStripes copies of every launch's three words (maxima of ~start, ~dry, end)
        std::vector<unsigned long long> raw((size_t)n_launches * ptd::kStatStripes * 3);
        HIPCHK(hipMemcpy(raw.data(), (const char*)s->d_counters + ptd::kStatStripeOff, raw.size() * 8, hipMemcpyDeviceToHost));
        for (int l = 0; l < n_launches; l++)
            for (int k = 0; k < 3; k++) {
                unsigned long long m = 0;
                for (int st = 0; st < ptd::kStatStripes; st++) { const unsigned long long v = raw[((size_t)l * ptd::kStatStripes + st) * 3 + k]; if (v > m) m = v; }
                out3n[(size_t)l * 3 + k] = (int64_t)m;
            }
    } else
    HIPCHK(hipMemcpy(out3n, (const char*)s->d_counters + 64, (size_t)n_launches * 24, hipMemcpyDeviceToHost));
    if (n_launches == 0 && out3n) HIPCHK(hipMemcpy(out3n, (const char*)s->d_counters + 64 + 2700 * 24, 32 * 8, hipMemcpyDeviceToHost));   // n = 0: the 32-bin histogram of wave lifetimes (32 us bins)
    return PT_OK;
}

PT_API int pt_enable_counters(PtScene* s, int32_t on) { if (!s) return PT_ERR_INVALID; s->count_next = on != 0; return PT_OK; }

// ---- parity hooks ------------------------------------------------------------------------
int pt_dbg_raycast(PtScene* s, const float* rays8, int32_t n, float* out_hits29, int32_t* out_prim)
{
    if (!s || !rays8 || n < 0 || !out_hits29 || !out_prim) { pt_set_error("pt_dbg_raycast: bad argument"); return PT_ERR_INVALID; }
    return with_buffers(s->device, rays8, (size_t)n * 32, out_hits29, (size_t)n * 29 * 4, out_prim, (size_t)n * 4,
                        [&](void* i, void* o, void* o2) { return ptk_dbg_raycast(&s->dev, (const float*)i, n, (float*)o, (int*)o2, nullptr); });
}
int pt_dbg_bxdf(int32_t device, int32_t lobe, const float* in28, int32_t n, float* out12)
{
    if (!in28 || !out12 || n < 0 || lobe < 0 || lobe > 3) { pt_set_error("pt_dbg_bxdf: bad argument"); return PT_ERR_INVALID; }
    return with_buffers(device, in28, (size_t)n * 28 * 4, out12, (size_t)n * 12 * 4, nullptr, 0,
                        [&](void* i, void* o, void*) { return ptk_dbg_bxdf(lobe, (const float*)i, n, (float*)o, nullptr); });
}
int pt_dbg_rng(int32_t device, uint64_t seed, int32_t n, uint32_t* raw_out, float* uniform_out)
{
    if (!raw_out || !uniform_out || n < 0) { pt_set_error("pt_dbg_rng: bad argument"); return PT_ERR_INVALID; }
    return with_buffers(device, nullptr, 0, raw_out, (size_t)n * 4, uniform_out, (size_t)n * 4,
                        [&](void*, void* o, void* o2) { return ptk_dbg_rng(seed, n, (uint32_t*)o, (float*)o2, nullptr); });
}
__global__ __launch_bounds__(256) void triad_kernel(float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float s, size_t n)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        const float4 x = b[i], y = c[i];
        a[i] = make_float4(x.x + s * y.x, x.y + s * y.y, x.z + s * y.z, x.w + s * y.w);
    }
}

int pt_dbg_triad(int32_t device, int64_t bytes_per_array, int32_t iters, double* gb_per_s)
{
    if (!gb_per_s || bytes_per_array < 4096 || iters < 1) { pt_set_error("pt_dbg_triad: bad argument"); return PT_ERR_INVALID; }
    HIPCHK(hipSetDevice(device));
    const size_t n = (size_t)bytes_per_array / 16;
    float4 *a = nullptr, *b = nullptr, *c = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PT_OK;
    do {
        if (hipMalloc((void**)&a, n * 16) != hipSuccess || hipMalloc((void**)&b, n * 16) != hipSuccess || hipMalloc((void**)&c, n * 16) != hipSuccess) { pt_set_error("pt_dbg_triad: out of device memory"); rc = PT_ERR_DEVICE; break; }
        (void)hipMemset(b, 0, n * 16); (void)hipMemset(c, 0, n * 16);
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        const unsigned blocks = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(triad_kernel, dim3(blocks), dim3(256), 0, 0, a, b, c, 0.5f, n);       // warm-up
        (void)hipEventRecord(e0, 0);
        for (int k = 0; k < iters; k++) hipLaunchKernelGGL(triad_kernel, dim3(blocks), dim3(256), 0, 0, a, b, c, 0.5f, n);
        (void)hipEventRecord(e1, 0);
        if (hipEventSynchronize(e1) != hipSuccess) { pt_set_error("pt_dbg_triad: kernel failed"); rc = PT_ERR_DEVICE; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        *gb_per_s = 3.0 * (double)n * 16.0 * iters / ((double)ms * 1e-3) / 1e9;
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    if (c) (void)hipFree(c);
    return rc;
}

int pt_dbg_pixel_dir(int32_t device, const PtCamera* cam, const int32_t* pxpypass, int32_t n, float* out8)
{
    if (!cam || !pxpypass || !out8 || n < 0 || cam->W < 2 || cam->H < 2) { pt_set_error("pt_dbg_pixel_dir: bad argument"); return PT_ERR_INVALID; }
    ptd::DevCamera c;
    fill_camera(cam, c);
    return with_buffers(device, pxpypass, (size_t)n * 12, out8, (size_t)n * 32, nullptr, 0,
                        [&](void* i, void* o, void*) { return ptk_dbg_pixel_dir(&c, (const int*)i, n, (float*)o, nullptr); });
}
int pt_dbg_nee(PtScene* s, const float* in5, int32_t n, float* out12)
{
    if (!s || !in5 || !out12 || n < 0) { pt_set_error("pt_dbg_nee: bad argument"); return PT_ERR_INVALID; }
    if (s->n_lights < 1) { pt_set_error("pt_dbg_nee: scene has no light"); return PT_ERR_NO_LIGHT; }
    return with_buffers(s->device, in5, (size_t)n * 20, out12, (size_t)n * 48, nullptr, 0,
                        [&](void* i, void* o, void*) { return ptk_dbg_nee(&s->dev, (const float*)i, n, (float*)o, nullptr); });
}

int pt_dbg_ray_setup(int32_t device, const float* dir3, int32_t n, float* out5)
{
    if (!dir3 || !out5 || n < 0) { pt_set_error("pt_dbg_ray_setup: bad argument"); return PT_ERR_INVALID; }
    return with_buffers(device, dir3, (size_t)n * 12, out5, (size_t)n * 20, nullptr, 0,
                        [&](void* i, void* o, void*) { return ptk_dbg_ray_setup((const float*)i, n, (float*)o, nullptr); });
}
int pt_dbg_math(int32_t device, const float* in, int32_t n, float* out8)
{
    if (!in || !out8 || n < 0) { pt_set_error("pt_dbg_math: bad argument"); return PT_ERR_INVALID; }
    return with_buffers(device, in, (size_t)n * 4, out8, (size_t)n * 8 * 4, nullptr, 0,
                        [&](void* i, void* o, void*) { return ptk_dbg_math((const float*)i, n, (float*)o, nullptr); });
}
int pt_dbg_sincos(int32_t device, const float* in, int32_t n, float* out2)
{
    if (!in || !out2 || n < 0) { pt_set_error("pt_dbg_sincos: bad argument"); return PT_ERR_INVALID; }
    return with_buffers(device, in, (size_t)n * 4, out2, (size_t)n * 2 * 4, nullptr, 0,
                        [&](void* i, void* o, void*) { return ptk_dbg_sincos((const float*)i, n, (float*)o, nullptr); });
}

}  // extern "C"
