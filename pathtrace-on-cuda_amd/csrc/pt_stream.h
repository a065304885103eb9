// pt_stream.h — a stream (one pixel of one pass) as data: its HBM-resident state, the one-bounce
// step function, and the prologue that starts it (used by the queue-driven pipeline, pt_wavefront.hip).
#pragma once
#include "pt_device.h"
#include "pt_math.h"
#include "pt_bxdf.h"
#include "pt_shade.h"

namespace ptd {

// Stream state, rays and hit records are touched once per kernel by one lane each: PT_STREAM_NT = 1 moves them with non-temporal
// loads / stores (the `nt` cache hint) so that they do not push the traversal tree and the surface table out of the L2s.
#ifndef PT_STREAM_NT
#define PT_STREAM_NT 0
#endif
#ifndef PT_STREAM_NT_LD
#define PT_STREAM_NT_LD PT_STREAM_NT
#endif
#ifndef PT_STREAM_NT_ST
#define PT_STREAM_NT_ST PT_STREAM_NT
#endif
typedef float pt_v4f __attribute__((ext_vector_type(4)));
typedef float pt_v2f __attribute__((ext_vector_type(2)));
typedef uint32_t pt_v4u __attribute__((ext_vector_type(4)));
PT_DEV float4 ld_s(const float4* p) {
#if PT_STREAM_NT_LD
    const pt_v4f v = __builtin_nontemporal_load((const pt_v4f*)p); return make_float4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
PT_DEV uint4 ld_s(const uint4* p) {
#if PT_STREAM_NT_LD
    const pt_v4u v = __builtin_nontemporal_load((const pt_v4u*)p); return make_uint4(v.x, v.y, v.z, v.w);
#else
    return *p;
#endif
}
PT_DEV float2 ld_s(const float2* p) {
#if PT_STREAM_NT_LD
    const pt_v2f v = __builtin_nontemporal_load((const pt_v2f*)p); return make_float2(v.x, v.y);
#else
    return *p;
#endif
}
PT_DEV uint32_t ld_s(const uint32_t* p) {
#if PT_STREAM_NT_LD
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
// A/B: PT_STREAM_ST_MODE = cache-policy modifiers of the state stores: 1 "sc1", 2 "sc0 sc1", 3 "sc0", 4 "nt sc1"
#if defined(PT_STREAM_ST_MODE) && PT_STREAM_ST_MODE == 1
#define PT_STREAM_ST_MOD "sc1"
#elif defined(PT_STREAM_ST_MODE) && PT_STREAM_ST_MODE == 2
#define PT_STREAM_ST_MOD "sc0 sc1"
#elif defined(PT_STREAM_ST_MODE) && PT_STREAM_ST_MODE == 3
#define PT_STREAM_ST_MOD "sc0"
#elif defined(PT_STREAM_ST_MODE) && PT_STREAM_ST_MODE == 4
#define PT_STREAM_ST_MOD "nt sc1"
#endif
#ifdef PT_STREAM_ST_MOD
PT_DEV void st_s(float4* p, const float4 v) { const pt_v4f w = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off " PT_STREAM_ST_MOD : : "v"(p), "v"(w) : "memory"); }
PT_DEV void st_s(uint4* p, const uint4 v) { const pt_v4u w = {v.x, v.y, v.z, v.w}; asm volatile("global_store_dwordx4 %0, %1, off " PT_STREAM_ST_MOD : : "v"(p), "v"(w) : "memory"); }
PT_DEV void st_s(float2* p, const float2 v) { const pt_v2f w = {v.x, v.y}; asm volatile("global_store_dwordx2 %0, %1, off " PT_STREAM_ST_MOD : : "v"(p), "v"(w) : "memory"); }
#else
PT_DEV void st_s(float4* p, const float4 v) {
#if PT_STREAM_NT_ST
    const pt_v4f w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (pt_v4f*)p);
#else
    *p = v;
#endif
}
PT_DEV void st_s(uint4* p, const uint4 v) {
#if PT_STREAM_NT_ST
    const pt_v4u w = {v.x, v.y, v.z, v.w}; __builtin_nontemporal_store(w, (pt_v4u*)p);
#else
    *p = v;
#endif
}
PT_DEV void st_s(float2* p, const float2 v) {
#if PT_STREAM_NT_ST
    const pt_v2f w = {v.x, v.y}; __builtin_nontemporal_store(w, (pt_v2f*)p);
#else
    *p = v;
#endif
}
#endif

enum : uint32_t {
    F_REFR = 1,       // bRefracted of the current sample (loop-carried, Q8)
    F_NEEOK = 2,      // !isnan(brdfcos) of the current sample's pending NEE term
    F_SHADOW = 4,     // the current sample has a pending NEE term: shadow ray (kind 1) traced
    F_PATH = 8,       // the current sample's path ray (kind 0) traced
    F_PRIMARY = 16,   // that path ray is the pixel's camera ray (first iteration only): its hit is cached
    F_SHADOWA = 32,   // an OLDER sample, already closed, still has its last NEE term pending: shadow ray (kind 2) traced
    F_NEEOKA = 64,    // !isnan(brdfcos) of that term
    F_CUR = 128,      // a current sample exists (started, not yet added to the pixel)
};
constexpr uint32_t kResumeBit = 0x80000000u;      // in a ray-queue entry: the stream's ray of that kind is a suspended traversal to resume
constexpr int kNotReady = (int)0x80000000;      // hit.prim of a ray that has been emitted but not traced yet (set when a shade launch runs with MARK)
constexpr int kRayKinds = 3;     // 0 path, 1 shadow of the current sample, 2 shadow of the older closed sample

// Shadow rays only decide whether the closest hit is the sampled light point (GetLightColor, CudaUtil.cuh:150-166: visible iff
// |hit.p - P| < EPS with t_max = |P - p| + 1).  Any hit at t < (t_max - 1) - margin proves that the closest hit lies at least
// margin - EPS in front of P, so the traversal may stop there: margin = 5e-4, or 64 ulps of the largest coordinate involved where
// that is more (coordinates in the thousands: the rounding of org + t*dir must not eat the reference's EPS = 1e-4).  Computed where the
// ray is emitted (the traversal kernel is the one short of issue slots) and carried in ray_d.w.
PT_DEV float shadow_stop_t(const f3& o, float tmax)
{
    const float mag = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(o.x), __builtin_fabsf(o.y)), __builtin_fmaxf(__builtin_fabsf(o.z), tmax));
    return (tmax - 1.0f) - __builtin_fmaxf(5e-4f, mag * 7.6293945e-6f);
}

// Queue order by ray class (round 3).  The launch tail of wf_trace is the longest ray still in flight when the queue runs dry
// (~60-100 steps at ~3 us: ~0.3 ms whatever the launch size).  A ray whose segment misses the box around the scene's small triangles
// (DevScene::core) can only meet the handful of big ones — <= 10 trips on the config scenes, 70 % of all rays — so such rays are
// queued LAST: everything long has been started (and, in a large launch, finished) by the time the queue runs dry, and what is left
// in flight is short.  Pure scheduling: the test may be as sloppy as it likes (approximate reciprocals), it decides no result.
PT_DEV bool ray_is_short(const DevScene& sc, const f3& o, const f3& d, float tmax)
{
    if (sc.core == nullptr) return false;
    const float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
    const float ax = (sc.core[0] - o.x) * ix, bx = (sc.core[3] - o.x) * ix;
    const float ay = (sc.core[1] - o.y) * iy, by = (sc.core[4] - o.y) * iy;
    const float az = (sc.core[2] - o.z) * iz, bz = (sc.core[5] - o.z) * iz;
    const float t0 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fminf(ax, bx), __builtin_fminf(ay, by)), __builtin_fmaxf(__builtin_fminf(az, bz), 0.f));
    const float t1 = __builtin_fminf(__builtin_fminf(__builtin_fmaxf(ax, bx), __builtin_fmaxf(ay, by)), __builtin_fminf(__builtin_fmaxf(az, bz), tmax));
    return !(t0 <= t1);
}

struct WfCounters {      // one slot per iteration parity (3 rotating slots); every hot word on its own 128-B line
    uint32_t nActive, padA[31];
    uint32_t nRays[kRayKinds][32];                 // [kind][0]: rays queued per kind from the front of its queue, [kind][16]: short rays queued from the back (kShortWord)
    uint32_t nSusp, padD[31];
    struct { uint32_t v, pad[31]; } head[16];     // sharded ray-queue heads, one 128-B line each
};
constexpr int kWfShards = 16;
constexpr int kShortWord = 16;      // WfCounters::nRays[kind][kShortWord]: the kind's short-class rays
constexpr int kWfSlotBytes = 128 * (2 + kRayKinds + 16);
static_assert(sizeof(WfCounters) == kWfSlotBytes, "counter slot layout");

struct WfBuf {
    uint4* rng0;         // x0 x1 x2 x3
    uint4* rng1;         // x4 d | samplesToStart<<16 | depth<<8 | refractCnt | flags
    float4* weight;      // weight.xyz | cosA of the pending NEE term
    float4* rad;         // radiance.xyz | denom of the pending NEE term
    float4* pix;         // pixelColor.xyz
    float4* dir0;        // camera ray direction of this pixel & pass (every sample of the pass shares it, Q2)
    float2* hit0;        // its closest hit, traced once: (t, primitive)
    float4* wb;          // weight*brdfcos of the pending NEE term
    float4* lp;          // sampled light point of the pending NEE term
    float4* radA;        // older closed sample: radiance.xyz | denom
    float4* wbA;         //                      weight*brdfcos | cosA
    float4* lpA;         //                      light point
    float4* ray_o[kRayKinds];    // org.xyz | tmax
    float4* ray_d[kRayKinds];    // dir.xyz | shadow rays: the t below which any hit ends the traversal (shadow_stop_t); path rays: -inf
    float2* hit[kRayKinds];      // t | primitive index (int bits); prim <= -2: traversal suspended, record -2-prim
    uint32_t* active[2];         // live stream ids, ping-pong
    uint32_t* rq[kRayKinds];     // ray queues (stream ids)
    WfCounters* cnt;     // [3]
    float* staging;      // per-pass means, [stream][3]
    int* ovf;            // traversal stack overflow (entries >= kWfLdsStack), [level][thread]
    int* susp[2];        // suspended traversals (ping-pong by iteration): [record][kSuspInts]
    uint32_t suspCap;    // records per pool
    uint8_t* res;        // per live-list position: what wf_shade's early phase did with the stream (R_* bits)
};

// StartRender prologue for one pixel & pass (srcs/pathtracer.cu:70-74): seeds the RNG, draws the
// jittered camera direction and queues the camera ray, whose hit all samples of the pass share.
PT_DEV void init_stream(const DevCamera& cam, const DevParams& prm, const WfBuf& b, uint32_t slot, int px, int py, int pass)
{
    const int offset = py * cam.W + px;
    Rng rng;
    rng.init((uint64_t)(int64_t)(offset + pass * cam.W * cam.H));
    float u1, u2;
    const f3 d0 = pixel_direction(cam, px, py, rng, u1, u2);      // GetPixelDirection + Ray ctor (pt_shade.h)
    b.rng0[slot] = make_uint4(rng.x0, rng.x1, rng.x2, rng.x3);
    b.rng1[slot] = make_uint4(rng.x4, rng.d, ((uint32_t)prm.spp_per_pass << 16), F_PATH | F_PRIMARY);
    b.weight[slot] = make_float4(1.f, 1.f, 1.f, 0.f);
    b.rad[slot] = make_float4(0.f, 0.f, 0.f, 1.f);
    b.pix[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    b.dir0[slot] = make_float4(d0.x, d0.y, d0.z, 0.f);
    b.ray_o[0][slot] = make_float4(cam.pos[0], cam.pos[1], cam.pos[2], 999999.f);
    b.ray_d[0][slot] = make_float4(d0.x, d0.y, d0.z, -__builtin_inff());
    b.hit[0][slot] = make_float2(0.f, __int_as_float(kNotReady));      // the camera ray: emitted, not traced yet
    for (int k = 1; k < kRayKinds; k++) b.hit[k][slot] = make_float2(0.f, __int_as_float(-1));
}

// ---------------------------------------------------------------------------------------
// One step of one stream.
//
// The reference runs, per pixel and pass, NUM_SAMPLE paths one after the other
// (srcs/pathtracer.cu:77-81), each a loop of bounces (GetColor_iter, include/CudaUtil.cuh:216-380):
// closest hit -> emission -> NEE sample + shadow ray -> BSDF sample -> roulette.  A stream keeps
// that order of random draws and of floating-point additions, but overlaps what does not depend
// on each other:
//   * the camera ray is the same for every sample of the pass (Q2), so it is traced once and its
//     hit cached; a new sample starts directly with the shading of that hit;
//   * when a path ends, only its last NEE term is still waiting for a shadow ray.  The next
//     sample starts IN THE SAME STEP (its random draws simply follow), and the closed sample's
//     term rides along in slot "A" until the shadow ray is back, one step later; only then is the
//     closed sample added to the pixel — before anything of the newer one, so pixelColor sees
//     the samples in order.
// A step therefore shades up to two hits (the current path's, then the next sample's first) and
// leaves at most three rays to trace: path, shadow, shadow A.  Per sample it takes (bounces - 1)
// steps instead of `bounces`.
// ---------------------------------------------------------------------------------------
struct SState {
    Rng rng;
    int toStart, depth, refractCnt;
    uint32_t flags;                 // F_* of the rays that were traced for this step
    f3 weight, radiance;
    f3 pixelColor; bool pixLoaded;  // loaded lazily: only a closing sample touches it
    float cosA, denom;              // pending NEE term of the current sample ...
    f3 wb, lightP;                  // ... weight*brdfcos, sampled light point
    f3 pathO, pathD;                // path ray (traced if F_PATH)
    f3 shO, shD; float shTmax;      // shadow ray (traced if F_SHADOW)
    uint32_t cls;                   // ray classes for the queue order (bit k: the ray of kind k is short, ray_is_short); scheduling only
    // The older closed sample (slot A: radiance, pending NEE term, shadow ray) never sits in registers across the
    // step: shade_step reads it where it is consumed and writes the new one where it is produced.
};

// ---------------------------------------------------------------------------------------
// One bounce at a hit: everything GetColor_iter does between finding the closest hit and casting the next ray,
// except visibility (include/CudaUtil.cuh:216-373): emission, the NEE sample (its term is parked in st.wb / lightP /
// cosA / denom until the shadow ray is back), the BSDF sample and the roulette.  Leaves the shadow ray in st.sh* and,
// unless the path ends here (returns true), the next path ray in st.pathO / pathD.
// ---------------------------------------------------------------------------------------
PT_DEV bool bounce(const DevScene& sc, const DevParams& prm, int prim, float t, const f3& rorg, const f3& rdir,
                   SState& st, bool& bRefracted, bool& neeOk, bool& needShadow)
{
    Surf s;
    if (prim < sc.n_tris) { SurfRec rec; load_surf(sc, prim, rec); surf_from_rec(rec, t, rorg, rdir, s); }
    else surf_sphere(sc, prim - sc.n_tris, t, rorg, rdir, s);
    if (sqlen(s.m.emittance) > kEps) st.radiance += st.weight * s.m.emittance;   // :220-224
    const float ior = ior_of(s.m);                                          // :231
    const int lobe = lobe_of(s.m);
    const f3 wo = -rdir;
    // NEE sample (:235-245, SamplePrimitive :38-48)
    const NeeSample ns = nee_sample(sc, st.rng, s.p);
    const f3 lightP = ns.lightP, toL = ns.toL, wl = ns.wl;
    const float pdfLight = ns.pdfLight;
    st.cosA = ns.cosA;
    const f3 brdfcos = lobe_eval(lobe, s.m, ior, s.fr, wo, wl);
    neeOk = !anynan(brdfcos);
    st.wb = st.weight * brdfcos;
    st.lightP = lightP;
    st.denom = sqlen(s.p - lightP) * pdfLight;
    // Dead NEE terms need no shadow ray.  The term ((wb * Le) * cosA) / denom (GetLightColor's result times :271-272) depends on the
    // shadow ray only through Le, which is 0 or a primitive's emittance (finite, >= 0 and <= 1e8: checked at upload, DevScene::nee_prune).
    //  * brdfcos NaN: the reference skips the term (:271), nobody reads the ray;
    //  * every component of wb an exact zero (the light is below the surface's horizon: eval returns 0), cosA finite and
    //    0 < denom < inf: wb * Le = +-0 whatever Le is, so the term is +-0, and adding +-0 never changes a radiance (a radiance
    //    component is never -0: it starts as +0 and only (-0) + (-0) gives -0);
    //  * cosA exactly 0 with |wb| < 1e30 and 0 < denom < inf: (wb * Le) is finite (nee_prune also says every Le <= 1e8), times 0 is +-0,
    //    same argument.
    // Everything else — including the NaN-producing corner cases, which must be reproduced — keeps its ray.
    {
        const float kInf = __builtin_inff();
        const bool denomOk = st.denom > 0.f && st.denom < kInf;
        const bool wbZero = st.wb.x == 0.f && st.wb.y == 0.f && st.wb.z == 0.f;
        const bool wbSmall = __builtin_fabsf(st.wb.x) < 1e30f && __builtin_fabsf(st.wb.y) < 1e30f && __builtin_fabsf(st.wb.z) < 1e30f;
        const bool dead = !neeOk || (denomOk && ((wbZero && st.cosA < kInf) || (st.cosA == 0.f && wbSmall)));
        needShadow = !(dead && sc.nee_prune);
    }
    // BSDF sample (:283-338)
    const f3 wi = lobe_sample(lobe, s.m, ior, s.fr, wo, st.rng);
    const f3 w1 = lobe_eval(lobe, s.m, ior, s.fr, wo, wi);
    float w2 = lobe_pdf(lobe, s.m, ior, s.fr, wo, wi);
    w2 = selmax(w2, 1e-2f);
    const f3 cw = w1 / w2;
    if (lobe >= LOBE_REFRACTIVE) bRefracted = (dot(s.fr.n, wo) * dot(s.fr.n, wi)) <= 0.f;   // :307 (loop-carried, Q8)
    bool terminate = false;
    f3 nOrg(0.f, 0.f, 0.f);
    if (sqlen(wi) > kEps) st.weight *= cw; else terminate = true;
    if (!terminate) {
        nOrg = s.p + s.fr.n * (bRefracted ? -kEps : kEps);                  // :349-350
        if (bRefracted) {
            if (st.refractCnt++ > prm.max_refract) terminate = true;        // :351-359 (Depth unchanged)
        } else {
            if (st.depth >= prm.rr_bounce) {                                // :361-373
                const float u = st.rng.uniform();
                const float q = selmax(selmin(maxcomp(st.weight), 1.f), prm.rr_floor);
                if (u < q) st.weight *= (1.f / q); else terminate = true;
            }
            st.depth++;
            if (st.depth >= prm.max_bounce) terminate = true;
        }
    }
    // shadow ray: Ray(p, P - p), t_max = |P - p| + 1 (GetLightColor :152-157)
    st.shO = s.p; st.shD = wl; st.shTmax = length(toL) + 1.0f;
    if (!terminate) { st.pathO = nOrg; st.pathD = wi; }
    st.cls = (st.cls & 4u) | ((!terminate && ray_is_short(sc, nOrg, wi, 3.0e38f)) ? 1u : 0u) | (ray_is_short(sc, st.shO, st.shD, st.shTmax) ? 2u : 0u);
    return terminate;
}

// Returns true when the stream has added its last sample to the pixel.  On return st.flags
// describes the rays to trace next and the ray fields hold them.
template <bool TWO>
PT_DEV bool shade_step_t(const DevScene& sc, const DevCamera& cam, const DevParams& prm, const WfBuf& b, uint32_t sid, SState& st,
                         float2 hitP, float2 hitS, float2 hitA)
{
    const float4* __restrict__ pixPtr = &b.pix[sid];
    const float4* __restrict__ dir0Ptr = &b.dir0[sid];
    float2* __restrict__ hit0Ptr = &b.hit0[sid];
    const uint32_t flags = st.flags;
    bool bRefracted = (flags & F_REFR) != 0;
    const int primS = (flags & F_SHADOW) ? __float_as_int(hitS.y) : -1;
    const int primA = (flags & F_SHADOWA) ? __float_as_int(hitA.y) : -1;
    // scene fetches that do not depend on anything computed below, issued together
    const float4 emS = tri_emit4(sc, (primS >= 0 && primS < sc.n_tris) ? primS : 0);
    const float4 emA = tri_emit4(sc, (primA >= 0 && primA < sc.n_tris) ? primA : 0);
    float2 h0 = hitP;
    if (!(flags & F_PRIMARY)) h0 = *hit0Ptr;

    auto add_to_pixel = [&](const f3& r) {                                  // pathtracer.cu:79
        if (!st.pixLoaded) { const float4 pq = ld_s(pixPtr); st.pixelColor = f3(pq.x, pq.y, pq.z); st.pixLoaded = true; }
        st.pixelColor += r;
    };
    // ---- a. the older closed sample: its last NEE term, then it joins the pixel ----
    if (flags & F_SHADOWA) {
        const float4 ra = ld_s(&b.radA[sid]), wa = ld_s(&b.wbA[sid]), la = ld_s(&b.lpA[sid]), ao = ld_s(&b.ray_o[2][sid]), ad = ld_s(&b.ray_d[2][sid]);
        f3 radA(ra.x, ra.y, ra.z);
        const f3 Le = nee_light_color(f3(ao.x, ao.y, ao.z), f3(ad.x, ad.y, ad.z), f3(la.x, la.y, la.z), hitA.x, primA,
                                      (primA < sc.n_tris) ? f3(emA.x, emA.y, emA.z) : prim_emittance(sc, primA < 0 ? 0 : primA));
        if (flags & F_NEEOKA) radA += ((f3(wa.x, wa.y, wa.z) * Le) * wa.w) / ra.w;      // GetLightColor tail + CudaUtil.cuh:271-272
        add_to_pixel(radA);
    }
    // ---- b. pending NEE term of the current sample ----
    if (flags & F_SHADOW) {
        const f3 Le = nee_light_color(st.shO, st.shD, st.lightP, hitS.x, primS,
                                      (primS < sc.n_tris) ? f3(emS.x, emS.y, emS.z) : prim_emittance(sc, primS < 0 ? 0 : primS));
        if (flags & F_NEEOK) st.radiance += ((st.wb * Le) * st.cosA) / st.denom;
    }
    bool cur = (flags & F_CUR) != 0;       // a current sample exists
    bool closing = false;                  // it has just shaded its last bounce (its NEE term is in the current slot)
    bool shCur = false, neeCur = false, pathCur = false, shA = false, neeA = false;
    if (flags & F_PRIMARY) *hit0Ptr = hitP;                                  // the camera ray's hit, shared by every sample
    else if (!(flags & F_PATH) && cur) { add_to_pixel(st.radiance); cur = false; }   // the current sample closed one step ago

    if constexpr (TWO) {
    // ---- c. up to two hits to shade: round 0 the current path's, round 1 the first hit of the next sample ----
#pragma unroll
    for (int round = 0; round < 2; round++) {
        bool go;
        int prim = -1; float t = 0.f;
        f3 rorg(0.f, 0.f, 0.f), rdir(0.f, 0.f, 1.f);
        if (round == 0) {
            go = (flags & F_PATH) && !(flags & F_PRIMARY);
            prim = __float_as_int(hitP.y); t = hitP.x; rorg = st.pathO; rdir = st.pathD;
            if (go && prim < 0) {
                st.radiance += st.weight * f3(0.1f, 0.1f, 0.1f);               // CudaUtil.cuh:375-379: the path left the scene
                add_to_pixel(st.radiance);
                cur = false; go = false;
            }
        } else {
            go = (!cur || closing) && st.toStart > 0;
            if (go) {
                if (closing) {
                    // the closed sample waits in slot A for its shadow ray; the pixel gets it first thing next step
                    st_s(&b.radA[sid], make_float4(st.radiance.x, st.radiance.y, st.radiance.z, st.denom));
                    st_s(&b.wbA[sid], make_float4(st.wb.x, st.wb.y, st.wb.z, st.cosA));
                    st_s(&b.lpA[sid], make_float4(st.lightP.x, st.lightP.y, st.lightP.z, 0.f));
                    st_s(&b.ray_o[2][sid], make_float4(st.shO.x, st.shO.y, st.shO.z, st.shTmax));
                    st_s(&b.ray_d[2][sid], make_float4(st.shD.x, st.shD.y, st.shD.z, shadow_stop_t(st.shO, st.shTmax)));
                    shA = true; neeA = neeCur; shCur = false; neeCur = false; closing = false;
                    st.cls = (st.cls & ~4u) | ((st.cls & 2u) << 1);
                }
                prim = __float_as_int(h0.y); t = h0.x;
                const float4 d0 = ld_s(dir0Ptr);
                rorg = f3(cam.pos[0], cam.pos[1], cam.pos[2]); rdir = f3(d0.x, d0.y, d0.z);
                if (prim < 0) {
                    // the pixel looks past the scene: every remaining sample is the ambient term (no draws, no rays)
                    do { st.radiance = f3(0.f, 0.f, 0.f); st.radiance += f3(1.f, 1.f, 1.f) * f3(0.1f, 0.1f, 0.1f); add_to_pixel(st.radiance); } while (--st.toStart > 0);
                    cur = false; go = false;
                } else {
                    st.toStart--;                                            // pathtracer.cu:77-78: next sample
                    st.weight = f3(1.f, 1.f, 1.f); st.radiance = f3(0.f, 0.f, 0.f);
                    st.depth = 0; st.refractCnt = 0; bRefracted = false; cur = true;
                }
            }
        }
        if (go) {
            bool needSh;
            const bool terminate = bounce(sc, prm, prim, t, rorg, rdir, st, bRefracted, neeCur, needSh);
            shCur = needSh;
            if (!terminate) { pathCur = true; closing = false; }
            else {
                pathCur = false;
                // A path that ends with no NEE term pending has nothing to wait for.  In round 0 it joins the pixel at once (the older
                // closed sample was added at the top of this step, so the order of additions is kept) and the next sample can start;
                // in round 1 an older sample may still sit in slot A, so the new one waits as a current sample without rays and
                // is added by the next step's "closed one step ago" branch.
                if (needSh || round == 1) closing = true;
                else { add_to_pixel(st.radiance); cur = false; closing = false; }
            }
        }
    }
    } else {
    // ---- c'. ONE hit to shade per step (TWO = false): the lane continues its current path if it has a path hit, or else
    // starts the next sample at the pixel's cached camera hit — the same bounce code for both, so a wave runs it once.  A sample
    // then takes `bounces` steps instead of `bounces - 1`, but every step's dependent chain is one bounce long instead of two.
    // The operations a stream goes through, and their order, are those of the two-round step; only the step they happen in
    // differs, so both schedules (and any mix of them) give the same bits. ----
    {
        bool go = false, fresh = false;
        int prim = -1; float t = 0.f;
        f3 rorg(0.f, 0.f, 0.f), rdir(0.f, 0.f, 1.f);
        if ((flags & F_PATH) && !(flags & F_PRIMARY)) {
            prim = __float_as_int(hitP.y); t = hitP.x; rorg = st.pathO; rdir = st.pathD;
            if (prim < 0) {
                st.radiance += st.weight * f3(0.1f, 0.1f, 0.1f);               // CudaUtil.cuh:375-379: the path left the scene
                add_to_pixel(st.radiance);
                cur = false;
            } else go = true;
        }
        if (!go && (!cur || closing) && st.toStart > 0) {
            if (closing) {
                // the closed sample waits in slot A for its shadow ray; the pixel gets it first thing next step
                st_s(&b.radA[sid], make_float4(st.radiance.x, st.radiance.y, st.radiance.z, st.denom));
                st_s(&b.wbA[sid], make_float4(st.wb.x, st.wb.y, st.wb.z, st.cosA));
                st_s(&b.lpA[sid], make_float4(st.lightP.x, st.lightP.y, st.lightP.z, 0.f));
                st_s(&b.ray_o[2][sid], make_float4(st.shO.x, st.shO.y, st.shO.z, st.shTmax));
                st_s(&b.ray_d[2][sid], make_float4(st.shD.x, st.shD.y, st.shD.z, shadow_stop_t(st.shO, st.shTmax)));
                shA = true; neeA = neeCur; shCur = false; neeCur = false; closing = false;
                st.cls = (st.cls & ~4u) | ((st.cls & 2u) << 1);
            }
            prim = __float_as_int(h0.y); t = h0.x;
            const float4 d0 = ld_s(dir0Ptr);
            rorg = f3(cam.pos[0], cam.pos[1], cam.pos[2]); rdir = f3(d0.x, d0.y, d0.z);
            if (prim < 0) {
                // the pixel looks past the scene: every remaining sample is the ambient term (no draws, no rays)
                do { st.radiance = f3(0.f, 0.f, 0.f); st.radiance += f3(1.f, 1.f, 1.f) * f3(0.1f, 0.1f, 0.1f); add_to_pixel(st.radiance); } while (--st.toStart > 0);
                cur = false;
            } else {
                st.toStart--;                                            // pathtracer.cu:77-78: next sample
                st.weight = f3(1.f, 1.f, 1.f); st.radiance = f3(0.f, 0.f, 0.f);
                st.depth = 0; st.refractCnt = 0; bRefracted = false; cur = true;
                go = true; fresh = true;
            }
        }
        if (go) {
            bool needSh;
            const bool terminate = bounce(sc, prm, prim, t, rorg, rdir, st, bRefracted, neeCur, needSh);
            shCur = needSh;
            if (!terminate) { pathCur = true; closing = false; }
            else {
                pathCur = false;
                // A path that ends with no NEE term pending has nothing to wait for.  If it is the path this step continued, it joins
                // the pixel at once (the older closed sample was added at the top of this step, so the order of additions is kept) and
                // the next sample can start; a sample started in this step may still have an older one sitting in slot A, so it waits
                // as a current sample without rays and is added by the next step's "closed one step ago" branch.
                if (needSh || fresh) closing = true;
                else { add_to_pixel(st.radiance); cur = false; closing = false; }
            }
        }
    }
    }
    st.flags = (cur ? F_CUR : 0u) | (shCur ? F_SHADOW : 0u) | (neeCur ? F_NEEOK : 0u) | (pathCur ? F_PATH : 0u) |
               (shA ? F_SHADOWA : 0u) | (neeA ? F_NEEOKA : 0u) | (bRefracted ? F_REFR : 0u);
    return !cur && !shA && st.toStart == 0;
}

PT_DEV bool shade_step(const DevScene& sc, const DevCamera& cam, const DevParams& prm, const WfBuf& b, uint32_t sid, SState& st,
                       float2 hitP, float2 hitS, float2 hitA)
{
    return shade_step_t<true>(sc, cam, prm, b, sid, st, hitP, hitS, hitA);
}

PT_DEV void load_state(const WfBuf& b, uint32_t sid, SState& st)
{
    const uint4 r0 = ld_s(&b.rng0[sid]), r1 = ld_s(&b.rng1[sid]);
    st.rng.x0 = r0.x; st.rng.x1 = r0.y; st.rng.x2 = r0.z; st.rng.x3 = r0.w; st.rng.x4 = r1.x; st.rng.d = r1.y;
    st.toStart = (int)(r1.z >> 16); st.depth = (int)((r1.z >> 8) & 0xff); st.refractCnt = (int)(r1.z & 0xff);
    st.flags = r1.w;
    const float4 wq = ld_s(&b.weight[sid]), rq4 = ld_s(&b.rad[sid]);
    st.weight = f3(wq.x, wq.y, wq.z); st.radiance = f3(rq4.x, rq4.y, rq4.z);
    st.cosA = wq.w; st.denom = rq4.w;
    st.pixelColor = f3(0.f, 0.f, 0.f); st.pixLoaded = false; st.cls = 0u;
    // the current sample's rays and pending NEE term are fetched whatever the flags say (stale values are
    // never used): nearly every step has them, and waiting for the flags first only adds latency
    const float4 po = ld_s(&b.ray_o[0][sid]), pd = ld_s(&b.ray_d[0][sid]);
    st.pathO = f3(po.x, po.y, po.z); st.pathD = f3(pd.x, pd.y, pd.z);
    const float4 so = ld_s(&b.ray_o[1][sid]), sd = ld_s(&b.ray_d[1][sid]), lpq = ld_s(&b.lp[sid]), wbq = ld_s(&b.wb[sid]);
    st.shO = f3(so.x, so.y, so.z); st.shD = f3(sd.x, sd.y, sd.z); st.shTmax = so.w;
    st.lightP = f3(lpq.x, lpq.y, lpq.z); st.wb = f3(wbq.x, wbq.y, wbq.z);
}

PT_DEV void write_mean(const WfBuf& b, const DevParams& prm, uint32_t sid, const SState& st)
{
    const f3 mean = st.pixelColor / (float)prm.spp_per_pass;          // pathtracer.cu:81
    b.staging[3 * (size_t)sid + 0] = mean.x; b.staging[3 * (size_t)sid + 1] = mean.y; b.staging[3 * (size_t)sid + 2] = mean.z;
}

// state + the rays the next trace must serve, back to the slot
PT_DEV void store_state(const WfBuf& b, uint32_t slot, const SState& st)
{
    const uint32_t nf = st.flags;
    st_s(&b.rng0[slot], make_uint4(st.rng.x0, st.rng.x1, st.rng.x2, st.rng.x3));
    st_s(&b.rng1[slot], make_uint4(st.rng.x4, st.rng.d, ((uint32_t)st.toStart << 16) | ((uint32_t)st.depth << 8) | (uint32_t)st.refractCnt, nf));
    st_s(&b.weight[slot], make_float4(st.weight.x, st.weight.y, st.weight.z, st.cosA));
    st_s(&b.rad[slot], make_float4(st.radiance.x, st.radiance.y, st.radiance.z, st.denom));
    if (st.pixLoaded) st_s(&b.pix[slot], make_float4(st.pixelColor.x, st.pixelColor.y, st.pixelColor.z, 0.f));
    if (nf & F_SHADOW) {
        st_s(&b.ray_o[1][slot], make_float4(st.shO.x, st.shO.y, st.shO.z, st.shTmax));
        st_s(&b.ray_d[1][slot], make_float4(st.shD.x, st.shD.y, st.shD.z, shadow_stop_t(st.shO, st.shTmax)));
        st_s(&b.wb[slot], make_float4(st.wb.x, st.wb.y, st.wb.z, 0.f));
        st_s(&b.lp[slot], make_float4(st.lightP.x, st.lightP.y, st.lightP.z, 0.f));
    }
    if (nf & F_PATH) {
        st_s(&b.ray_o[0][slot], make_float4(st.pathO.x, st.pathO.y, st.pathO.z, 999999.f));
        st_s(&b.ray_d[0][slot], make_float4(st.pathD.x, st.pathD.y, st.pathD.z, -__builtin_inff()));
    }
}

}  // namespace ptd
