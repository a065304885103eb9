// pt_stream.h — a stream (one pixel of one pass) as data: its HBM-resident state, the one-bounce
// step function, and the prologue that starts it.  Shared by the queue-driven pipeline
// (pt_wavefront.hip) and the persistent workgroup-local pipeline (pt_persist.hip).
#pragma once
#include "pt_device.h"
#include "pt_math.h"
#include "pt_bxdf.h"
#include "pt_shade.h"

namespace ptd {

enum : uint32_t {
    F_REFR = 1,      // bRefracted (loop-carried, Q8)
    F_NEEOK = 2,     // !isnan(brdfcos) of the pending NEE term
    F_SHADOW = 4,    // a shadow ray was traced for this stream: NEE term pending
    F_PATH = 8,      // a path ray was traced for this stream
    F_NEWPATH = 16,  // that path ray is the NEXT sample's camera ray: retire the old path first
};

struct WfCounters {      // one slot per iteration parity (3 rotating slots); every hot word on its own 128-B line
    uint32_t nActive, padA[31];
    uint32_t nPath, padB[31];
    uint32_t nShadow, padC[31];
    uint32_t nSusp, padD[31];
    struct { uint32_t v, pad[31]; } head[16];     // sharded ray-queue heads, one 128-B line each
};
constexpr int kWfShards = 16;
constexpr int kWfSlotBytes = 2560;
static_assert(sizeof(WfCounters) == kWfSlotBytes, "counter slot layout");

struct WfBuf {
    uint4* rng0;         // x0 x1 x2 x3
    uint4* rng1;         // x4 d | samplesLeft<<16 | depth<<8 | refractCnt | flags
    float4* weight;      // weight.xyz | cosA
    float4* rad;         // radiance.xyz | denom
    float4* pix;         // pixelColor.xyz
    float4* dir0;        // camera ray direction of this pixel & pass
    float4* wb;          // weight*brdfcos of the pending NEE term
    float4* lp;          // sampled light point of the pending NEE term
    float4* ray_o[2];    // [0] path, [1] shadow: org.xyz | tmax
    float4* ray_d[2];    // dir.xyz
    float2* hit[2];      // t | primitive index (int bits)
    uint32_t* active[2]; // live stream ids, ping-pong
    uint32_t* rq[2];     // ray queues (stream ids): [0] path, [1] shadow
    WfCounters* cnt;     // [3]
    float* staging;      // per-pass means, [stream][3]
    int* ovf;            // traversal stack overflow (entries >= kWfLdsStack), [level][thread]
    int* susp[2];        // suspended traversals (ping-pong by iteration): [record][kSuspInts]
    uint32_t suspCap;    // records per pool
};

// StartRender prologue for one pixel & pass (srcs/pathtracer.cu:70-74): seeds the RNG, draws the
// jittered camera direction and writes the initial state of slot `slot`.
PT_DEV void init_stream(const DevCamera& cam, const DevParams& prm, const WfBuf& b, uint32_t slot, int px, int py, int pass)
{
    const f3 camF(cam.forward[0], cam.forward[1], cam.forward[2]);
    const f3 camU(cam.up[0], cam.up[1], cam.up[2]);
    const f3 camR(cam.right[0], cam.right[1], cam.right[2]);
    const int offset = py * cam.W + px;
    Rng rng;
    rng.init((uint64_t)(int64_t)(offset + pass * cam.W * cam.H));
    const float u1 = rng.uniform();
    const float u2 = rng.uniform();
    const f3 offR = ((2.f * (((float)px + u1) / (float)(cam.W - 1) - 0.5f)) * cam.tan_half_fovx) * camR;
    const f3 offU = ((-2.f * (((float)py + u2) / (float)(cam.H - 1) - 0.5f)) * cam.tan_half_fovy) * camU;
    const f3 direction = normalize(camF + offR + offU);      // GetPixelDirection, pathtracer.cu:33-40
    const f3 d0 = normalize(direction);                      // Ray ctor normalises again, CudaRay.cuh:12
    b.rng0[slot] = make_uint4(rng.x0, rng.x1, rng.x2, rng.x3);
    b.rng1[slot] = make_uint4(rng.x4, rng.d, ((uint32_t)prm.spp_per_pass << 16), F_PATH);
    b.weight[slot] = make_float4(1.f, 1.f, 1.f, 0.f);
    b.rad[slot] = make_float4(0.f, 0.f, 0.f, 1.f);
    b.pix[slot] = make_float4(0.f, 0.f, 0.f, 0.f);
    b.dir0[slot] = make_float4(d0.x, d0.y, d0.z, 0.f);
    b.ray_o[0][slot] = make_float4(cam.pos[0], cam.pos[1], cam.pos[2], 999999.f);
    b.ray_d[0][slot] = make_float4(d0.x, d0.y, d0.z, 0.f);
    b.hit[0][slot] = make_float2(0.f, __int_as_float(-1));
    b.hit[1][slot] = make_float2(0.f, __int_as_float(-1));
}

// ---------------------------------------------------------------------------------------
// One bounce of one stream — the body of GetColor_iter's loop (include/CudaUtil.cuh:216-380)
// plus StartRender's sample-loop bookkeeping (srcs/pathtracer.cu:77-81) — on a register-resident
// stream state.  Shared by wf_shade (state in HBM, one bounce per launch) and wf_drain (state in
// registers, runs a stream to its end).
// ---------------------------------------------------------------------------------------
struct SState {
    Rng rng;
    int samplesLeft, depth, refractCnt;
    uint32_t flags;                 // F_* of the rays that were traced for this bounce
    f3 weight, radiance;
    f3 pixelColor; bool pixLoaded;  // loaded lazily: only a retiring path touches it
    float cosA, denom;              // pending NEE term ...
    f3 wb, lightP;                  // ... weight*brdfcos, sampled light point
    f3 pathO, pathD;                // path ray (traced if F_PATH)
    f3 shO, shD; float shTmax;      // shadow ray (traced if F_SHADOW)
};

// Returns true when the stream has finished its last sample.  On return st.flags describes the
// rays to trace next (F_PATH / F_SHADOW / F_NEWPATH) and the ray fields hold them.
PT_DEV bool shade_step(const DevScene& sc, const DevCamera& cam, const DevParams& prm, SState& st,
                       float2 hitP, float2 hitS, const float4* __restrict__ pixPtr, const float4* __restrict__ dir0Ptr)
{
    const uint32_t flags = st.flags;
    bool bRefracted = (flags & F_REFR) != 0;
    const int Nl = sc.n_lights;
    // ---- 0. the scene fetches of the bounce, issued together ----
    // The path hit's surface record (one 192-B record: no index chasing), the shadow hit's emittance and
    // the light the NEE draw will pick (its random word is peeked) are requested up front with clamped
    // indices instead of branches, so the compiler can keep them all in flight.  The two per-pixel
    // values a retiring path needs stay lazy: ~40 % of the steps use them, and the kernel is limited by
    // memory throughput (its time does not change between 2, 3 and 4 waves/SIMD), not by fetch depth.
    const int primP = (flags & F_PATH) ? __float_as_int(hitP.y) : -1;
    const int primS = (flags & F_SHADOW) ? __float_as_int(hitS.y) : -1;
    const bool triP = primP >= 0 && primP < sc.n_tris;
    SurfRec rec;
    load_surf(sc, triP ? primP : 0, rec);
    const float4 emS = tri_emit4(sc, (primS >= 0 && primS < sc.n_tris) ? primS : 0);
    Rng peek = st.rng;
    const int li = (int)(peek.next() % (uint32_t)Nl);
    const float4 l0 = sc.lights[4 * li], l1 = sc.lights[4 * li + 1], l2 = sc.lights[4 * li + 2], l3 = sc.lights[4 * li + 3];
    // ---- 1. pending NEE term (GetLightColor tail + CudaUtil.cuh:271-272) ----
    if (flags & F_SHADOW) {
        f3 Le(0.f, 0.f, 0.f);
        if (primS >= 0) {
            const f3 hp = st.shO + hitS.x * st.shD;
            if (length(hp - st.lightP) < kEps) Le = (primS < sc.n_tris) ? f3(emS.x, emS.y, emS.z) : prim_emittance(sc, primS);
        }
        if (flags & F_NEEOK) st.radiance += ((st.wb * Le) * st.cosA) / st.denom;
    }
    // ---- 2. the traced path ray belongs to the next sample: retire the old path first ----
    bool streamDone = false;
    auto retire = [&]() {                                        // pathtracer.cu:79
        if (!st.pixLoaded) { const float4 pix0 = *pixPtr; st.pixelColor = f3(pix0.x, pix0.y, pix0.z); st.pixLoaded = true; }
        st.pixelColor += st.radiance;
        st.samplesLeft--;
        st.weight = f3(1.f, 1.f, 1.f); st.radiance = f3(0.f, 0.f, 0.f);
        st.depth = 0; st.refractCnt = 0; bRefracted = false;
    };
    if (flags & F_NEWPATH) retire();

    uint32_t nflags = 0;
    if (flags & F_PATH) {
        const f3 rorg = st.pathO, rdir = st.pathD;
        if (primP < 0) {
            st.radiance += st.weight * f3(0.1f, 0.1f, 0.1f);               // CudaUtil.cuh:375-379
            retire();
            if (st.samplesLeft > 0) nflags = F_PATH; else streamDone = true;
        } else {
            // ---- shade a PATH hit: the whole bounce except visibility ----
            Surf s;
            if (triP) surf_from_rec(rec, hitP.x, rorg, rdir, s);
            else surf_sphere(sc, primP - sc.n_tris, hitP.x, rorg, rdir, s);
            if (sqlen(s.m.emittance) > kEps) st.radiance += st.weight * s.m.emittance;   // :220-224
            const float ior = ior_of(s.m);                                          // :231
            const int lobe = lobe_of(s.m);
            const f3 wo = -rdir;
            // NEE sample (:235-245, SamplePrimitive :38-48)
            st.rng = peek;                                  // the light index drawn above: first draw of the bounce
            const f3 LV0(l0.x, l0.y, l0.z), LV1(l0.w, l1.x, l1.y), LV2(l1.z, l1.w, l2.x), LN(l2.y, l2.z, l2.w);
            const float r1u = __builtin_sqrtf(st.rng.uniform());
            const float r2u = st.rng.uniform();
            const f3 lightP = (1.f - r1u) * LV0 + (r1u * (1.f - r2u)) * LV1 + (r1u * r2u) * LV2;
            const float pdfLight = (1.f / l3.x) / ((float)Nl);
            const f3 toL = lightP - s.p;
            const f3 wl = normalize(toL);
            const float ca = dot(LN, normalize(s.p - lightP));
            st.cosA = (ca < 0.f) ? 0.f : ca;
            const f3 brdfcos = lobe_eval(lobe, s.m, ior, s.fr, wo, wl);
            const bool neeOk = !anynan(brdfcos);
            st.wb = st.weight * brdfcos;
            st.lightP = lightP;
            st.denom = sqlen(s.p - lightP) * pdfLight;
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
            nflags = F_SHADOW | (neeOk ? F_NEEOK : 0u);
            if (!terminate) {
                st.pathO = nOrg; st.pathD = wi;
                nflags |= F_PATH;
            } else if (st.samplesLeft > 1) {
                nflags |= F_PATH | F_NEWPATH;      // pre-launch the next sample's camera ray beside the shadow ray
            }
        }
    } else {
        // only a shadow ray was traced: the last path of the stream ended at the previous bounce
        retire();
        streamDone = true;
    }
    if (((nflags & F_PATH) && !(nflags & F_SHADOW)) || (nflags & F_NEWPATH)) {
        const float4 cam0 = *dir0Ptr;
        st.pathO = f3(cam.pos[0], cam.pos[1], cam.pos[2]);
        st.pathD = f3(cam0.x, cam0.y, cam0.z);
    }
    if (bRefracted) nflags |= F_REFR;
    st.flags = nflags;
    return streamDone;
}

PT_DEV void load_state(const WfBuf& b, uint32_t sid, SState& st)
{
    const uint4 r0 = b.rng0[sid], r1 = b.rng1[sid];
    st.rng.x0 = r0.x; st.rng.x1 = r0.y; st.rng.x2 = r0.z; st.rng.x3 = r0.w; st.rng.x4 = r1.x; st.rng.d = r1.y;
    st.samplesLeft = (int)(r1.z >> 16); st.depth = (int)((r1.z >> 8) & 0xff); st.refractCnt = (int)(r1.z & 0xff);
    st.flags = r1.w;
    const float4 wq = b.weight[sid], rq4 = b.rad[sid];
    st.weight = f3(wq.x, wq.y, wq.z); st.radiance = f3(rq4.x, rq4.y, rq4.z);
    st.cosA = wq.w; st.denom = rq4.w;
    st.pixelColor = f3(0.f, 0.f, 0.f); st.pixLoaded = false;
    st.wb = f3(0.f, 0.f, 0.f); st.lightP = f3(0.f, 0.f, 0.f);
    st.pathO = st.pathD = st.shO = st.shD = f3(0.f, 0.f, 0.f); st.shTmax = 0.f;
    // rays and the pending NEE term are fetched whatever the flags say (stale values are never used):
    // waiting for the flags first would add a level to the kernel's dependent-load chain
    const float4 po = b.ray_o[0][sid], pd = b.ray_d[0][sid];
    st.pathO = f3(po.x, po.y, po.z); st.pathD = f3(pd.x, pd.y, pd.z);
    const float4 so = b.ray_o[1][sid], sd = b.ray_d[1][sid], lpq = b.lp[sid], wbq = b.wb[sid];
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
    b.rng0[slot] = make_uint4(st.rng.x0, st.rng.x1, st.rng.x2, st.rng.x3);
    b.rng1[slot] = make_uint4(st.rng.x4, st.rng.d, ((uint32_t)st.samplesLeft << 16) | ((uint32_t)st.depth << 8) | (uint32_t)st.refractCnt, nf);
    b.weight[slot] = make_float4(st.weight.x, st.weight.y, st.weight.z, st.cosA);
    b.rad[slot] = make_float4(st.radiance.x, st.radiance.y, st.radiance.z, st.denom);
    if (st.pixLoaded) b.pix[slot] = make_float4(st.pixelColor.x, st.pixelColor.y, st.pixelColor.z, 0.f);
    if (nf & F_SHADOW) {
        b.ray_o[1][slot] = make_float4(st.shO.x, st.shO.y, st.shO.z, st.shTmax);
        b.ray_d[1][slot] = make_float4(st.shD.x, st.shD.y, st.shD.z, 0.f);
        b.wb[slot] = make_float4(st.wb.x, st.wb.y, st.wb.z, 0.f);
        b.lp[slot] = make_float4(st.lightP.x, st.lightP.y, st.lightP.z, 0.f);
    }
    if (nf & F_PATH) {
        b.ray_o[0][slot] = make_float4(st.pathO.x, st.pathO.y, st.pathO.z, 999999.f);
        b.ray_d[0][slot] = make_float4(st.pathD.x, st.pathD.y, st.pathD.z, 0.f);
    }
}

}  // namespace ptd
