// pt_persist.hip — the integrator as ONE persistent launch of workgroup-local pipelines (mode 2).
//
// The queue-driven pipeline (pt_wavefront.hip, mode 1) advances all streams one bounce per
// launch pair; every launch ends in a device-wide drain (the last rays in flight: ~280 us on
// MI355X, 14 % of a 16M-stream iteration but half of a 2M-stream one), and the stream state
// makes a round trip through HBM each bounce.  Here a workgroup of 8 waves OWNS 2048 stream
// slots and runs the same two phases on them by itself, separated by workgroup barriers only:
//
//   refill  free slots take new (tile, pass) units from one global counter — a slot is reused
//           as soon as its stream retires, so a workgroup always works on ~2048 live streams
//           and the slow pixels of a frame never leave the rest of the GPU idle;
//   trace   the 8 waves drain the workgroup's ray list (LDS) with lane refill and the
//           node/triangle vote of wf_trace;
//   shade   one bounce of every live slot (shade_step), retire finished streams, push rays.
//
// Workgroups never talk to each other, so there is nothing to deadlock on and no device-wide
// dependency; a workgroup's drain is covered by the other workgroup resident on its CU.  Slot
// state (~224 B x 2048 x 512 workgroups = 235 MB) is indexed by slot, not by stream, so it
// stays Infinity-Cache resident however many passes are rendered.
//
// Per-stream arithmetic is that of the other modes (same shade_step, same traversal step
// functions): frames are bit-identical (tests run the whole GPU suite under PTAMD_MODE=2 too).
//
// STATUS: experimental, not the default.  Measured on MI355X (config 3): 535 Msamples/s on 8 passes
// and 403 on one pass, against 930 / 610 for mode 1.  Fusing both phases into one kernel costs the
// traversal loop its occupancy (128 VGPRs, 360 B/lane of spills -> 4 waves/SIMD instead of 8), and a
// workgroup's 8 waves still wait for its slowest ray every phase.  Kept as a third, independently
// scheduled implementation for parity cross-checks and as the starting point for wave-specialised
// variants (trace waves and shade waves with separate register budgets).
#include <hip/hip_runtime.h>
#include <cstdlib>
#include "pt_device.h"
#include "pt_math.h"
#include "pt_bxdf.h"
#include "pt_trace.h"
#include "pt_shade.h"
#include "pt_stream.h"

namespace ptd {

constexpr int kPwThreads = 512;
constexpr int kPwWaves = kPwThreads / 64;
constexpr int kPwSlots = 2048;               // stream slots per workgroup (4 per thread); 4096 measured no better
constexpr int kPwLdsStack = 16;
constexpr int kPwOvfLevels = 32;
constexpr int kPwDone = (int)0x80000000;

__global__ __launch_bounds__(kPwThreads, 4)
void wf_persist(DevScene sc, DevCamera cam, DevParams prm, WfBuf b, uint32_t* __restrict__ sidOf,
                unsigned int* __restrict__ unitCounter, int* __restrict__ ovfBase, int ovfStride)
{
    __shared__ int lds_stack[kPwWaves][kPwLdsStack * 64];
    __shared__ unsigned short rayq[2 * kPwSlots];      // slot | (kind << 15)
    __shared__ unsigned short freeList[kPwSlots];
    __shared__ unsigned short retList[kPwSlots];
    __shared__ unsigned char live[kPwSlots];
    __shared__ unsigned int s_nRays, s_qHead, s_nFree, s_nRet, s_unitStart, s_unitCount, s_more;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t gbase = (uint32_t)blockIdx.x * kPwSlots;
    int* stack = &lds_stack[wave][lane];
    int* ovf = ovfBase + ((size_t)blockIdx.x * kPwThreads + tid);

    for (int s = tid; s < kPwSlots; s += kPwThreads) { live[s] = 0; freeList[s] = (unsigned short)s; }
    if (tid == 0) { s_nRays = 0; s_qHead = 0; s_nFree = kPwSlots; s_nRet = 0; s_more = 1; }
    __syncthreads();

    for (;;) {
        // ---------------- refill: free slots take new (tile, pass) units ----------------
        if (tid == 0) {
            const unsigned int want = s_nFree / 64u;
            unsigned int start = 0, got = 0;
            if (want > 0 && s_more) {
                start = atomicAdd(unitCounter, want);
                if (start >= (unsigned)prm.n_units) { s_more = 0; }
                else { got = ((unsigned)prm.n_units - start < want) ? (unsigned)prm.n_units - start : want; if (start + want >= (unsigned)prm.n_units) s_more = 0; }
            }
            s_unitStart = start; s_unitCount = got;
        }
        __syncthreads();
        const unsigned int nNew = s_unitCount * 64u, nFree0 = s_nFree, unit0 = s_unitStart;
        for (unsigned int i = tid; i < nNew; i += kPwThreads) {
            const unsigned int slot = freeList[nFree0 - 1u - i];
            const uint32_t unit = (uint32_t)prm.unit_base + unit0 + (i >> 6), l = i & 63u;
            const int pass_rel = (int)(unit / (uint32_t)prm.n_tiles_local);
            const int lt = (int)(unit % (uint32_t)prm.n_tiles_local);
            const int tile = lt * prm.world + prm.rank;
            const int tx = tile % prm.tiles_x, ty = tile / prm.tiles_x;
            const int px = tx * kTile + (int)(l & 7), py = ty * kTile + (int)(l >> 3);
            const uint32_t sid = unit * 64u + l;                      // staging index of this stream
            if ((tile < prm.n_tiles_total) && (px < cam.W) && (py < cam.H)) {
                init_stream(cam, prm, b, gbase + slot, px, py, prm.first_pass + pass_rel);
                sidOf[gbase + slot] = sid;
                live[slot] = 1;
                rayq[atomicAdd(&s_nRays, 1u)] = (unsigned short)slot;
            } else {
                b.staging[3 * (size_t)sid + 0] = 0.f; b.staging[3 * (size_t)sid + 1] = 0.f; b.staging[3 * (size_t)sid + 2] = 0.f;
                retList[atomicAdd(&s_nRet, 1u)] = (unsigned short)slot;     // pixel outside the frame: the slot stays free
            }
        }
        __syncthreads();
        {
            const unsigned int nRet = s_nRet, keep = nFree0 - nNew;
            for (unsigned int j = tid; j < nRet; j += kPwThreads) freeList[keep + j] = retList[j];
            __syncthreads();
            if (tid == 0) { s_nFree = keep + nRet; s_nRet = 0; }
        }
        __syncthreads();
        const unsigned int nR = s_nRays;
        if (nR == 0) break;          // no live stream and no unit left for this workgroup (uniform)

        // ---------------- trace: the workgroup's rays, lanes refill from the LDS list ----------------
        {
            unsigned int chunkPos = 0, chunkEnd = 0;
            bool exhausted = false, hasRay = false, shadow = false, degenerate = false;
            uint32_t gs = 0;
            f3 org(0.f, 0.f, 0.f), dir(0.f, 0.f, 1.f), invD(0.f, 0.f, 0.f);
            float bestT = 0.f, cullB = 0.f, kcull = 0.f, stopBelow = 0.f;
            int bestPrim = -1, cur = kPwDone, sp = 0;
            for (;;) {
                const unsigned long long idle = __ballot(!hasRay);
                const int nIdle = __builtin_popcountll(idle);
                if (!exhausted && nIdle >= 16) {
                    if (chunkPos == chunkEnd) {
                        unsigned int start = 0;
                        if (lane == 0) start = atomicAdd(&s_qHead, 64u);
                        start = __builtin_amdgcn_readfirstlane(start);
                        if (start >= nR) exhausted = true;
                        else { chunkPos = start; chunkEnd = (start + 64u < nR) ? start + 64u : nR; }
                    }
                    if (!exhausted) {
                        const unsigned int avail = chunkEnd - chunkPos;
                        const unsigned int take = ((unsigned)nIdle < avail) ? (unsigned)nIdle : avail;
                        if (!hasRay) {
                            const unsigned int r = (unsigned)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
                            if (r < take) {
                                const unsigned int e = rayq[chunkPos + r];
                                shadow = (e & 0x8000u) != 0;
                                gs = gbase + (e & 0x7fffu);
                                const float4 o = (shadow ? b.ray_o[1] : b.ray_o[0])[gs], d = (shadow ? b.ray_d[1] : b.ray_d[0])[gs];
                                org = f3(o.x, o.y, o.z); dir = f3(d.x, d.y, d.z);
                                const f3 inv(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);                 // inv(), CudaUtil.cuh:60-63
                                const float L = __builtin_sqrtf(inv.x * inv.x + inv.y * inv.y + inv.z * inv.z);
                                invD = inv / L;                                                      // Normalize(inv(dir)), :70
                                degenerate = !(L < __builtin_inff());
                                kcull = degenerate ? 1.0078125f : 1.0078125f / L;
                                stopBelow = shadow ? (o.w - 1.0f) - 5e-4f : -__builtin_inff();      // see wf_trace
                                bestT = o.w; bestPrim = -1; cur = 0; sp = 0;
                                cullB = bestT * kcull;
                                hasRay = true;
                            }
                        }
                        chunkPos += take;
                    }
                }
                if (__ballot(hasRay) == 0ull) { if (exhausted) break; else continue; }
                if (hasRay) {
                    const int nNode = __builtin_popcountll(__ballot(cur >= 0));
                    const int nTri = __builtin_popcountll(__ballot(hasRay && cur < 0 && cur != kPwDone));
                    const bool doNode = nNode >= nTri;
                    if (doNode && cur >= 0) {
                        const float4 q0 = sc.nodes[4 * cur + 0];
                        const float4 q1 = sc.nodes[4 * cur + 1];
                        const float4 q2 = sc.nodes[4 * cur + 2];
                        const float4 q3 = sc.nodes[4 * cur + 3];
                        float tnL, tnR;
                        bool okL, okR;
                        if (!degenerate) {
                            okL = box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, org, invD, cullB, tnL);
                            okR = box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, org, invD, cullB, tnR);
                        } else {
                            const f3 inv(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);
                            okL = box_test_robust(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, org, dir, inv, cullB, tnL);
                            okR = box_test_robust(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, org, dir, inv, cullB, tnR);
                        }
                        const int refL = __float_as_int(q3.x), refR = __float_as_int(q3.y);
                        if (okL & okR) {
                            const bool lNear = tnL <= tnR;
                            const int farRef = lNear ? refR : refL;
                            if (sp < kPwLdsStack) stack[sp * 64] = farRef; else ovf[(size_t)(sp - kPwLdsStack) * ovfStride] = farRef;
                            sp++;
                            cur = lNear ? refL : refR;
                        } else if (okL) {
                            cur = refL;
                        } else if (okR) {
                            cur = refR;
                        } else if (sp == 0) {
                            cur = kPwDone;
                        } else {
                            sp--;
                            cur = (sp < kPwLdsStack) ? stack[sp * 64] : ovf[(size_t)(sp - kPwLdsStack) * ovfStride];
                        }
                    } else if (!doNode && cur < 0 && cur != kPwDone) {
                        const int code = ~cur, first = code >> 3, cnt = code & 7;
                        bool pop = true;
                        if (cnt > 0) {
                            tri_test(sc, first, org, dir, invD, degenerate, bestT, bestPrim);
                            cullB = bestT * kcull;
                            if (bestPrim >= 0 && bestT < stopBelow) { cur = kPwDone; pop = false; }
                            else if (cnt > 1) { cur = ~(((first + 1) << 3) | (cnt - 1)); pop = false; }
                        }
                        if (pop) {
                            if (sp == 0) cur = kPwDone;
                            else { sp--; cur = (sp < kPwLdsStack) ? stack[sp * 64] : ovf[(size_t)(sp - kPwLdsStack) * ovfStride]; }
                        }
                    }
                    if (cur == kPwDone) {
                        for (int s = 0; s < sc.n_spheres; s++) {            // spheres, in order (CudaUtil.cuh:137-145)
                            const float4 c = sc.spheres[4 * s];
                            float root;
                            if (sphere_root(f3(c.x, c.y, c.z), c.w, org, dir, bestT, root)) { bestT = root; bestPrim = sc.n_tris + s; }
                        }
                        (shadow ? b.hit[1] : b.hit[0])[gs] = make_float2(bestT, __int_as_float(bestPrim));
                        hasRay = false;
                    }
                }
            }
        }
        __syncthreads();
        if (tid == 0) { s_nRays = 0; s_qHead = 0; }
        __syncthreads();

        // ---------------- shade: one bounce of every live slot ----------------
        for (int k = 0; k < kPwSlots / kPwThreads; k++) {
            const int slot = tid + k * kPwThreads;
            if (!live[slot]) continue;
            const uint32_t g = gbase + (uint32_t)slot;
            SState st;
            load_state(b, g, st);
            const float2 hitP = (st.flags & F_PATH) ? b.hit[0][g] : make_float2(0.f, __int_as_float(-1));
            const float2 hitS = (st.flags & F_SHADOW) ? b.hit[1][g] : make_float2(0.f, __int_as_float(-1));
            if (shade_step(sc, cam, prm, st, hitP, hitS, &b.pix[g], &b.dir0[g])) {
                const f3 mean = st.pixelColor / (float)prm.spp_per_pass;          // pathtracer.cu:81
                const size_t sid = sidOf[g];
                b.staging[3 * sid + 0] = mean.x; b.staging[3 * sid + 1] = mean.y; b.staging[3 * sid + 2] = mean.z;
                live[slot] = 0;
                freeList[atomicAdd(&s_nFree, 1u)] = (unsigned short)slot;
            } else {
                store_state(b, g, st);
                if (st.flags & F_PATH) rayq[atomicAdd(&s_nRays, 1u)] = (unsigned short)slot;
                if (st.flags & F_SHADOW) rayq[atomicAdd(&s_nRays, 1u)] = (unsigned short)(slot | 0x8000);
            }
        }
        __syncthreads();
    }
}

}  // namespace ptd

extern "C" {

// device scratch: [ staging (all streams of the call) | slot state for `groups` workgroups | sidOf | overflow stacks | unit counter ]
static size_t pw_staging_bytes(size_t nStreams) { return ((nStreams * 12 + 16) + 255) & ~(size_t)255; }
static size_t pw_state_bytes(size_t slots)
{
    size_t b = 0;
    b += slots * 16 * 8;     // 8 state arrays
    b += slots * 16 * 4;     // ray_o/ray_d x2
    b += slots * 8 * 2;      // hits
    b += slots * 4;          // sidOf
    return (b + 255) & ~(size_t)255;
}
int ptk_pw_groups(size_t nUnits, int numCUs)
{
    size_t g = (nUnits * 64 + ptd::kPwSlots - 1) / ptd::kPwSlots;
    const size_t cap = (size_t)numCUs * 2;           // two 512-thread workgroups per CU (128 VGPRs -> 4 waves/SIMD)
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}
size_t ptk_pw_work_bytes(size_t nUnits, int numCUs)
{
    const size_t groups = (size_t)ptk_pw_groups(nUnits, numCUs);
    const size_t slots = groups * ptd::kPwSlots;
    return pw_staging_bytes(nUnits * 64) + pw_state_bytes(slots) + groups * ptd::kPwThreads * ptd::kPwOvfLevels * 4 + 512;
}
const float* ptk_pw_staging(void* work) { return (const float*)work; }

hipError_t ptk_pw_render(const ptd::DevScene* sc, const ptd::DevCamera* cam, const ptd::DevParams* prm, void* work, int numCUs,
                         hipStream_t stream, hipEvent_t ev_begin, hipEvent_t ev_end)
{
    using namespace ptd;
    const size_t nUnits = (size_t)prm->n_units;
    const int groups = ptk_pw_groups(nUnits, numCUs);
    const size_t slots = (size_t)groups * kPwSlots;
    char* p = (char*)work;
    WfBuf b{};
    b.staging = (float*)p; p += pw_staging_bytes(nUnits * 64);
    auto take = [&](size_t bytes) { char* q = p; p += bytes; return q; };
    b.rng0 = (uint4*)take(slots * 16); b.rng1 = (uint4*)take(slots * 16);
    b.weight = (float4*)take(slots * 16); b.rad = (float4*)take(slots * 16); b.pix = (float4*)take(slots * 16);
    b.dir0 = (float4*)take(slots * 16); b.wb = (float4*)take(slots * 16); b.lp = (float4*)take(slots * 16);
    for (int k = 0; k < 2; k++) { b.ray_o[k] = (float4*)take(slots * 16); b.ray_d[k] = (float4*)take(slots * 16); }
    for (int k = 0; k < 2; k++) b.hit[k] = (float2*)take(slots * 8);
    uint32_t* sidOf = (uint32_t*)take(slots * 4);
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    int* ovf = (int*)take((size_t)groups * kPwThreads * kPwOvfLevels * 4);
    p = (char*)(((uintptr_t)p + 255) & ~(uintptr_t)255);
    unsigned int* unitCounter = (unsigned int*)p;
    hipError_t e;
    if ((e = hipMemsetAsync(unitCounter, 0, 64, stream)) != hipSuccess) return e;
    if (ev_begin) { if ((e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e; }
    hipLaunchKernelGGL(wf_persist, dim3(groups), dim3(kPwThreads), 0, stream, *sc, *cam, *prm, b, sidOf, unitCounter, ovf, groups * kPwThreads);
    if ((e = hipGetLastError()) != hipSuccess) return e;
    if (ev_end) { if ((e = hipEventRecord(ev_end, stream)) != hipSuccess) return e; }
    return hipSuccess;
}

}  // extern "C"
