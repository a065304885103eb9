// pt_wavefront.hip — the integrator as a queue-driven pipeline (the default render path).
//
// Why: the one-kernel state machine (pt_kernels.hip, kept as `mode 0`) is issue-bound at
// ~11 % SIMD lane utilisation (profiles/r01_pmc_megakernel_v1.json): rays of very different
// length share a wave, leaf code runs for a few lanes at a time, and the fat shading code
// holds 190 VGPRs (2 waves/SIMD).  Here every (pixel, pass) is a *stream* whose state lives
// in HBM as SoA float4 arrays; each iteration advances every live stream by one step:
//
//   wf_trace  closest hits of all pending path rays and visibility of all pending NEE shadow
//             rays (one queue index space): lean kernel, 7 waves/SIMD, lanes refill from the queue
//   wf_shade  one step of every live stream (pt_stream.h): apply the NEE terms whose shadow rays are back,
//             shade the path hit — and, when that path ends, the first hit of the next sample too
//             (the camera ray's hit is cached) — emit the shadow and path rays, or retire the stream
//             and write its per-pass mean.
//
// The traversal kernel is persistent: a wave takes ray ids from a 16-way sharded queue in
// chunks of <= 128 and, whenever >= 24 of its lanes have finished their ray, hands them new
// ones (ballot + prefix-popcount compaction), so lanes do not idle for the longest ray of the
// wave.  It walks the 4-wide quantised tree (pt_device.h).  Each trip of its loop the wave runs
// ONE of two code paths, a node step or a triangle test; a ray that reaches a leaf parks it and
// keeps walking, so both kinds of trip run fuller.  Rays that exceed a node budget are suspended
// and resumed by the next launch (time slicing).  Shadow rays stop at the first hit that is
// provably in front of the sampled light point (result-neutral, see wf_trace).
// For small renders (one rank of an 8-way tile split, a single full-frame pass) the shade step starts on a second stream beside
// the draining traversal kernel ("early shade", wf_shade PHASE 1 / 2): the launch tail of wf_trace is then not idle time.
//
// Per-stream arithmetic — order of random draws, every float operation — is exactly that of
// render_units / the reference's GetColor_iter, so images are bit-identical across modes.
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <thread>
#include <type_traits>
#include <vector>
#include "pt_device.h"
#include "pt_math.h"
#include "pt_bxdf.h"
#include "pt_trace.h"
#include "pt_shade.h"
#include "pt_stream.h"

namespace ptd {

constexpr int kWfLdsStack = 16;      // stack entries per lane kept in LDS (4 KB / wave)
constexpr int kWfOvfLevels = 48;     // further levels spill to global memory (never needed on the config scenes: 4-wide depth 12 -> at most 38 entries)
constexpr int kWfChunk = 128;        // most ray ids a wave takes from a queue shard per atomic (measured optimum 116-229)
constexpr int kWfRefill = 24;        // refill lanes once this many are idle (measured: 8..16 -2 %, 32 -0.4 %)
constexpr int kDone = (int)0x80000000;
#ifndef SHARD_BLOCK
#define SHARD_BLOCK 2048
#endif
constexpr uint32_t kShardBlock = SHARD_BLOCK;   // queue indices per block of the shard interleave (a power of two)
// wf_trace's waves per SIMD.  7 (72 VGPRs) rather than 8 (64): the two-triangle leaf test needs the room, and
// the kernel is bound by VALU issue, not by latency hiding (measured: 8 waves with 6-9 spilled registers and 6 waves
// with none are both slower; round 2 again: 8 waves +5 % kernel time).
#ifndef TRACE_WAVES
#define TRACE_WAVES 7
#endif
// Time slicing: every launch is followed by a device-wide dependency (the shade kernel needs all
// hits), so one ray that visits thousands of nodes would hold up the whole iteration (measured:
// ~800 us per launch).  A ray that has visited kWfBudget nodes is therefore suspended — cur, sp,
// closest hit and stack go to a record — and its stream simply waits one iteration; the next
// launch resumes it.  hit.prim <= -2 encodes "pending, record = -2 - prim".
constexpr int kWfBudget = 256;        // least node steps a ray may take per launch (measured: 96 cost 8 % on a 2M-stream render, >= 192 is flat)
constexpr int kSuspInts = 4 + kWfLdsStack + kWfOvfLevels;
// Top of the tree in LDS.  Every ray walks the first levels of the quad tree, and the kernel's vector-memory path is its
// busiest unit (TA / TD ~70 % busy: each lane fetches its own 64-byte node, 4 x 16 B per lane per node step), so the
// first kTopNodes nodes (breadth-first numbering, host/accel_build.cpp) are copied into LDS by every workgroup and
// node steps on them read LDS instead.  80-byte stride: consecutive nodes start 20 banks apart, so the 16 lanes of
// a ds_read_b128 group rarely collide.  Size: 16 KB of stacks + 6 KB of tree per workgroup, 7 workgroups per CU.
// Node step: 1 = sort the four (entry distance, child) pairs (rounds 1-2), 0 = nearest child exactly, the others in slot order (round 3)
#ifndef TRACE_SORT4
#define TRACE_SORT4 1
#endif
#ifndef TRACE_TOP_NODES
#define TRACE_TOP_NODES 0
#endif
constexpr int kTopNodes = TRACE_TOP_NODES > 0 ? TRACE_TOP_NODES : 1;
[[maybe_unused]] constexpr int kTopStride = 5;         // uint4 per staged node

// block-aggregated append to four lists at once (live streams + one ray queue per kind): one atomicAdd
// per list per block.  (One atomic per wave was the shade kernel's bottleneck: ~100k returning atomics
// per launch on one cache line serialise at ~88 per microsecond; with one atomic per 256-thread block and the
// counters on separate lines they no longer show.)
// Must be called by every thread of the block.
constexpr int kShadeThreads = 1024;      // largest wf_shade workgroup (the default runs 512-thread workgroups, two per CU: 4 waves/SIMD at 126 VGPRs)
constexpr int kLists = 1 + 2 * kRayKinds;      // live streams + per ray kind a front list (through the core box / unknown) and a back list (short rays)
template <int N>
PT_DEV void block_append(const bool (&e)[N], const uint32_t (&id)[N], uint32_t* const (&c)[N], uint32_t* const (&l)[N], const uint32_t (&top)[N])
{
    // top[k] = 0: list k grows upwards from index 0; top[k] = T > 0: it grows DOWNWARDS from index T (the short rays of a kind share
    // the kind's queue array with its other rays, from the other end)
    constexpr int W = kShadeThreads / 64;
    __shared__ uint32_t s_cnt[N][W];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nw = (int)(blockDim.x >> 6);
    unsigned long long m[N];
#pragma unroll
    for (int k = 0; k < N; k++) { m[k] = __ballot(e[k]); if (lane == 0) s_cnt[k][wave] = __builtin_popcountll(m[k]); }
    __syncthreads();
    if (threadIdx.x < N) {
        // one thread per list: the block's total -> one atomic, and every wave's count turned into its start position in place
        // (every thread summing the counts of the waves below its own: 7 lists x up to 7 LDS reads per thread, -1.3 % overall, r03_b26.log)
        uint32_t tot = 0;
        for (int w = 0; w < nw; w++) tot += s_cnt[threadIdx.x][w];
        uint32_t run = tot ? atomicAdd(c[threadIdx.x], tot) : 0u;
        for (int w = 0; w < nw; w++) { const uint32_t v = s_cnt[threadIdx.x][w]; s_cnt[threadIdx.x][w] = run; run += v; }
    }
    __syncthreads();
    const unsigned long long below = (1ull << lane) - 1ull;
#pragma unroll
    for (int k = 0; k < N; k++) {
        const uint32_t pos = s_cnt[k][wave] + (uint32_t)__builtin_popcountll(m[k] & below);
        if (e[k]) l[k][top[k] ? top[k] - pos : pos] = id[k];
    }
}

// ---------------------------------------------------------------------------------------
// wf_init: StartRender prologue for every stream (pathtracer.cu:70-74)
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256)
void wf_init(DevScene sc, DevCamera cam, DevParams prm, WfBuf b, uint32_t nStreams)
{
    const uint32_t sid = blockIdx.x * 256u + threadIdx.x;
    bool live = false, camShort = false;
    if (sid < nStreams) {
        const uint32_t unit = (uint32_t)prm.unit_base + (sid >> 6), lane = sid & 63;
        const int pass_rel = (int)(unit / (uint32_t)prm.n_tiles_local);
        const int lt = (int)(unit % (uint32_t)prm.n_tiles_local);
        const int tile = lt * prm.world + prm.rank;
        const int tx = tile % prm.tiles_x, ty = tile / prm.tiles_x;
        const int px = tx * kTile + (int)(lane & 7), py = ty * kTile + (int)(lane >> 3);
        const int pass = prm.first_pass + pass_rel;
        live = (tile < prm.n_tiles_total) && (px < cam.W) && (py < cam.H);
        if (live) {
            init_stream(cam, prm, b, sid, px, py, pass);
            const float4 d0 = b.ray_d[0][sid];
            camShort = ray_is_short(sc, f3(cam.pos[0], cam.pos[1], cam.pos[2]), f3(d0.x, d0.y, d0.z), 3.0e38f);
        } else {
            b.staging[3 * (size_t)sid + 0] = 0.f; b.staging[3 * (size_t)sid + 1] = 0.f; b.staging[3 * (size_t)sid + 2] = 0.f;
        }
    }
    // the camera ray: queued by class like every other ray (a pixel that looks past the mesh has a short ray)
    const bool shortRay = live && camShort;
    const uint32_t topIdx = (uint32_t)(b.hit[1] - b.hit[0]) - 1u;      // n16 - 1: the last entry of a queue array
    const bool e[kLists] = {live, live && !shortRay, false, false, shortRay, false, false};
    uint32_t* const c[kLists] = {&b.cnt[0].nActive, &b.cnt[0].nRays[0][0], &b.cnt[0].nRays[1][0], &b.cnt[0].nRays[2][0],
                                 &b.cnt[0].nRays[0][kShortWord], &b.cnt[0].nRays[1][kShortWord], &b.cnt[0].nRays[2][kShortWord]};
    uint32_t* const l[kLists] = {b.active[0], b.rq[0], b.rq[1], b.rq[2], b.rq[0], b.rq[1], b.rq[2]};
    const uint32_t ids[kLists] = {sid, sid, sid, sid, sid, sid, sid};
    const uint32_t top[kLists] = {0u, 0u, 0u, 0u, topIdx, topIdx, topIdx};
    block_append<kLists>(e, ids, c, l, top);
}

// ---------------------------------------------------------------------------------------
// wf_trace: persistent closest-hit kernel with lane refill.
// Shadow rays (kinds 1 and 2, queue indices >= nPath): the ray only decides whether the closest hit is the sampled light point
// (GetLightColor, CudaUtil.cuh:150-166: visible iff |hit.p - P| < EPS, with t_max = |P-p|+1).
// Any hit at t < (t_max - 1) - 5e-4 proves the closest hit is at least ~4e-4 in front of P,
// hence not within EPS = 1e-4 of it, so traversal may stop there; what is reported is then
// some occluder, for which wf_shade's |hit.p - P| < EPS test fails exactly as it would for the
// closest one.
// ---------------------------------------------------------------------------------------
// STAT: a diagnostic build that also counts trips and the lanes they serve (pt_last_counters; PTAMD_TSTAT=1).
template <int MODE, bool PUBLISH = false>      // PUBLISH: hits are stored device-coherently (wf_shade PHASE 1 reads them while this kernel drains);  MODE: 0 production, 1 trip counters + histograms + timeline (PTAMD_TSTAT=1), 2 timeline only (PTAMD_TSTAT=2), 3 trip counters + section clocks, no per-step atomics (PTAMD_TSTAT=3)
__global__ __launch_bounds__(256, TRACE_WAVES)
void wf_trace(DevScene sc, WfBuf b, int slot, int ovfStride, int parity, int chunkShift, int budgetShift, int budgetMin, int guideShift, int triTrig, int refillMin,
              int topWant, unsigned long long* stat, int statLaunch, int helpShards, int lateBudget)
{
    constexpr bool STAT = MODE == 1 || MODE == 3, HIST = MODE == 1, timeline = MODE != 0;
    // PUBLISH launches share the chip with wf_shade's early phase: the traversal is the critical path of the iteration (its last waves
    // run a serial chain), so its waves issue ahead of the shading waves (statLaunch carries the priority in the production build)
    if (PUBLISH && MODE == 0) { if (statLaunch == 1) __builtin_amdgcn_s_setprio(1); else if (statLaunch == 2) __builtin_amdgcn_s_setprio(2); else if (statLaunch == 3) __builtin_amdgcn_s_setprio(3); }
    const unsigned long long stT0 = timeline ? __builtin_amdgcn_s_memrealtime() : 0ull;      // 100 MHz
    unsigned long long stTExh = 0;
    unsigned long long stClk[5] = {0, 0, 0, 0, 0}, stMark = 0;      // STAT: shader clocks in refill / vote + budget / node step / triangle step / ray epilogue
    unsigned long long stNodeTrips = 0, stNodeLanes = 0, stTriTrips = 0, stTriLanes = 0, stRefills = 0, stRefillLanes = 0, stNoRayLanes = 0, stRays = 0;
    unsigned int stTrips = 0, stTripsDry = 0;      // MODE 2: trips of this wave in all, and after it found the queue dry
    // MODE 2, one chosen launch (PTAMD_TDUMP; stat[6] = launch + 1): a record per wave (kStatWaveRec) and a per-trip log of every 112th wave
    const bool stDump = MODE == 2 && stat[6] == (unsigned long long)statLaunch + 1ull;
    const uint32_t stWave = blockIdx.x * 4u + (threadIdx.x >> 6);
    const bool stLog = stDump && stWave % kStatLogEvery == 0 && stWave / kStatLogEvery < (uint32_t)kStatLogWaves;
    __shared__ int lds_stack[4][kWfLdsStack * 64];
    // one queue index space, six segments: the rays of kind 0 (path), 1 and 2 (shadow) queued from the front of their arrays — rays through
    // the scene's core box, or of unknown length —, then the SHORT rays of each kind, queued from the back of the same arrays
    // (pt_stream.h: ray_is_short): what is in flight when the queue runs dry is then short, and so is the launch tail
    const uint32_t p1 = b.cnt[slot].nRays[0][0];
    const uint32_t p2 = p1 + b.cnt[slot].nRays[1][0];
    const uint32_t p3 = p2 + b.cnt[slot].nRays[2][0];
    const uint32_t p4 = p3 + b.cnt[slot].nRays[0][kShortWord];
    const uint32_t p5 = p4 + b.cnt[slot].nRays[1][kShortWord];
    const uint32_t n = p5 + b.cnt[slot].nRays[2][kShortWord];
    if ((uint32_t)blockIdx.x * 256u >= n) return;      // surplus blocks leave before touching the queue (fewer rays per workgroup: no faster, r02_b21.log)
#if TRACE_TOP_NODES > 0
    __shared__ uint4 lds_top[kTopNodes * kTopStride];
    const int topN = min(min(topWant, kTopNodes), sc.n_quad);
    for (int i = threadIdx.x; i < topN * 4; i += 256) lds_top[(i >> 2) * kTopStride + (i & 3)] = sc.quad[i];
    __syncthreads();
#endif

    const int lane = threadIdx.x & 63;
    int* stack = &lds_stack[threadIdx.x >> 6][lane];
    int* ovf = b.ovf + (blockIdx.x * 256 + threadIdx.x);
    // Stack entry k of this lane: LDS below kWfLdsStack, global memory above (5e-7 of the node steps).  Written as a plain select of
    // the two places the compiler merges them into ONE flat_load (LDS through the texture path, waited for with vmcnt(0), i.e. behind
    // every outstanding store); the empty asm pins the LDS read down as a ds_read of its own.
    auto stack_at = [&](int k) -> int {
        int v = stack[(k < kWfLdsStack ? k : 0) * 64];
        asm volatile("" : "+v"(v));
        if (k >= kWfLdsStack) v = ovf[(k - kWfLdsStack) * ovfStride];
        return v;
    };
    // node steps a ray may take in this launch before it is suspended: large launches hide long rays,
    // small (latency-bound) launches must not wait for them
    const int budget = (n >> budgetShift) < (uint32_t)budgetMin ? budgetMin : ((n >> budgetShift) > 1024u ? 1024 : (int)(n >> budgetShift));
    const int* __restrict__ suspIn = b.susp[parity ^ 1];
    int* __restrict__ suspOut = b.susp[parity];
    // rays a wave takes per queue access: ~n / (4 x resident waves), between 16 and kWfChunk (one word
    // saturates near 88 returning atomics per microsecond, so large launches take large chunks)
    const uint32_t kChunk = (n >> chunkShift) < 16u ? 16u : ((n >> chunkShift) > (uint32_t)kWfChunk ? (uint32_t)kWfChunk : (n >> chunkShift));

    uint32_t chunkPos = 0, chunkEnd = 0;   // wave-uniform
    uint32_t seenLeft = 0xffffffffu;       // rays this wave last saw left in its current shard (wave-uniform)
    int shard = (int)(blockIdx.x % kWfShards), shardsTried = 0;   // wave-uniform; blockIdx % 8 shares an XCD, so a shard stays in one L2
    bool exhausted = false;                // wave-uniform
    bool hasRay = false;
    // per-ray registers
    // inv: the reference's Normalize(inv(dir)) (CudaUtil.cuh:70) — what its leaf-box test uses, and a perfectly good
    // inverse direction for the tree walk, which then measures t in units of 1/|inv(dir)|; cscale converts the
    // closest hit into those units.  Degenerate rays (a zero direction component): 1/dir clamped to +-1e30, true units.
    f3 org(0.f, 0.f, 0.f), dir(0.f, 0.f, 1.f), inv(0.f, 0.f, 0.f);
    float bestT = 0.f, cscale = 0.f, stopBelow = 0.f;
    int bestPrim = -1, cur = kDone, sp = 0, steps = 0;
    int pend = 0;        // a leaf this ray has reached but not yet tested (0 = none): see "postponed leaves" below
    bool degenerate = false;
    // The per-kind arrays of WfBuf lie back to back (hit[k] = hit[0] + k * n16, rq[k] likewise, ray_o[k] = ray_o[0] + 2 k n16, ray_d[k] = ray_o[k] + n16):
    // a ray is known by hs = kind * n16 + stream id, its hit slot is hit[0][hs], and no pointer is ever selected by kind.
    const uint32_t n16 = (uint32_t)(b.hit[1] - b.hit[0]);
    uint32_t hs = 0;

    for (;;) {
        // ---- hand new rays to idle lanes (ballot + mbcnt compaction) ----
        if (STAT) stMark = __builtin_amdgcn_s_memtime();
#define PT_STCLK(k) if (STAT) { const unsigned long long now = __builtin_amdgcn_s_memtime(); stClk[k] += now - stMark; stMark = now; }
        uint32_t stW0 = 0, stAtomics = 0, stTook = 0;      // MODE 2 trip log: time at the top of the loop, queue atomics and rays taken in this iteration
        uint32_t stW1 = 0, stW3 = 0;
        if (MODE == 2 && stLog) {
            // A = top of the loop, on the constant 100 MHz clock, and the shader clock counter at the same moment (their ratio is the clock the SIMD runs at)
            stW0 = (uint32_t)((__builtin_amdgcn_s_memrealtime() - stT0) & 0xfffffull);
            stW1 = (uint32_t)__builtin_amdgcn_s_memtime();
        }
        const unsigned long long idle = __ballot(!hasRay);
        const int nIdle = __builtin_popcountll(idle);
        if (!exhausted && (nIdle >= refillMin)) {
            if (chunkPos == chunkEnd) {
                // The queue index space is cut into kWfShards ranges, each with its own head word (a
                // single word saturates near 88 returning atomics per microsecond, which throttled
                // launches of a few million rays).  A wave drains its home shard, then helps the next.
                for (;;) {
                    // A shard owns every 16th block of kShardBlock queue indices (block b -> shard b % 16), so all shards walk the index
                    // space front to back together: path rays are started first and the (shorter, any-hit) shadow rays last, which
                    // is what is still in flight when the queue runs dry.  chunkPos / chunkEnd are shard-local positions.
#ifdef PT_SHARD_CONTIG      // A/B build: a shard owns one contiguous sixteenth of the index space
                    const uint32_t cLo = (uint32_t)(((unsigned long long)n * (unsigned)shard) / kWfShards);
                    const uint32_t cnt = (uint32_t)(((unsigned long long)n * (unsigned)(shard + 1)) / kWfShards) - cLo;
#else
                    const uint32_t rounds = n / (kShardBlock * kWfShards), rem = n % (kShardBlock * kWfShards);
                    const uint32_t part = rem > (uint32_t)shard * kShardBlock ? rem - (uint32_t)shard * kShardBlock : 0u;
                    const uint32_t cnt = rounds * kShardBlock + (part < kShardBlock ? part : kShardBlock);      // indices this shard owns
#endif
                    // guided self-scheduling: the chunk shrinks with what this wave last saw left in the shard, so the
                    // last rays of a launch are spread over many waves instead of queuing behind one
                    uint32_t want = seenLeft >> guideShift;
                    want = want < 16u ? 16u : (want > kChunk ? kChunk : want);
                    uint32_t start = 0;
                    if (lane == 0) start = atomicAdd(&b.cnt[slot].head[shard].v, want);
                    if (MODE == 2) stAtomics++;
                    start = __builtin_amdgcn_readfirstlane(start);
                    if (start < cnt) {
                        chunkPos = start; chunkEnd = (cnt - start > want) ? start + want : cnt; seenLeft = cnt - start; break; }
                    seenLeft = 0xffffffffu;
                    shard = (shard + 1) % kWfShards;
                    if (++shardsTried >= helpShards) { exhausted = true; if (timeline) stTExh = __builtin_amdgcn_s_memrealtime(); break; }
                }
            }
            if (!exhausted) {
                const uint32_t avail = chunkEnd - chunkPos;
                const uint32_t take = ((uint32_t)nIdle < avail) ? (uint32_t)nIdle : avail;
                if (STAT && take) { stRefills++; stRefillLanes += take; }
                if (MODE == 2) stTook = take;
                if (!hasRay) {
                    const uint32_t r = (uint32_t)__builtin_popcountll(idle & ((1ull << lane) - 1ull));
                    if (r < take) {
                        const uint32_t j = chunkPos + r;      // shard-local -> queue index
#ifdef PT_SHARD_CONTIG
                        const uint32_t q = (uint32_t)(((unsigned long long)n * (unsigned)shard) / kWfShards) + j;
#else
                        const uint32_t q = ((j / kShardBlock) * kWfShards + (uint32_t)shard) * kShardBlock + (j % kShardBlock);
#endif
                        // segment of q -> kind, position in the kind's queue array (front part upwards, short part downwards from n16 - 1)
                        const bool shortSeg = q >= p3;
                        const uint32_t b0 = shortSeg ? p3 : 0u, b1 = shortSeg ? p4 : p1, b2 = shortSeg ? p5 : p2;      // segment starts of kinds 0, 1, 2
                        const bool k0 = q < b1, k1 = q < b2;                                   // kind 0 / kind 0 or 1
                        const uint32_t kn = k0 ? 0u : (k1 ? n16 : 2u * n16);                  // kind * n16
                        const uint32_t local = q - (k0 ? b0 : (k1 ? b1 : b2));
                        const uint32_t qid = b.rq[0][kn + (shortSeg ? n16 - 1u - local : local)];
                        hs = kn + (qid & ~kResumeBit);
                        const float4 o = ld_s(&b.ray_o[0][hs + kn]), d = ld_s(&b.ray_d[0][hs + kn]);
                        const int kind = k0 ? 0 : 1;      // all that is still asked of it: path ray or not
                        org = f3(o.x, o.y, o.z); dir = f3(d.x, d.y, d.z);
                        ray_setup(dir, inv, cscale, degenerate);      // pt_trace.h
                        stopBelow = kind != 0 ? d.w : -__builtin_inff();      // shadow rays: any hit below this t ends the traversal (pt_stream.h: shadow_stop_t)
                        steps = 0;
                        if (qid & kResumeBit) {
                            // resume a suspended traversal (wf_shade flags the queue entry: the hit slot then holds the record number;
                            // a fresh ray's hit slot is not read at all — 8 scattered bytes per ray that nothing else would fetch)
                            const float2 prev = b.hit[0][hs];
                            const int pp = __float_as_int(prev.y);
                            const int* rec = suspIn + (size_t)(-2 - pp) * kSuspInts;
                            cur = rec[0]; sp = rec[1]; bestT = __int_as_float(rec[2]); bestPrim = rec[3];
                            for (int k = 0; k < sp; k++) {
                                const int v = rec[4 + k];
                                if (k < kWfLdsStack) stack[k * 64] = v; else ovf[(k - kWfLdsStack) * ovfStride] = v;
                            }
                        } else {
                            bestT = o.w; bestPrim = -1; cur = 0; sp = 0;
                        }
                        pend = 0;
                        hasRay = true;
                    }
                }
                chunkPos += take;
            }
        }
        PT_STCLK(0)
        if (__ballot(hasRay) == 0ull) { if (exhausted) break; else continue; }
        if (MODE == 2) { stTrips++; if (exhausted) stTripsDry++; }
        uint32_t* stLogAt = nullptr;
        if (MODE == 2 && stLog && 4 * stTrips <= (unsigned)kStatLogTrips) {
            // four words per trip (10-ns ticks since the wave started in the low 20 bits).  w0: A, top of the loop | lanes with a ray (7 bits) | queue
            // already dry | queue atomics of the refill (2 bits, saturating); w1: the shader clock counter at A (low 32 bits);
            // w2: C, after the refill | rays taken (7) | some lane holds a leaf; w3: D, node data of a node trip arrived (0 for a triangle trip)
            const uint32_t lanesNow = (uint32_t)__builtin_popcountll(__ballot(hasRay)), pendNow = (uint32_t)__builtin_popcountll(__ballot(hasRay && pend != 0));
            stLogAt = (uint32_t*)(stat + kStatWords + (size_t)kStatWaves * 8) + (size_t)(stWave / kStatLogEvery) * kStatLogTrips + 4 * (stTrips - 1);
            if (lane == 0) {
                stLogAt[0] = stW0 | (lanesNow << 20) | (exhausted ? 1u << 27 : 0u) | ((stAtomics > 3u ? 3u : stAtomics) << 28);
                stLogAt[1] = stW1;
                stLogAt[2] = (uint32_t)((__builtin_amdgcn_s_memrealtime() - stT0) & 0xfffffull) | ((stTook > 127u ? 127u : stTook) << 20) | (pendNow ? 1u << 27 : 0u);
                stLogAt[3] = 0u;
            }
        }
        if (hasRay) {
            // Only one code path runs per trip: a node step or ONE triangle test per lane (the vote is below).
            // (The classic while-while shape made 64 lanes wait for the slowest lane to reach a leaf every
            // round — 24 % VALU lane utilisation; running both paths every trip, "if-if", gave 29 %.)
            if (cur >= 0) {
                // lateBudget (PTAMD_LB, default 64): once the queue is dry and this wave is down to its last two rays, a ray that has already done
                // that many node steps is suspended like one that has spent its budget — the launch then does not wait for a lone ray of several
                // hundred steps (it goes on in the next launch, from the start and among full waves)
                if (steps >= ((lateBudget > 0 && exhausted && nIdle >= 62) ? lateBudget : budget)) {
                    // node budget spent: suspend (or, if the pool is full, carry on)
                    const uint32_t rec = atomicAdd(&b.cnt[slot].nSusp, 1u);
                    if (rec < b.suspCap) {
                        int* r = suspOut + (size_t)rec * kSuspInts;
                        if (pend != 0) { if (sp < kWfLdsStack) stack[sp * 64] = pend; else ovf[(sp - kWfLdsStack) * ovfStride] = pend; sp++; pend = 0; }
                        r[0] = cur; r[1] = sp; r[2] = __float_as_int(bestT); r[3] = bestPrim;
                        for (int k = 0; k < sp; k++) r[4 + k] = (k < kWfLdsStack) ? stack[k * 64] : ovf[(k - kWfLdsStack) * ovfStride];
                        if (PUBLISH) __hip_atomic_store((unsigned long long*)&b.hit[0][hs], (unsigned long long)__float_as_uint(bestT) | ((unsigned long long)(uint32_t)(-2 - (int)rec) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        else b.hit[0][hs] = make_float2(bestT, __int_as_float(-2 - (int)rec));
                        hasRay = false;
                        cur = kDone;
                    } else {
                        steps = -(1 << 28);
                    }
                }
            }
            // Wave vote with postponed leaves.  A ray that reaches a leaf does not wait for a triangle trip:
            // it parks the leaf in `pend` and goes on with the next entry of its stack; only a ray that
            // reaches a SECOND leaf (or has nothing else left) needs a triangle trip.  A triangle trip
            // then serves every ray with a parked leaf, a node trip every ray that still holds a node:
            // both kinds of trip run fuller than when each ray blocked at its first leaf
            // (59 % of lane-trips useful before).  Order of tests does not matter to the result: the tie rule
            // makes the closest hit independent of it; parked leaves only delay the tightening of the cull.
            const int nNode = __builtin_popcountll(__ballot(cur >= 0));
            const int nTri = __builtin_popcountll(__ballot(hasRay && pend != 0));
            const int nBlk = __builtin_popcountll(__ballot(hasRay && pend != 0 && cur < 0));
            const bool doTri = nTri > 0 && (nTri >= triTrig || nBlk >= nNode);
            const bool doNode = !doTri;
            if (STAT) {
                if (doNode) { stNodeTrips++; stNodeLanes += nNode; } else { stTriTrips++; stTriLanes += nTri; }
                stNoRayLanes += 64 - __builtin_popcountll(__ballot(hasRay));
            }
            PT_STCLK(1)
            if (doNode && cur >= 0) {
                steps++;
                // ---- one 4-wide node: conservative slab test of its four quantised child boxes ----
                // Child box = origin + 2^e * q.  Along each axis t(q) = q*A + B with A = 2^e * inv and
                // B = (origin - org) * inv; both are widened by `sl` (2^-20 of their magnitudes, ~16 ulps),
                // the near/far bytes are picked by the sign of the direction once per node (all four children
                // of a coordinate share a dword), and the far side is cut at the closest hit.  The boxes only
                // steer the search — acceptance is Triangle::hit + the reference's leaf box — so all that
                // matters is that no box containing a point the ray reaches is ever rejected.
                uint4 n0, n1, n2, n3;
#if TRACE_TOP_NODES > 0
                if (cur < topN) {
                    // explicit LDS reads: left to itself the compiler merges the two address spaces into flat_load instructions,
                    // which cost the whole kernel 18 % (every node fetch then waits on both counters and the ray origin spills)
                    const uint32_t la = (uint32_t)(uintptr_t)(lds_top + cur * kTopStride);
                    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:16\n\tds_read_b128 %2, %4 offset:32\n\tds_read_b128 %3, %4 offset:48\n\ts_waitcnt lgkmcnt(0)"
                                 : "=&v"(n0), "=&v"(n1), "=&v"(n2), "=&v"(n3) : "v"(la) : "memory");
                } else
#endif
                {
                    const uint4* np = sc.quad + 4 * (size_t)cur;
                    n0 = np[0]; n1 = np[1]; n2 = np[2]; n3 = np[3];
                }
                if (MODE == 2 && stLogAt) {
                    asm volatile("s_waitcnt vmcnt(0)" : "+v"(n0.x), "+v"(n1.x), "+v"(n2.x), "+v"(n3.x) :: "memory");
                    stW3 = (uint32_t)((__builtin_amdgcn_s_memrealtime() - stT0) & 0xfffffull);
                    stLogAt[3] = stW3;      // every lane of the node trip writes the same word
                }
                const float Ax = inv.x * __uint_as_float(n0.w), Ay = inv.y * __uint_as_float(n3.z), Az = inv.z * __uint_as_float(n3.w);      // scales are powers of two
                const float Bx = (__uint_as_float(n0.x) - org.x) * inv.x;
                const float By = (__uint_as_float(n0.y) - org.y) * inv.y;
                const float Bz = (__uint_as_float(n0.z) - org.z) * inv.z;
                const float kSl = 9.5367431640625e-7f;                           // 2^-20
                const float sx = (__builtin_fabsf(Bx) + 255.f * __builtin_fabsf(Ax)) * kSl;
                const float sy = (__builtin_fabsf(By) + 255.f * __builtin_fabsf(Ay)) * kSl;
                const float sz = (__builtin_fabsf(Bz) + 255.f * __builtin_fabsf(Az)) * kSl;
                const float Bnx = Bx - sx, Bfx = Bx + sx, Bny = By - sy, Bfy = By + sy, Bnz = Bz - sz, Bfz = Bz + sz;
                // near / far bytes by the sign of the direction, as a masked swap: on this chip a second v_cndmask on the same vcc costs
                // ~23 clocks (tools/valu_probe.py), xor / and / arithmetic shift ~2.3 each.  (A zero component of either sign gives
                // A = B = +-0 on that axis: both planes at t = 0, so either assignment is the same test.)
                const uint32_t mx = (uint32_t)(__float_as_int(inv.x) >> 31), my = (uint32_t)(__float_as_int(inv.y) >> 31), mz = (uint32_t)(__float_as_int(inv.z) >> 31);
                const uint32_t swx = (n2.x ^ n2.w) & mx, swy = (n2.y ^ n3.x) & my, swz = (n2.z ^ n3.y) & mz;
                const uint32_t nqx = n2.x ^ swx, fqx = n2.w ^ swx;   // lo.x = n2.x, hi.x = n2.w
                const uint32_t nqy = n2.y ^ swy, fqy = n3.x ^ swy;   // lo.y = n2.y, hi.y = n3.x
                const uint32_t nqz = n2.z ^ swz, fqz = n3.y ^ swz;   // lo.z = n2.z, hi.z = n3.y
                const float cullT = bestT * cscale;
                int key[4];
#pragma unroll
                for (int k = 0; k < 4; k++) {
                    const float tnx = __builtin_fmaf((float)((nqx >> (8 * k)) & 0xffu), Ax, Bnx);
                    const float tny = __builtin_fmaf((float)((nqy >> (8 * k)) & 0xffu), Ay, Bny);
                    const float tnz = __builtin_fmaf((float)((nqz >> (8 * k)) & 0xffu), Az, Bnz);
                    const float tfx = __builtin_fmaf((float)((fqx >> (8 * k)) & 0xffu), Ax, Bfx);
                    const float tfy = __builtin_fmaf((float)((fqy >> (8 * k)) & 0xffu), Ay, Bfy);
                    const float tfz = __builtin_fmaf((float)((fqz >> (8 * k)) & 0xffu), Az, Bfz);
                    const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, 0.f));
                    const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, cullT));
                    key[k] = (tn <= tf) ? __float_as_int(tn) : 0x7fffffff;   // tn >= 0: its bits order like ints
                }
#if TRACE_SORT4
                // sort (entry distance, ref) pairs: 5 compare-exchanges, each one compare + four selects (equal distances: any order will do)
                int k0 = key[0], k1 = key[1], k2 = key[2], k3 = key[3], r0 = (int)n1.x, r1 = (int)n1.y, r2 = (int)n1.z, r3 = (int)n1.w;
#define PT_CE(ka, ra, kb, rb) { const bool sw = ka > kb; const int tk = sw ? kb : ka, tr = sw ? rb : ra; kb = sw ? ka : kb; rb = sw ? ra : rb; ka = tk; ra = tr; }
                PT_CE(k0, r0, k1, r1) PT_CE(k2, r2, k3, r3) PT_CE(k0, r0, k2, r2) PT_CE(k1, r1, k3, r3) PT_CE(k1, r1, k2, r2)
#undef PT_CE
                if (k0 != 0x7fffffff) {
                    // nearest child next; the other hit children go to the stack, farthest first
                    if (k3 != 0x7fffffff) { if (sp < kWfLdsStack) stack[sp * 64] = r3; else ovf[(sp - kWfLdsStack) * ovfStride] = r3; sp++; }
                    if (k2 != 0x7fffffff) { if (sp < kWfLdsStack) stack[sp * 64] = r2; else ovf[(sp - kWfLdsStack) * ovfStride] = r2; sp++; }
                    if (k1 != 0x7fffffff) { if (sp < kWfLdsStack) stack[sp * 64] = r1; else ovf[(sp - kWfLdsStack) * ovfStride] = r1; sp++; }
                    cur = r0;
                } else if (sp == 0) {
                    cur = kDone;
                } else {
                    sp--;
                    cur = stack_at(sp);
                }
#else
                // The nearest hit child next, found exactly (a 4-way minimum of the entry distances and four equality tests: 9 slow-class
                // instructions where sorting all four (distance, ref) pairs took 25); the other hit children go to the stack in slot order.
                // Measured on the config scenes' ray mix (tools/quad_order_lab.cpp): +0 ... +1.2 % node steps against the full sort — what
                // matters is which child is entered first, hardly in which order the others wait.  The closest hit does not depend on
                // the order of tests (tie rule), so this is result-neutral.
                const int kmin = min(min(key[0], key[1]), min(key[2], key[3]));
                if (kmin != 0x7fffffff) {
                    const bool m0 = key[0] == kmin, m1 = !m0 && key[1] == kmin, m2 = !m0 && !m1 && key[2] == kmin, m3 = !m0 && !m1 && !m2;
                    if (key[3] != 0x7fffffff && !m3) { if (sp < kWfLdsStack) stack[sp * 64] = (int)n1.w; else ovf[(sp - kWfLdsStack) * ovfStride] = (int)n1.w; sp++; }
                    if (key[2] != 0x7fffffff && !m2) { if (sp < kWfLdsStack) stack[sp * 64] = (int)n1.z; else ovf[(sp - kWfLdsStack) * ovfStride] = (int)n1.z; sp++; }
                    if (key[1] != 0x7fffffff && !m1) { if (sp < kWfLdsStack) stack[sp * 64] = (int)n1.y; else ovf[(sp - kWfLdsStack) * ovfStride] = (int)n1.y; sp++; }
                    if (key[0] != 0x7fffffff && !m0) { if (sp < kWfLdsStack) stack[sp * 64] = (int)n1.x; else ovf[(sp - kWfLdsStack) * ovfStride] = (int)n1.x; sp++; }
                    cur = m0 ? (int)n1.x : (m1 ? (int)n1.y : (m2 ? (int)n1.z : (int)n1.w));
                } else if (sp == 0) {
                    cur = kDone;
                } else {
                    sp--;
                    cur = stack_at(sp);
                }
#endif
                if (HIST) atomicAdd(&stat[8 + 3 * 2700 + 32 + 2700 + 64 + (sp > 31 ? 31 : sp)], 1ull);      // stack depth after this node step
                if (cur < 0 && cur != kDone && pend == 0) {
                    // park the leaf, carry on with the next stack entry
                    pend = cur;
                    if (sp == 0) cur = kDone;
                    else { sp--; cur = stack_at(sp); }
                }
            } else if (doTri && hasRay && pend != 0) {
                // ---- the parked leaf: its (up to) two triangles in one go ----
                const int code = ~pend, first = code >> 3, cnt = code & 7;
                pend = 0;
                if (cnt > 0) {
                    tri_test_pairrec(sc, first, cnt > 1, org, dir, inv, degenerate, bestT, bestPrim);
                    if (bestPrim >= 0 && bestT < stopBelow) { cur = kDone; sp = 0; }          // shadow ray: any occluder in front of the light will do
                    else if (cnt > 2) pend = ~(((first + 2) << 3) | (cnt - 2));
                }
                if (pend == 0 && cur < 0 && cur != kDone) {
                    // the ray was waiting with a second leaf: park that one, take the next stack entry
                    pend = cur;
                    if (sp == 0) cur = kDone;
                    else { sp--; cur = stack_at(sp); }
                }
            }
            PT_STCLK(doNode ? 2 : 3)
            if (hasRay && cur == kDone && pend == 0) {
                // spheres, in order, against the triangles' closest t (CudaUtil.cuh:137-145)
                for (int s = 0; s < sc.n_spheres; s++) {
                    const float4 c = sc.spheres[4 * s];
                    float root;
                    if (sphere_root(f3(c.x, c.y, c.z), c.w, org, dir, bestT, root)) { bestT = root; bestPrim = sc.n_tris + s; }
                }
                if (PUBLISH) __hip_atomic_store((unsigned long long*)&b.hit[0][hs], (unsigned long long)__float_as_uint(bestT) | ((unsigned long long)(uint32_t)bestPrim << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else st_s(&b.hit[0][hs], make_float2(bestT, __int_as_float(bestPrim)));
                hasRay = false;
                if (STAT || MODE == 2) stRays++;
                if (HIST) atomicAdd(&stat[8 + 3 * 2700 + 32 + 2700 + (steps >= 252 ? 63 : steps >> 2)], 1ull);      // node steps of this ray (this launch), bins of 4
            }
            PT_STCLK(4)
        }
    }
#undef PT_STCLK
    if (timeline) {
        // per-lane ray count -> wave total
        unsigned long long r = stRays;
        if (STAT || MODE == 2) for (int o = 32; o > 0; o >>= 1) r += __shfl_xor(r, o);
        if (lane == 0 && stDump) {
            // the dumped launch: one record per wave and none of the pooled statistics below (their atomics on a few hot words come from
            // every wave as it leaves, i.e. all through the launch tail that is being looked at)
            if (stWave < (uint32_t)kStatWaves) {
                unsigned long long* w = stat + kStatWords + (size_t)stWave * 8;
                w[0] = stT0; w[1] = stTExh; w[2] = __builtin_amdgcn_s_memrealtime(); w[3] = stTrips; w[4] = stTripsDry; w[5] = r;
                w[6] = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_ID: wave slot [3:0], SIMD [5:4], CU [11:8], SH [12], SE [15:13]
                w[7] = __builtin_amdgcn_s_getreg(20 | (31 << 11));      // XCC_ID
            }
        } else if (lane == 0) {
            if (STAT) { atomicAdd(&stat[0], stNodeTrips); atomicAdd(&stat[1], stNodeLanes); atomicAdd(&stat[2], stTriTrips); atomicAdd(&stat[3], stTriLanes); }
            // launch timeline (100 MHz ticks): earliest wave start, earliest "queue empty", latest wave exit
            // MODE 2 keeps it in kStatStripes copies and nothing else unless stat[5] asks for the pooled histograms (PTAMD_TPOOL=1): atomics from
            // every leaving wave on a handful of words stretched the very tail they were meant to measure (r03_b27.log: a launch of 100 us
            // became one of 287 us)
            unsigned long long* tl = MODE == 2 ? stat + kStatStripeOff / 8 + 3 * ((size_t)statLaunch * kStatStripes + (blockIdx.x % kStatStripes)) : stat + 8 + 3 * (size_t)statLaunch;
            const unsigned long long tEnd = __builtin_amdgcn_s_memrealtime();
            atomicMax(&tl[0], ~stT0); if (stTExh) atomicMax(&tl[1], ~stTExh); atomicMax(&tl[2], tEnd);
            if (blockIdx.x == 0 && threadIdx.x == 0) stat[8 + 3 * 2700 + 32 + statLaunch] = n;      // rays of this launch
            if (MODE == 2 && stat[5] == 0ull) return;
            // distribution of wave exit times over the launch, all launches pooled (absolute: 32 us bins)
            unsigned long long* hist = stat + 8 + 3 * 2700;
            const unsigned long long dtk = (tEnd - stT0) / 3200ull;      // 32 us bins (100 MHz ticks)
            atomicAdd(&hist[dtk < 31 ? dtk : 31], 1ull);
            if (MODE == 2) {
                // per-wave work, all launches pooled: trips per wave (64 bins of 4), trips after the wave found the queue dry (32 bins of 2: in the
                // slots of MODE 1's per-ray histograms), and rays per wave summed into stat[7] / trips into stat[0] for averages
                atomicAdd(&stat[8 + 3 * 2700 + 32 + 2700 + (stTrips >= 252 ? 63 : stTrips >> 2)], 1ull);
                atomicAdd(&stat[8 + 3 * 2700 + 32 + 2700 + 64 + (stTripsDry >= 62 ? 31 : stTripsDry >> 1)], 1ull);
                atomicAdd(&stat[7], r); atomicAdd(&stat[0], (unsigned long long)stTrips); atomicAdd(&stat[2], (unsigned long long)stTripsDry);
            }
            if (STAT) { atomicAdd(&stat[4], stRefills); atomicAdd(&stat[5], stRefillLanes); atomicAdd(&stat[6], stNoRayLanes); atomicAdd(&stat[7], r); }
            if (STAT) for (int k = 0; k < 5; k++) atomicAdd(&stat[8 + 3 * 2700 + 32 + 2700 + 64 + 32 + k], stClk[k]);
        }
    }
}

// ---------------------------------------------------------------------------------------
// wf_shade: one thread per live stream, one bounce.
// ---------------------------------------------------------------------------------------
// PHASE (round 3, "early shade"): the launch tail of wf_trace — the last rays of a launch finishing in ever emptier waves — leaves the
// chip idle for ~0.3 ms per iteration, which is most of an iteration for one rank of an 8-way tile split.  With PHASE 1 / 2 the shade
// step of an iteration is cut in two launches:
//   PHASE 1 (early)  runs on a second HIP stream BESIDE the draining wf_trace: its workgroups get wave slots as traversal waves leave.
//                    A stream is shaded only if every ray it is waiting for is already back — wf_trace publishes a hit with a
//                    device-scope store, the slot held kNotReady since the ray was emitted (MARK) and is read here with a device-scope
//                    load (the XCDs' L2s are not coherent with each other) —, otherwise it is left alone.  Nothing is appended here: the
//                    outcome of a shaded stream (alive / which rays it emitted) is parked in res[list position].  No waiting, no
//                    spinning: a stream whose rays are not back is simply skipped.
//   PHASE 2 (rest)   runs after both have finished: shades the streams phase 1 skipped and does ALL the appends, in list order — so the
//                    live list and the ray queues keep stream order exactly as with one launch.
// A stream goes through the same shade_step either way: result-neutral.
enum : uint32_t { R_ALIVE = 1, R_EMIT0 = 2, R_EMIT1 = 4, R_EMIT2 = 8, R_SHORT0 = 16, R_SHORT1 = 32, R_SHORT2 = 64, R_DONE = 128 };      // res[]: what phase 1 did with a list position

PT_DEV float2 load_hit_coherent(const float2* p)
{
    const unsigned long long v = __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return make_float2(__uint_as_float((uint32_t)v), __uint_as_float((uint32_t)(v >> 32)));
}

template <int WAVES, bool TWO, int PHASE = 0, bool MARK = false>
__global__ __launch_bounds__(WAVES * 256, WAVES)
void wf_shade(DevScene sc, DevCamera cam, DevParams prm, WfBuf b, int slotIn, int slotOut, int slotClear, int listIn)
{
    static_assert(PHASE == 0 || MARK, "the two-phase step needs the not-ready marks");
    const uint32_t nIn = b.cnt[slotIn].nActive;
    if (PHASE != 1 && blockIdx.x == 0) for (int k = threadIdx.x; k < kWfSlotBytes / 4; k += blockDim.x) ((uint32_t*)&b.cnt[slotClear])[k] = 0;
    if ((uint32_t)blockIdx.x * blockDim.x >= nIn) return;
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    const bool have = idx < nIn;
    bool alive = false, emit[kRayKinds] = {false, false, false};
    uint32_t sid = 0, resume = 0;      // resume: the queued rays are suspended traversals (wf_trace then reads their records)
    uint32_t cls = 0;                  // bit k: the ray of kind k this step emitted is short (queued from the back of its queue)
    bool step = have;
    if (have) {
        // While no stream has retired yet (more than half of a render's iterations) every stream is alive, so list position idx can
        // simply take stream idx: one dependent fetch level less for the whole step (the list itself is in stream order only inside the
        // blocks that appended to it; any one-to-one assignment of streams to lanes gives the same result).
        sid = (nIn == (uint32_t)prm.n_units * 64u) ? idx : ld_s(&b.active[listIn][idx]);
        if (PHASE == 2) {
            const uint32_t r = b.res[idx];
            if (r & R_DONE) {      // shaded by phase 1: only the appends are left
                step = false;
                alive = (r & R_ALIVE) != 0; emit[0] = (r & R_EMIT0) != 0; emit[1] = (r & R_EMIT1) != 0; emit[2] = (r & R_EMIT2) != 0;
                cls = (r / R_SHORT0) & 7u;
            }
        }
    }
    if (step) {
        SState st;
        float2 hitP, hitS, hitA;
        if (PHASE == 1) { hitP = load_hit_coherent(&b.hit[0][sid]); hitS = load_hit_coherent(&b.hit[1][sid]); hitA = load_hit_coherent(&b.hit[2][sid]); }
        else { hitP = ld_s(&b.hit[0][sid]); hitS = ld_s(&b.hit[1][sid]); hitA = ld_s(&b.hit[2][sid]); }      // same fetch level as the state
        load_state(b, sid, st);
        // a ray of this stream is still being traversed (time-sliced): wait one iteration
        const int pendP = (st.flags & F_PATH) ? __float_as_int(hitP.y) : -1, pendS = (st.flags & F_SHADOW) ? __float_as_int(hitS.y) : -1;
        const int pendA = (st.flags & F_SHADOWA) ? __float_as_int(hitA.y) : -1;
        if (PHASE == 1 && (pendP <= -2 || pendS <= -2 || pendA <= -2)) {
            // phase 1: a ray is not back yet (kNotReady), or a traversal is suspended (its slot keeps the record number until wf_trace
            // resumes it, so the slot cannot tell "back" from "not yet"): phase 2 takes the stream
            step = false;
        } else if (pendP <= -2 || pendS <= -2 || pendA <= -2) {
            alive = true; emit[0] = pendP <= -2; emit[1] = pendS <= -2; emit[2] = pendA <= -2; resume = kResumeBit;
        } else {
            const bool done = shade_step_t<TWO>(sc, cam, prm, b, sid, st, hitP, hitS, hitA);
            if (done) {
                write_mean(b, prm, sid, st);
            } else {
                const uint32_t nf = st.flags;
                store_state(b, sid, st);
                alive = true;
                emit[0] = (nf & F_PATH) != 0; emit[1] = (nf & F_SHADOW) != 0; emit[2] = (nf & F_SHADOWA) != 0;
                cls = st.cls;
            }
        }
    }
    // MARK: the hit slot of every ray this step emitted says "not traced yet" until wf_trace publishes its hit (a suspended traversal
    // that is re-queued keeps its slot: it holds the record number).  Written here, at the end, where nothing else is live.
    if (MARK && step && !resume) {
#pragma unroll
        for (int k = 0; k < kRayKinds; k++) if (emit[k]) b.hit[k][sid] = make_float2(0.f, __int_as_float(kNotReady));
    }
    if (PHASE == 1) {
        if (have) b.res[idx] = (uint8_t)(step ? (R_DONE | (alive ? R_ALIVE : 0u) | (emit[0] ? R_EMIT0 : 0u) | (emit[1] ? R_EMIT1 : 0u) | (emit[2] ? R_EMIT2 : 0u) | (cls & 7u) * R_SHORT0) : 0u);
        return;
    }
    // a re-queued suspended traversal is long by definition: cls = 0 for it (it never went through the step)
    const bool s0 = (cls & 1u) != 0, s1 = (cls & 2u) != 0, s2 = (cls & 4u) != 0;
    const uint32_t topIdx = (uint32_t)(b.hit[1] - b.hit[0]) - 1u;      // n16 - 1
    const bool e[kLists] = {alive, emit[0] && !s0, emit[1] && !s1, emit[2] && !s2, emit[0] && s0, emit[1] && s1, emit[2] && s2};
    uint32_t* const c[kLists] = {&b.cnt[slotOut].nActive, &b.cnt[slotOut].nRays[0][0], &b.cnt[slotOut].nRays[1][0], &b.cnt[slotOut].nRays[2][0],
                                 &b.cnt[slotOut].nRays[0][kShortWord], &b.cnt[slotOut].nRays[1][kShortWord], &b.cnt[slotOut].nRays[2][kShortWord]};
    uint32_t* const l[kLists] = {b.active[listIn ^ 1], b.rq[0], b.rq[1], b.rq[2], b.rq[0], b.rq[1], b.rq[2]};
    const uint32_t ids[kLists] = {sid, sid | resume, sid | resume, sid | resume, sid, sid, sid};
    const uint32_t top[kLists] = {0u, 0u, 0u, 0u, topIdx, topIdx, topIdx};
    block_append<kLists>(e, ids, c, l, top);
}

// ---------------------------------------------------------------------------------------
// wf_drain: run every remaining stream to its end inside one launch.
// The pipeline's bounce iterations each carry a device-wide dependency, and the last third of
// them serve only the few pixels whose every path runs the full depth (a render of 256 spp
// needs ~1800 iterations while the average stream is done after ~730).  Once few streams are
// left, per-launch latency — not throughput — sets the pace, so they are handed to this kernel:
// one lane per stream (every 2^spreadShift-th lane: the kernel is bound by latency, so thinner waves are faster), state in
// registers, rays traced in place (quad_step on the 4-wide tree, or trace_closest on the binary one when that walk would not fit
// the per-lane stack), same shade_step.  Pending time-sliced traversals are simply redone (they are deterministic).
// On by default for the last 80,000 live streams of a render (pt_api.hip: drain_below).
// ---------------------------------------------------------------------------------------
constexpr int kDrainQuadStack = 40;      // per-lane stack entries of wf_drain's 4-wide walk (40 KB of LDS per workgroup): trees up to 12 levels, the config scenes' depth
#ifndef DRAIN_MINBLOCKS
#define DRAIN_MINBLOCKS 1      // workgroups per CU the register allocation of wf_drain must allow: 1 = free (189 VGPRs with the 4-wide walk: 2 waves/SIMD), 3 = 168 VGPRs (16 spilled)
#endif
// One step of one lane's walk through the 4-wide quantised tree, for wf_drain: wf_trace's node step (the box arithmetic is that kernel's,
// statement by statement) or its pair-record leaf test, whichever `cur` asks for; returns true when the ray is finished.  The closest
// hit does not depend on the order of tests (tie rule), and a shadow ray may stop at any hit below `stopBelow` (pt_stream.h:
// shadow_stop_t) exactly as it does there.  The caller guarantees 3 * quad_depth + 2 <= kDrainQuadStack.
PT_DEV bool quad_step(const DevScene& sc, const f3& org, const f3& dir, const f3& inv, float cscale, bool degenerate, float stopBelow,
                      int* stack, int& cur, int& sp, float& bestT, int& bestPrim)
{
    if (cur >= 0) {
        const uint4* np = sc.quad + 4 * (size_t)cur;
        const uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3];
        const float Ax = inv.x * __uint_as_float(n0.w), Ay = inv.y * __uint_as_float(n3.z), Az = inv.z * __uint_as_float(n3.w);
        const float Bx = (__uint_as_float(n0.x) - org.x) * inv.x;
        const float By = (__uint_as_float(n0.y) - org.y) * inv.y;
        const float Bz = (__uint_as_float(n0.z) - org.z) * inv.z;
        const float kSl = 9.5367431640625e-7f;                           // 2^-20
        const float sx = (__builtin_fabsf(Bx) + 255.f * __builtin_fabsf(Ax)) * kSl;
        const float sy = (__builtin_fabsf(By) + 255.f * __builtin_fabsf(Ay)) * kSl;
        const float sz = (__builtin_fabsf(Bz) + 255.f * __builtin_fabsf(Az)) * kSl;
        const float Bnx = Bx - sx, Bfx = Bx + sx, Bny = By - sy, Bfy = By + sy, Bnz = Bz - sz, Bfz = Bz + sz;
        const uint32_t mx = (uint32_t)(__float_as_int(inv.x) >> 31), my = (uint32_t)(__float_as_int(inv.y) >> 31), mz = (uint32_t)(__float_as_int(inv.z) >> 31);
        const uint32_t swx = (n2.x ^ n2.w) & mx, swy = (n2.y ^ n3.x) & my, swz = (n2.z ^ n3.y) & mz;
        const uint32_t nqx = n2.x ^ swx, fqx = n2.w ^ swx, nqy = n2.y ^ swy, fqy = n3.x ^ swy, nqz = n2.z ^ swz, fqz = n3.y ^ swz;
        const float cullT = bestT * cscale;
        int key[4];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const float tnx = __builtin_fmaf((float)((nqx >> (8 * k)) & 0xffu), Ax, Bnx);
            const float tny = __builtin_fmaf((float)((nqy >> (8 * k)) & 0xffu), Ay, Bny);
            const float tnz = __builtin_fmaf((float)((nqz >> (8 * k)) & 0xffu), Az, Bnz);
            const float tfx = __builtin_fmaf((float)((fqx >> (8 * k)) & 0xffu), Ax, Bfx);
            const float tfy = __builtin_fmaf((float)((fqy >> (8 * k)) & 0xffu), Ay, Bfy);
            const float tfz = __builtin_fmaf((float)((fqz >> (8 * k)) & 0xffu), Az, Bfz);
            const float tn = __builtin_fmaxf(__builtin_fmaxf(tnx, tny), __builtin_fmaxf(tnz, 0.f));
            const float tf = __builtin_fminf(__builtin_fminf(tfx, tfy), __builtin_fminf(tfz, cullT));
            key[k] = (tn <= tf) ? __float_as_int(tn) : 0x7fffffff;
        }
        int k0 = key[0], k1 = key[1], k2 = key[2], k3 = key[3], r0 = (int)n1.x, r1 = (int)n1.y, r2 = (int)n1.z, r3 = (int)n1.w;
#define PT_CE(ka, ra, kb, rb) { const bool sw = ka > kb; const int tk = sw ? kb : ka, tr = sw ? rb : ra; kb = sw ? ka : kb; rb = sw ? ra : rb; ka = tk; ra = tr; }
        PT_CE(k0, r0, k1, r1) PT_CE(k2, r2, k3, r3) PT_CE(k0, r0, k2, r2) PT_CE(k1, r1, k3, r3) PT_CE(k1, r1, k2, r2)
#undef PT_CE
        if (k0 != 0x7fffffff) {
            if (k3 != 0x7fffffff) { stack[sp * 64] = r3; sp++; }
            if (k2 != 0x7fffffff) { stack[sp * 64] = r2; sp++; }
            if (k1 != 0x7fffffff) { stack[sp * 64] = r1; sp++; }
            cur = r0;
            return false;
        }
    } else {
        const int code = ~cur;
        int first = code >> 3, cnt = code & 7;
        while (cnt > 0) { tri_test_pairrec(sc, first, cnt > 1, org, dir, inv, degenerate, bestT, bestPrim); first += 2; cnt -= 2; }
        if (bestPrim >= 0 && bestT < stopBelow) return true;
    }
    if (sp == 0) return true;
    sp--;
    cur = stack[sp * 64];
    return false;
}

template <bool QUAD>      // QUAD: walk the 4-wide tree (quad_step); the host picks it when the walk fits the per-lane stack
__global__ __launch_bounds__(kBlockThreads, DRAIN_MINBLOCKS)
void wf_drain(DevScene sc, DevCamera cam, DevParams prm, WfBuf b, int slotIn, int listIn, int spreadShift)
{
    __shared__ int lds_stack[kWavesPerBlock][(QUAD ? kDrainQuadStack : kStackDepth) * 64];
    const uint32_t nIn = b.cnt[slotIn].nActive;
    // spreadShift: only every 2^s-th lane carries a stream.  The kernel is bound by latency (a wave steps at the pace of its slowest
    // lane, every bounce), and the chip is far from full at this point: thinner waves wait for the maximum of fewer paths
    const uint32_t t = blockIdx.x * (uint32_t)kBlockThreads + threadIdx.x;
    if (t & ((1u << spreadShift) - 1u)) return;
    const uint32_t idx = t >> spreadShift;
    if (idx >= nIn) return;
    int* stack = &lds_stack[threadIdx.x >> 6][threadIdx.x & 63];
    const uint32_t sid = b.active[listIn][idx];
    SState st;
    load_state(b, sid, st);
    for (;;) {
        float2 hitP = make_float2(0.f, __int_as_float(-1)), hitS = hitP, hitA = hitP;
        TraceStats ts{0, 0, 0};
        if (QUAD) {
            // The rays of this bounce (second-to-last shadow ray, shadow ray, path ray: any subset) in ONE flat loop: a lane that has finished a ray
            // sets up its next one inside the loop, so the wave waits for the lane with the most steps in all — not, as with one
            // loop per ray kind, for the slowest lane of each kind in turn.
            int todo = ((st.flags & F_SHADOWA) ? 1 : 0) | ((st.flags & F_SHADOW) ? 2 : 0) | ((st.flags & F_PATH) ? 4 : 0);
            f3 org(0.f, 0.f, 0.f), dir(0.f, 0.f, 1.f), inv(0.f, 0.f, 0.f);
            float cscale = 0.f, bestT = 0.f, stopBelow = 0.f;
            bool degenerate = false;
            int bestPrim = -1, cur = 0, sp = 0, kind = -1;
            for (;;) {
                if (kind < 0) {
                    if (todo == 0) break;
                    kind = __builtin_ctz((unsigned)todo); todo &= todo - 1;
                    if (kind == 0) {
                        const float4 ao = b.ray_o[2][sid], ad = b.ray_d[2][sid];
                        org = f3(ao.x, ao.y, ao.z); dir = f3(ad.x, ad.y, ad.z); bestT = ao.w; stopBelow = ad.w;
                    } else if (kind == 1) { org = st.shO; dir = st.shD; bestT = st.shTmax; stopBelow = shadow_stop_t(st.shO, st.shTmax); }
                    else { org = st.pathO; dir = st.pathD; bestT = 999999.f; stopBelow = -__builtin_inff(); }
                    ray_setup(dir, inv, cscale, degenerate);
                    bestPrim = -1; cur = 0; sp = 0;
                }
                if (quad_step(sc, org, dir, inv, cscale, degenerate, stopBelow, stack, cur, sp, bestT, bestPrim)) {
                    for (int s = 0; s < sc.n_spheres; s++) {      // spheres, in order, against the triangles' closest t (CudaUtil.cuh:137-145)
                        const float4 c = sc.spheres[4 * s];
                        float root;
                        if (sphere_root(f3(c.x, c.y, c.z), c.w, org, dir, bestT, root)) { bestT = root; bestPrim = sc.n_tris + s; }
                    }
                    const float2 h = make_float2(bestT, __int_as_float(bestPrim));
                    if (kind == 0) hitA = h; else if (kind == 1) hitS = h; else hitP = h;
                    kind = -1;
                }
            }
        } else {
        if (st.flags & F_SHADOWA) {
            const float4 ao = b.ray_o[2][sid], ad = b.ray_d[2][sid];
            float t; const int prim = trace_closest<false>(sc, f3(ao.x, ao.y, ao.z), f3(ad.x, ad.y, ad.z), ao.w, stack, t, ts); hitA = make_float2(t, __int_as_float(prim));
        }
        if (st.flags & F_SHADOW) { float t; const int prim = trace_closest<false>(sc, st.shO, st.shD, st.shTmax, stack, t, ts); hitS = make_float2(t, __int_as_float(prim)); }
        if (st.flags & F_PATH) { float t; const int prim = trace_closest<false>(sc, st.pathO, st.pathD, 999999.f, stack, t, ts); hitP = make_float2(t, __int_as_float(prim)); }
        }
        if (shade_step(sc, cam, prm, b, sid, st, hitP, hitS, hitA)) break;
    }
    write_mean(b, prm, sid, st);
}

// ---------------------------------------------------------------------------------------
// Parity hooks for the integrator's sub-functions (include/pt_api.h: pt_dbg_pixel_dir, pt_dbg_nee): the SAME device
// functions the render kernels call (pt_shade.h), run on rows of inputs so that tests can compare them with the oracle
// one function at a time.
// ---------------------------------------------------------------------------------------
__global__ void dbg_ray_setup(const float* __restrict__ dir3, int n, float* __restrict__ out5)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    f3 inv; float cscale; bool deg;
    ray_setup(f3(dir3[3 * i], dir3[3 * i + 1], dir3[3 * i + 2]), inv, cscale, deg);
    out5[5 * i] = inv.x; out5[5 * i + 1] = inv.y; out5[5 * i + 2] = inv.z; out5[5 * i + 3] = cscale; out5[5 * i + 4] = deg ? 1.f : 0.f;
}
__global__ void dbg_pixel_dir(DevCamera cam, const int* __restrict__ pxpypass, int n, float* __restrict__ out8)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int px = pxpypass[3 * i], py = pxpypass[3 * i + 1], pass = pxpypass[3 * i + 2];
    Rng rng;
    rng.init((uint64_t)(int64_t)(py * cam.W + px + pass * cam.W * cam.H));      // srcs/pathtracer.cu:70-71
    float u1, u2;
    const f3 d = pixel_direction(cam, px, py, rng, u1, u2);
    float* o = out8 + (size_t)i * 8;
    o[0] = u1; o[1] = u2; o[2] = d.x; o[3] = d.y; o[4] = d.z; o[5] = rng.uniform(); o[6] = 0.f; o[7] = 0.f;      // o[5]: the next draw (RNG position)
}

// in5: surface point p.xyz | seed lo | seed hi (uint32 bits).  out12: light index | lightP.xyz | pdfLight | cosA | shadow-ray t_max |
// closest-hit primitive of the shadow ray (int bits) | Le.xyz (GetLightColor) | next uniform draw.
__global__ __launch_bounds__(kBlockThreads)
void dbg_nee(DevScene sc, const float* __restrict__ in5, int n, float* __restrict__ out12)
{
    __shared__ int lds_stack[kWavesPerBlock][kStackDepth * 64];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int* stack = &lds_stack[threadIdx.x >> 6][threadIdx.x & 63];
    if (i >= n) return;
    const float* r = in5 + (size_t)i * 5;
    const f3 p(r[0], r[1], r[2]);
    Rng rng;
    rng.init(((uint64_t)__float_as_uint(r[4]) << 32) | (uint64_t)__float_as_uint(r[3]));
    const NeeSample ns = nee_sample(sc, rng, p);
    const float tmax = length(ns.toL) + 1.0f;                               // GetLightColor, CudaUtil.cuh:152-157
    float t; TraceStats ts{0, 0, 0};
    const int prim = trace_closest<false>(sc, p, ns.wl, tmax, stack, t, ts);
    const f3 Le = nee_light_color(p, ns.wl, ns.lightP, t, prim, prim_emittance(sc, prim < 0 ? 0 : prim));
    float* o = out12 + (size_t)i * 12;
    o[0] = __int_as_float(ns.li); o[1] = ns.lightP.x; o[2] = ns.lightP.y; o[3] = ns.lightP.z; o[4] = ns.pdfLight; o[5] = ns.cosA;
    o[6] = tmax; o[7] = __int_as_float(prim); o[8] = Le.x; o[9] = Le.y; o[10] = Le.z; o[11] = rng.uniform();
}

}  // namespace ptd

// ---------------------------------------------------------------------------------------
// Host driver
// ---------------------------------------------------------------------------------------
extern "C" {

// entries a ray's traversal stack can hold (LDS + global overflow); a 4-wide walk needs at most 3 per level + 2
int ptk_wf_stack_capacity(void) { return ptd::kWfLdsStack + ptd::kWfOvfLevels; }

// ---- work buffer: [ staging (all streams) | cohort 0 | cohort 1 | ... ] ----------------------
static size_t cohort_bytes(size_t nStreams, int traceBlocks)
{
    const size_t n16 = (nStreams + 3) & ~(size_t)3;
    size_t b = 0;
    b += n16 * 16 * 11;                       // 11 float4 state arrays
    b += n16 * 16 * 2 * ptd::kRayKinds;       // ray_o/ray_d per kind
    b += n16 * 8 * (ptd::kRayKinds + 1);      // hits per kind + the cached camera-ray hit
    b += n16 * 4 * (2 + ptd::kRayKinds);      // active x2, one ray queue per kind
    b += 3 * ptd::kWfSlotBytes; // counters
    b += (size_t)traceBlocks * 256 * ptd::kWfOvfLevels * 4;
    b += 2 * ((nStreams / 4 + 1024) * ptd::kSuspInts * 4 + 16);
    b += n16 + 16;                            // res: one byte per live-list position (early shade)
    return b + 512;
}
static size_t staging_bytes(size_t nStreams) { return ((nStreams * 12 + 16) + 255) & ~(size_t)255; }

// Streams can be split into cohorts that run the pipeline concurrently on separate HIP streams
// (contiguous unit ranges, so each is a whole number of (tile, pass) units).  Measured on MI355X:
// before the queue heads were sharded two cohorts gained 8 % on 16.6M streams (one cohort's
// shade kernel overlapping the other's trace kernel); with sharded heads one cohort is as fast
// (927 vs 920 Msamples/s) and small renders lose, so the default is one.  PTAMD_COHORTS=2..4
// turns it on for experiments.
int ptk_wf_cohorts(size_t nUnits)
{
    static const int forced = getenv("PTAMD_COHORTS") ? atoi(getenv("PTAMD_COHORTS")) : 0;
    int c = forced > 0 ? forced : 1;
    if (c > 4) c = 4;
    if ((size_t)c > nUnits) c = (int)(nUnits ? nUnits : 1);
    return c;
}

size_t ptk_wf_work_bytes(size_t nUnits, int traceBlocks)
{
    const int C = ptk_wf_cohorts(nUnits);
    const size_t per = (nUnits + C - 1) / C;
    return staging_bytes(nUnits * 64) + (size_t)C * cohort_bytes(per * 64, traceBlocks) + 256;
}

static void carve(char* p, size_t nStreams, int traceBlocks, ptd::WfBuf& b)
{
    const size_t n16 = (nStreams + 3) & ~(size_t)3;
    auto take = [&](size_t bytes) { char* q = p; p += (bytes + 15) & ~(size_t)15; return q; };
    b.rng0 = (uint4*)take(n16 * 16); b.rng1 = (uint4*)take(n16 * 16);
    b.weight = (float4*)take(n16 * 16); b.rad = (float4*)take(n16 * 16); b.pix = (float4*)take(n16 * 16);
    b.dir0 = (float4*)take(n16 * 16); b.wb = (float4*)take(n16 * 16); b.lp = (float4*)take(n16 * 16);
    b.radA = (float4*)take(n16 * 16); b.wbA = (float4*)take(n16 * 16); b.lpA = (float4*)take(n16 * 16);
    for (int k = 0; k < ptd::kRayKinds; k++) { b.ray_o[k] = (float4*)take(n16 * 16); b.ray_d[k] = (float4*)take(n16 * 16); }
    for (int k = 0; k < ptd::kRayKinds; k++) b.hit[k] = (float2*)take(n16 * 8);
    b.hit0 = (float2*)take(n16 * 8);
    for (int k = 0; k < 2; k++) b.active[k] = (uint32_t*)take(n16 * 4);
    for (int k = 0; k < ptd::kRayKinds; k++) b.rq[k] = (uint32_t*)take(n16 * 4);
    b.cnt = (ptd::WfCounters*)take(3 * ptd::kWfSlotBytes);
    b.ovf = (int*)take((size_t)traceBlocks * 256 * ptd::kWfOvfLevels * 4);
    b.suspCap = (uint32_t)(nStreams / 4 + 1024);
    for (int k = 0; k < 2; k++) b.susp[k] = (int*)take((size_t)b.suspCap * ptd::kSuspInts * 4);
    b.res = (uint8_t*)take(n16);
}

const float* ptk_wf_staging(void* work) { return (const float*)work; }

// Scheduling constants of the pipeline, read from the environment ONCE per process (tuning and A/B sweeps only — none of them can
// change a result).  Defaults are the measured optima on MI355X.
struct WfTuning {
    int chunkShift;      // PTAMD_CS   chunk  = clamp(n >> CS, 16, kWfChunk) ray ids per queue access
    int guideShift;      // PTAMD_GS   guided = the chunk shrinks to (rays left in the shard) >> GS
    int budgetShift;     // PTAMD_BS   node budget = clamp(n >> BS, BM, 1024) steps per launch
    int budgetMin;       // PTAMD_BM
    int refillMin;       // PTAMD_RF   idle lanes that trigger a refill
    int triTrig;         // PTAMD_TT   parked leaves that force a triangle trip (64 = only when blocked rays outnumber walking ones)
    int helpShards;      // PTAMD_HELP shards a wave tries (its own included) before it takes the queue to be dry: 4 (16 = all: every wave then
                         // spends 16 returning atomics on hot words at the end of every launch; the shards are interleaved and equally long, so
                         // there is little to help with: 16 -> 4 is +1 % on configs[2], +3 % on configs[1], +4.5 % for an 8-way rank, r03_b20.log)
    int lateBudget;      // PTAMD_LB   node steps after which a ray is suspended once the queue is dry and its wave holds at most two rays: 64 (0 = never).
                         // An 8-way rank's big launches wait 15 us on average (20 % of them > 25 us, 3 % > 100 us) for the latest of their 64 stripes of
                         // waves — one ray of several hundred steps, alone on its SIMD (tools/straggler_cost.py, r03_b41.log); cut there it goes on in
                         // the next launch among full waves, and no iteration is added: 4-way rank +3 %, 8-way +0.4 ... +2.5 %, configs[2] / [3] / [4]
                         // +0.3 / +0.5 / +1.2 % (32: -4 %, 48 / 96 / 128 within 1 % of 64; r03_b42.log, r03_b43.log)
    int topNodes;        // PTAMD_TOP  quad nodes staged in LDS (TRACE_TOP_NODES builds only)
    // wf_shade: 4 waves/SIMD (126 VGPRs, nothing spilled since the library is built without the SLP vectoriser) in 512-thread workgroups =
    // two per CU; other shapes: 256 threads -4 %, 384 / 768 -13 %, 1024 -8 %, 3 waves/SIMD -9...-13 % (r02_t16_shade_shapes_after_noslp.log)
    int shadeWaves;      // PTAMD_SW
    int shadeThreads;    // PTAMD_ST
    // wf_shade: may a stream whose path has just ended start its next sample in the same step (a second trip through the bounce code)?
    // It saves one iteration per sample but doubles the step's dependent chain: shadeRounds 0 / 1 forces it (pt_set_shade_rounds,
    // PTAMD_TR), -1 switches at trStreams live streams.  The result does not depend on it (pt_stream.h: shade_step_t).
    uint32_t trStreams;  // PTAMD_TRS
    int earlyThreads;    // PTAMD_EST  threads per workgroup of wf_shade's early phase (64: one free wave slot is enough; 512 / 256 / 128 / 64 -> 0.488 / 0.481 / 0.473 / 0.472 s for an 8-way rank)
    int earlyPrio;       // PTAMD_EPRIO issue priority of the traversal waves while the early phase runs beside them (no effect measured)
    bool pubOnly;        // PTAMD_EPUB  A/B: device-scope hit stores and marks, but no early phase
    int drainSpread;     // PTAMD_DSPREAD  wf_drain: at most every 2^this-th lane carries a stream (3)
    int drainQuad;       // PTAMD_DQUAD    wf_drain walks the 4-wide tree (1; 0 = the binary tree of the one-kernel mode, A/B)
    bool tracePool;      // PTAMD_TPOOL with PTAMD_TSTAT=2: also the pooled per-wave histograms (tools/wave_exit_hist.py) — their atomics lengthen the launch tail
    int traceDump;       // PTAMD_TDUMP with PTAMD_TSTAT=2: the wf_trace launch (iteration) whose waves are dumped one by one (pt_dbg_trace_timeline -3003 / -3004)
    int traceStat;       // PTAMD_TSTAT 1 trip counters + histograms (slower build), 2 timeline only (production code path), 3 trip counters + section clocks
};
static const WfTuning& wf_tuning()
{
    static const WfTuning t = [] {
        auto num = [](const char* name, long long def) -> long long { const char* v = getenv(name); return v ? atoll(v) : def; };
        auto threads = [&](const char* name, int def) { const long long v = num(name, def); return (v >= 64 && v <= ptd::kShadeThreads) ? (int)(v & ~63LL) : def; };
        WfTuning w;
        w.chunkShift = (int)num("PTAMD_CS", 12); w.guideShift = (int)num("PTAMD_GS", 9);
        w.budgetShift = (int)num("PTAMD_BS", 14); w.budgetMin = (int)num("PTAMD_BM", ptd::kWfBudget);
        w.refillMin = (int)num("PTAMD_RF", ptd::kWfRefill); w.triTrig = (int)num("PTAMD_TT", 64); w.topNodes = (int)num("PTAMD_TOP", ptd::kTopNodes);
        w.lateBudget = (int)num("PTAMD_LB", 64);
        w.helpShards = (int)num("PTAMD_HELP", 4); if (w.helpShards < 1) w.helpShards = 1; if (w.helpShards > ptd::kWfShards) w.helpShards = ptd::kWfShards;
        w.shadeWaves = (int)num("PTAMD_SW", 4); w.shadeThreads = threads("PTAMD_ST", 512);
        w.trStreams = (uint32_t)num("PTAMD_TRS", 4000000);
        w.earlyThreads = threads("PTAMD_EST", 64); w.earlyPrio = (int)num("PTAMD_EPRIO", 0); w.pubOnly = num("PTAMD_EPUB", 0) != 0;
        w.traceStat = (int)num("PTAMD_TSTAT", 0);
        w.traceDump = (int)num("PTAMD_TDUMP", -1);
        w.tracePool = num("PTAMD_TPOOL", 0) != 0;
        w.drainQuad = num("PTAMD_DQUAD", 1) != 0 ? 1 : 0;
        w.drainSpread = (int)num("PTAMD_DSPREAD", 3); if (w.drainSpread < 0) w.drainSpread = 0; if (w.drainSpread > 5) w.drainSpread = 5;
        return w;
    }();
    return t;
}

// One cohort's pipeline on its own stream.  Blocks the calling host thread until the cohort has
// drained (it polls the live-stream count every 16..64 iterations).
static hipError_t run_cohort(int device, const ptd::DevScene* sc, const ptd::DevCamera* cam, ptd::DevParams prm,
                             ptd::WfBuf b, int traceBlocks, uint32_t* h_cnt, hipStream_t stream,
                             hipEvent_t* trace_ev, int trace_ev_pairs, int* trace_ev_used, int drainBelow, int shadeRounds, int* iters_out, unsigned long long* traceStat,
                             hipStream_t aux, hipEvent_t* evOvl, int earlyBelow)
{
    using namespace ptd;
    hipError_t e;
    if ((e = hipSetDevice(device)) != hipSuccess) return e;
    const size_t nStreams = (size_t)prm.n_units * 64;
    if ((e = hipMemsetAsync(b.cnt, 0, 3 * kWfSlotBytes, stream)) != hipSuccess) return e;
    const int nb = (int)((nStreams + 255) / 256);
    hipLaunchKernelGGL(wf_init, dim3(nb), dim3(256), 0, stream, *sc, *cam, prm, b, (uint32_t)nStreams);
    const int ovfStride = traceBlocks * 256;
    const int tb = traceBlocks < nb ? traceBlocks : nb;
    // every sample needs at most max_bounce + (max_refract + 2) bounces, +1 iteration to retire
    // (time-sliced rays add iterations; 64x is far beyond anything a finite tree can need)
    const long long hardCap = ((long long)prm.spp_per_pass * (prm.max_bounce + prm.max_refract + 3) + 8) * 64;
    const WfTuning& tn = wf_tuning();
    const int guideShift = tn.guideShift, budgetShift = tn.budgetShift, budgetMin = tn.budgetMin, shadeWaves = tn.shadeWaves, shadeThreads = tn.shadeThreads;
    const int earlyPrio = tn.earlyPrio, earlyThreads = tn.earlyThreads, refillMin = tn.refillMin, triTrig = tn.triTrig, chunkShift = tn.chunkShift, topNodes = tn.topNodes;
    const bool pubOnly = tn.pubOnly, traceStatClk = tn.traceStat == 3, traceStatFull = tn.traceStat == 1;
    const uint32_t trStreams = tn.trStreams;
    if (traceStat && tn.traceStat == 2 && tn.tracePool) {
        static const unsigned long long one = 1ull;
        if ((e = hipMemcpyAsync(traceStat + 5, &one, 8, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    }
    if (traceStat && tn.traceStat == 2 && tn.traceDump >= 0) {
        // diagnostic: wf_trace dumps a record per wave for this one launch (the word was cleared with the rest of the buffer by the caller)
        static unsigned long long dumpWord; dumpWord = (unsigned long long)tn.traceDump + 1ull;
        if ((e = hipMemcpyAsync(traceStat + 6, &dumpWord, 8, hipMemcpyHostToDevice, stream)) != hipSuccess) return e;
    }
    int it = 0;
    int poll = 16;
    // streams only ever retire, so the live count of the last poll bounds every later one: the shade grid
    // shrinks with it instead of launching thousands of workgroups that find nothing to do
    uint32_t liveBound = (uint32_t)nStreams;
    for (;;) {
        for (int k = 0; k < poll; k++, it++) {
            const int sIn = it % 3, sOut = (it + 1) % 3, sClr = (it + 2) % 3;
            const bool timed = trace_ev && it < trace_ev_pairs;
            if (timed) (void)hipEventRecord(trace_ev[3 * it], stream);
            // early shade (wf_shade PHASE 1 / 2): for a render of few enough streams that the traversal's launch tail is a large part of
            // every iteration (one rank of an 8-way tile split), the shade step starts on `aux` beside the draining wf_trace and the rest
            // follows both (result-neutral).  Decided once per render: a large render gains nothing from it in its last iterations.
            // (not below ~1/8 of the limit either: a render that small is bound by launch latency, and this adds a launch and two waits per iteration)
            const bool early = aux != nullptr && earlyBelow > 0 && nStreams <= (size_t)earlyBelow && nStreams >= (size_t)earlyBelow / 8 && !traceStat && !pubOnly;
            const bool marks = early || pubOnly;
            if (early) {
                // aux may start once the previous iteration's shade (everything on `stream` so far) is done
                if ((e = hipEventRecord(evOvl[it & 1], stream)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(aux, evOvl[it & 1], 0)) != hipSuccess) return e;
            }
            if (traceStat && traceStatClk) hipLaunchKernelGGL(wf_trace<3>, dim3(tb), dim3(256), 0, stream, *sc, b, sIn, ovfStride, it & 1, chunkShift, budgetShift, budgetMin, guideShift, triTrig, refillMin, topNodes, traceStat, it < 2700 ? it : 2699, tn.helpShards, tn.lateBudget);
            else if (traceStat && traceStatFull) hipLaunchKernelGGL(wf_trace<1>, dim3(tb), dim3(256), 0, stream, *sc, b, sIn, ovfStride, it & 1, chunkShift, budgetShift, budgetMin, guideShift, triTrig, refillMin, topNodes, traceStat, it < 2700 ? it : 2699, tn.helpShards, tn.lateBudget);
            else if (traceStat) hipLaunchKernelGGL(wf_trace<2>, dim3(tb), dim3(256), 0, stream, *sc, b, sIn, ovfStride, it & 1, chunkShift, budgetShift, budgetMin, guideShift, triTrig, refillMin, topNodes, traceStat, it < 2700 ? it : 2699, tn.helpShards, tn.lateBudget);
            else if (early || pubOnly) hipLaunchKernelGGL((wf_trace<0, true>), dim3(tb), dim3(256), 0, stream, *sc, b, sIn, ovfStride, it & 1, chunkShift, budgetShift, budgetMin, guideShift, triTrig, refillMin, topNodes, (unsigned long long*)nullptr, earlyPrio, tn.helpShards, tn.lateBudget);
            else hipLaunchKernelGGL(wf_trace<0>, dim3(tb), dim3(256), 0, stream, *sc, b, sIn, ovfStride, it & 1, chunkShift, budgetShift, budgetMin, guideShift, triTrig, refillMin, topNodes, (unsigned long long*)nullptr, 0, tn.helpShards, tn.lateBudget);
            if (timed) (void)hipEventRecord(trace_ev[3 * it + 1], stream);
            const dim3 sg((liveBound + shadeThreads - 1) / shadeThreads), sb(shadeThreads);
            const bool twoRounds = shadeRounds >= 0 ? (shadeRounds != 0) : (liveBound < trStreams);
#define PT_SHADE(W, T, P, M, S) hipLaunchKernelGGL((wf_shade<W, T, P, M>), sg, sb, 0, S, *sc, *cam, prm, b, sIn, sOut, sClr, it & 1)
            if (early) {
                // phase 1 in small workgroups: a 256-thread workgroup needs one free wave slot per SIMD, i.e. two traversal workgroups of
                // the CU gone, a 512-thread one four — it gets onto the chip earlier in the drain
                { const dim3 sg((liveBound + earlyThreads - 1) / earlyThreads), sb(earlyThreads);
                  if (twoRounds) PT_SHADE(4, true, 1, true, aux); else PT_SHADE(4, false, 1, true, aux); }
                if ((e = hipEventRecord(evOvl[2 + (it & 1)], aux)) != hipSuccess) return e;
                if ((e = hipStreamWaitEvent(stream, evOvl[2 + (it & 1)], 0)) != hipSuccess) return e;
                if (twoRounds) PT_SHADE(4, true, 2, true, stream); else PT_SHADE(4, false, 2, true, stream);
            }
            else if (marks) { if (twoRounds) PT_SHADE(4, true, 0, true, stream); else PT_SHADE(4, false, 0, true, stream); }
            else if (twoRounds) { if (shadeWaves == 2) PT_SHADE(2, true, 0, false, stream); else if (shadeWaves == 3) PT_SHADE(3, true, 0, false, stream); else PT_SHADE(4, true, 0, false, stream); }
            else { if (shadeWaves == 2) PT_SHADE(2, false, 0, false, stream); else if (shadeWaves == 3) PT_SHADE(3, false, 0, false, stream); else PT_SHADE(4, false, 0, false, stream); }
#undef PT_SHADE
            if (timed) (void)hipEventRecord(trace_ev[3 * it + 2], stream);      // [3it+1, 3it+2] brackets this iteration's wf_shade
        }
        if ((e = hipGetLastError()) != hipSuccess) return e;
        if ((e = hipMemcpyAsync(h_cnt, &b.cnt[it % 3].nActive, 4, hipMemcpyDeviceToHost, stream)) != hipSuccess) return e;
        if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
        if (h_cnt[0] == 0) break;
        liveBound = h_cnt[0];
        if (h_cnt[0] <= (uint32_t)drainBelow) {
            // few streams left: finish them in one launch instead of hundreds of latency-bound iterations
            // the 4-wide tree if its walk fits the per-lane stack (any tree the builder makes for the config scenes does), else the binary one
            const bool quadWalk = tn.drainQuad && 3 * sc->quad_depth + 2 <= kDrainQuadStack;
            // 2 (4-wide walk: 189 VGPRs) or 3 (165) waves per SIMD of wf_drain fit: spread the streams over at most that many lanes
            const size_t drainLanes = (size_t)((quadWalk && DRAIN_MINBLOCKS < 3) ? 2 : 3) * 4 * 256 * 64;
            int spread = 0;
            while (spread < tn.drainSpread && ((size_t)h_cnt[0] << (spread + 1)) <= drainLanes) spread++;
            const int db = (int)((((size_t)h_cnt[0] << spread) + kBlockThreads - 1) / kBlockThreads);
            if (quadWalk) hipLaunchKernelGGL(wf_drain<true>, dim3(db), dim3(kBlockThreads), 0, stream, *sc, *cam, prm, b, it % 3, it & 1, spread);
            else hipLaunchKernelGGL(wf_drain<false>, dim3(db), dim3(kBlockThreads), 0, stream, *sc, *cam, prm, b, it % 3, it & 1, spread);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if ((e = hipStreamSynchronize(stream)) != hipSuccess) return e;
            break;
        }
        if (it > hardCap) return hipErrorLaunchFailure;      // cannot happen for a well-formed scene; never spin forever
        if (poll < 64) poll *= 2;
        if (drainBelow > 0 && (unsigned long long)h_cnt[0] <= (unsigned long long)drainBelow * 8ull) poll = 16;      // near the hand-over: look again soon
    }
    if (iters_out) *iters_out = it;
    if (trace_ev_used) *trace_ev_used = trace_ev ? (it < trace_ev_pairs ? it : trace_ev_pairs) : 0;
    return hipSuccess;
}

// Runs the whole pipeline for one pt_render_tiles call.  `stream` is the caller's stream;
// `xstreams` are up to 3 extra streams owned by the scene; `h_cnt` holds one pinned poll word
// (64 B apart) per cohort.  ev_begin/ev_end bracket the whole render on `stream`.  trace_ev:
// optional event triples (before wf_trace, after it, after wf_shade), split evenly between cohorts; trace_ev_used[c] = triples used by cohort c.
// Blocks the host until the render has drained.
hipError_t ptk_wf_render(int device, const ptd::DevScene* sc, const ptd::DevCamera* cam, const ptd::DevParams* prm,
                         void* work, int traceBlocks, uint32_t* h_cnt, hipStream_t stream, hipStream_t* xstreams,
                         hipEvent_t ev_begin, hipEvent_t ev_end, hipEvent_t ev_fork, hipEvent_t* ev_join, int* iters_out,
                         hipEvent_t* trace_ev, int trace_ev_pairs, int* trace_ev_used, int drainBelow, int shadeRounds, void* traceStat, int earlyBelow)
{
    using namespace ptd;
    const size_t nUnits = (size_t)prm->n_units;
    const int C = ptk_wf_cohorts(nUnits);
    const size_t per = (nUnits + C - 1) / C;
    char* base = (char*)work;
    float* staging = (float*)base;
    char* p = base + staging_bytes(nUnits * 64);
    hipError_t e;
    if (ev_begin) { if ((e = hipEventRecord(ev_begin, stream)) != hipSuccess) return e; }
    if (C > 1) { if ((e = hipEventRecord(ev_fork, stream)) != hipSuccess) return e; }
    std::vector<hipError_t> rc((size_t)C, hipSuccess);
    std::vector<int> iters((size_t)C, 0);
    std::vector<std::thread> th;
    const int evPer = trace_ev ? trace_ev_pairs / C : 0;
    for (int c = 0; c < C; c++) {
        DevParams cp = *prm;
        const size_t u0 = (size_t)c * per, u1 = (u0 + per < nUnits) ? u0 + per : nUnits;
        cp.unit_base = (int)u0; cp.n_units = (int)(u1 > u0 ? u1 - u0 : 0);
        WfBuf b; carve(p + (size_t)c * cohort_bytes(per * 64, traceBlocks), per * 64, traceBlocks, b);
        b.staging = staging + u0 * 64 * 3;
        hipStream_t cs = (c == 0) ? stream : xstreams[c - 1];
        if (c > 0) { if ((e = hipStreamWaitEvent(cs, ev_fork, 0)) != hipSuccess) return e; }
        hipEvent_t* tev = trace_ev ? trace_ev + (size_t)3 * evPer * c : nullptr;
        int* used = trace_ev_used ? &trace_ev_used[c] : nullptr;
        if (cp.n_units == 0) { if (used) *used = 0; continue; }
        // early shade needs a second stream and four events: with one cohort the scene's extra streams and fork / join events are free
        hipStream_t aux = (C == 1) ? xstreams[0] : nullptr;
        auto job = [=, &rc, &iters]() {
            hipEvent_t evOvl[4] = {ev_fork, ev_join[0], ev_join[1], ev_join[2]};
            rc[(size_t)c] = run_cohort(device, sc, cam, cp, b, traceBlocks, h_cnt + 16 * c, cs, tev, evPer, used, drainBelow, shadeRounds, &iters[(size_t)c], (unsigned long long*)traceStat, aux, evOvl, earlyBelow);
        };
        if (C == 1) job(); else th.emplace_back(job);
    }
    for (auto& t : th) t.join();
    for (int c = 0; c < C; c++) if (rc[(size_t)c] != hipSuccess) return rc[(size_t)c];
    // every cohort stream has been synchronised by its poll loop; order the caller's stream after them anyway
    for (int c = 1; c < C; c++) {
        if ((e = hipEventRecord(ev_join[c - 1], xstreams[c - 1])) != hipSuccess) return e;
        if ((e = hipStreamWaitEvent(stream, ev_join[c - 1], 0)) != hipSuccess) return e;
    }
    if (ev_end) { if ((e = hipEventRecord(ev_end, stream)) != hipSuccess) return e; }
    int mx = 0; for (int v : iters) mx = v > mx ? v : mx;
    if (iters_out) *iters_out = mx;
    return hipSuccess;
}

hipError_t ptk_dbg_ray_setup(const float* dir3, int n, float* out5, hipStream_t stream)
{
    const int nb = (n + 255) / 256;
    if (nb > 0) hipLaunchKernelGGL(ptd::dbg_ray_setup, dim3(nb), dim3(256), 0, stream, dir3, n, out5);
    return hipGetLastError();
}
hipError_t ptk_dbg_pixel_dir(const ptd::DevCamera* cam, const int* pxpypass, int n, float* out8, hipStream_t stream)
{
    const int nb = (n + 255) / 256;
    if (nb > 0) hipLaunchKernelGGL(ptd::dbg_pixel_dir, dim3(nb), dim3(256), 0, stream, *cam, pxpypass, n, out8);
    return hipGetLastError();
}
hipError_t ptk_dbg_nee(const ptd::DevScene* sc, const float* in5, int n, float* out12, hipStream_t stream)
{
    const int nb = (n + ptd::kBlockThreads - 1) / ptd::kBlockThreads;
    if (nb > 0) hipLaunchKernelGGL(ptd::dbg_nee, dim3(nb), dim3(ptd::kBlockThreads), 0, stream, *sc, in5, n, out12);
    return hipGetLastError();
}

}  // extern "C"
