// pt_kernels.hip — the radiance integrator for MI355X (gfx950, wave64).
//
// Replaces the reference's StartRender kernel + GetColor_iter device function
// (srcs/pathtracer.cu:42-83, include/CudaUtil.cuh:193-382).  Same estimator, same random
// number consumption order, same float operation order — different machine mapping:
//
//   * persistent waves: each wavefront pulls (tile, pass) units from one global counter;
//     a unit is an 8x8-pixel tile (one pixel per lane) for one pass.
//   * inside a unit every lane runs its pixel's `spp_per_pass` paths as a small state
//     machine (PATH ray -> shade -> SHADOW ray -> accumulate -> next ray / next sample), so a
//     lane whose path ends regenerates its next camera path immediately instead of idling
//     until the longest path of the wave finishes (the reference's nested loops).
//   * one traversal loop serves both ray kinds; the shading that follows a PATH ray draws
//     all random numbers of the bounce (NEE first, then the BSDF sample, then roulette — the
//     reference's order) and precomputes the NEE contribution, so only 17 floats survive
//     the shadow-ray traversal.
//   * per-pass means go to a staging slab and are summed in pass order afterwards, which
//     reproduces `image[offset] += mean` (pathtracer.cu:81) bit for bit while letting all
//     passes share one launch.
#include <hip/hip_runtime.h>
#include "pt_device.h"
#include "pt_math.h"
#include "pt_bxdf.h"
#include "pt_trace.h"
#include "pt_shade.h"

namespace ptd {

// ---------------------------------------------------------------------------------------
// The render kernel.
// ---------------------------------------------------------------------------------------
enum : int { KIND_PATH = 0, KIND_SHADOW = 1 };

struct Counters {           // int64 x 8, see include/pt_api.h pt_last_counters
    unsigned long long v[8];
};

template <bool COUNT>
__global__ __launch_bounds__(kBlockThreads)
void render_units(DevScene sc, DevCamera cam, DevParams prm, float* __restrict__ staging,
                  unsigned int* __restrict__ unit_counter, Counters* __restrict__ counters)
{
    __shared__ int lds_stack[kWavesPerBlock][kStackDepth * 64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    int* stack = &lds_stack[wave][lane];

    const f3 camPos(cam.pos[0], cam.pos[1], cam.pos[2]);
    const f3 camF(cam.forward[0], cam.forward[1], cam.forward[2]);
    const f3 camU(cam.up[0], cam.up[1], cam.up[2]);
    const f3 camR(cam.right[0], cam.right[1], cam.right[2]);
    const int W = cam.W, H = cam.H;
    const int Nl = sc.n_lights;

    unsigned long long c_rays = 0, c_nodes = 0, c_tris = 0, c_sph = 0, c_hits = 0, c_paths = 0, c_trips = 0, c_active = 0;

    for (;;) {
        unsigned int unit = 0;
        if (lane == 0) unit = atomicAdd(unit_counter, 1u);
        unit = __builtin_amdgcn_readfirstlane(unit);
        if (unit >= (unsigned)prm.n_units) break;

        // unit -> (pass, local tile): tile-major within a pass keeps neighbouring waves on
        // neighbouring tiles (shared BVH working set in L2).
        const int pass_rel = (int)(unit / (unsigned)prm.n_tiles_local);
        const int lt = (int)(unit % (unsigned)prm.n_tiles_local);
        const int tile = lt * prm.world + prm.rank;
        const int tx = tile % prm.tiles_x, ty = tile / prm.tiles_x;
        const int px = tx * kTile + (lane & 7), py = ty * kTile + (lane >> 3);
        const int pass = prm.first_pass + pass_rel;
        bool active = (tile < prm.n_tiles_total) && (px < W) && (py < H);

        // ---- StartRender prologue for this pixel & pass (pathtracer.cu:70-74) ----
        Rng rng;
        f3 dir0(0.f, 0.f, -1.f);
        if (active) {
            const int offset = py * W + px;
            rng.init((uint64_t)(int64_t)(offset + pass * W * H));
            const float u1 = rng.uniform();
            const float u2 = rng.uniform();
            const f3 offR = ((2.f * (((float)px + u1) / (float)(W - 1) - 0.5f)) * cam.tan_half_fovx) * camR;
            const f3 offU = ((-2.f * (((float)py + u2) / (float)(H - 1) - 0.5f)) * cam.tan_half_fovy) * camU;
            const f3 direction = normalize(camF + offR + offU);      // GetPixelDirection, pathtracer.cu:33-40
            dir0 = normalize(direction);                             // Ray(CameraPos, direction) normalises again, CudaRay.cuh:12
        }

        // ---- per-lane path state ----
        f3 pixelColor(0.f, 0.f, 0.f);
        int samplesLeft = prm.spp_per_pass;
        f3 weight(1.f, 1.f, 1.f), radiance(0.f, 0.f, 0.f);
        int depth = 0, refractCnt = 0;
        bool bRefracted = false;
        int kind = KIND_PATH;
        f3 rorg = camPos, rdir = dir0;
        float rtmax = 999999.f;
        // context that survives a SHADOW traversal
        f3 nOrg(0.f, 0.f, 0.f), nDir(0.f, 0.f, 0.f);   // next PATH ray
        f3 wb(0.f, 0.f, 0.f), lightP(0.f, 0.f, 0.f);   // weight*brdfcos, sampled light point
        float cosA = 0.f, denom = 1.f;
        bool neeOk = false, terminate = false;
        if (COUNT && active) c_paths++;

        while (__ballot(active) != 0ull) {
            if (COUNT) { c_trips++; if (active) c_active++; }
            float t = 0.f;
            int prim = -1;
            if (active) {
                TraceStats st{0, 0, 0};
                prim = trace_closest<COUNT>(sc, rorg, rdir, rtmax, stack, t, st);
                if (COUNT) { c_rays++; c_nodes += st.nodes; c_tris += st.tris; c_sph += st.spheres; if (prim >= 0) c_hits++; }
            }
            bool pathDone = false;
            if (active && kind == KIND_SHADOW) {
                // ---- NEE accumulate (GetLightColor tail + CudaUtil.cuh:271-272) ----
                f3 Le(0.f, 0.f, 0.f);
                if (prim >= 0) {
                    const f3 hp = rorg + t * rdir;
                    if (length(hp - lightP) < kEps) Le = prim_emittance(sc, prim);
                }
                if (neeOk) radiance += ((wb * Le) * cosA) / denom;
                if (terminate) pathDone = true;
                else { kind = KIND_PATH; rorg = nOrg; rdir = nDir; rtmax = 999999.f; }
            } else if (active) {
                if (prim < 0) {
                    radiance += weight * f3(0.1f, 0.1f, 0.1f);               // CudaUtil.cuh:375-379
                    pathDone = true;
                } else {
                    // ---- shade a PATH hit: everything of the bounce except visibility ----
                    Surf s;
                    make_surf(sc, prim, t, rorg, rdir, s);
                    if (sqlen(s.m.emittance) > kEps) radiance += weight * s.m.emittance;   // :220-224
                    const float ior = ior_of(s.m);                                          // :231
                    const int lobe = lobe_of(s.m);
                    const f3 wo = -rdir;
                    // NEE sample (:235-245, SamplePrimitive :38-48)
                    const int li = (int)(rng.next() % (uint32_t)Nl);
                    const float4 l0 = sc.lights[4 * li], l1 = sc.lights[4 * li + 1], l2 = sc.lights[4 * li + 2], l3 = sc.lights[4 * li + 3];
                    const f3 LV0(l0.x, l0.y, l0.z), LV1(l0.w, l1.x, l1.y), LV2(l1.z, l1.w, l2.x), LN(l2.y, l2.z, l2.w);
                    const float r1 = __builtin_sqrtf(rng.uniform());
                    const float r2 = rng.uniform();
                    lightP = (1.f - r1) * LV0 + (r1 * (1.f - r2)) * LV1 + (r1 * r2) * LV2;
                    const float pdfLight = (1.f / l3.x) / ((float)Nl);
                    const f3 toL = lightP - s.p;
                    const f3 wl = normalize(toL);
                    float ca = dot(LN, normalize(s.p - lightP));
                    cosA = (ca < 0.f) ? 0.f : ca;
                    const f3 brdfcos = lobe_eval(lobe, s.m, ior, s.fr, wo, wl);
                    neeOk = !anynan(brdfcos);
                    wb = weight * brdfcos;
                    denom = sqlen(s.p - lightP) * pdfLight;
                    // BSDF sample (:283-338)
                    const f3 wi = lobe_sample(lobe, s.m, ior, s.fr, wo, rng);
                    const f3 w1 = lobe_eval(lobe, s.m, ior, s.fr, wo, wi);
                    float w2 = lobe_pdf(lobe, s.m, ior, s.fr, wo, wi);
                    w2 = selmax(w2, 1e-2f);
                    const f3 cw = w1 / w2;
                    if (lobe >= LOBE_REFRACTIVE) bRefracted = (dot(s.fr.n, wo) * dot(s.fr.n, wi)) <= 0.f;   // :307 (loop-carried, Q8)
                    terminate = false;
                    if (sqlen(wi) > kEps) weight *= cw; else terminate = true;
                    if (!terminate) {
                        nOrg = s.p + s.fr.n * (bRefracted ? -kEps : kEps);                  // :349-350
                        nDir = wi;
                        if (bRefracted) {
                            if (refractCnt++ > prm.max_refract) terminate = true;           // :351-359 (Depth unchanged)
                        } else {
                            if (depth >= prm.rr_bounce) {                                   // :361-373
                                const float u = rng.uniform();
                                const float q = selmax(selmin(maxcomp(weight), 1.f), prm.rr_floor);
                                if (u < q) weight *= (1.f / q); else terminate = true;
                            }
                            depth++;
                            if (depth >= prm.max_bounce) terminate = true;
                        }
                    }
                    // shadow ray: Ray(p, P - p), t_max = |P - p| + 1 (GetLightColor :152-157)
                    kind = KIND_SHADOW;
                    rorg = s.p; rdir = wl; rtmax = length(toL) + 1.0f;
                }
            }
            if (pathDone) {
                pixelColor += radiance;                                      // pathtracer.cu:79
                samplesLeft--;
                if (samplesLeft > 0) {
                    weight = f3(1.f, 1.f, 1.f); radiance = f3(0.f, 0.f, 0.f);
                    depth = 0; refractCnt = 0; bRefracted = false;
                    kind = KIND_PATH; rorg = camPos; rdir = dir0; rtmax = 999999.f;
                    if (COUNT) c_paths++;
                } else {
                    active = false;
                }
            }
        }

        // per-pass mean of this pixel (pathtracer.cu:81); summed in pass order by sum_passes
        const bool inFrame = (tile < prm.n_tiles_total) && (px < W) && (py < H);
        const f3 mean = inFrame ? pixelColor / (float)prm.spp_per_pass : f3(0.f, 0.f, 0.f);
        float* o = staging + (((size_t)pass_rel * prm.n_tiles_local + lt) * kTilePixels + lane) * 3;
        o[0] = mean.x; o[1] = mean.y; o[2] = mean.z;
    }

    if (COUNT) {
        atomicAdd(&counters->v[0], c_rays); atomicAdd(&counters->v[1], c_nodes); atomicAdd(&counters->v[2], c_tris);
        atomicAdd(&counters->v[3], c_sph); atomicAdd(&counters->v[4], c_hits); atomicAdd(&counters->v[5], c_paths);
        if (lane == 0) atomicAdd(&counters->v[6], c_trips);
        atomicAdd(&counters->v[7], c_active);
    }
}

// image[offset] += mean, pass after pass, starting from 0 (pathtracer.cu:81, 211-213).
__global__ void sum_passes(const float* __restrict__ staging, int passes, long long floats_per_pass, float* __restrict__ tiles)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= floats_per_pass) return;
    float acc = 0.f;
    for (int p = 0; p < passes; p++) acc += staging[(long long)p * floats_per_pass + i];
    tiles[i] = acc;
}

// gathered tile buffers (rank-major) -> row-major frame
__global__ void untile(const float* __restrict__ gathered, int W, int H, int tiles_x, int n_tiles_total, int world,
                       long long floats_per_rank, float* __restrict__ frame)
{
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)W * H) return;
    const int px = (int)(i % W), py = (int)(i / W);
    const int tile = (py / kTile) * tiles_x + (px / kTile);
    const int rank = tile % world, lt = tile / world;
    const int lane = (py % kTile) * kTile + (px % kTile);
    const float* src = gathered + (long long)rank * floats_per_rank + ((long long)lt * kTilePixels + lane) * 3;
    frame[3 * i + 0] = src[0]; frame[3 * i + 1] = src[1]; frame[3 * i + 2] = src[2];
}

// ---------------------------------------------------------------------------------------
// Parity hooks (pt_dbg_*): single device functions, one record per thread.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlockThreads)
void dbg_raycast(DevScene sc, const float* __restrict__ rays8, int n, float* __restrict__ out29, int* __restrict__ out_prim)
{
    __shared__ int lds_stack[kWavesPerBlock][kStackDepth * 64];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    int* stack = &lds_stack[threadIdx.x >> 6][threadIdx.x & 63];
    if (i >= n) return;
    const float* r = rays8 + (size_t)i * 8;
    const f3 org(r[0], r[1], r[2]), dir(r[3], r[4], r[5]);
    float t; TraceStats st{0, 0, 0};
    const int prim = trace_closest<false>(sc, org, dir, r[7], stack, t, st);
    float* o = out29 + (size_t)i * 29;
    out_prim[i] = prim;
    if (prim < 0) { for (int k = 0; k < 29; k++) o[k] = 0.f; return; }
    Surf s;
    make_surf(sc, prim, t, org, dir, s);
    o[0] = 1.f; o[1] = t; o[2] = 0.f; o[3] = 0.f; o[4] = s.fr.front ? 1.f : 0.f;
    o[5] = s.p.x; o[6] = s.p.y; o[7] = s.p.z;
    o[8] = s.fr.n.x; o[9] = s.fr.n.y; o[10] = s.fr.n.z;
    o[11] = s.fr.t.x; o[12] = s.fr.t.y; o[13] = s.fr.t.z;
    o[14] = s.fr.b.x; o[15] = s.fr.b.y; o[16] = s.fr.b.z;
    o[17] = s.m.emittance.x; o[18] = s.m.emittance.y; o[19] = s.m.emittance.z;
    o[20] = s.m.albedo.x; o[21] = s.m.albedo.y; o[22] = s.m.albedo.z;
    o[23] = s.m.specular.x; o[24] = s.m.specular.y; o[25] = s.m.specular.z;
    o[26] = s.m.opacity; o[27] = s.m.roughness; o[28] = s.m.metallic;
}

__global__ void dbg_bxdf(int lobe, const float* __restrict__ in28, int n, float* __restrict__ out12)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* f = in28 + (size_t)i * 28;
    Frame h; h.n = f3(f[0], f[1], f[2]); h.t = f3(f[3], f[4], f[5]); h.b = f3(f[6], f[7], f[8]); h.front = f[9] != 0.f;
    Mat m; m.emittance = f3(0.f, 0.f, 0.f); m.albedo = f3(f[10], f[11], f[12]); m.specular = f3(f[13], f[14], f[15]);
    m.roughness = f[16]; m.metallic = f[17]; m.opacity = 1.f;
    const f3 wo(f[18], f[19], f[20]), wi(f[21], f[22], f[23]);
    const uint32_t lo = __float_as_uint(f[24]), hi = __float_as_uint(f[25]);
    Rng s; s.init(((uint64_t)hi << 32) | lo);
    const Rng s0 = s;
    const float ior = ior_of(m);
    const f3 e = lobe_eval(lobe, m, ior, h, wo, wi);
    const float p = lobe_pdf(lobe, m, ior, h, wo, wi);
    const f3 ws = lobe_sample(lobe, m, ior, h, wo, s);
    const f3 es = lobe_eval(lobe, m, ior, h, wo, ws);
    const float ps = lobe_pdf(lobe, m, ior, h, wo, ws);
    // number of draws = how far the Weyl counter moved
    const uint32_t draws = (s.d - s0.d) / 362437U;
    float* o = out12 + (size_t)i * 12;
    o[0] = e.x; o[1] = e.y; o[2] = e.z; o[3] = p;
    o[4] = ws.x; o[5] = ws.y; o[6] = ws.z;
    o[7] = es.x; o[8] = es.y; o[9] = es.z; o[10] = ps; o[11] = (float)draws;
}

__global__ void dbg_rng(unsigned long long seed, int n, uint32_t* __restrict__ raw, float* __restrict__ uni)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    Rng r; r.init(seed);
    for (int i = 0; i < n; i++) raw[i] = r.next();
    r.init(seed);
    for (int i = 0; i < n; i++) uni[i] = r.uniform();
}

// in: x per record; out8: sin cos atan pow5(clamped) sqrt 1/x x/3 length(x,x+1,x+2)
__global__ void dbg_math(const float* __restrict__ in, int n, float* __restrict__ out8)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = in[i];
    float* o = out8 + (size_t)i * 8;
    o[0] = cr_sin(x); o[1] = cr_cos(x); o[2] = cr_atan(x);
    o[3] = cr_pow5(clampf(__builtin_fabsf(x), kEps, 0.999f));
    o[4] = __builtin_sqrtf(__builtin_fabsf(x)); o[5] = 1.f / x; o[6] = x / 3.f;
    o[7] = length(f3(x, x + 1.f, x + 2.f));
}

// the samplers' sin / cos pair (pt_math.h: cr_sincos, angles in [0, 2 pi]): out2 = sin cos
__global__ void dbg_sincos(const float* __restrict__ in, int n, float* __restrict__ out2)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s, c;
    cr_sincos(in[i], s, c);
    out2[2 * (size_t)i] = s; out2[2 * (size_t)i + 1] = c;
}

}  // namespace ptd

// ---------------------------------------------------------------------------------------
// Launchers (called from pt_api.hip)
// ---------------------------------------------------------------------------------------
extern "C" {

hipError_t ptk_render_units(const ptd::DevScene* sc, const ptd::DevCamera* cam, const ptd::DevParams* prm,
                            float* staging, unsigned int* unit_counter, void* counters, int grid_blocks,
                            int count, hipStream_t stream)
{
    if (count)
        hipLaunchKernelGGL(ptd::render_units<true>, dim3(grid_blocks), dim3(ptd::kBlockThreads), 0, stream,
                           *sc, *cam, *prm, staging, unit_counter, (ptd::Counters*)counters);
    else
        hipLaunchKernelGGL(ptd::render_units<false>, dim3(grid_blocks), dim3(ptd::kBlockThreads), 0, stream,
                           *sc, *cam, *prm, staging, unit_counter, (ptd::Counters*)counters);
    return hipGetLastError();
}

hipError_t ptk_sum_passes(const float* staging, int passes, long long floats_per_pass, float* tiles, hipStream_t stream)
{
    const int bs = 256;
    const long long nb = (floats_per_pass + bs - 1) / bs;
    if (nb > 0) hipLaunchKernelGGL(ptd::sum_passes, dim3((unsigned)nb), dim3(bs), 0, stream, staging, passes, floats_per_pass, tiles);
    return hipGetLastError();
}

hipError_t ptk_untile(const float* gathered, int W, int H, int tiles_x, int n_tiles_total, int world,
                      long long floats_per_rank, float* frame, hipStream_t stream)
{
    const int bs = 256;
    const long long n = (long long)W * H;
    const long long nb = (n + bs - 1) / bs;
    if (nb > 0) hipLaunchKernelGGL(ptd::untile, dim3((unsigned)nb), dim3(bs), 0, stream, gathered, W, H, tiles_x, n_tiles_total, world, floats_per_rank, frame);
    return hipGetLastError();
}

hipError_t ptk_dbg_raycast(const ptd::DevScene* sc, const float* rays8, int n, float* out29, int* out_prim, hipStream_t stream)
{
    const int nb = (n + ptd::kBlockThreads - 1) / ptd::kBlockThreads;
    if (nb > 0) hipLaunchKernelGGL(ptd::dbg_raycast, dim3(nb), dim3(ptd::kBlockThreads), 0, stream, *sc, rays8, n, out29, out_prim);
    return hipGetLastError();
}
hipError_t ptk_dbg_bxdf(int lobe, const float* in28, int n, float* out12, hipStream_t stream)
{
    const int nb = (n + 255) / 256;
    if (nb > 0) hipLaunchKernelGGL(ptd::dbg_bxdf, dim3(nb), dim3(256), 0, stream, lobe, in28, n, out12);
    return hipGetLastError();
}
hipError_t ptk_dbg_rng(unsigned long long seed, int n, uint32_t* raw, float* uni, hipStream_t stream)
{
    hipLaunchKernelGGL(ptd::dbg_rng, dim3(1), dim3(64), 0, stream, seed, n, raw, uni);
    return hipGetLastError();
}
hipError_t ptk_dbg_math(const float* in, int n, float* out8, hipStream_t stream)
{
    const int nb = (n + 255) / 256;
    if (nb > 0) hipLaunchKernelGGL(ptd::dbg_math, dim3(nb), dim3(256), 0, stream, in, n, out8);
    return hipGetLastError();
}
hipError_t ptk_dbg_sincos(const float* in, int n, float* out2, hipStream_t stream)
{
    const int nb = (n + 255) / 256;
    if (nb > 0) hipLaunchKernelGGL(ptd::dbg_sincos, dim3(nb), dim3(256), 0, stream, in, n, out2);
    return hipGetLastError();
}

}  // extern "C"
