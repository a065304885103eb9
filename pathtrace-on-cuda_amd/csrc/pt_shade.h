// pt_shade.h — surface reconstruction at a closest hit, shared by the render kernels.
#pragma once
#include "pt_device.h"
#include "pt_math.h"
#include "pt_bxdf.h"

namespace ptd {

// ---------------------------------------------------------------------------------------
// Surface data at a hit (what the reference keeps in HitResult).
// ---------------------------------------------------------------------------------------
struct Surf {
    f3 p;
    Frame fr;
    Mat m;
};

PT_DEV Mat mat_from(const float4 a, const float4 b, const float4 c)
{
    Mat m;
    m.emittance = f3(a.x, a.y, a.z);
    m.albedo = f3(a.w, b.x, b.y);
    m.specular = f3(b.z, b.w, c.x);
    m.opacity = c.y; m.roughness = c.z; m.metallic = c.w;
    return m;
}
PT_DEV Mat sphere_mat(const float4* __restrict__ sph, int s) { return mat_from(sph[4 * s + 1], sph[4 * s + 2], sph[4 * s + 3]); }

// One triangle's surface record (pt_device.h): 12 independent 16-byte loads, no index chasing.
struct SurfRec { float4 q[12]; };
PT_DEV void load_surf(const DevScene& sc, int prim, SurfRec& r)      // 0 <= prim < n_tris
{
    const float4* __restrict__ p = sc.surf + 12 * (size_t)prim;
#pragma unroll
    for (int k = 0; k < 12; k++) r.q[k] = p[k];
}
// emittance of a triangle (first material quad of its record: emittance.xyz | albedo.x)
PT_DEV float4 tri_emit4(const DevScene& sc, int prim) { return sc.surf[12 * (size_t)prim + 9]; }

// emittance of whatever primitive `prim` is (GetLightColor returns hitResult.mat.emittance)
PT_DEV f3 prim_emittance(const DevScene& sc, int prim)
{
    const float4 a = (prim < sc.n_tris) ? tri_emit4(sc, prim) : sc.spheres[4 * (prim - sc.n_tris) + 1];
    return f3(a.x, a.y, a.z);
}

// Rebuild the reference's HitResult for a closest hit on a triangle (Triangle::hit tail,
// CudaPrimitive.cuh:117-156) from its surface record.  u,v are recomputed with the same
// operations the traversal used, so they carry the same bits.
PT_DEV void surf_from_rec(const SurfRec& r, float t, const f3& org, const f3& dir, Surf& s)
{
    const float4* q = r.q;
    s.p = org + t * dir;                                              // Ray::at
    const f3 V0(q[0].x, q[0].y, q[0].z), E1(q[0].w, q[1].x, q[1].y), E2(q[1].z, q[1].w, q[2].x);
    const f3 T = org - V0;
    const f3 P = cross(dir, E2);
    const f3 Q = cross(T, E1);
    const float det = dot(P, E1);
    const float invDet = 1.f / det;
    float u = dot(P, T);
    float v = dot(Q, dir);
    u *= invDet;
    v *= invDet;
    const float w = 1.f - v - u;
    const f3 N0(q[2].y, q[2].z, q[2].w), N1(q[3].x, q[3].y, q[3].z), N2(q[3].w, q[4].x, q[4].y);
    const f3 T0(q[4].z, q[4].w, q[5].x), T1(q[5].y, q[5].z, q[5].w), T2(q[6].x, q[6].y, q[6].z);
    const f3 B0(q[6].w, q[7].x, q[7].y), B1(q[7].z, q[7].w, q[8].x), B2(q[8].y, q[8].z, q[8].w);
    const f3 outward = normalize(w * N0 + v * N1 + u * N2);          // weights swapped on purpose (Q5)
    s.fr.front = dot(dir, outward) < 0.f;                             // HitResult::SetNormal
    s.fr.n = s.fr.front ? outward : -outward;
    s.fr.t = normalize(w * T0 + v * T1 + u * T2);
    s.fr.b = normalize(w * B0 + v * B1 + u * B2);
    s.m = mat_from(q[9], q[10], q[11]);
}

// Sphere::hit tail (CudaPrimitive.cuh:274-301)
PT_DEV void surf_sphere(const DevScene& sc, int si, float t, const f3& org, const f3& dir, Surf& s)
{
    s.p = org + t * dir;
    const float4 c = sc.spheres[4 * si];
    const f3 outward = (s.p - f3(c.x, c.y, c.z)) / c.w;
    s.fr.front = dot(dir, outward) < 0.f;
    s.fr.n = s.fr.front ? outward : -outward;
    s.fr.t = normalize(cross(f3(0.f, 1.f, 0.f), s.fr.n));
    s.fr.b = cross(s.fr.n, s.fr.t);
    s.m = sphere_mat(sc.spheres, si);
}

PT_DEV void make_surf(const DevScene& sc, int prim, float t, const f3& org, const f3& dir, Surf& s)
{
    if (prim < sc.n_tris) {
        SurfRec r;
        load_surf(sc, prim, r);
        surf_from_rec(r, t, org, dir, s);
    } else {
        surf_sphere(sc, prim - sc.n_tris, t, org, dir, s);
    }
}

// ---------------------------------------------------------------------------------------
// Next-event estimation pieces and the pixel direction, shared by the render kernels and by the parity hooks
// (pt_dbg_nee / pt_dbg_pixel_dir) that compare them with the oracle row by row.
// ---------------------------------------------------------------------------------------
struct NeeSample {
    int li;              // light picked: curand(s) % Nl, CudaUtil.cuh:235
    f3 lightP;           // SamplePrimitive's point, :38-48
    f3 toL, wl;          // lightP - p and its normalisation (the shadow ray's direction, Ray ctor CudaRay.cuh:12)
    float pdfLight;      // (1 / area) / Nl, :236
    float cosA;          // max(dot(light normal, normalize(p - lightP)), 0), :240-241
};
PT_DEV NeeSample nee_sample(const DevScene& sc, Rng& rng, const f3& p)
{
    const int Nl = sc.n_lights;
    NeeSample n;
    n.li = (int)(rng.next() % (uint32_t)Nl);
    const float4 l0 = sc.lights[4 * n.li], l1 = sc.lights[4 * n.li + 1], l2 = sc.lights[4 * n.li + 2], l3 = sc.lights[4 * n.li + 3];
    const f3 LV0(l0.x, l0.y, l0.z), LV1(l0.w, l1.x, l1.y), LV2(l1.z, l1.w, l2.x), LN(l2.y, l2.z, l2.w);
    const float r1u = __builtin_sqrtf(rng.uniform());
    const float r2u = rng.uniform();
    n.lightP = (1.f - r1u) * LV0 + (r1u * (1.f - r2u)) * LV1 + (r1u * r2u) * LV2;
    n.pdfLight = (1.f / l3.x) / ((float)Nl);
    n.toL = n.lightP - p;
    n.wl = normalize(n.toL);
    const float ca = dot(LN, normalize(p - n.lightP));
    n.cosA = (ca < 0.f) ? 0.f : ca;
    return n;
}

// GetLightColor's verdict (CudaUtil.cuh:157-165) from the shadow ray's closest hit (t, prim): the hit primitive's emittance (passed in:
// the shade kernel fetches it ahead of time) if the hit point lies within EPS of the sampled light point, else black.
PT_DEV f3 nee_light_color(const f3& shO, const f3& shD, const f3& lightP, float t, int prim, const f3& primEmittance)
{
    f3 Le(0.f, 0.f, 0.f);
    if (prim >= 0) {
        const f3 hp = shO + t * shD;
        if (length(hp - lightP) < kEps) Le = primEmittance;
    }
    return Le;
}

// GetPixelDirection (srcs/pathtracer.cu:33-40) with the two jitter draws of StartRender (:72-73), then the Ray constructor's second
// normalisation (CudaRay.cuh:12).
PT_DEV f3 pixel_direction(const DevCamera& cam, int px, int py, Rng& rng, float& u1, float& u2)
{
    const f3 camF(cam.forward[0], cam.forward[1], cam.forward[2]);
    const f3 camU(cam.up[0], cam.up[1], cam.up[2]);
    const f3 camR(cam.right[0], cam.right[1], cam.right[2]);
    u1 = rng.uniform();
    u2 = rng.uniform();
    const f3 offR = ((2.f * (((float)px + u1) / (float)(cam.W - 1) - 0.5f)) * cam.tan_half_fovx) * camR;
    const f3 offU = ((-2.f * (((float)py + u2) / (float)(cam.H - 1) - 0.5f)) * cam.tan_half_fovy) * camU;
    const f3 direction = normalize(camF + offR + offU);
    return normalize(direction);
}

}  // namespace ptd
