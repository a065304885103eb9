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

PT_DEV Mat load_mat(const float4* __restrict__ mats, int idx)
{
    const float4 a = mats[3 * idx], b = mats[3 * idx + 1], c = mats[3 * idx + 2];
    Mat m;
    m.emittance = f3(a.x, a.y, a.z);
    m.albedo = f3(a.w, b.x, b.y);
    m.specular = f3(b.z, b.w, c.x);
    m.opacity = c.y; m.roughness = c.z; m.metallic = c.w;
    return m;
}
PT_DEV Mat sphere_mat(const float4* __restrict__ sph, int s) { return load_mat(sph + 4 * s + 1, 0); }

PT_DEV int tri_mat_index(const DevScene& sc, int prim) { return __float_as_int(sc.shade[7 * prim + 6].w); }

// emittance of whatever primitive `prim` is (GetLightColor returns hitResult.mat.emittance)
PT_DEV f3 prim_emittance(const DevScene& sc, int prim)
{
    if (prim < sc.n_tris) {
        const float4 a = sc.mats[3 * tri_mat_index(sc, prim)];
        return f3(a.x, a.y, a.z);
    }
    const float4 a = sc.spheres[4 * (prim - sc.n_tris) + 1];
    return f3(a.x, a.y, a.z);
}

// Rebuild the reference's HitResult for the closest hit (Triangle::hit tail,
// CudaPrimitive.cuh:117-156; Sphere::hit tail, :274-301).  u,v are recomputed with the
// same operations the traversal used, so they carry the same bits.
PT_DEV void make_surf(const DevScene& sc, int prim, float t, const f3& org, const f3& dir, Surf& s)
{
    s.p = org + t * dir;                                              // Ray::at
    if (prim < sc.n_tris) {
        const float4 a = sc.tri_ref[3 * prim], b = sc.tri_ref[3 * prim + 1], c = sc.tri_ref[3 * prim + 2];
        const f3 V0(a.x, a.y, a.z), E1(b.x, b.y, b.z), E2(c.x, c.y, c.z);
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
        const float4* sh = sc.shade + 7 * prim;
        const float4 s0 = sh[0], s1 = sh[1], s2 = sh[2], s3 = sh[3], s4 = sh[4], s5 = sh[5], s6 = sh[6];
        const f3 N0(s0.x, s0.y, s0.z), N1(s0.w, s1.x, s1.y), N2(s1.z, s1.w, s2.x);
        const f3 T0(s2.y, s2.z, s2.w), T1(s3.x, s3.y, s3.z), T2(s3.w, s4.x, s4.y);
        const f3 B0(s4.z, s4.w, s5.x), B1(s5.y, s5.z, s5.w), B2(s6.x, s6.y, s6.z);
        const f3 outward = normalize(w * N0 + v * N1 + u * N2);      // weights swapped on purpose (Q5)
        s.fr.front = dot(dir, outward) < 0.f;                         // HitResult::SetNormal
        s.fr.n = s.fr.front ? outward : -outward;
        s.fr.t = normalize(w * T0 + v * T1 + u * T2);
        s.fr.b = normalize(w * B0 + v * B1 + u * B2);
        s.m = load_mat(sc.mats, __float_as_int(s6.w));
    } else {
        const int si = prim - sc.n_tris;
        const float4 c = sc.spheres[4 * si];
        const f3 outward = (s.p - f3(c.x, c.y, c.z)) / c.w;
        s.fr.front = dot(dir, outward) < 0.f;
        s.fr.n = s.fr.front ? outward : -outward;
        s.fr.t = normalize(cross(f3(0.f, 1.f, 0.f), s.fr.n));
        s.fr.b = cross(s.fr.n, s.fr.t);
        s.m = sphere_mat(sc.spheres, si);
    }
}

}  // namespace ptd
