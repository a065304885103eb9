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

}  // namespace ptd
