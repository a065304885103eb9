// pt_trace.h — closest-hit query over the wide-node BVH, one ray per lane (wave64).
//
// What it must reproduce: RayCast (include/CudaUtil.cuh:93-148) — closest hit over all
// triangles in [t_min, t_max], then the linear sphere loop — including its tie rule: a hit
// with t == closestT REPLACES the incumbent (Triangle::hit rejects only t > t_max,
// include/CudaPrimitive.cuh:104-108), so among equal t the largest primitive index wins and
// spheres (tested last, in order) beat triangles.
//
// How it differs: the reference walks its own 40-byte-per-node tree
// depth-first in index order with a 128-entry local-memory stack and culls boxes only
// against an un-scaled closestT.  Here the tree that is walked is a SAH tree over the
// triangles built at upload (host/accel_build.cpp) with slightly padded boxes; one 64-byte
// record holds both children's boxes, children are visited near-first, far children go to a
// per-lane LDS stack, and boxes are also culled against the current closest hit (with 2^-7
// relative slack).  Acceptance stays the reference's: a triangle counts iff Triangle::hit
// passes AND the reference's own slab arithmetic (intersectionAABB, CudaUtil.cuh:65-88:
// inverse direction normalised, far side scaled by 1.00000024f) passes on that triangle's
// REFERENCE leaf box (ancestors are implied: float rounding is monotonic and a child's
// interval lies inside its parent's).  The final (t, primitive) is therefore the reference's.
#pragma once
#include "pt_device.h"
#include "pt_math.h"

namespace ptd {

struct TraceStats { uint32_t nodes, tris, spheres; };

// Reference slab arithmetic for one box (non-degenerate rays).  Returns the scaled entry
// distance in tn.  `cullB` is the closest hit so far in the same scaled units, with slack.
PT_DEV bool box_test(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
                     const f3& org, const f3& invD, float cullB, float& tn)
{
    float x1 = (bminx - org.x) * invD.x, x2 = (bmaxx - org.x) * invD.x;
    float y1 = (bminy - org.y) * invD.y, y2 = (bmaxy - org.y) * invD.y;
    float z1 = (bminz - org.z) * invD.z, z2 = (bmaxz - org.z) * invD.z;
    float txmin = __builtin_fminf(x1, x2), txmax = __builtin_fmaxf(x1, x2);
    float tymin = __builtin_fminf(y1, y2), tymax = __builtin_fmaxf(y1, y2);
    float tzmin = __builtin_fminf(z1, z2), tzmax = __builtin_fmaxf(z1, z2);
    tn = __builtin_fmaxf(tzmin, __builtin_fmaxf(tymin, __builtin_fmaxf(txmin, 0.f)));
    float tf = __builtin_fminf(tzmax, __builtin_fminf(tymax, txmax));
    tf *= 1.00000024f;
    return (tn <= tf) & (tn <= cullB);
}

// Degenerate rays: a direction component that is exactly 0 (or whose reciprocal overflows)
// makes the reference's normalised inverse direction (NaN, 0, 0)-like, and with its
// select-style max/min (CudaVector.cuh:226-234) its test then ACCEPTS EVERY BOX — the
// reference brute-forces all triangles for such rays (in the Cornell scenes: every NEE ray
// cast from a point on the light to another point on the light, 0.25 % of all rays, which is
// where ~95 % of the reference's triangle tests go).  The result it returns is simply the
// closest triangle that passes Triangle::hit.  This test finds the same triangles with a
// geometrically correct slab test that is padded by 2^-10 relative — three orders of
// magnitude above float rounding — so no triangle that Triangle::hit can accept is missed.
// tn / cullB are in true (un-scaled) ray units here.
PT_DEV bool box_test_robust(float bminx, float bminy, float bminz, float bmaxx, float bmaxy, float bmaxz,
                            const f3& org, const f3& dir, const f3& inv, float cullB, float& tn)
{
    float lo = 0.f, hi = __builtin_inff();
    bool ok = true;
    const float kPad = 0.0009765625f;
#define PT_AXIS(c, mn, mx)                                                                               \
    if (dir.c != 0.f && __builtin_fabsf(inv.c) < __builtin_inff()) {                                     \
        const float a = (mn - org.c) * inv.c, b = (mx - org.c) * inv.c;                                  \
        lo = __builtin_fmaxf(lo, __builtin_fminf(a, b));                                                 \
        hi = __builtin_fminf(hi, __builtin_fmaxf(a, b));                                                 \
    } else {                                                                                             \
        const float pad = kPad * __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(mn), __builtin_fabsf(mx)), __builtin_fabsf(org.c)) + 1e-30f; \
        ok = ok & (org.c >= mn - pad) & (org.c <= mx + pad);                                             \
    }
    PT_AXIS(x, bminx, bmaxx)
    PT_AXIS(y, bminy, bmaxy)
    PT_AXIS(z, bminz, bmaxz)
#undef PT_AXIS
    tn = lo;
    return ok & (lo * (1.f - kPad) <= hi * (1.f + kPad) + 1e-30f) & (lo <= cullB);
}

// Per-ray set-up of the quad-tree walk (wf_trace's refill; parity hook pt_dbg_ray_setup): inv = the reference's Normalize(inv(dir))
// (CudaUtil.cuh:60-63, :70) — what its leaf-box test uses, and a perfectly good inverse direction for the tree walk, which then
// measures t in units of 1/|inv(dir)| —, cscale = what converts the closest hit into those units, degenerate = a direction with a
// zero component (L = inf: 1/dir clamped to +-1e30, true units, no leaf-box test; see box_test_robust).
PT_DEV void ray_setup(const f3& dir, f3& inv, float& cscale, bool& degenerate)
{
    inv = f3(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);                     // inv(), CudaUtil.cuh:60-63
    const float L = __builtin_sqrtf(inv.x * inv.x + inv.y * inv.y + inv.z * inv.z);
    degenerate = !(L < __builtin_inff());
    if (!degenerate) {
        inv = inv / L;                                                   // Normalize(inv(dir)), :70
        cscale = __builtin_amdgcn_rcpf(L) * 1.0000019f;                  // 1/L, rounded up a little: the cull must not bite early
    } else {
        inv.x = (__builtin_fabsf(inv.x) <= 1e30f) ? inv.x : __builtin_copysignf(1e30f, dir.x);
        inv.y = (__builtin_fabsf(inv.y) <= 1e30f) ? inv.y : __builtin_copysignf(1e30f, dir.y);
        inv.z = (__builtin_fabsf(inv.z) <= 1e30f) ? inv.z : __builtin_copysignf(1e30f, dir.z);
        cscale = 1.0000019f;
    }
}

// Moeller-Trumbore with the reference's back-face cull and test order,
// Triangle::hit, include/CudaPrimitive.cuh:89-118.  `q` indexes the tree-ordered records; the
// record carries the triangle's index in the reference's order (tie rule: among equal t the
// largest reference index wins) and its reference leaf.  A triangle only counts if the
// reference would have reached it, i.e. if its reference leaf box passes the reference's slab
// test (degenerate rays: the reference accepts every box).
PT_DEV void tri_test(const DevScene& sc, int q, const f3& org, const f3& dir, const f3& invD, bool degenerate,
                     float& bestT, int& bestPrim)
{
    const float4 a = sc.tri[3 * q], b = sc.tri[3 * q + 1], c = sc.tri[3 * q + 2];
    const f3 V0(a.x, a.y, a.z), E1(b.x, b.y, b.z), E2(c.x, c.y, c.z);
    const f3 T = org - V0;
    const f3 P = cross(dir, E2);
    const f3 Q = cross(T, E1);
    const float det = dot(P, E1);
    if (det < kEps) return;
    const float invDet = 1.f / det;
    const float t = dot(Q, E2) * invDet;
    if (t < 0.f || t > bestT) return;
    const float u = dot(P, T);
    if (u < 0.f || u > det) return;
    const float v = dot(Q, dir);
    if (v < 0.f || (v + u) > det) return;
    const int prim = __float_as_int(a.w);
    if (!(t < bestT || prim > bestPrim)) return;
    if (!degenerate) {
        const int leaf = __float_as_int(b.w);
        const float4 l0 = sc.leafbox[2 * leaf], l1 = sc.leafbox[2 * leaf + 1];
        float tn;
        if (!box_test(l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, org, invD, __builtin_inff(), tn)) return;
    }
    bestT = t; bestPrim = prim;
}

// Two triangles of one leaf in one go (wf_trace).  The arithmetic of Triangle::hit up to det, t's numerator, u and v is done on both
// triangles at once with 2-wide vectors, which the compiler turns into packed v_pk_mul_f32 / v_pk_add_f32: one instruction serves both
// triangles, and each lane of a packed operation is an ordinary IEEE operation (no contraction: -ffp-contract=off), so every value has
// the bits tri_test computes.  `two` = false: the second triangle of the record is ignored.
// The operands come from a pair record (pt_device.h: tripair) — interleaved in memory, so no register moves are needed to form the
// pairs —, and the per-triangle decisions are straight-line mask arithmetic, in index order exactly as two calls of tri_test:
//   g_j   triangle j passes the tests of Triangle::hit that do not involve the closest hit so far (det, t >= 0, u, v);
//   c0    g_0 and t0 / prim0 beat the closest hit on entry;   c1  the same for triangle 1.
// The reference's leaf box (inline in the pair record) is then evaluated ONCE for most lanes (block A: the box of triangle 0 if c0,
// else of triangle 1);
// only a lane where both triangles are candidates goes on to block B, which redoes triangle 1's comparison against
// the closest hit as triangle 0 left it (in index order, exactly as a second call of tri_test would) and evaluates its box.
// Every floating-point value is produced by the same IEEE operations as in tri_test.
PT_DEV bool pair_box_ok(const float* rec, bool second, const f3& org, const f3& invD)
{
    const float* bx = rec + (second ? 26 : 20);      // same 128-byte line as the record itself
    const float4 l0 = *(const float4*)bx; const float2 l1 = *(const float2*)(bx + 4);
    float tn;
    return box_test(l0.x, l0.y, l0.z, l0.w, l1.x, l1.y, org, invD, __builtin_inff(), tn);
}

PT_DEV void tri_test_pairrec(const DevScene& sc, int q, bool two, const f3& org, const f3& dir, const f3& invD, bool degenerate,
                             float& bestT, int& bestPrim)
{
    typedef float f2v __attribute__((ext_vector_type(2)));
    const float4* rec = sc.tripair + 8 * (size_t)q;
    const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2], r3 = rec[3], r4 = rec[4];
    const f2v ox = {org.x, org.x}, oy = {org.y, org.y}, oz = {org.z, org.z};
    const f2v dx = {dir.x, dir.x}, dy = {dir.y, dir.y}, dz = {dir.z, dir.z};
    const f2v E1x = {r1.z, r1.w}, E1y = {r2.x, r2.y}, E1z = {r2.z, r2.w};
    const f2v E2x = {r3.x, r3.y}, E2y = {r3.z, r3.w}, E2z = {r4.x, r4.y};
    const f2v Tx = ox - (f2v){r0.x, r0.y}, Ty = oy - (f2v){r0.z, r0.w}, Tz = oz - (f2v){r1.x, r1.y};
    const f2v Px = dy * E2z - dz * E2y, Py = -(dx * E2z - dz * E2x), Pz = dx * E2y - dy * E2x;
    const f2v Qx = Ty * E1z - Tz * E1y, Qy = -(Tx * E1z - Tz * E1x), Qz = Tx * E1y - Ty * E1x;
    const f2v det = Px * E1x + Py * E1y + Pz * E1z;
    const f2v tnum = Qx * E2x + Qy * E2y + Qz * E2z;
    // u and v as scalar dot products: a packed instruction occupies the slow pipe for 4.2 clocks, two scalar ones issue in 2 x 2.1 on the
    // fast one, and this block is short of the former (-0.8 % wf_trace; the same operations either way)
    const f2v uu = {Px.x * Tx.x + Py.x * Ty.x + Pz.x * Tz.x, Px.y * Tx.y + Py.y * Ty.y + Pz.y * Tz.y};
    const f2v vv = {Qx.x * dir.x + Qy.x * dir.y + Qz.x * dir.z, Qx.y * dir.x + Qy.y * dir.y + Qz.y * dir.z};
    const float t0 = tnum.x * (1.f / det.x), t1 = tnum.y * (1.f / det.y);
    const int prim0 = __float_as_int(r4.z), prim1 = __float_as_int(r4.w);
    const bool g0 = !(det.x < kEps) & !(t0 < 0.f) & !((uu.x < 0.f) | (uu.x > det.x)) & !((vv.x < 0.f) | ((vv.x + uu.x) > det.x));
    const bool g1 = two & !(det.y < kEps) & !(t1 < 0.f) & !((uu.y < 0.f) | (uu.y > det.y)) & !((vv.y < 0.f) | ((vv.y + uu.y) > det.y));
    const bool c0 = g0 & !(t0 > bestT) & ((t0 < bestT) | (prim0 > bestPrim));
    const bool c1 = g1 & !(t1 > bestT) & ((t1 < bestT) | (prim1 > bestPrim));
    if (c0 | c1) {
        const bool okA = degenerate ? true : pair_box_ok((const float*)rec, !c0, org, invD);
        if (c0 & c1) {
            // both: triangle 0 first, then triangle 1 against what it left
            if (okA) { bestT = t0; bestPrim = prim0; }
            if (!(t1 > bestT) & ((t1 < bestT) | (prim1 > bestPrim))) {
                const bool okB = degenerate ? true : pair_box_ok((const float*)rec, true, org, invD);
                if (okB) { bestT = t1; bestPrim = prim1; }
            }
        } else if (okA) {
            bestT = c0 ? t0 : t1; bestPrim = c0 ? prim0 : prim1;
        }
    }
}

// Sphere::hit root selection, include/CudaPrimitive.cuh:255-272.
PT_DEV bool sphere_root(const f3& center, float rad, const f3& org, const f3& dir, float tmax, float& root)
{
    const f3 oc = org - center;
    const float a = sqlen(dir);
    const float half_b = dot(oc, dir);
    const float c = sqlen(oc) - rad * rad;
    const float disc = half_b * half_b - a * c;
    if (disc < 0.f) return false;
    const float sq = __builtin_sqrtf(disc);
    root = (-half_b - sq) / a;
    if (root < 0.f || tmax < root) {
        root = (-half_b + sq) / a;
        if (root < 0.f || tmax < root) return false;
    }
    return true;
}

// Closest hit of (org, dir) in [0, tmax].  `stack` points at this lane's column of the
// wave's LDS stack: entry k lives at stack[k * 64].
// Returns primitive index (triangle i, n_tris + sphere j) or -1; bestT = its t.
template <bool COUNT>
PT_DEV int trace_closest(const DevScene& sc, const f3& org, const f3& dir, float tmax,
                         int* stack, float& bestT, TraceStats& st)
{
    const f3 inv(1.f / dir.x, 1.f / dir.y, 1.f / dir.z);                    // inv(), CudaUtil.cuh:60-63
    const float L = __builtin_sqrtf(inv.x * inv.x + inv.y * inv.y + inv.z * inv.z);
    const f3 invD = inv / L;                                                 // Normalize(inv(dir)), :70
    const bool degenerate = !(L < __builtin_inff());
    // un-scales the reference's scaled entry distance (degenerate rays work un-scaled), +2^-7 slack
    const float kcull = degenerate ? 1.0078125f : 1.0078125f / L;
    bestT = tmax;
    int bestPrim = -1;
    float cullB = bestT * kcull;

    int sp = 0;
    int cur = 0;
    for (;;) {
        const float4 q0 = sc.nodes[4 * cur + 0];
        const float4 q1 = sc.nodes[4 * cur + 1];
        const float4 q2 = sc.nodes[4 * cur + 2];
        const float4 q3 = sc.nodes[4 * cur + 3];
        if (COUNT) st.nodes++;
        float tnL, tnR;
        bool okL, okR;
        if (!degenerate) {
            okL = box_test(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, org, invD, cullB, tnL);
            okR = box_test(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, org, invD, cullB, tnR);
        } else {
            okL = box_test_robust(q0.x, q0.y, q0.z, q0.w, q1.x, q1.y, org, dir, inv, cullB, tnL);
            okR = box_test_robust(q1.z, q1.w, q2.x, q2.y, q2.z, q2.w, org, dir, inv, cullB, tnR);
        }
        const int refL = __float_as_int(q3.x), refR = __float_as_int(q3.y);

        if (okL && refL < 0) {
            const int code = ~refL, first = code >> 3, cnt = code & 7;
            for (int k = 0; k < cnt; k++) { tri_test(sc, first + k, org, dir, invD, degenerate, bestT, bestPrim); if (COUNT) st.tris++; }
            cullB = bestT * kcull;
            okL = false;
        }
        if (okR && refR < 0) {
            // re-check against the possibly tightened bound (result-neutral: skips boxes beyond the hit)
            if (tnR <= cullB) {
                const int code = ~refR, first = code >> 3, cnt = code & 7;
                for (int k = 0; k < cnt; k++) { tri_test(sc, first + k, org, dir, invD, degenerate, bestT, bestPrim); if (COUNT) st.tris++; }
                cullB = bestT * kcull;
            }
            okR = false;
        }
        if (okL & okR) {
            const bool lNear = tnL <= tnR;
            stack[sp * 64] = lNear ? refR : refL;
            sp++;
            cur = lNear ? refL : refR;
        } else if (okL) {
            cur = refL;
        } else if (okR) {
            cur = refR;
        } else {
            if (sp == 0) break;
            sp--;
            cur = stack[sp * 64];
        }
    }

    // spheres, in order, against the triangles' closest t (CudaUtil.cuh:137-145)
    for (int s = 0; s < sc.n_spheres; s++) {
        const float4 c = sc.spheres[4 * s];
        float root;
        if (COUNT) st.spheres++;
        if (sphere_root(f3(c.x, c.y, c.z), c.w, org, dir, bestT, root)) { bestT = root; bestPrim = sc.n_tris + s; }
    }
    return bestPrim;
}

}  // namespace ptd
