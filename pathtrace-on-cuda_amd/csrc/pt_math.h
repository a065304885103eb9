// pt_math.h — device-side scalar/vector arithmetic of the integrator (gfx950).
//
// Numerics contract: every float operation the reference's
// integrator performs is performed here in the same order, un-contracted (the library is
// built with -ffp-contract=off) and with IEEE-correct division and square root (hipcc's
// default -fhip-fp32-correctly-rounded-divide-sqrt).  Transcendentals are *correctly
// rounded* float results, obtained by evaluating in fp64 and rounding once — MI355X's
// fp64 vector rate (78 TF) makes that cheap next to traversal, and it is what makes the
// HIP image reproducible against a CPU checker bit for bit.
//
// Reference arithmetic this mirrors: include/CudaVector.cuh:11-303 (vec3 operators,
// reflect, refract, Normalize, saturate), cuRAND XORWOW call sites.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define PT_DEV __device__ __forceinline__
#define PT_SC_FN __device__ __forceinline__
#include "pt_sincos.h"

namespace ptd {

constexpr float kEps = 0.0001f;              // EPS, include/CudaPrimitive.cuh:11
constexpr float kPi = 3.141592f;             // pif / PI (truncated on purpose), include/Bxdf.cuh:10
constexpr float kInvPi = 1.f / 3.141592f;    // invPif, include/Bxdf.cuh:11

struct f3 {
    float x, y, z;
    PT_DEV f3() {}
    PT_DEV f3(float a, float b, float c) : x(a), y(b), z(c) {}
};

PT_DEV f3 operator-(const f3& a) { return f3(-a.x, -a.y, -a.z); }
PT_DEV f3 operator+(const f3& a, const f3& b) { return f3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV f3 operator-(const f3& a, const f3& b) { return f3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV f3 operator*(const f3& a, const f3& b) { return f3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV f3 operator/(const f3& a, const f3& b) { return f3(a.x / b.x, a.y / b.y, a.z / b.z); }
PT_DEV f3 operator*(float t, const f3& v) { return f3(t * v.x, t * v.y, t * v.z); }
PT_DEV f3 operator*(const f3& v, float t) { return f3(t * v.x, t * v.y, t * v.z); }
PT_DEV f3 operator/(const f3& v, float t) { return f3(v.x / t, v.y / t, v.z / t); }
PT_DEV f3 operator+(float t, const f3& v) { return f3(t + v.x, t + v.y, t + v.z); }
PT_DEV f3 operator-(float t, const f3& v) { return f3(t - v.x, t - v.y, t - v.z); }
PT_DEV f3& operator+=(f3& a, const f3& b) { a.x += b.x; a.y += b.y; a.z += b.z; return a; }
PT_DEV f3& operator*=(f3& a, const f3& b) { a.x *= b.x; a.y *= b.y; a.z *= b.z; return a; }
PT_DEV f3& operator*=(f3& a, float t) { a.x *= t; a.y *= t; a.z *= t; return a; }

PT_DEV float dot(const f3& a, const f3& b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
PT_DEV f3 cross(const f3& a, const f3& b) {
    return f3(a.y * b.z - a.z * b.y, -(a.x * b.z - a.z * b.x), a.x * b.y - a.y * b.x);
}
PT_DEV float sqlen(const f3& a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
PT_DEV float length(const f3& a) { return __builtin_sqrtf(a.x * a.x + a.y * a.y + a.z * a.z); }
PT_DEV f3 normalize(const f3& a) { return a / length(a); }
// `a > b ? a : b` semantics (a NaN in `a` yields b) — cudamax/cudamin, CudaVector.cuh:226-234
PT_DEV float selmax(float a, float b) { return (a > b) ? a : b; }
PT_DEV float selmin(float a, float b) { return (a < b) ? a : b; }
PT_DEV float clampf(float v, float lo, float hi) { return selmin(selmax(v, lo), hi); }
PT_DEV float maxcomp(const f3& v) { float m = (v.x > v.y) ? v.x : v.y; return (m > v.z) ? m : v.z; }
PT_DEV bool anynan(const f3& v) { return (v.x != v.x) || (v.y != v.y) || (v.z != v.z); }
PT_DEV f3 reflect(const f3& w, const f3& n) { return -w + (2.f * dot(n, w)) * n; }
PT_DEV f3 refract(const f3& w, const f3& n, float inv_eta) {
    float c = dot(n, w);
    float k = 1.f + (inv_eta * inv_eta) * (c * c - 1.f);
    if (k < 0.f) return f3(0.f, 0.f, 0.f);
    return (-w) * inv_eta + (inv_eta * c - __builtin_sqrtf(k)) * n;
}

// ---- correctly rounded float transcendentals through fp64 ---------------------------
PT_DEV float cr_sin(float x) { return (float)::sin((double)x); }
PT_DEV float cr_cos(float x) { return (float)::cos((double)x); }
PT_DEV float cr_atan(float x) { return (float)::atan((double)x); }
// sin and cos of a sampler's angle (always in [0, 2 pi]: phi = 2 pi u, theta = atan(.) >= 0), correctly rounded to float: pt_sincos.h —
// fp64, reduced to what that domain needs (≈50 instead of OCML sincos's 157 VALU instructions a pair) and checked on the host against
// the oracle's definition for EVERY float of the domain (tools/sincos_check.c: zero mismatches).  PT_OCML_SINCOS = 1 builds rounds 1-2's
// call of OCML's general double sincos instead (A/B).
#ifndef PT_OCML_SINCOS
#define PT_OCML_SINCOS 0
#endif
PT_DEV void cr_sincos(float x, float& s, float& c)
{
    double ds, dc;
#if PT_OCML_SINCOS
    ::sincos((double)x, &ds, &dc);
#else
    pt_sincos_0_2pi((double)x, &ds, &dc);
#endif
    s = (float)ds; c = (float)dc;
}
// x^5 for x in [1e-4, 0.999]: two exact-ish fp64 products (error 2^-52) rounded once.
PT_DEV float cr_pow5(float x) { double d = (double)x; double d2 = d * d; return (float)((d2 * d2) * d); }
// x^2 rounded once == the float product.
PT_DEV float cr_pow2(float x) { return x * x; }

// ---- XORWOW, seed scramble with rocRAND's constants, subsequence 0, offset 0 ----------
struct Rng {
    uint32_t x0, x1, x2, x3, x4, d;
    PT_DEV void init(uint64_t seed) {
        const uint32_t s0 = (uint32_t)seed ^ 0x2c7f967fU;
        const uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xa03697cbU;
        const uint32_t t0 = 1228688033U * s0;
        const uint32_t t1 = 2073658381U * s1;
        x0 = 123456789U + t0;
        x1 = 362436069U ^ t0;
        x2 = 521288629U + t1;
        x3 = 88675123U ^ t1;
        x4 = 5783321U + t0;
        d = 6615241U + t1 + t0;
    }
    PT_DEV uint32_t next() {
        const uint32_t t = x0 ^ (x0 >> 2);
        x0 = x1; x1 = x2; x2 = x3; x3 = x4;
        x4 = (x4 ^ (x4 << 4)) ^ (t ^ (t << 1));
        d += 362437U;
        return d + x4;
    }
    PT_DEV float uniform() { return 2.3283064e-10f + (float)next() * 2.3283064e-10f; }
};

}  // namespace ptd
