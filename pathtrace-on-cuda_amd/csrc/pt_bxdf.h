// pt_bxdf.h — the four scattering lobes of the integrator, device side.
//
// Computes what include/Bxdf.cuh:13-370 computes (gltfpbr, reflective, refractive,
// pure_refractive: eval / sample / pdf, plus the fresnel / GGX helpers), operation for
// operation, so results are bit-identical to the CPU checker.  Organisation is this
// repo's: a lobe is selected once per hit (`lobe_of`) and the hit frame travels as a
// small POD instead of the reference's 136-byte HitResult.
#pragma once
#include "pt_math.h"

namespace ptd {

struct Frame {            // shading frame at a hit: what the BxDFs read from HitResult
    f3 n, t, b;           // normal (facing the ray), tangent, bitangent
    bool front;           // HitResult::bFrontFace
};

struct Mat {              // Material, include/CudaPrimitive.cuh:15-23
    f3 emittance, albedo, specular;
    float opacity, roughness, metallic;
};

enum Lobe : int { LOBE_GLTFPBR = 0, LOBE_REFLECTIVE = 1, LOBE_REFRACTIVE = 2, LOBE_PURE_REFRACTIVE = 3 };

// the 4-way branch of GetColor_iter, include/CudaUtil.cuh:247-270 / 283-334
PT_DEV int lobe_of(const Mat& m) {
    if (m.opacity < (1.f - kEps)) return (m.roughness < 1e-2f) ? LOBE_PURE_REFRACTIVE : LOBE_REFRACTIVE;
    return (m.roughness < 1e-2f) ? LOBE_REFLECTIVE : LOBE_GLTFPBR;
}

PT_DEV f3 lerp3(const f3& x, const f3& y, float a) { return x * (1.f - a) + y * a; }          // Bxdf.cuh:13
PT_DEV float mean3(const f3& v) { return (v.x + v.y + v.z) * 0.333333f; }                     // Bxdf.cuh:18

// reflectivity_to_eta(...)[0], Bxdf.cuh:53-56 (only the first component is ever used, CudaUtil.cuh:231)
PT_DEV float ior_of(const Mat& m) {
    float r = m.specular.x;
    r = (r > 0.0f) ? r : 0.0f;
    r = (r < 0.99f) ? r : 0.99f;
    float s = __builtin_sqrtf(r);
    return (1.f + s) / (1.f - s);
}

PT_DEV float fresnel_dielectric(float eta, const f3& normal, const f3& outgoing) {           // Bxdf.cuh:59-79
    float cosw = __builtin_fabsf(dot(normal, outgoing));
    float sin2 = 1.f - cosw * cosw;
    float eta2 = eta * eta;
    float cos2t = 1.f - sin2 / eta2;
    if (cos2t < 0.f) return 1.f;
    float t0 = __builtin_sqrtf(cos2t);
    float t1 = eta * t0;
    float t2 = eta * cosw;
    float rs = (cosw - t1) / (cosw + t1);
    float rp = (t0 - t2) / (t0 + t2);
    return (rs * rs + rp * rp) / 2.f;
}

PT_DEV f3 fresnel_schlick(const f3& specular, const f3& normal, const f3& outgoing) {        // Bxdf.cuh:81-87
    if (sqlen(specular) < kEps) return f3(0.f, 0.f, 0.f);
    float cosine = dot(normal, outgoing);
    float p = cr_pow5(clampf(1.f - __builtin_fabsf(cosine), kEps, 0.999f));
    return specular + (1.f - specular) * p;
}

PT_DEV float ggx_d(float roughness, const f3& normal, const f3& halfway) {                   // Bxdf.cuh:89-101 (ggx branch)
    float cosine = dot(normal, halfway);
    if (cosine <= kEps) return 0.f;
    float r2 = roughness * roughness;
    float c2 = cosine * cosine;
    float divisor = (c2 * r2 + 1.f - c2);
    divisor = selmax(divisor, 1e-2f);
    return r2 / (kPi * divisor * divisor);
}

PT_DEV float ggx_g1(float roughness, const f3& normal, const f3& halfway, const f3& dir) {   // Bxdf.cuh:109-122 (ggx branch)
    float cosine = dot(normal, dir);
    float cosineh = dot(halfway, dir);
    if (cosine * cosineh <= 0.f) return 0.f;
    float r2 = roughness * roughness;
    float c2 = cosine * cosine;
    return 2.f * __builtin_fabsf(cosine) / (__builtin_fabsf(cosine) + __builtin_sqrtf(c2 - r2 * c2 + r2));
}
PT_DEV float ggx_g(float roughness, const f3& normal, const f3& halfway, const f3& o, const f3& i) { // Bxdf.cuh:132-137
    return ggx_g1(roughness, normal, halfway, o) * ggx_g1(roughness, normal, halfway, i);
}

PT_DEV f3 sample_microfacet(float roughness, const Frame& h, Rng& s) {                       // Bxdf.cuh:140-150
    float phi = (2.f * kPi) * s.uniform();
    float ry = s.uniform();
    float theta = cr_atan(roughness * __builtin_sqrtf(ry / (1.f - ry)));
    float cp, sp, ct, st;
    cr_sincos(phi, sp, cp);
    cr_sincos(theta, st, ct);
    float lx = cp * st, ly = sp * st, lz = ct;
    return lx * h.t + ly * h.b + lz * h.n;
}
PT_DEV float sample_microfacet_pdf(float roughness, const Frame& h, const f3& halfway) {     // Bxdf.cuh:153-158
    float cosine = dot(h.n, halfway);
    if (cosine < 0.f) return 0.f;
    return ggx_d(roughness, h.n, halfway) * cosine;
}

PT_DEV f3 sample_hemisphere(Rng& s, const Frame& h) {                                        // Bxdf.cuh:23-41
    float phi = (2.f * kPi) * s.uniform();
    float cosTheta = __builtin_sqrtf(s.uniform());
    float sinTheta = __builtin_sqrtf(1.f - cosTheta * cosTheta);
    float cosPhi, sinPhi;
    cr_sincos(phi, sinPhi, cosPhi);
    float x = cosPhi * sinTheta, y = sinPhi * sinTheta, z = cosTheta;
    return normalize(x * h.t + y * h.b + z * h.n);
}

// ---- gltfpbr ---------------------------------------------------------------------------
PT_DEV f3 eval_gltfpbr(const Mat& m, const Frame& h, const f3& o, const f3& i) {             // Bxdf.cuh:160-176
    float ni = dot(h.n, i), no = dot(h.n, o);
    if (ni * no <= 0.f) return f3(0.f, 0.f, 0.f);
    f3 reflectivity = lerp3(m.specular, m.albedo, m.metallic);
    f3 F1 = fresnel_schlick(reflectivity, h.n, o);
    f3 halfway = normalize(i + o);
    f3 F = fresnel_schlick(reflectivity, halfway, i);
    float D = ggx_d(m.roughness, h.n, halfway);
    float G = ggx_g(m.roughness, h.n, halfway, o, i);
    f3 k = (1.f - m.metallic) * (1.f - F1);
    float ani = __builtin_fabsf(ni);
    return ((m.albedo * k) * kInvPi) * ani + (((F * D) * G) / ((4.f * no) * ni)) * ani;
}
PT_DEV f3 sample_gltfpbr(const Mat& m, const Frame& h, const f3& o, Rng& s) {                // Bxdf.cuh:179-194
    f3 reflectivity = lerp3(m.specular, m.albedo, m.metallic);
    if (s.uniform() < mean3(fresnel_schlick(reflectivity, h.n, o))) {
        f3 halfway = sample_microfacet(m.roughness, h, s);
        f3 i = reflect(o, halfway);
        if (dot(h.n, i) * dot(h.n, o) < -kEps) return f3(0.f, 0.f, 0.f);
        return i;
    }
    return sample_hemisphere(s, h);
}
PT_DEV float pdf_gltfpbr(const Mat& m, const Frame& h, const f3& o, const f3& i) {           // Bxdf.cuh:197-207
    if (dot(h.n, i) * dot(h.n, o) <= 0.f) return 0.f;
    f3 halfway = normalize(o + i);
    f3 reflectivity = lerp3(m.specular, m.albedo, m.metallic);
    float F = mean3(fresnel_schlick(reflectivity, h.n, o));
    return (F * sample_microfacet_pdf(m.roughness, h, halfway)) / (4.f * __builtin_fabsf(dot(o, halfway))) +
           ((1.f - F) * dot(h.n, i)) * kInvPi;
}

// ---- reflective (delta mirror treated as non-delta by NEE) --------------------------------
PT_DEV f3 eval_reflective(const Mat& m, const Frame& h, const f3& o, const f3& i) {          // Bxdf.cuh:211-222
    float ni = dot(h.n, i), no = dot(h.n, o);
    if (ni * no <= 0.f) return f3(0.f, 0.f, 0.f);
    f3 reflectivity = lerp3(m.specular, m.albedo, m.metallic);
    f3 F1 = fresnel_schlick(reflectivity, h.n, o);
    f3 F = fresnel_schlick(reflectivity, h.n, i);
    f3 k = (1.f - m.metallic) * (1.f - F1);
    float ani = __builtin_fabsf(ni);
    return ((m.albedo * k) * kInvPi) * ani + F * ani;
}

// ---- refractive family ----------------------------------------------------------------------
struct RFrame { f3 normal, up; bool entering; float rel_ior; };
PT_DEV RFrame rframe(float ior, const Frame& h, const f3& o) {                                // Bxdf.cuh:238-241 (and its 5 repeats)
    RFrame f;
    f.normal = h.front ? h.n : -h.n;
    f.entering = dot(f.normal, o) >= 0.f;
    f.up = f.entering ? f.normal : -f.normal;
    f.rel_ior = f.entering ? ior : (1.f / ior);
    return f;
}
PT_DEV f3 refr_halfway(const RFrame& f, const f3& o, const f3& i) {                           // Bxdf.cuh:253-254
    return (-normalize(f.rel_ior * i + o)) * (f.entering ? 1.0f : -1.0f);
}

PT_DEV f3 eval_refractive(const Mat& m, float ior, const Frame& h, const f3& o, const f3& i) { // Bxdf.cuh:236-268
    RFrame f = rframe(ior, h, o);
    float ni = dot(f.normal, i), no = dot(f.normal, o);
    if (ni * no >= 0.f) {
        f3 halfway = normalize(i + o);
        float F = fresnel_dielectric(f.rel_ior, halfway, o);
        float D = ggx_d(m.roughness, f.up, halfway);
        float G = ggx_g(m.roughness, f.up, halfway, o, i);
        return ((((m.albedo * F) * D) * G) / __builtin_fabsf((4.f * no) * ni)) * __builtin_fabsf(ni);
    } else {
        f3 halfway = refr_halfway(f, o, i);
        float F = fresnel_dielectric(f.rel_ior, halfway, o);
        float D = ggx_d(m.roughness, f.up, halfway);
        float G = ggx_g(m.roughness, f.up, halfway, o, i);
        float A = __builtin_fabsf((dot(o, halfway) * dot(i, halfway)) / (dot(o, f.normal) * dot(i, f.normal)));
        float den = cr_pow2(f.rel_ior * dot(halfway, i) + dot(halfway, o));
        return (((((m.albedo * A) * (1.f - F)) * D) * G) / den) * __builtin_fabsf(ni);
    }
}
PT_DEV f3 sample_refractive(const Mat& m, float ior, const Frame& h, const f3& o, Rng& s) {   // Bxdf.cuh:271-288
    RFrame f = rframe(ior, h, o);
    f3 halfway = sample_microfacet(m.roughness, h, s);
    if (s.uniform() < fresnel_dielectric(f.entering ? ior : (1.f / ior), halfway, o)) {
        f3 i = reflect(o, halfway);
        if (!(dot(f.normal, o) * dot(f.normal, i) >= 0.f)) return f3(0.f, 0.f, 0.f);
        return i;
    } else {
        f3 i = refract(o, halfway, f.entering ? (1.f / ior) : ior);
        if (dot(f.normal, o) * dot(f.normal, i) >= 0.f) return f3(0.f, 0.f, 0.f);
        return i;
    }
}
PT_DEV float pdf_refractive(const Mat& m, float ior, const Frame& h, const f3& o, const f3& i) { // Bxdf.cuh:291-315
    RFrame f = rframe(ior, h, o);
    if (dot(f.normal, i) * dot(f.normal, o) >= 0.f) {
        f3 halfway = normalize(i + o);
        return (fresnel_dielectric(f.rel_ior, halfway, o) * sample_microfacet_pdf(m.roughness, h, halfway)) /
               (4.f * __builtin_fabsf(dot(o, halfway)));
    } else {
        f3 halfway = refr_halfway(f, o, i);
        return (((1.f - fresnel_dielectric(f.rel_ior, halfway, o)) * sample_microfacet_pdf(m.roughness, h, halfway)) *
                __builtin_fabsf(dot(halfway, i))) /
               cr_pow2(f.rel_ior * dot(halfway, i) + dot(halfway, o));
    }
}

PT_DEV f3 eval_pure_refractive(const Mat& m, float ior, const Frame& h, const f3& o, const f3& i) { // Bxdf.cuh:317-334
    RFrame f = rframe(ior, h, o);
    if (dot(f.normal, i) * dot(f.normal, o) >= 0.f) {
        f3 halfway = normalize(i + o);
        float F = fresnel_dielectric(f.rel_ior, halfway, o);
        return m.albedo * F;
    } else {
        f3 halfway = refr_halfway(f, o, i);
        float F = fresnel_dielectric(f.rel_ior, halfway, o);
        return (m.albedo * (1.f - F)) / (f.rel_ior * f.rel_ior);
    }
}
PT_DEV f3 sample_pure_refractive(float ior, const Frame& h, const f3& o, Rng& s) {            // Bxdf.cuh:337-351
    RFrame f = rframe(ior, h, o);
    if (s.uniform() < fresnel_dielectric(f.entering ? ior : (1.f / ior), f.up, o)) return reflect(o, f.up);
    return refract(o, f.up, f.entering ? (1.f / ior) : ior);
}
PT_DEV float pdf_pure_refractive(float ior, const Frame& h, const f3& o, const f3& i) {       // Bxdf.cuh:354-370
    RFrame f = rframe(ior, h, o);
    if (dot(f.normal, i) * dot(f.normal, o) >= 0.f) {
        f3 halfway = normalize(i + o);
        return fresnel_dielectric(f.rel_ior, halfway, o);
    } else {
        f3 halfway = refr_halfway(f, o, i);
        return 1.f - fresnel_dielectric(f.rel_ior, halfway, o);
    }
}

// ---- lobe dispatch ---------------------------------------------------------------------------
PT_DEV f3 lobe_eval(int lobe, const Mat& m, float ior, const Frame& h, const f3& o, const f3& i) {
    switch (lobe) {
    case LOBE_GLTFPBR: return eval_gltfpbr(m, h, o, i);
    case LOBE_REFLECTIVE: return eval_reflective(m, h, o, i);
    case LOBE_REFRACTIVE: return eval_refractive(m, ior, h, o, i);
    default: return eval_pure_refractive(m, ior, h, o, i);
    }
}
PT_DEV f3 lobe_sample(int lobe, const Mat& m, float ior, const Frame& h, const f3& o, Rng& s) {
    switch (lobe) {
    case LOBE_GLTFPBR: return sample_gltfpbr(m, h, o, s);
    case LOBE_REFLECTIVE: return reflect(o, h.n);                                             // Bxdf.cuh:225-228
    case LOBE_REFRACTIVE: return sample_refractive(m, ior, h, o, s);
    default: return sample_pure_refractive(ior, h, o, s);
    }
}
PT_DEV float lobe_pdf(int lobe, const Mat& m, float ior, const Frame& h, const f3& o, const f3& i) {
    switch (lobe) {
    case LOBE_GLTFPBR: return pdf_gltfpbr(m, h, o, i);
    case LOBE_REFLECTIVE: return 1.f;                                                         // Bxdf.cuh:231-234
    case LOBE_REFRACTIVE: return pdf_refractive(m, ior, h, o, i);
    default: return pdf_pure_refractive(ior, h, o, i);
    }
}

}  // namespace ptd
