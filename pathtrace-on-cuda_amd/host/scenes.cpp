// scenes.cpp — procedural scenes (C-ABI pt_scene_gen).
//
// The reference ships no scene: its models are git-ignored and loaded from absolute
// Windows paths (srcs/renderer.cpp:102-115, .gitignore:365).  These generators produce the
// scenes SURVEY.md Appendix A specifies, as `Primitive` arrays in the order a
// `BVH::AddModel` call sequence would have appended them (insertion order is part of the
// result: it is the initial order the unstable sort of the BVH build sees).
//
// Vertex attributes follow what Model::processMesh produces for a mesh without tangents
// (include/model.h:159-171): flat normal, tangent = normalize(-n.z,0,n.x) if |n.x|>|n.y|
// else normalize(0,n.z,-n.y), bitangent = cross(n, tangent); material defaults
// specular .04, opacity 1 (include/model.h:174-186).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/pt_api.h"

void pt_set_error(const char* fmt, ...);

namespace {

struct V { float x, y, z; };
inline V sub(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V crossg(V x, V y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }   // glm::cross
inline V normg(V v) { const float k = 1.0f / std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z); return {v.x * k, v.y * k, v.z * k}; } // glm::normalize

struct Gen {
    PtPrimitive* out;
    int cap;
    int count = 0;

    void tri(V a, V b, V c, V albedo, V emit, float rough, float metal)
    {
        if (out && count < cap) {
            const V n = normg(crossg(sub(b, a), sub(c, a)));
            V t;
            if (std::fabs(n.x) > std::fabs(n.y)) t = normg(V{-n.z, 0.f, n.x});
            else t = normg(V{0.f, n.z, -n.y});
            const V bt = crossg(n, t);
            PtPrimitive p;
            memset(&p, 0, sizeof(p));
            const V pos[3] = {a, b, c};
            PtVertex* vs[3] = {&p.v1, &p.v2, &p.v3};
            for (int i = 0; i < 3; i++) {
                PtVertex& v = *vs[i];
                v.Position = {pos[i].x, pos[i].y, pos[i].z};
                v.Normal = {n.x, n.y, n.z};
                v.TexCoords = {0.f, 0.f};
                v.Tangent = {t.x, t.y, t.z};
                v.Bitangent = {bt.x, bt.y, bt.z};
                v.mat.emittance = {emit.x, emit.y, emit.z};
                v.mat.albedo = {albedo.x, albedo.y, albedo.z};
                v.mat.specular = {0.04f, 0.04f, 0.04f};
                v.mat.opacity = 1.f; v.mat.metallic = metal; v.mat.roughness = rough;
                v.u = 0.f; v.v = 0.f;
            }
            out[count] = p;
        }
        count++;
    }
    void quad(V a, V b, V c, V d, V albedo, V emit)
    {
        tri(a, b, c, albedo, emit, 1.f, 0.f);
        tri(a, c, d, albedo, emit, 1.f, 0.f);
    }

    void cornell()
    {
        const float s = 20.f;
        const V white{.73f, .73f, .73f}, red{.65f, .05f, .05f}, green{.12f, .45f, .15f}, black{0.f, 0.f, 0.f}, none{0.f, 0.f, 0.f};
        quad({-s, 0, s}, {s, 0, s}, {s, 0, -s}, {-s, 0, -s}, white, none);                    // floor
        quad({-s, 2 * s, -s}, {s, 2 * s, -s}, {s, 2 * s, s}, {-s, 2 * s, s}, white, none);     // ceiling
        quad({-s, 0, -s}, {s, 0, -s}, {s, 2 * s, -s}, {-s, 2 * s, -s}, white, none);           // back
        quad({-s, 0, s}, {-s, 0, -s}, {-s, 2 * s, -s}, {-s, 2 * s, s}, red, none);             // left
        quad({s, 0, -s}, {s, 0, s}, {s, 2 * s, s}, {s, 2 * s, -s}, green, none);               // right
        quad({-5, 39.98f, -5}, {5, 39.98f, -5}, {5, 39.98f, 5}, {-5, 39.98f, 5}, black, V{15.f, 15.f, 15.f});   // light
    }

    // "bunny stand-in": a bumpy UV sphere, LAT = LON = n
    void standin(int n, float R, V C)
    {
        auto P = [&](int i, int j) -> V {
            const float th = 3.14159265f * (float)i / (float)n;
            const float ph = 6.2831853f * (float)j / (float)n;
            const float r = R * (1.f + 0.08f * sinf(7.f * th) * cosf(5.f * ph));
            return V{C.x + r * (sinf(th) * cosf(ph)), C.y + r * cosf(th), C.z + r * (sinf(th) * sinf(ph))};
        };
        const V albedo{.8f, .6f, .2f}, none{0.f, 0.f, 0.f};
        for (int i = 0; i < n; i++)
            for (int j = 0; j < n; j++) {
                const V a = P(i, j), b = P(i + 1, j), c = P(i + 1, j + 1), d = P(i, j + 1);
                if (i > 0) tri(a, d, c, albedo, none, 0.f, 1.f);
                if (i < n - 1) tri(a, c, b, albedo, none, 0.f, 1.f);
            }
    }
};

}  // namespace

extern "C" int32_t pt_scene_gen(int32_t kind, int32_t lat_lon, PtPrimitive* prims, int32_t cap)
{
    if (kind < 0 || kind > 2 || (kind > 0 && lat_lon < 3)) { pt_set_error("pt_scene_gen: bad kind/lat_lon"); return PT_ERR_INVALID; }
    Gen g;
    g.out = prims; g.cap = prims ? cap : 0;
    g.cornell();
    if (kind == 1) g.standin(lat_lon, 10.f, V{0.f, 11.f, 0.f});
    if (kind == 2)
        for (int k = 0; k < 4; k++) g.standin(lat_lon, 6.f, V{(k & 1) ? 8.f : -8.f, 7.f, (k & 2) ? 6.f : -8.f});
    return g.count;
}
