// obj_loader.cpp — Wavefront OBJ(+MTL) ingest without assimp (C-ABI pt_load_obj).
//
// SURVEY.md §8(f) row 1.  Produces the `Primitive` records BVH::AddModel (srcs/bvh.cpp:153-189)
// would append for a Model loaded by include/model.h:74-207, for the subset of OBJ the
// reference's scenes use (triangles/polygons, optional normals/texcoords, MTL colours):
//   * faces are triangulated as fans (aiProcess_Triangulate);
//   * missing normals become smooth per-position normals: the normalised sum of the
//     un-normalised face normals of all faces sharing the position (aiProcess_GenSmoothNormals);
//   * without texture coordinates the tangent frame is the reference's own fallback
//     (include/model.h:159-171): t = normalize(-n.z,0,n.x) if |n.x| > |n.y| else normalize(0,n.z,-n.y),
//     b = cross(n,t); with texture coordinates a per-triangle UV tangent is used;
//   * material: Kd -> albedo, Ke -> emittance, Ks -> specular, d -> opacity, Pm -> metallic,
//     Pr -> roughness, with the reference's defaults when a key is absent
//     (include/model.h:174-186: albedo 0, emittance 0, specular .04, metallic 0, roughness 0, opacity 1);
//   * the model matrix T*S (uniform scale, then translate; the reference's scenes use no rotation,
//     srcs/renderer.cpp:102-118) is baked into positions (w=1) and normal/tangent/bitangent (w=0,
//     left un-normalised exactly like AddModel; LoadFromBVH normalises them later).
// PARITY UNPINNED: assimp is not available in this image, so vertex order inside assimp's
// meshes, its exact smoothing weights and tangent-space results cannot be compared.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/pt_api.h"

void pt_set_error(const char* fmt, ...);

namespace {

struct V3 { float x, y, z; };
inline V3 sub(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 add(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 mul(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V3 crossg(V3 x, V3 y) { return {x.y * y.z - y.y * x.z, x.z * y.x - y.z * x.x, x.x * y.y - y.x * x.y}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 normg(V3 v) { const float l2 = dot(v, v); if (!(l2 > 0.f)) return {0.f, 1.f, 0.f}; const float k = 1.0f / std::sqrt(l2); return mul(v, k); }

struct Mtl { V3 kd{0, 0, 0}, ke{0, 0, 0}, ks{0.04f, 0.04f, 0.04f}; float d = 1.f, pm = 0.f, pr = 0.f; };

struct Corner { int v, vt, vn; };

int fix_index(int i, int n) { return i > 0 ? i - 1 : (i < 0 ? n + i : -1); }

bool load_mtl(const std::string& path, std::map<std::string, Mtl>& out)
{
    std::ifstream f(path);
    if (!f) return false;
    std::string line, cur;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string k; ss >> k;
        if (k == "newmtl") { ss >> cur; out[cur] = Mtl(); }
        else if (cur.empty()) continue;
        else if (k == "Kd") ss >> out[cur].kd.x >> out[cur].kd.y >> out[cur].kd.z;
        else if (k == "Ke") ss >> out[cur].ke.x >> out[cur].ke.y >> out[cur].ke.z;
        else if (k == "Ks") ss >> out[cur].ks.x >> out[cur].ks.y >> out[cur].ks.z;
        else if (k == "d") ss >> out[cur].d;
        else if (k == "Tr") { float tr; if (ss >> tr) out[cur].d = 1.f - tr; }
        else if (k == "Pm") ss >> out[cur].pm;
        else if (k == "Pr") ss >> out[cur].pr;
    }
    return true;
}

}  // namespace

extern "C" int32_t pt_load_obj(const char* path, float scale, const float translate[3], PtPrimitive* prims, int32_t cap)
{
    if (!path) { pt_set_error("pt_load_obj: NULL path"); return PT_ERR_INVALID; }
    std::ifstream f(path);
    if (!f) { pt_set_error("pt_load_obj: cannot open %s", path); return PT_ERR_IO; }
    const V3 T = translate ? V3{translate[0], translate[1], translate[2]} : V3{0, 0, 0};
    std::string dir(path);
    const size_t slash = dir.find_last_of("/\\");
    dir = (slash == std::string::npos) ? std::string() : dir.substr(0, slash + 1);

    std::vector<V3> P, N; std::vector<float> UV;
    struct Tri { Corner c[3]; int mtl; };
    std::vector<Tri> tris;
    std::map<std::string, Mtl> mtls;
    std::vector<Mtl> mtlList(1);           // index 0 = the reference's defaults
    std::map<std::string, int> mtlIndex;
    int curMtl = 0;
    std::string line;
    int lineNo = 0;
    while (std::getline(f, line)) {
        lineNo++;
        if (line.empty() || line[0] == '#') continue;
        std::istringstream ss(line);
        std::string k; ss >> k;
        if (k == "v") { V3 p; if (!(ss >> p.x >> p.y >> p.z)) { pt_set_error("pt_load_obj: %s:%d bad vertex", path, lineNo); return PT_ERR_IO; } P.push_back(p); }
        else if (k == "vn") { V3 n; ss >> n.x >> n.y >> n.z; N.push_back(n); }
        else if (k == "vt") { float u = 0, v = 0; ss >> u >> v; UV.push_back(u); UV.push_back(1.f - v); }   // aiProcess_FlipUVs
        else if (k == "mtllib") { std::string m; ss >> m; load_mtl(dir + m, mtls); }
        else if (k == "usemtl") {
            std::string m; ss >> m;
            auto it = mtlIndex.find(m);
            if (it != mtlIndex.end()) curMtl = it->second;
            else { curMtl = (int)mtlList.size(); mtlIndex[m] = curMtl; mtlList.push_back(mtls.count(m) ? mtls[m] : Mtl()); }
        } else if (k == "f") {
            std::vector<Corner> cs;
            std::string tok;
            while (ss >> tok) {
                Corner c{-1, -1, -1};
                int a = 0, b = 0, d = 0;
                if (sscanf(tok.c_str(), "%d/%d/%d", &a, &b, &d) == 3) { c.v = a; c.vt = b; c.vn = d; }
                else if (sscanf(tok.c_str(), "%d//%d", &a, &d) == 2) { c.v = a; c.vn = d; c.vt = 0; }
                else if (sscanf(tok.c_str(), "%d/%d", &a, &b) == 2) { c.v = a; c.vt = b; c.vn = 0; }
                else if (sscanf(tok.c_str(), "%d", &a) == 1) { c.v = a; c.vt = 0; c.vn = 0; }
                else { pt_set_error("pt_load_obj: %s:%d bad face token '%s'", path, lineNo, tok.c_str()); return PT_ERR_IO; }
                c.v = fix_index(c.v, (int)P.size()); c.vt = fix_index(c.vt, (int)UV.size() / 2); c.vn = fix_index(c.vn, (int)N.size());
                if (c.v < 0 || c.v >= (int)P.size()) { pt_set_error("pt_load_obj: %s:%d vertex index out of range", path, lineNo); return PT_ERR_IO; }
                if (c.vn >= (int)N.size()) c.vn = -1;
                if (c.vt >= (int)UV.size() / 2) c.vt = -1;
                cs.push_back(c);
            }
            for (size_t i = 2; i < cs.size(); i++) tris.push_back(Tri{{cs[0], cs[i - 1], cs[i]}, curMtl});
        }
    }
    if (tris.empty()) { pt_set_error("pt_load_obj: %s has no faces", path); return PT_ERR_IO; }

    // smooth normals per position where the file gives none
    std::vector<V3> smooth;
    bool needSmooth = false;
    for (auto& t : tris) for (auto& c : t.c) if (c.vn < 0) needSmooth = true;
    if (needSmooth) {
        smooth.assign(P.size(), V3{0, 0, 0});
        for (auto& t : tris) {
            const V3 fn = crossg(sub(P[(size_t)t.c[1].v], P[(size_t)t.c[0].v]), sub(P[(size_t)t.c[2].v], P[(size_t)t.c[0].v]));
            for (auto& c : t.c) smooth[(size_t)c.v] = add(smooth[(size_t)c.v], fn);
        }
        for (auto& n : smooth) n = normg(n);
    }

    const int32_t total = (int32_t)tris.size();
    if (!prims) return total;
    const int32_t n = total < cap ? total : cap;
    for (int32_t i = 0; i < n; i++) {
        const Tri& t = tris[(size_t)i];
        const Mtl& m = mtlList[(size_t)t.mtl];
        PtPrimitive pr;
        memset(&pr, 0, sizeof(pr));
        PtVertex* vs[3] = {&pr.v1, &pr.v2, &pr.v3};
        // per-triangle UV tangent (used only when all three corners have texture coordinates)
        bool hasUV = t.c[0].vt >= 0 && t.c[1].vt >= 0 && t.c[2].vt >= 0;
        V3 uvT{0, 0, 0}, uvB{0, 0, 0};
        if (hasUV) {
            const V3 e1 = sub(P[(size_t)t.c[1].v], P[(size_t)t.c[0].v]), e2 = sub(P[(size_t)t.c[2].v], P[(size_t)t.c[0].v]);
            const float du1 = UV[2 * (size_t)t.c[1].vt] - UV[2 * (size_t)t.c[0].vt], dv1 = UV[2 * (size_t)t.c[1].vt + 1] - UV[2 * (size_t)t.c[0].vt + 1];
            const float du2 = UV[2 * (size_t)t.c[2].vt] - UV[2 * (size_t)t.c[0].vt], dv2 = UV[2 * (size_t)t.c[2].vt + 1] - UV[2 * (size_t)t.c[0].vt + 1];
            const float det = du1 * dv2 - du2 * dv1;
            if (std::fabs(det) > 1e-20f) {
                const float r = 1.f / det;
                uvT = normg(mul(sub(mul(e1, dv2), mul(e2, dv1)), r));
                uvB = normg(mul(sub(mul(e2, du1), mul(e1, du2)), r));
            } else hasUV = false;
        }
        for (int k = 0; k < 3; k++) {
            const Corner& c = t.c[k];
            PtVertex& v = *vs[k];
            const V3 p = P[(size_t)c.v];
            const V3 nrm = c.vn >= 0 ? N[(size_t)c.vn] : smooth[(size_t)c.v];
            V3 tan, bit;
            if (hasUV) {
                v.TexCoords = {UV[2 * (size_t)c.vt], UV[2 * (size_t)c.vt + 1]};
                tan = uvT; bit = uvB;
            } else {
                if (std::fabs(nrm.x) > std::fabs(nrm.y)) tan = normg(V3{-nrm.z, 0.f, nrm.x});
                else tan = normg(V3{0.f, nrm.z, -nrm.y});
                bit = crossg(nrm, tan);
            }
            // BVH::AddModel: M * vec4(p,1), M * vec4(n,0) with M = translate * scale
            v.Position = {scale * p.x + T.x, scale * p.y + T.y, scale * p.z + T.z};
            v.Normal = {scale * nrm.x, scale * nrm.y, scale * nrm.z};
            v.Tangent = {scale * tan.x, scale * tan.y, scale * tan.z};
            v.Bitangent = {scale * bit.x, scale * bit.y, scale * bit.z};
            v.mat.emittance = {m.ke.x, m.ke.y, m.ke.z};
            v.mat.albedo = {m.kd.x, m.kd.y, m.kd.z};
            v.mat.specular = {m.ks.x, m.ks.y, m.ks.z};
            v.mat.opacity = m.d; v.mat.metallic = m.pm; v.mat.roughness = m.pr;
            v.u = 0.f; v.v = 0.f;
        }
        prims[i] = pr;
    }
    return total;
}
