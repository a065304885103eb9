// integration/pathtracer_mi355x.cpp — the reference-side binding of libptamd.so.
//
// A maintainer of WaterPlease/PathTrace-on-CUDA adds THIS file to the viewer's build in place of
//   srcs/pathtracer.cu      (PathTracer::Render + the StartRender kernel, :24-259)
//   srcs/CudaPrimitive.cu   (the three global vectors + LoadFromBVH, :3-145)
// puts include/pt_api.h of this repo on the include path and links libptamd.so.  Everything else of the viewer — Camera
// (srcs/camera.cpp), BVH / SAHBVH (srcs/bvh.cpp), Image (srcs/image.cpp + the vendored stb), Renderer, Model — stays the
// reference's own code, compiled as it is today.  include/CudaUtil.cuh, include/Bxdf.cuh and cuRAND are no longer needed.
//
// It is compiled here against the reference's REAL headers (oracle/Makefile: `make ref`, into the git-ignored oracle/_ref/;
// tests/test_integration_binding.py) and linked with the reference's own camera.cpp / bvh.cpp / image.cpp into a headless host
// (integration/headless_host.cpp) that the GPU tests run.  The product never loads it.
//
// What it provides, with the reference's own types (include/CudaPrimitive.cuh — so there is exactly one Material / Sphere /
// CudaSpheres in the program, the ones srcs/renderer.cpp:16,126-144 already uses):
//   std::vector<CudaBVHNode> CudaBVH; std::vector<Triangle> CudaPrims; std::vector<Sphere> CudaSpheres;   (CudaPrimitive.cuh:325-327)
//   void LoadFromBVH(BVH*)                                                                                (CudaPrimitive.cuh:337)
//   void PathTracer::Render(Camera&, BVH*)                                                                (pathtracer.cuh:3-7)
#include "pathtracer.cuh"          // reference: class PathTracer
#include "CudaPrimitive.cuh"       // reference: Material, Sphere, Triangle, CudaBVHNode, the extern globals; pulls bvh.h / mesh.h / glm
#include "camera.h"                // reference: class Camera
#include "image.h"                 // reference: class Image, ConverToUint8
#include "pt_api.h"                // this repo: the C-ABI

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <vector>

// Layout facts the C-ABI relies on (sizes printed by the real reference build: oracle/_ref/ptref sizes).
static_assert(sizeof(Primitive) == sizeof(PtPrimitive), "Vertex layout (include/mesh.h:21-37) must match PtVertex");
static_assert(sizeof(Material) == sizeof(PtMaterial), "Material (include/CudaPrimitive.cuh:15-23) must match PtMaterial");
static_assert(sizeof(Sphere) == sizeof(PtSphere), "Sphere data members (include/CudaPrimitive.cuh:318-322) must match PtSphere");
static_assert(sizeof(CudaBVHNode) == sizeof(PtBVHNode), "CudaBVHNode (include/CudaPrimitive.cuh:237-247) must match PtBVHNode");
static_assert(sizeof(vec3) == 12, "vec3 is three floats");

// The one definition of the globals the app links against (they lived in srcs/CudaPrimitive.cu:3-5).
std::vector<CudaBVHNode> CudaBVH;
std::vector<Triangle>    CudaPrims;
std::vector<Sphere>      CudaSpheres;

namespace {

// Compile-time tunables of the reference (include/CudaUtil.cuh:15-19) are run-time PtParams here.  A maintainer who wants other
// sample counts edits this one function, exactly where the #defines used to be edited.  The environment overrides exist for
// headless runs and tests only.
PtParams render_params()
{
    PtParams p;
    pt_params_default(&p);        // NUM_MULTI_SAMPLE 8, NUM_SAMPLE 1024, MAX_BOUNCE 8, RUSSIAN_ROULETTE_BOUNCE 3, PROB_STOP_BOUNCE 0.5
    if (const char* e = getenv("PT_NUM_MULTI_SAMPLE")) p.passes = atoi(e);
    if (const char* e = getenv("PT_NUM_SAMPLE")) p.spp_per_pass = atoi(e);
    if (const char* e = getenv("PT_MAX_BOUNCE")) p.max_bounce = atoi(e);
    return p;
}

[[noreturn]] void die(const char* what)      // the reference's convention: message on stderr, exit(99) (include/CudaUtil.cuh:28-36)
{
    std::cerr << "GPU error in " << what << " : " << pt_last_error() << std::endl;
    exit(99);
}

PtFlatBVH* g_flat = nullptr;      // what the last LoadFromBVH built; PathTracer::Render uploads it

vec3 v3(const float* f) { return vec3(f[0], f[1], f[2]); }
Material mat_of(const PtMaterial& m)
{
    Material r;
    r.emittance = v3(m.emittance); r.albedo = v3(m.albedo); r.specular = v3(m.specular);
    r.opacity = m.opacity; r.roughness = m.roughness; r.metallic = m.metallic;
    return r;
}

// exportImage, srcs/pathtracer.cu:94-122: raw / SampleCnt -> ACESFilm -> ConverToUint8, then Image::WriteTo (the reference's own
// srcs/image.cpp + stb_image_write.h, which stay in the viewer's build).
void exportImage(Image& img, const float* rawData, const char* path, int H, int W, int SampleCnt)
{
    if (pt_tonemap_u8(rawData, (int64_t)H * W, SampleCnt, img.GetData()) != PT_OK) die("pt_tonemap_u8");
    if (img.WriteTo(path)) std::cout << "Export Success" << std::endl;
    else std::cout << "Export failed" << std::endl;
}

}  // namespace

// LoadFromBVH, srcs/CudaPrimitive.cu:8-145: flatten bvh->rootBVH into CudaBVH / CudaPrims.  Here the flat arrays are rebuilt from
// bvh->primitives (SAHBVH::GenBVHTree's input, srcs/bvh.cpp:426-511) by pt_bvh_build_sah; they are byte-identical to what the
// reference's own GenBVHTree + LoadFromBVH produce (tests/test_oracle_ref.py), so the pointer tree is not walked.
void LoadFromBVH(BVH* bvh)
{
    if (g_flat) { pt_bvh_free(g_flat); g_flat = nullptr; }
    if (pt_bvh_build_sah((const PtPrimitive*)bvh->primitives.data(), (int32_t)bvh->primitives.size(), &g_flat) != PT_OK) die("pt_bvh_build_sah");
    const int nN = pt_bvh_num_nodes(g_flat), nT = pt_bvh_num_tris(g_flat);
    CudaBVH.resize((size_t)nN);
    if (nN) memcpy((void*)CudaBVH.data(), pt_bvh_nodes(g_flat), (size_t)nN * sizeof(CudaBVHNode));
    // CudaPrims keeps the reference's Triangle objects (vptr and all) for whatever else in the app looks at them
    CudaPrims.clear();
    CudaPrims.resize((size_t)nT);
    const PtTriangle* t = pt_bvh_tris(g_flat);
    for (int i = 0; i < nT; i++)
        CudaPrims[(size_t)i].Copy(v3(t[i].V0), v3(t[i].V1), v3(t[i].V2), v3(t[i].T0), v3(t[i].T1), v3(t[i].T2),
                                  v3(t[i].B0), v3(t[i].B1), v3(t[i].B2), v3(t[i].N0), v3(t[i].N1), v3(t[i].N2),
                                  mat_of(t[i].mat0), mat_of(t[i].mat1), mat_of(t[i].mat2),
                                  t[i].u0, t[i].u1, t[i].u2, t[i].v0, t[i].v1, t[i].v2);
    std::cout << "Maximum depth of tree : " << pt_bvh_max_depth(g_flat) << std::endl;      // srcs/CudaPrimitive.cu:144
}

// PathTracer::Render, srcs/pathtracer.cu:124-259.  Same console lines, same two PNG side effects (temp.png after every pass,
// result.png at the end, both in the CWD), same error convention.
void PathTracer::Render(Camera& camera, BVH* bvh)
{
    using namespace std::chrono;
    std::cout << "Camera : " << camera.Screen_W << " x " << camera.Screen_H << std::endl;
    const int W = (int)camera.Screen_W, H = (int)camera.Screen_H, C = 3;
    Image img(W, H, C);

    LoadFromBVH(bvh);
    std::cout << "Tree on GPU Size : " << CudaBVH.size() << std::endl;
    std::cout << "Prim on GPU Size : " << CudaPrims.size() << std::endl;
    std::cout << "Prim on CPU Size : " << bvh->primitives.size() << std::endl;

    std::cout << "Upload world on GPU" << std::endl;                                  // :142-188
    int device = 0;
    if (const char* e = getenv("PT_DEVICE")) device = atoi(e);
    PtScene* scene = nullptr;
    if (pt_scene_create(pt_bvh_nodes(g_flat), pt_bvh_num_nodes(g_flat), pt_bvh_tris(g_flat), pt_bvh_num_tris(g_flat),
                        CudaSpheres.empty() ? nullptr : (const PtSphere*)CudaSpheres.data(), (int32_t)CudaSpheres.size(),
                        device, &scene) != PT_OK) die("pt_scene_create");
    for (int i = 0; i < pt_scene_num_lights(scene); i++) std::cout << "ADD light" << std::endl;      // :171

    std::cout << "Upload camera configuration on GPU" << std::endl << std::endl << std::endl;      // :191-210
    PtCamera cam;
    const glm::vec3 p = camera.pos, f = camera.GetForward(), u = camera.GetUp(), r = camera.GetRight();
    memcpy(cam.pos, &p, 12); memcpy(cam.forward, &f, 12); memcpy(cam.up, &u, 12); memcpy(cam.right, &r, 12);
    cam.fovy_deg = camera.fovy; cam.aspect = camera.aspect; cam.W = W; cam.H = H;

    const PtParams prm = render_params();
    std::vector<float> rawData((size_t)W * H * 3, 0.f), pass((size_t)W * H * 3);

    const auto t0 = system_clock::now();
    for (int i = 0; i < prm.passes; i++) {                                             // the StartRender launch loop, :236-246
        // one call per pass, summed in pass order: bit-identical to one multi-pass call (the device side adds the per-pass
        // means in the same order), and temp.png can be rewritten after every pass as the reference does
        PtParams one = prm; one.passes = 1; one.first_pass = i;                        // SampleIDX = i, :71
        if (pt_render(scene, &cam, &one, pass.data()) != PT_OK) die("pt_render");
        for (size_t k = 0; k < rawData.size(); k++) rawData[k] += pass[k];            // image[offset] += mean, :81
        std::cout << "Sample " << i << " : Delta time : " << duration_cast<milliseconds>(system_clock::now() - t0).count() << " (ms)" << std::endl;
        exportImage(img, rawData.data(), "temp.png", H, W, i + 1);
    }
    std::cout << "Delta time : " << duration_cast<milliseconds>(system_clock::now() - t0).count() << " (ms)" << std::endl;
    exportImage(img, rawData.data(), "result.png", H, W, prm.passes);

    // headless runs and tests: the float accumulation buffer itself, so a frame can be compared bit for bit
    if (const char* rp = getenv("PT_RAW_OUT")) {
        if (FILE* fp = fopen(rp, "wb")) { fwrite(rawData.data(), 4, rawData.size(), fp); fclose(fp); }
    }
    pt_scene_destroy(scene);                                                            // :253-258
}
