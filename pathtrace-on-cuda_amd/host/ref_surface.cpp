// ref_surface.cpp — see ref_surface.h.  Thin adaptors over the C-ABI.
#include "ref_surface.h"
#include <cstdio>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>

std::vector<CudaBVHNode> CudaBVH;
std::vector<Triangle> CudaPrims;
std::vector<Sphere> CudaSpheres;

void Sphere::memset_zero() { memset(static_cast<PtSphere*>(this), 0, sizeof(PtSphere)); }

Camera::Camera() : pos(0.f, 0.f, 0.f), rotation(0.f, 90.f, 0.f)
{
    fovy = 45.f; aspect = 16.f / 9.f; near = 1.f; far = 5000.f;
    SetRotation(rotation);
}
Camera::Camera(vec3f _pos) : Camera() { pos = _pos; }
const Camera& Camera::AddRotation(vec3f d) { return SetRotation(vec3f(rotation.x + d.x, rotation.y + d.y, rotation.z + d.z)); }
const Camera& Camera::SetRotation(vec3f r)
{
    rotation = r;
    const float rot[3] = {r.x, r.y, r.z};
    float f[3], u[3], rt[3];
    pt_camera_basis(rot, f, u, rt);
    forward = vec3f(f[0], f[1], f[2]); up = vec3f(u[0], u[1], u[2]); right = vec3f(rt[0], rt[1], rt[2]);
    return *this;
}

BVH::~BVH() { if (flat) pt_bvh_free(flat); }
void BVH::GenBVHTree()
{
    if (flat) { pt_bvh_free(flat); flat = nullptr; }
    if (pt_bvh_build_sah(primitives.data(), (int)primitives.size(), &flat) != PT_OK) {
        std::cerr << "BVH build failed: " << pt_last_error() << std::endl;
        exit(99);
    }
}
bool BVH::AddOBJ(const std::string& path, float scale, vec3f t)
{
    const float tr[3] = {t.x, t.y, t.z};
    const int n = pt_load_obj(path.c_str(), scale, tr, nullptr, 0);
    if (n < 0) { std::cerr << "ERROR::OBJ:: " << pt_last_error() << std::endl; return false; }
    const size_t base = primitives.size();
    primitives.resize(base + (size_t)n);
    return pt_load_obj(path.c_str(), scale, tr, primitives.data() + base, n) == n;
}

void LoadFromBVH(BVH* bvh)
{
    if (!bvh->flat) bvh->GenBVHTree();
    CudaBVH.assign(pt_bvh_nodes(bvh->flat), pt_bvh_nodes(bvh->flat) + pt_bvh_num_nodes(bvh->flat));
    CudaPrims.assign(pt_bvh_tris(bvh->flat), pt_bvh_tris(bvh->flat) + pt_bvh_num_tris(bvh->flat));
    std::cout << "Maximum depth of tree : " << pt_bvh_max_depth(bvh->flat) << std::endl;      // srcs/CudaPrimitive.cu:144
}

Image::Image(int W, int H, int C) : width(W), height(H), nrChannels(C) { data = (unsigned char*)malloc((size_t)W * H * C); }
Image::~Image() { free(data); }
bool Image::WriteTo(const char* path) { return pt_write_png(path, data, width, height, nrChannels) == PT_OK; }

static void check(int rc, const char* what)
{
    if (rc != PT_OK) {      // the reference's convention: message on stderr, exit(99) (include/CudaUtil.cuh:28-36)
        std::cerr << "GPU error in " << what << " : " << pt_last_error() << std::endl;
        exit(99);
    }
}

static void exportImage(Image& img, const float* raw, const char* path, int H, int W, int SampleCnt)
{
    pt_tonemap_u8(raw, (int64_t)H * W, SampleCnt, img.GetData());          // srcs/pathtracer.cu:94-112
    std::cout << (img.WriteTo(path) ? "Export Success" : "Export failed") << std::endl;   // :114-121
}

static void write_raw(const std::string& path, const std::vector<float>& raw)
{
    if (path.empty()) return;
    const std::string tmp = path + ".tmp";
    FILE* f = fopen(tmp.c_str(), "wb");
    const bool ok = f && fwrite(raw.data(), 4, raw.size(), f) == raw.size();
    if (f) fclose(f);
    if (!ok || rename(tmp.c_str(), path.c_str()) != 0) std::cout << "Export failed (" << path << ")" << std::endl;
}

void PathTracer::Render(Camera& camera, BVH* bvh)
{
    using clk = std::chrono::system_clock;
    std::cout << "Camera : " << camera.Screen_W << " x " << camera.Screen_H << std::endl;
    const int W = (int)camera.Screen_W, H = (int)camera.Screen_H;
    Image img(W, H, 3);
    LoadFromBVH(bvh);
    std::cout << "Tree on GPU Size : " << CudaBVH.size() << std::endl;
    std::cout << "Prim on GPU Size : " << CudaPrims.size() << std::endl;
    std::cout << "Prim on CPU Size : " << bvh->primCnt() << std::endl;
    std::cout << "Upload world on GPU" << std::endl;
    PtScene* scene = nullptr;
    check(pt_scene_create(CudaBVH.data(), (int)CudaBVH.size(), CudaPrims.data(), (int)CudaPrims.size(),
                          CudaSpheres.empty() ? nullptr : CudaSpheres.data(), (int)CudaSpheres.size(), device, &scene), "pt_scene_create");
    for (int i = 0; i < pt_scene_num_lights(scene); i++) std::cout << "ADD light" << std::endl;
    std::cout << "Upload camera configuration on GPU" << std::endl << std::endl << std::endl;
    PtCamera cam;
    const vec3f f = camera.GetForward(), u = camera.GetUp(), r = camera.GetRight();
    cam.pos[0] = camera.pos.x; cam.pos[1] = camera.pos.y; cam.pos[2] = camera.pos.z;
    cam.forward[0] = f.x; cam.forward[1] = f.y; cam.forward[2] = f.z;
    cam.up[0] = u.x; cam.up[1] = u.y; cam.up[2] = u.z;
    cam.right[0] = r.x; cam.right[1] = r.y; cam.right[2] = r.z;
    cam.fovy_deg = camera.fovy; cam.aspect = camera.aspect; cam.W = W; cam.H = H;

    std::vector<float> raw((size_t)W * H * 3, 0.f), pass((size_t)W * H * 3);
    const auto t0 = clk::now();
    last_render_ms = 0.0;
    if (world > 1) {
        // one process per GPU: this rank's tiles for all passes, one gather (RCCL), frame on rank 0
        PtComm* comm = nullptr;
        check(pt_comm_create_from_file_tagged(id_file.c_str(), job_tag, rank, world, device, 120, &comm), "pt_comm_create_from_file_tagged");
        if (pt_render_split(scene, &cam, &params, comm, raw.data()) != PT_OK) {
            // every rank gets an error when any rank failed (status exchange before the gather): nobody is left in a collective
            std::cerr << "GPU error in pt_render_split (rank " << rank << ") : " << pt_last_error() << std::endl;
            pt_comm_destroy(comm); pt_scene_destroy(scene);
            exit(99);
        }
        float ms = 0.f; pt_last_render_ms(scene, &ms); last_render_ms = ms;
        pt_comm_destroy(comm);
        if (rank != 0) { pt_scene_destroy(scene); return; }
    } else if (progressive) {
        // one call per pass, summed in pass order: bit-identical to a single multi-pass call, and temp.png
        // can be rewritten after every pass as the reference does (srcs/pathtracer.cu:236-246)
        for (int i = 0; i < params.passes; i++) {
            PtParams p = params; p.passes = 1; p.first_pass = params.first_pass + i;
            check(pt_render(scene, &cam, &p, pass.data()), "pt_render");
            float ms = 0.f; pt_last_render_ms(scene, &ms); last_render_ms += ms;
            for (size_t k = 0; k < raw.size(); k++) raw[k] += pass[k];
            std::cout << "Sample " << i << " : Delta time : "
                      << std::chrono::duration_cast<std::chrono::milliseconds>(clk::now() - t0).count() << " (ms)" << std::endl;
            exportImage(img, raw.data(), "temp.png", H, W, i + 1);
            write_raw(raw_path, raw);
        }
    } else {
        check(pt_render(scene, &cam, &params, raw.data()), "pt_render");
        float ms = 0.f; pt_last_render_ms(scene, &ms); last_render_ms = ms;
    }
    std::cout << "Delta time : " << std::chrono::duration_cast<std::chrono::milliseconds>(clk::now() - t0).count() << " (ms)" << std::endl;
    exportImage(img, raw.data(), "result.png", H, W, params.passes);
    write_raw(raw_path, raw);
    pt_scene_destroy(scene);
}
