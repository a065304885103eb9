// ref_surface.h — the reference's C++ host surface, re-declared on top of the C-ABI.
//
// For users of WaterPlease/PathTrace-on-CUDA who have no viewer to drop the library behind:
// the same class names, public members and call sequence as include/camera.h, include/bvh.h
// (SAH path), include/CudaPrimitive.cuh (host-visible parts), include/image.h and
// include/pathtracer.cuh, with glm::vec3 replaced by a plain 3-float struct and all GL members
// removed.  Everything forwards to include/pt_api.h; nothing here computes radiance.
#pragma once
#include <string>
#include <vector>
#include "../../include/pt_api.h"

struct vec3f { float x, y, z; vec3f() : x(0), y(0), z(0) {} vec3f(float a, float b, float c) : x(a), y(b), z(c) {} };

// include/camera.h:9-40
class Camera {
public:
    Camera();                               // fovy 45, aspect 16/9, rotation (0,90,0), pos 0 (srcs/camera.cpp:7-15)
    explicit Camera(vec3f _pos);
    vec3f pos;
    unsigned int Screen_W = 1920, Screen_H = 1080;
    float fovy, aspect, near, far;
    const vec3f GetForward() { return forward; }
    const vec3f GetUp() { return up; }
    const vec3f GetRight() { return right; }
    const Camera& AddRotation(vec3f deltaRotation);
    const Camera& SetRotation(vec3f _rotation);   // (roll, pitch, yaw) in degrees (srcs/camera.cpp:42-66)
private:
    vec3f rotation, forward, up, right;
};

// include/mesh.h:11-37, include/bvh.h:8-13 — byte-identical to the C-ABI mirrors
typedef PtMaterialOnCPU MaterialOnCPU;
typedef PtVertex Vertex;
typedef PtPrimitive Primitive;

// include/CudaPrimitive.cuh:15-23, 237-247, 249-323 (data members)
typedef PtMaterial Material;
typedef PtBVHNode CudaBVHNode;
typedef PtTriangle Triangle;
struct Sphere : PtSphere {
    Sphere() { memset_zero(); }
    Sphere(float x, float y, float z, float r, Material m) { center[0] = x; center[1] = y; center[2] = z; rad = r; mat = m; }
private:
    void memset_zero();
};

// include/bvh.h:39-77,123-131 (SAH path; GL members dropped)
class BVH {
public:
    std::vector<Primitive> primitives;
    virtual ~BVH();
    virtual void GenBVHTree();              // build + flatten (the reference keeps a pointer tree in rootBVH; here the flat arrays)
    void AddPrimitives(const Primitive* p, size_t n) { primitives.insert(primitives.end(), p, p + n); }
    bool AddOBJ(const std::string& path, float scale, vec3f translation);   // Model + BVH::AddModel (srcs/bvh.cpp:153-189)
    unsigned int primCnt() const { return (unsigned int)primitives.size(); }
    PtFlatBVH* flat = nullptr;
};
class SAHBVH : public BVH {};

// include/CudaPrimitive.cuh:325-337
extern std::vector<CudaBVHNode> CudaBVH;
extern std::vector<Triangle> CudaPrims;
extern std::vector<Sphere> CudaSpheres;
void LoadFromBVH(BVH* bvh);

// include/image.h
typedef unsigned char Pixel;
inline unsigned char ConverToUint8(float value) { return (unsigned char)(value * 255.99f); }
class Image {
public:
    Image(int W, int H, int C);
    ~Image();
    bool WriteTo(const char* path);
    unsigned char* GetData() noexcept { return data; }
    int GetWidth() noexcept { return width; }
    int GetHeight() noexcept { return height; }
    int GetNrChannels() noexcept { return nrChannels; }
private:
    unsigned char* data = nullptr;
    int width, height, nrChannels;
};

// include/pathtracer.cuh + the compile-time tunables of include/CudaUtil.cuh:15-19 as members
class PathTracer {
public:
    PtParams params;                        // defaults = the reference's #defines
    int device = 0;
    bool progressive = true;                // rewrite temp.png after every pass (srcs/pathtracer.cu:245)
    PathTracer() { pt_params_default(&params); }
    void Render(Camera& camera, BVH* bvh);  // writes temp.png / result.png in the CWD; GPU error -> message + exit(99)
    double last_render_ms = 0.0;            // kernel time of the last Render (sum over passes)
    // Viewer hook (new; the reference's viewer only sees temp.png, srcs/renderer.cpp:283-293): when non-empty, the float accumulation
    // buffer (W*H*3 float32, row-major, sum of per-pass means so far) is written to this path after every pass and at the end,
    // atomically (temporary name + rename) — point it at /dev/shm and a viewer can map the frame while the render goes on.
    std::string raw_path;
    // Tile split over several processes, one per GPU (new; the reference is single-device): this process renders tiles
    // t % world == rank; the ranks meet through `id_file` (pt_comm_create_from_file_tagged: `job_tag` is any value the ranks of one
    // job share and other jobs do not) and rank 0 assembles and exports the frame.  If any rank fails, every rank exits 99.
    int rank = 0, world = 1;
    std::string id_file;
    unsigned long long job_tag = 0;
};
