// TEST INFRASTRUCTURE ONLY (oracle/): driver for a PARTIAL build of the real reference.
//
// What it is: a small command-line program, written for this repo, that is linked
// against the reference's own unmodified sources where they lie under /root/reference
//   srcs/bvh.cpp           (SAHBVH::GenBVHTree / SplitNode / ConvertToBVH, IntoBVHNode)
//   srcs/CudaPrimitive.cu  (LoadFromBVH flatten; compiled as host C++)
//   srcs/glad.c            (GL loader table bvh.cpp refers to; never initialised)
//   srcs/camera.cpp        (Camera::SetRotation / GetForward / GetUp / GetRight)
//   include/image.h        (ConverToUint8) and srcs/image.cpp (Image::WriteTo / Image(path), with the reference's own vendored
//                          include/stb_image_write.h / stb_image.h)
// and the headers include/CudaPrimitive.cuh, CudaVector.cuh, CudaRay.cuh
// (Triangle::Copy/hit, Sphere::hit, HitResult::SetNormal, vec3, reflect, refract).
// <cuda_runtime.h> is the REAL header bundled with this image's triton wheel; no CUDA
// or cuRAND header is faked.  Everything that includes <curand_kernel.h>
// (include/CudaUtil.cuh, include/Bxdf.cuh, srcs/pathtracer.cu) is NOT buildable in this
// image and is therefore not part of this program: RayCast, the BxDFs and the
// integrator are covered by the restatement in oracle/pt_oracle.cpp only.
//
// The one thing the recipe adds is `lp64_min` below: srcs/bvh.cpp:204 (deprecated
// K-means path, out of scope) calls min(unsigned long long, size_t), which only
// resolves where size_t == unsigned long long (MSVC x64).  It is never executed.
//
// Output goes to oracle/_ref/ only (git-ignored).  Used by oracle/gen_golden.py to
// produce tests/golden/*.bin and by tests/ to cross-check the restatement live.
//
// File formats are raw little-endian arrays, documented next to each command.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <string>

#include "CudaPrimitive.cuh"   // reference header (pulls bvh.h, mesh.h, glm)
#include "camera.h"            // reference header: class Camera (srcs/camera.cpp)
#include "image.h"             // reference header: ConverToUint8

// ---------------------------------------------------------------------------------
static std::vector<unsigned char> slurp(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "ptref: cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "ptref: short read %s\n", path); exit(2); }
    fclose(f);
    return b;
}
static void spit(const char* path, const void* p, size_t n)
{
    FILE* f = fopen(path, "wb");
    if (!f) { fprintf(stderr, "ptref: cannot write %s\n", path); exit(2); }
    if (n && fwrite(p, 1, n, f) != n) { fprintf(stderr, "ptref: short write %s\n", path); exit(2); }
    fclose(f);
}
static void put3(std::vector<float>& o, const vec3& v) { o.push_back(v[0]); o.push_back(v[1]); o.push_back(v[2]); }
static void putm(std::vector<float>& o, const Material& m)
{
    put3(o, m.emittance); put3(o, m.albedo); put3(o, m.specular);
    o.push_back(m.opacity); o.push_back(m.roughness); o.push_back(m.metallic);
}

// Packed triangle record written by `bvh` (float32 x TRI_FLOATS), independent of the
// reference's in-memory layout (which has a vptr):
//   V0 V1 V2 | T0 T1 T2 | B0 B1 B2 | N0 N1 N2 | normal | E1 | E2 | u0 v0 u1 v1 u2 v2 |
//   mat0(12) mat1(12) mat2(12) | area          => 9*3+3*3... = 90 floats
static const int TRI_FLOATS = 12 * 3 + 3 * 3 + 6 + 36 + 1;   // 88
static void put_tri(std::vector<float>& o, const Triangle& t)
{
    put3(o, t.V0); put3(o, t.V1); put3(o, t.V2);
    put3(o, t.T0); put3(o, t.T1); put3(o, t.T2);
    put3(o, t.B0); put3(o, t.B1); put3(o, t.B2);
    put3(o, t.N0); put3(o, t.N1); put3(o, t.N2);
    put3(o, t.normal); put3(o, t.E1); put3(o, t.E2);
    o.push_back(t.u0); o.push_back(t.v0); o.push_back(t.u1); o.push_back(t.v1); o.push_back(t.u2); o.push_back(t.v2);
    putm(o, t.mat0); putm(o, t.mat1); putm(o, t.mat2);
    o.push_back(t.area);
}

// HitResult record (float32 x HIT_FLOATS): hit(0/1) t u v frontface | p | normal | tangent | bitangent | mat(12)
static const int HIT_FLOATS = 5 + 12 + 12;
static void put_hit(std::vector<float>& o, bool hit, const HitResult& h)
{
    if (!hit) { for (int i = 0; i < HIT_FLOATS; i++) o.push_back(0.f); return; }
    o.push_back(1.f); o.push_back(h.t); o.push_back(h.u); o.push_back(h.v); o.push_back(h.bFrontFace ? 1.f : 0.f);
    put3(o, h.p); put3(o, h.normal); put3(o, h.tangent); put3(o, h.bitangent);
    putm(o, h.mat);
}

// ---------------------------------------------------------------------------------
// bvh <prims.bin> <nodes.bin> <tris.bin>
//   prims.bin : N x reference `Primitive` (3 x `Vertex`, include/mesh.h:21-37), raw.
//   nodes.bin : CudaBVH as raw CudaBVHNode[ ] (40 B each, include/CudaPrimitive.cuh:237-247)
//   tris.bin  : CudaPrims after Triangle::Copy (as PathTracer::Render does,
//               srcs/pathtracer.cu:166), TRI_FLOATS float32 each.
static int cmd_bvh(int argc, char** argv)
{
    if (argc != 5) return 1;
    auto in = slurp(argv[2]);
    if (in.size() % sizeof(Primitive)) { fprintf(stderr, "ptref: prims.bin not a multiple of %zu\n", sizeof(Primitive)); return 2; }
    size_t n = in.size() / sizeof(Primitive);
    SAHBVH* bvh = new SAHBVH();               // GL members stay untouched (Init() is never called)
    bvh->primitives.resize(n);
    memcpy((void*)bvh->primitives.data(), in.data(), in.size());
    bvh->GenBVHTree(new Cluster());          // srcs/renderer.cpp:153
    LoadFromBVH(bvh);                        // srcs/pathtracer.cu:133
    spit(argv[3], CudaBVH.data(), CudaBVH.size() * sizeof(CudaBVHNode));
    std::vector<float> o; o.reserve(CudaPrims.size() * TRI_FLOATS);
    for (size_t i = 0; i < CudaPrims.size(); i++) {
        Triangle t; t.Copy(CudaPrims[i]);    // srcs/pathtracer.cu:166
        put_tri(o, t);
    }
    spit(argv[4], o.data(), o.size() * sizeof(float));
    fprintf(stderr, "ptref bvh: %zu prims -> %zu nodes, %zu tris\n", n, CudaBVH.size(), CudaPrims.size());
    return 0;
}

// trihit <tris9.bin> <rays.bin> <out.bin>
//   tris9.bin : M x { V0 V1 V2 N0 N1 N2 T0 T1 T2 B0 B1 B2 (36 f) mat0 (12 f) } = 48 float32
//   rays.bin  : R x { triIndex(as float) org(3) dir(3) tmin tmax normalise(0/1) } = 10 float32
//               normalise=1 builds the ray with Ray(org,dir) (normalises, CudaRay.cuh:12),
//               0 assigns .dir raw (as GetColor_iter does for bounce rays, CudaUtil.cuh:350).
//   out.bin   : R x HIT_FLOATS
static int cmd_trihit(int argc, char** argv)
{
    if (argc != 5) return 1;
    auto tb = slurp(argv[2]); auto rb = slurp(argv[3]);
    const float* tf = (const float*)tb.data(); size_t M = tb.size() / (48 * 4);
    const float* rf = (const float*)rb.data(); size_t R = rb.size() / (10 * 4);
    std::vector<Triangle> tris(M);
    for (size_t i = 0; i < M; i++) {
        const float* f = tf + i * 48;
        auto v = [&](int k) { return vec3(f[3 * k], f[3 * k + 1], f[3 * k + 2]); };
        Material m; m.emittance = v(12); m.albedo = v(13); m.specular = v(14);
        m.opacity = f[45]; m.roughness = f[46]; m.metallic = f[47];
        tris[i].Copy(v(0), v(1), v(2), v(6), v(7), v(8), v(9), v(10), v(11), v(3), v(4), v(5),
                     m, m, m, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
    }
    std::vector<float> o; o.reserve(R * HIT_FLOATS);
    for (size_t r = 0; r < R; r++) {
        const float* f = rf + r * 10;
        size_t ti = (size_t)f[0];
        Ray ray;
        if (f[9] != 0.f) ray = Ray(vec3(f[1], f[2], f[3]), vec3(f[4], f[5], f[6]));
        else { ray.org = vec3(f[1], f[2], f[3]); ray.dir = vec3(f[4], f[5], f[6]); }
        HitResult h; memset((void*)&h, 0, sizeof(h));
        bool hit = tris[ti].hit(ray, f[7], f[8], h);
        put_hit(o, hit, h);
    }
    spit(argv[4], o.data(), o.size() * sizeof(float));
    return 0;
}

// sphit <spheres.bin> <rays.bin> <out.bin>
//   spheres.bin : S x { center(3) rad mat(12) } = 16 float32 ; rays as in trihit (index = sphere)
static int cmd_sphit(int argc, char** argv)
{
    if (argc != 5) return 1;
    auto sb = slurp(argv[2]); auto rb = slurp(argv[3]);
    const float* sf = (const float*)sb.data(); size_t S = sb.size() / (16 * 4);
    const float* rf = (const float*)rb.data(); size_t R = rb.size() / (10 * 4);
    std::vector<Sphere> sph;
    for (size_t i = 0; i < S; i++) {
        const float* f = sf + i * 16;
        Material m; m.emittance = vec3(f[4], f[5], f[6]); m.albedo = vec3(f[7], f[8], f[9]); m.specular = vec3(f[10], f[11], f[12]);
        m.opacity = f[13]; m.roughness = f[14]; m.metallic = f[15];
        sph.push_back(Sphere(f[0], f[1], f[2], f[3], m));
    }
    std::vector<float> o; o.reserve(R * HIT_FLOATS);
    for (size_t r = 0; r < R; r++) {
        const float* f = rf + r * 10;
        size_t si = (size_t)f[0];
        Ray ray;
        if (f[9] != 0.f) ray = Ray(vec3(f[1], f[2], f[3]), vec3(f[4], f[5], f[6]));
        else { ray.org = vec3(f[1], f[2], f[3]); ray.dir = vec3(f[4], f[5], f[6]); }
        HitResult h; memset((void*)&h, 0, sizeof(h));
        bool hit = sph[si].hit(ray, f[7], f[8], h);
        put_hit(o, hit, h);
    }
    spit(argv[4], o.data(), o.size() * sizeof(float));
    return 0;
}

// vecmath <in.bin> <out.bin>
//   in.bin  : K x { a(3) b(3) s } = 7 float32
//   out.bin : K x { Normalize(a)(3) reflect(a,b)(3) refract(a,b,s)(3) cross(a,b)(3) dot(a,b) a.length()
//                   (a/=s)(3) MaxFrom(a) saturate(a)(3) } = 21 float32   (include/CudaVector.cuh)
static int cmd_vecmath(int argc, char** argv)
{
    if (argc != 4) return 1;
    auto ib = slurp(argv[2]);
    const float* f0 = (const float*)ib.data(); size_t K = ib.size() / (7 * 4);
    std::vector<float> o; o.reserve(K * 21);
    for (size_t k = 0; k < K; k++) {
        const float* f = f0 + k * 7;
        vec3 a(f[0], f[1], f[2]), b(f[3], f[4], f[5]); float s = f[6];
        put3(o, Normalize(a)); put3(o, reflect(a, b)); put3(o, refract(a, b, s)); put3(o, cross(a, b));
        o.push_back(dot(a, b)); o.push_back(a.length());
        vec3 c = a; c /= s; put3(o, c);
        o.push_back(MaxFrom(a)); put3(o, saturate(a));
    }
    spit(argv[3], o.data(), o.size() * sizeof(float));
    return 0;
}

// sizes : prints the reference's struct sizes (layout facts the C-ABI mirrors).
static int cmd_sizes()
{
    printf("{\"Vertex\": %zu, \"Primitive\": %zu, \"MaterialOnCPU\": %zu, \"CudaBVHNode\": %zu, "
           "\"Triangle\": %zu, \"Sphere\": %zu, \"Material\": %zu, \"HitResult\": %zu, \"vec3\": %zu, \"Ray\": %zu}\n",
           sizeof(Vertex), sizeof(Primitive), sizeof(MaterialOnCPU), sizeof(CudaBVHNode),
           sizeof(Triangle), sizeof(Sphere), sizeof(Material), sizeof(HitResult), sizeof(vec3), sizeof(Ray));
    return 0;
}

// camera rot.bin out.bin : rot = N x 3 floats (roll, pitch, yaw in degrees); out = N x 9 floats
//   forward | up | right of a default-constructed reference Camera after SetRotation(rot)   (srcs/camera.cpp:22-66)
static int cmd_camera(int argc, char** argv)
{
    if (argc != 4) return 1;
    std::vector<unsigned char> in = slurp(argv[2]);
    const size_t n = in.size() / 12;
    const float* r = (const float*)in.data();
    std::vector<float> out;
    for (size_t i = 0; i < n; i++) {
        Camera cam;
        cam.SetRotation(glm::vec3(r[3 * i], r[3 * i + 1], r[3 * i + 2]));
        const glm::vec3 f = cam.GetForward(), u = cam.GetUp(), rt = cam.GetRight();
        const float v[9] = {f.x, f.y, f.z, u.x, u.y, u.z, rt.x, rt.y, rt.z};
        out.insert(out.end(), v, v + 9);
    }
    spit(argv[3], out.data(), out.size() * 4);
    return 0;
}

// u8 in.bin out.bin : N floats -> N bytes through ConverToUint8 (include/image.h:5-8)
static int cmd_u8(int argc, char** argv)
{
    if (argc != 4) return 1;
    std::vector<unsigned char> in = slurp(argv[2]);
    const size_t n = in.size() / 4;
    const float* v = (const float*)in.data();
    std::vector<unsigned char> out(n);
    for (size_t i = 0; i < n; i++) out[i] = ConverToUint8(v[i]);
    spit(argv[3], out.data(), out.size());
    return 0;
}

// pngwrite in.bin W H C out.png : W*H*C bytes -> PNG through the reference's Image(W,H,C) + Image::WriteTo (srcs/image.cpp:17-25)
static int cmd_pngwrite(int argc, char** argv)
{
    if (argc != 7) return 1;
    std::vector<unsigned char> in = slurp(argv[2]);
    const int W = atoi(argv[3]), H = atoi(argv[4]), C = atoi(argv[5]);
    if (in.size() != (size_t)W * H * C) { fprintf(stderr, "ptref: pngwrite size mismatch\n"); return 2; }
    Image img(W, H, C);
    memcpy(img.GetData(), in.data(), in.size());
    return img.WriteTo(argv[6]) ? 0 : 3;
}

// pngread in.png out.bin : decode through the reference's Image(path) (stbi_load, srcs/image.cpp:12-15); out = W H C (int32) + pixels
static int cmd_pngread(int argc, char** argv)
{
    if (argc != 4) return 1;
    Image img(argv[2]);
    if (!img.GetData()) { fprintf(stderr, "ptref: cannot decode %s\n", argv[2]); return 3; }
    const int hdr[3] = {img.GetWidth(), img.GetHeight(), img.GetNrChannels()};
    std::vector<unsigned char> out((const unsigned char*)hdr, (const unsigned char*)hdr + 12);
    out.insert(out.end(), img.GetData(), img.GetData() + (size_t)hdr[0] * hdr[1] * hdr[2]);
    spit(argv[3], out.data(), out.size());
    return 0;
}

int main(int argc, char** argv)
{
    int rc = 1;
    if (argc >= 2) {
        std::string c = argv[1];
        if (c == "bvh") rc = cmd_bvh(argc, argv);
        else if (c == "trihit") rc = cmd_trihit(argc, argv);
        else if (c == "sphit") rc = cmd_sphit(argc, argv);
        else if (c == "vecmath") rc = cmd_vecmath(argc, argv);
        else if (c == "sizes") rc = cmd_sizes();
        else if (c == "camera") rc = cmd_camera(argc, argv);
        else if (c == "u8") rc = cmd_u8(argc, argv);
        else if (c == "pngwrite") rc = cmd_pngwrite(argc, argv);
        else if (c == "pngread") rc = cmd_pngread(argc, argv);
    }
    if (rc == 1) fprintf(stderr, "usage: ptref bvh|trihit|sphit|vecmath|sizes|camera|u8|pngwrite|pngread ...\n");
    return rc;
}
