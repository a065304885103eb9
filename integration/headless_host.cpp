// integration/headless_host.cpp — TEST HARNESS: the part of the reference's viewer that leads to PathTracer::Render, without GL.
//
// Linked (oracle/Makefile: `make ref` -> oracle/_ref/ptviewer, git-ignored) from
//   the reference's own, unmodified srcs/bvh.cpp (SAHBVH::GenBVHTree), srcs/camera.cpp (Camera), srcs/image.cpp (Image, stb),
//   srcs/glad.c (loader table bvh.cpp refers to; never initialised),
//   integration/pathtracer_mi355x.cpp (the binding under test) and libptamd.so.
// It does what srcs/renderer.cpp:28-30,102-153 and srcs/main.cpp's P-key handler (renderer.cpp:283-288) do: place the camera, fill
// bvh.primitives (from a file instead of assimp: the reference ships no assets), push analytic spheres into the global CudaSpheres
// exactly as renderer.cpp:126-144 does, build the tree with the reference's own GenBVHTree, and call PathTracer::Render.
//
//   ptviewer <prims.bin> <W> <H> [spheres.bin]
//     prims.bin   N x reference `Primitive` (include/bvh.h:8-13), raw
//     spheres.bin S x { x y z rad | Material (12 floats: emittance albedo specular opacity roughness metallic) }
//   Output: temp.png / result.png in the CWD (and $PT_RAW_OUT), as PathTracer::Render writes them.
#include "pathtracer.cuh"
#include "CudaPrimitive.cuh"
#include "camera.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

static std::vector<unsigned char> slurp(const char* path)
{
    FILE* f = fopen(path, "rb");
    if (!f) { fprintf(stderr, "ptviewer: cannot open %s\n", path); exit(2); }
    fseek(f, 0, SEEK_END); const long n = ftell(f); fseek(f, 0, SEEK_SET);
    std::vector<unsigned char> b((size_t)n);
    if (n && fread(b.data(), 1, (size_t)n, f) != (size_t)n) { fprintf(stderr, "ptviewer: short read %s\n", path); exit(2); }
    fclose(f);
    return b;
}

int main(int argc, char** argv)
{
    if (argc < 4) { fprintf(stderr, "usage: ptviewer prims.bin W H [spheres.bin]\n"); return 1; }
    const int W = atoi(argv[2]), H = atoi(argv[3]);

    // Renderer::Renderer, srcs/renderer.cpp:28-30
    Camera camera(glm::vec3(0.f, 20.f, 60.f));
    camera.Screen_W = (unsigned)W; camera.Screen_H = (unsigned)H;
    camera.aspect = (float)W / (float)H;

    SAHBVH bvh;                                       // GL members untouched: Init() is never called
    const auto in = slurp(argv[1]);
    if (in.size() % sizeof(Primitive)) { fprintf(stderr, "ptviewer: prims.bin is not a whole number of Primitives\n"); return 2; }
    bvh.primitives.resize(in.size() / sizeof(Primitive));
    memcpy((void*)bvh.primitives.data(), in.data(), in.size());

    if (argc > 4) {                                   // srcs/renderer.cpp:126-144
        const auto sb = slurp(argv[4]);
        const float* f = (const float*)sb.data();
        for (size_t i = 0; i < sb.size() / 64; i++, f += 16) {
            Material mat;
            mat.emittance = Color(f[4], f[5], f[6]); mat.albedo = Color(f[7], f[8], f[9]); mat.specular = Color(f[10], f[11], f[12]);
            mat.opacity = f[13]; mat.roughness = f[14]; mat.metallic = f[15];
            Sphere s(f[0], f[1], f[2], f[3], mat);
            CudaSpheres.push_back(s);
        }
    }
    bvh.GenBVHTree(new Cluster());                    // srcs/renderer.cpp:153

    PathTracer tracer;                                // srcs/renderer.cpp:286-287
    tracer.Render(camera, &bvh);
    return 0;
}
