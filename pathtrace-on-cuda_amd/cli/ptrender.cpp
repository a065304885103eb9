// ptrender — headless command line for the path tracer (the reference has no CLI: its window size
// is a compile-time constant and rendering starts on key P, srcs/main.cpp:15-16, srcs/renderer.cpp:283-293).
// Uses only the reference-shaped host surface (host/ref_surface.h).
#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>
#include "../host/ref_surface.h"

static void usage()
{
    std::cout <<
        "usage: ptrender [--scene cornell|standin|standin4] [--obj FILE --obj-scale S --obj-translate X,Y,Z]\n"
        "                [--glass-sphere] [--width W] [--height H] [--passes N] [--spp N] [--depth N]\n"
        "                [--lat-lon N] [--device D] [--no-progressive] [--raw FILE]\n"
        "                [--world N --rank R --id-file PATH [--job-tag T]]   (one process per GPU; rank 0 writes the frame;\n"
        "                 T = a number the ranks of this job share and other jobs do not, default: the parent process id)\n"
        "Writes temp.png (per pass) and result.png in the current directory, like PathTracer::Render.\n"
        "--raw FILE: also writes the float accumulation buffer (W*H*3 float32) there after every pass (viewer hook).\n"
        "Defaults: scene cornell, 1920x1080, 8 passes x 64 spp, depth 8.\n";
}

int main(int argc, char** argv)
{
    std::string scene = "cornell", obj, rawPath;
    float objScale = 1.f; float objT[3] = {0, 0, 0};
    int W = 1920, H = 1080, passes = 8, spp = 64, depth = 8, latlon = 187, device = 0;
    bool glass = false, progressive = true;
    int rank = 0, world = 1; std::string idFile; unsigned long long jobTag = (unsigned long long)getppid();
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
        if (a == "--scene") scene = next();
        else if (a == "--obj") obj = next();
        else if (a == "--obj-scale") objScale = (float)atof(next());
        else if (a == "--obj-translate") { if (sscanf(next(), "%f,%f,%f", &objT[0], &objT[1], &objT[2]) != 3) { usage(); return 2; } }
        else if (a == "--glass-sphere") glass = true;
        else if (a == "--width") W = atoi(next());
        else if (a == "--height") H = atoi(next());
        else if (a == "--passes") passes = atoi(next());
        else if (a == "--spp") spp = atoi(next());
        else if (a == "--depth") depth = atoi(next());
        else if (a == "--lat-lon") latlon = atoi(next());
        else if (a == "--device") device = atoi(next());
        else if (a == "--no-progressive") progressive = false;
        else if (a == "--raw") rawPath = next();
        else if (a == "--world") world = atoi(next());
        else if (a == "--rank") rank = atoi(next());
        else if (a == "--id-file") idFile = next();
        else if (a == "--job-tag") jobTag = strtoull(next(), nullptr, 10);
        else if (a == "--help" || a == "-h") { usage(); return 0; }
        else { std::cerr << "unknown option " << a << "\n"; usage(); return 2; }
    }
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && idFile.empty())) { std::cerr << "--world N needs 0 <= --rank < N and --id-file PATH\n"; return 2; }
    const int kind = scene == "cornell" ? 0 : scene == "standin" ? 1 : scene == "standin4" ? 2 : -1;
    if (kind < 0) { std::cerr << "unknown scene " << scene << "\n"; return 2; }

    // the reference app's setup: camera at (0,20,60), rotation (0,90,0), aspect W/H (srcs/renderer.cpp:28-30,47-53)
    Camera camera(vec3f(0.f, 20.f, 60.f));
    camera.Screen_W = (unsigned)W; camera.Screen_H = (unsigned)H; camera.aspect = (float)W / (float)H;

    SAHBVH bvh;
    const int n = pt_scene_gen(kind, latlon, nullptr, 0);
    if (n < 0) { std::cerr << pt_last_error() << "\n"; return 2; }
    bvh.primitives.resize((size_t)n);
    pt_scene_gen(kind, latlon, bvh.primitives.data(), n);
    if (!obj.empty() && !bvh.AddOBJ(obj, objScale, vec3f(objT[0], objT[1], objT[2]))) return 2;
    if (glass) {   // BASELINE.json configs[3]: glass sphere r=6 at (10,6,8), opacity 0, roughness 0
        Material m; memset(&m, 0, sizeof(m));
        m.albedo[0] = m.albedo[1] = m.albedo[2] = 1.f; m.specular[0] = m.specular[1] = m.specular[2] = 0.04f;
        CudaSpheres.push_back(Sphere(10.f, 6.f, 8.f, 6.f, m));
    }
    std::cout << "Build BVH" << std::endl;
    bvh.GenBVHTree();
    std::cout << "Pre process done : Primitive CNT = " << bvh.primCnt() << std::endl;

    PathTracer tracer;
    tracer.params.passes = passes; tracer.params.spp_per_pass = spp; tracer.params.max_bounce = depth;
    tracer.device = device; tracer.progressive = progressive; tracer.raw_path = rawPath;
    tracer.rank = rank; tracer.world = world; tracer.id_file = idFile; tracer.job_tag = jobTag;
    tracer.Render(camera, &bvh);
    const double samples = (double)W * H * passes * spp;
    std::cout << "{\"msamples_per_s_kernel\": " << samples / (tracer.last_render_ms * 1e-3) / 1e6 << ", \"kernel_ms\": " << tracer.last_render_ms << "}" << std::endl;
    return 0;
}
