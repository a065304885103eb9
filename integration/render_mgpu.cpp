// integration/render_mgpu.cpp — the tile split from a C++ host program, one process per GPU, nothing but the C-ABI and the HIP runtime
// (no Python, no torch, no MPI).  The reference is single-device (srcs/pathtracer.cu:124-259 has no cudaSetDevice and no collective);
// this is what its maintainer would start N times on an N-GPU node:
//     for r in 0 .. N-1:   ./render_mgpu --rank $r --world N [--device $r] &
// Every rank builds and uploads its own replica of the scene (<= 110 MB: no broadcast needed), renders the 8x8-pixel tiles t with
// t % world == rank for all passes (pt_render_tiles), and ONE collective — pt_gather_frame: RCCL ncclGather + the de-interleave on
// rank 0 — assembles the frame, which rank 0 tone-maps and writes as PNG exactly as PathTracer::Render does.  The ranks meet through
// a file (pt_comm_create_from_file_tagged): rank 0 writes { world, job tag, id }, the others wait for a file carrying their job tag.
// Built by pathtrace-on-cuda_amd/Makefile (-> pathtrace-on-cuda_amd/render_mgpu); tests/test_ingest_cli.py runs it with world = 1 on the
// GPU (bit-identical to pt_render) — world > 1 needs one GPU per rank and has not run on real devices yet.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include "pt_api.h"

static void die(const char* what) { fprintf(stderr, "render_mgpu: %s: %s\n", what, pt_last_error()); exit(99); }      // the reference's convention (include/CudaUtil.cuh:28-36)
#define HIPOK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "render_mgpu: %s: %s\n", #x, hipGetErrorString(e_)); exit(99); } } while (0)

int main(int argc, char** argv)
{
    int rank = 0, world = 1, device = -1, W = 1920, H = 1080, passes = 8, spp = 256, kind = 1, latlon = 187;
    std::string out = "result.png", raw, idFile = "/dev/shm/ptamd_render_mgpu.id";
    unsigned long long jobTag = (unsigned long long)getppid();      // the ranks of one job share their launcher
    for (int i = 1; i < argc; i++) {
        const std::string a = argv[i];
        auto next = [&]() -> const char* { if (i + 1 >= argc) { fprintf(stderr, "render_mgpu: %s needs a value\n", a.c_str()); exit(2); } return argv[++i]; };
        if (a == "--rank") rank = atoi(next()); else if (a == "--world") world = atoi(next()); else if (a == "--device") device = atoi(next());
        else if (a == "--size") { if (sscanf(next(), "%dx%d", &W, &H) != 2) { fprintf(stderr, "--size WxH\n"); return 2; } }
        else if (a == "--passes") passes = atoi(next()); else if (a == "--spp") spp = atoi(next());
        else if (a == "--scene") kind = atoi(next()); else if (a == "--lat-lon") latlon = atoi(next());
        else if (a == "--out") out = next(); else if (a == "--raw") raw = next();
        else if (a == "--id-file") idFile = next(); else if (a == "--job-tag") jobTag = strtoull(next(), nullptr, 10);
        else {
            printf("usage: render_mgpu [--rank R --world N] [--device D] [--size WxH] [--passes P] [--spp S] [--scene 0|1|2] [--lat-lon L]\n"
                   "                   [--out result.png] [--raw frame.f32] [--id-file PATH] [--job-tag T]\n"
                   "one process per GPU; rank 0 writes the frame.  Defaults: configs[2] (1920x1080, 8 passes x 256 spp, bunny stand-in).\n");
            return a == "--help" ? 0 : 2;
        }
    }
    if (world < 1 || rank < 0 || rank >= world) { fprintf(stderr, "render_mgpu: need 0 <= rank < world\n"); return 2; }
    if (device < 0) device = rank;

    // scene: every rank its own replica
    const int n = pt_scene_gen(kind, latlon, nullptr, 0);
    if (n <= 0) die("pt_scene_gen");
    std::vector<PtPrimitive> prims((size_t)n);
    pt_scene_gen(kind, latlon, prims.data(), n);
    PtFlatBVH* flat = nullptr;
    if (pt_bvh_build_sah(prims.data(), n, &flat)) die("pt_bvh_build_sah");
    PtScene* scene = nullptr;
    if (pt_scene_create(pt_bvh_nodes(flat), pt_bvh_num_nodes(flat), pt_bvh_tris(flat), pt_bvh_num_tris(flat), nullptr, 0, device, &scene)) die("pt_scene_create");
    HIPOK(hipSetDevice(device));

    PtComm* comm = nullptr;
    if (pt_comm_create_from_file_tagged(idFile.c_str(), jobTag, rank, world, device, /*timeout_s*/ 120, &comm)) die("pt_comm_create_from_file_tagged");

    // the reference app's camera (srcs/renderer.cpp:28-30) and the config's sample counts
    PtCamera cam;
    const float rot[3] = {0.f, 90.f, 0.f};
    cam.pos[0] = 0.f; cam.pos[1] = 20.f; cam.pos[2] = 60.f;
    pt_camera_basis(rot, cam.forward, cam.up, cam.right);
    cam.fovy_deg = 45.f; cam.aspect = (float)W / (float)H; cam.W = W; cam.H = H;
    PtParams prm;
    pt_params_default(&prm);
    prm.passes = passes; prm.spp_per_pass = spp; prm.rank = rank; prm.world = world;

    const int64_t nt = pt_tiles_floats(&cam, &prm), wb = pt_work_bytes(&cam, &prm);
    if (nt < 0 || wb < 0) die("pt_tiles_floats / pt_work_bytes");
    float *d_tiles = nullptr, *d_gathered = nullptr, *d_frame = nullptr; void* d_work = nullptr;
    HIPOK(hipMalloc((void**)&d_tiles, (size_t)nt * 4)); HIPOK(hipMalloc(&d_work, (size_t)wb));
    if (rank == 0) { HIPOK(hipMalloc((void**)&d_gathered, (size_t)world * (size_t)nt * 4)); HIPOK(hipMalloc((void**)&d_frame, (size_t)W * H * 12)); }

    if (pt_render_tiles(scene, &cam, &prm, d_tiles, d_work, /*stream*/ nullptr)) die("pt_render_tiles");          // this rank's tiles, all passes
    if (pt_gather_frame(comm, d_tiles, &cam, &prm, d_gathered, d_frame, nullptr)) die("pt_gather_frame");         // the single collective + de-interleave on rank 0
    HIPOK(hipDeviceSynchronize());
    float ms = 0.f; pt_last_render_ms(scene, &ms);
    printf("rank %d of %d on device %d: %lld tile floats, render %.1f ms\n", rank, world, device, (long long)nt, ms);

    if (rank == 0) {      // exportImage, srcs/pathtracer.cu:94-122
        std::vector<float> frame((size_t)W * H * 3);
        HIPOK(hipMemcpy(frame.data(), d_frame, frame.size() * 4, hipMemcpyDeviceToHost));
        std::vector<uint8_t> rgb((size_t)W * H * 3);
        if (pt_tonemap_u8(frame.data(), (int64_t)W * H, passes, rgb.data())) die("pt_tonemap_u8");
        printf(pt_write_png(out.c_str(), rgb.data(), W, H, 3) == 0 ? "Export Success\n" : "Export failed\n");
        if (!raw.empty()) { if (FILE* f = fopen(raw.c_str(), "wb")) { fwrite(frame.data(), 4, frame.size(), f); fclose(f); } }
    }
    (void)hipFree(d_tiles); (void)hipFree(d_work); (void)hipFree(d_gathered); (void)hipFree(d_frame);
    pt_comm_destroy(comm); pt_scene_destroy(scene); pt_bvh_free(flat);
    return 0;
}
