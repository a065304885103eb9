// pt_comm.hip — the multi-GPU exchange step of the C-ABI (include/pt_api.h: pt_comm_*, pt_gather_tiles, pt_gather_frame).
//
// The reference is single-device (srcs/pathtracer.cu:124-259 never calls cudaSetDevice and has no collective).  Here one
// process drives one GPU, every rank renders its interleaved 8x8 tiles for all passes (pt_render_tiles) and the ONLY
// data-path collective is one gather of the finished tile buffers to rank 0 — RCCL's ncclGather (rccl.h:745) over xGMI:
// 3.1 MB per rank at 1080p, each of root's seven links carries exactly one peer's buffer — followed by pt_untile on
// rank 0.  This file gives the C++ host surface that step without Python or torch.distributed.
//
// RCCL is loaded with dlopen on first use by a communicator of world > 1, so a single-GPU user of libptamd.so never
// loads or initialises it.  Bootstrap: rank 0 makes a 128-byte id (pt_comm_unique_id) and hands it to the other
// processes by whatever channel the host application has; pt_comm_create_from_file does it through a file for
// processes of one node (rank 0 writes, the others wait for it; a job tag in the file keeps a reader from picking up the id of
// an earlier job, and rank 0 removes the file once every rank has joined).
//
// Failure behaviour of the multi-rank path (pt_render_split): every rank takes part in a 4-byte status all-reduce before the
// gather, whether its own render worked or not, so a rank-local error makes ALL ranks return an error instead of leaving the
// peers blocked in the collective.  HARDWARE STATUS: world > 1 has not run on real devices yet (no multi-GPU node was available
// to any round so far); what is tested is world = 1 and the argument / rendezvous-file logic.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>
#include <sys/stat.h>
#include <unistd.h>
#include <cstdio>
#include <cstring>
#include <string>

#include "../../include/pt_api.h"

void pt_set_error(const char* fmt, ...);   // pt_host.cpp

static_assert(PT_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "PT_COMM_ID_BYTES must match ncclUniqueId");

namespace {

struct Rccl {
    void* so = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Gather)(const void*, void*, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string why;      // dlopen / dlsym failure text, captured once
};

Rccl g_rccl;
const char* rccl_why() { return g_rccl.why.empty() ? "no further detail" : g_rccl.why.c_str(); }

Rccl* rccl()
{
    Rccl& r = g_rccl;
    static bool tried = false;
    if (!tried) {
        tried = true;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            r.so = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (r.so) break;
            const char* e = dlerror();      // read once: a second dlerror() call returns NULL
            if (e) { if (!r.why.empty()) r.why += "; "; r.why += e; }
        }
        if (r.so) {
            r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(r.so, "ncclGetUniqueId");
            r.CommInitRank = (decltype(r.CommInitRank))dlsym(r.so, "ncclCommInitRank");
            r.CommDestroy = (decltype(r.CommDestroy))dlsym(r.so, "ncclCommDestroy");
            r.Gather = (decltype(r.Gather))dlsym(r.so, "ncclGather");
            r.AllReduce = (decltype(r.AllReduce))dlsym(r.so, "ncclAllReduce");
            r.CommAbort = (decltype(r.CommAbort))dlsym(r.so, "ncclCommAbort");
            r.GetErrorString = (decltype(r.GetErrorString))dlsym(r.so, "ncclGetErrorString");
            if (!r.GetUniqueId || !r.CommInitRank || !r.CommDestroy || !r.Gather || !r.AllReduce || !r.CommAbort || !r.GetErrorString) {
                dlclose(r.so); r.so = nullptr; r.why = "librccl.so lacks a symbol this library needs";
            }
        }
    }
    return r.so ? &r : nullptr;
}

}  // namespace

struct PtComm {
    int rank = 0, world = 1, device = 0;
    ncclComm_t comm = nullptr;
    int32_t* d_status = nullptr;      // world > 1: one device word for the status all-reduce of pt_render_split
};

extern "C" int ptk_scene_device(const PtScene* s);      // pt_api.hip

#define NCCLCHK(expr)                                                                                   \
    do {                                                                                                \
        ncclResult_t r_ = (expr);                                                                       \
        if (r_ != ncclSuccess) { pt_set_error("RCCL error at %s:%d '%s': %s", __FILE__, __LINE__, #expr, R->GetErrorString(r_)); return PT_ERR_DEVICE; } \
    } while (0)
#define HIPCHK(expr)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (expr);                                                                         \
        if (e_ != hipSuccess) { pt_set_error("HIP error %d at %s:%d '%s': %s", (int)e_, __FILE__, __LINE__, #expr, hipGetErrorString(e_)); return PT_ERR_DEVICE; } \
    } while (0)

extern "C" {

int pt_comm_unique_id(uint8_t id[PT_COMM_ID_BYTES])
{
    if (!id) { pt_set_error("pt_comm_unique_id: NULL"); return PT_ERR_INVALID; }
    Rccl* R = rccl();
    if (!R) { pt_set_error("pt_comm_unique_id: librccl.so could not be loaded (%s)", rccl_why()); return PT_ERR_UNSUPPORTED; }
    ncclUniqueId u;
    NCCLCHK(R->GetUniqueId(&u));
    memcpy(id, u.internal, PT_COMM_ID_BYTES);
    return PT_OK;
}

int pt_comm_create(const uint8_t id[PT_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, PtComm** out)
{
    if (!out) { pt_set_error("pt_comm_create: out is NULL"); return PT_ERR_INVALID; }
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world || (world > 1 && !id)) { pt_set_error("pt_comm_create: bad rank %d / world %d", rank, world); return PT_ERR_INVALID; }
    PtComm* c = new PtComm();
    c->rank = rank; c->world = world; c->device = device;
    if (world > 1) {
        Rccl* R = rccl();
        if (!R) { delete c; pt_set_error("pt_comm_create: librccl.so could not be loaded (%s)", rccl_why()); return PT_ERR_UNSUPPORTED; }
        if (hipSetDevice(device) != hipSuccess) { delete c; pt_set_error("pt_comm_create: hipSetDevice(%d) failed", device); return PT_ERR_DEVICE; }
        ncclUniqueId u;
        memcpy(u.internal, id, PT_COMM_ID_BYTES);
        ncclResult_t r = R->CommInitRank(&c->comm, world, u, rank);
        if (r != ncclSuccess) { pt_set_error("ncclCommInitRank(rank %d of %d): %s", rank, world, R->GetErrorString(r)); delete c; return PT_ERR_DEVICE; }
        if (hipMalloc((void**)&c->d_status, 4) != hipSuccess) { (void)R->CommDestroy(c->comm); delete c; pt_set_error("pt_comm_create: out of device memory"); return PT_ERR_DEVICE; }
    }
    *out = c;
    return PT_OK;
}

// Rendezvous through a file, for the processes of one node.  File = { magic, world, job_tag, id[128] }.
//   rank 0   removes whatever is at `path` (a crashed earlier job may have left a file), makes the id, writes the file under a
//            temporary name and renames it (a reader never sees half of it), joins, and removes the file once
//            ncclCommInitRank has returned — every rank has read it by then.
//   rank > 0 polls for the file (up to timeout_s seconds) and takes the id only from a file whose magic, world AND job_tag are
//            its own; anything else — e.g. the file of an earlier job that rank 0 has not removed yet — is ignored and polled again.
// job_tag: any value all ranks of ONE job share and other jobs do not (the launcher's pid, a timestamp, a scheduler job id).
// pt_comm_create_from_file is the same with job_tag 0.
struct IdFile { uint32_t magic, world; uint64_t tag; uint8_t id[PT_COMM_ID_BYTES]; };
static const uint32_t kIdMagic = 0x50544944u;      // "PTID"

int pt_comm_create_from_file_tagged(const char* path, uint64_t job_tag, int32_t rank, int32_t world, int32_t device, int32_t timeout_s, PtComm** out)
{
    if (!path || !out) { pt_set_error("pt_comm_create_from_file: NULL argument"); return PT_ERR_INVALID; }
    *out = nullptr;
    if (world < 1 || rank < 0 || rank >= world) { pt_set_error("pt_comm_create_from_file: bad rank %d / world %d", rank, world); return PT_ERR_INVALID; }
    IdFile rec;
    memset(&rec, 0, sizeof(rec));
    if (world > 1) {
        if (rank == 0) {
            (void)unlink(path);      // never let a reader of THIS job find an earlier job's id while the new one is being made
            int rc = pt_comm_unique_id(rec.id);
            if (rc) return rc;
            rec.magic = kIdMagic; rec.world = (uint32_t)world; rec.tag = job_tag;
            const std::string tmp = std::string(path) + ".tmp";
            FILE* f = fopen(tmp.c_str(), "wb");
            if (!f || fwrite(&rec, 1, sizeof(rec), f) != sizeof(rec)) { if (f) fclose(f); pt_set_error("pt_comm_create_from_file: cannot write %s", tmp.c_str()); return PT_ERR_IO; }
            fclose(f);
            if (rename(tmp.c_str(), path) != 0) { pt_set_error("pt_comm_create_from_file: cannot rename to %s", path); return PT_ERR_IO; }
        } else {
            bool got = false;
            for (int waited_ms = 0; waited_ms <= timeout_s * 1000; waited_ms += 20) {
                FILE* f = fopen(path, "rb");
                if (f) {
                    IdFile r;
                    const bool whole = fread(&r, 1, sizeof(r), f) == sizeof(r);
                    fclose(f);
                    if (whole && r.magic == kIdMagic && r.world == (uint32_t)world && r.tag == job_tag) { rec = r; got = true; break; }
                }
                usleep(20000);
            }
            if (!got) { pt_set_error("pt_comm_create_from_file: rank %d timed out waiting for %s (world %d, job tag %llu)", rank, path, world, (unsigned long long)job_tag); return PT_ERR_IO; }
        }
    }
    const int rc = pt_comm_create(rec.id, rank, world, device, out);
    if (world > 1 && rank == 0) (void)unlink(path);      // joined (or failed): the id is of no further use to anyone
    return rc;
}

int pt_comm_create_from_file(const char* path, int32_t rank, int32_t world, int32_t device, int32_t timeout_s, PtComm** out)
{
    return pt_comm_create_from_file_tagged(path, 0, rank, world, device, timeout_s, out);
}

void pt_comm_destroy(PtComm* c)
{
    if (!c) return;
    if (c->comm) { Rccl* R = rccl(); if (R) (void)R->CommDestroy(c->comm); }
    if (c->d_status) { (void)hipSetDevice(c->device); (void)hipFree(c->d_status); }
    delete c;
}

int32_t pt_comm_rank(const PtComm* c) { return c ? c->rank : -1; }
int32_t pt_comm_world(const PtComm* c) { return c ? c->world : 0; }

// The single exchange step.  d_gathered (rank 0 only; may be NULL elsewhere) receives the world buffers of n_floats each,
// rank-major — the layout pt_untile reads.  Asynchronous on hip_stream.
int pt_gather_tiles(PtComm* c, const float* d_tiles, int64_t n_floats, float* d_gathered, void* hip_stream)
{
    if (!c || !d_tiles || n_floats < 0 || (c->rank == 0 && !d_gathered)) { pt_set_error("pt_gather_tiles: bad argument"); return PT_ERR_INVALID; }
    hipStream_t stream = (hipStream_t)hip_stream;
    if (c->world == 1) {
        if (d_gathered != d_tiles) HIPCHK(hipMemcpyAsync(d_gathered, d_tiles, (size_t)n_floats * 4, hipMemcpyDeviceToDevice, stream));
        return PT_OK;
    }
    Rccl* R = rccl();
    if (!R || !c->comm) { pt_set_error("pt_gather_tiles: communicator has no RCCL handle"); return PT_ERR_INVALID; }
    HIPCHK(hipSetDevice(c->device));
    NCCLCHK(R->Gather(d_tiles, d_gathered, (size_t)n_floats, ncclFloat, 0, c->comm, stream));
    return PT_OK;
}

// Gather + de-interleave: rank 0 ends with the row-major W*H*3 frame in d_frame_rgb.  d_gathered is scratch of
// world * pt_tiles_floats() floats on rank 0 (NULL elsewhere, where d_frame_rgb is ignored too).
int pt_gather_frame(PtComm* c, const float* d_tiles, const PtCamera* cam, const PtParams* prm, float* d_gathered, float* d_frame_rgb, void* hip_stream)
{
    if (!c || !cam || !prm) { pt_set_error("pt_gather_frame: NULL argument"); return PT_ERR_INVALID; }
    if (prm->world != c->world || prm->rank != c->rank) { pt_set_error("pt_gather_frame: params say rank %d of %d, communicator rank %d of %d", prm->rank, prm->world, c->rank, c->world); return PT_ERR_INVALID; }
    const int64_t n = pt_tiles_floats(cam, prm);
    if (n < 0) return PT_ERR_INVALID;
    int rc = pt_gather_tiles(c, d_tiles, n, d_gathered, hip_stream);
    if (rc) return rc;
    if (c->rank == 0) {
        if (!d_frame_rgb) { pt_set_error("pt_gather_frame: rank 0 needs d_frame_rgb"); return PT_ERR_INVALID; }
        return pt_untile(d_gathered, cam, c->world, d_frame_rgb, hip_stream);
    }
    return PT_OK;
}

// Whole-frame convenience for a host program without HIP code of its own (the multi-GPU sibling of pt_render): renders this
// rank's tiles for all passes, runs the single gather, and on rank 0 copies the assembled frame to h_accum_rgb[W*H*3]
// (ignored on the other ranks).  prm->rank / prm->world are taken from the communicator.  Synchronous.
int pt_render_split(PtScene* s, const PtCamera* cam, const PtParams* prm, PtComm* c, float* h_accum_rgb)
{
    if (!s || !cam || !prm || !c || (c->rank == 0 && !h_accum_rgb)) { pt_set_error("pt_render_split: NULL argument"); return PT_ERR_INVALID; }
    if (ptk_scene_device(s) != c->device) {
        pt_set_error("pt_render_split: the scene lives on device %d, the communicator on device %d", ptk_scene_device(s), c->device);
        return PT_ERR_INVALID;
    }
    PtParams p = *prm; p.rank = c->rank; p.world = c->world;
    const int64_t nt = pt_tiles_floats(cam, &p), wb = pt_work_bytes(cam, &p);
    if (nt < 0 || wb < 0) return PT_ERR_INVALID;      // the same on every rank (same camera and params): nobody enters a collective
    HIPCHK(hipSetDevice(c->device));
    float *d_tiles = nullptr, *d_gathered = nullptr, *d_frame = nullptr; void* d_work = nullptr;
    auto local = [&]() -> int {      // everything that can fail on this rank alone
        HIPCHK(hipMalloc((void**)&d_tiles, (size_t)nt * 4));
        HIPCHK(hipMalloc(&d_work, (size_t)wb));
        if (c->rank == 0) {
            HIPCHK(hipMalloc((void**)&d_gathered, (size_t)nt * 4 * (size_t)c->world));
            HIPCHK(hipMalloc((void**)&d_frame, (size_t)cam->W * cam->H * 12));
        }
        return pt_render_tiles(s, cam, &p, d_tiles, d_work, nullptr);
    };
    auto body = [&]() -> int {
        int rc = local();
        if (c->world > 1) {
            // Every rank reports how its render went BEFORE the gather (a 4-byte max all-reduce): a rank that failed still takes
            // part, so its peers learn of it and skip the gather instead of waiting in it for a buffer that will never come.
            Rccl* R = rccl();
            if (!R || !c->comm || !c->d_status) { pt_set_error("pt_render_split: communicator has no RCCL handle"); return PT_ERR_INVALID; }
            const int32_t mine = rc ? 1 : 0;
            int32_t any = 1;
            if (hipMemcpy(c->d_status, &mine, 4, hipMemcpyHostToDevice) != hipSuccess ||
                R->AllReduce(c->d_status, c->d_status, 1, ncclInt32, ncclMax, c->comm, nullptr) != ncclSuccess ||
                hipMemcpy(&any, c->d_status, 4, hipMemcpyDeviceToHost) != hipSuccess) {
                // the control collective itself failed: nothing sensible can follow on this communicator
                (void)R->CommAbort(c->comm); c->comm = nullptr;
                if (!rc) pt_set_error("pt_render_split: status exchange failed on rank %d; communicator aborted", c->rank);
                return rc ? rc : PT_ERR_DEVICE;
            }
            if (rc) return rc;                                  // this rank's own error (message already set)
            if (any) { pt_set_error("pt_render_split: another rank failed to render its tiles; gather skipped on rank %d", c->rank); return PT_ERR_DEVICE; }
        } else if (rc) return rc;
        rc = pt_gather_frame(c, d_tiles, cam, &p, d_gathered, d_frame, nullptr);
        if (rc) return rc;
        HIPCHK(hipDeviceSynchronize());
        if (c->rank == 0) HIPCHK(hipMemcpy(h_accum_rgb, d_frame, (size_t)cam->W * cam->H * 12, hipMemcpyDeviceToHost));
        return PT_OK;
    };
    const int rc = body();
    (void)hipFree(d_tiles); (void)hipFree(d_work); (void)hipFree(d_gathered); (void)hipFree(d_frame);
    return rc;
}

}  // extern "C"
