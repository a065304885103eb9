/* include/pt_api.h — C-ABI drop-in boundary of the MI355X-native path tracer.
 *
 * One shared library (libptamd.so, built from pathtrace-on-cuda_amd/) exports exactly the
 * entry points below.  Plain pointers and sizes only: no C++ types, no torch types.
 * Citations `file:line` are relative to the reference tree (WaterPlease/PathTrace-on-CUDA).
 * INTEGRATION.md shows the reference-side adaptor (`PathTracer::Render` re-implemented on
 * top of these calls) a maintainer would add.
 *
 * Every function returns PT_OK (0) or a negative PtStatus and records a message that
 * pt_last_error() returns (thread-local).  The reference's own convention for GPU errors
 * — print to stderr, reset the device, exit(99) (include/CudaUtil.cuh:28-36) — is kept by
 * the C++ adaptor `PathTracer::Render` (pathtrace-on-cuda_amd/host/ref_surface.cpp), not
 * imposed on C callers.
 */
#ifndef PT_API_H
#define PT_API_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PT_API __attribute__((visibility("default")))

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_INVALID = -1,      /* bad argument / inconsistent scene */
    PT_ERR_NO_LIGHT = -2,     /* no emissive triangle: `curand(s) % Nl` is UB in the reference (include/CudaUtil.cuh:235) */
    PT_ERR_DEVICE = -3,       /* HIP runtime error (message carries hipGetErrorString) */
    PT_ERR_IO = -4,           /* file could not be read / written */
    PT_ERR_UNSUPPORTED = -5   /* e.g. BVH deeper than the traversal stack */
} PtStatus;

/* ----------------------------------------------------------------------------------
 * Host scene types: byte-compatible mirrors of the reference's host structs, so the
 * adaptor can pass `bvh->primitives.data()` straight through.
 * -------------------------------------------------------------------------------- */
typedef struct PtVec3 { float x, y, z; } PtVec3;                 /* glm::vec3 */
typedef struct PtVec2 { float x, y; } PtVec2;                    /* glm::vec2 */

typedef struct PtMaterialOnCPU {                                 /* include/mesh.h:11-19 */
    PtVec3 emittance, albedo, specular;
    float opacity, metallic, roughness;
} PtMaterialOnCPU;

typedef struct PtVertex {                                        /* include/mesh.h:21-37, 112 B */
    PtVec3 Position, Normal;
    PtVec2 TexCoords;
    PtVec3 Tangent, Bitangent;
    PtMaterialOnCPU mat;
    float u, v;
} PtVertex;

typedef struct PtPrimitive { PtVertex v1, v2, v3; } PtPrimitive; /* include/bvh.h:8-13, 336 B */

typedef struct PtMaterial {                                      /* include/CudaPrimitive.cuh:15-23, 48 B */
    float emittance[3], albedo[3], specular[3];
    float opacity, roughness, metallic;
} PtMaterial;

typedef struct PtSphere {                                        /* include/CudaPrimitive.cuh:249-323 (data members), 64 B */
    float center[3];
    float rad;
    PtMaterial mat;
} PtSphere;

typedef struct PtBVHNode {                                       /* CudaBVHNode, include/CudaPrimitive.cuh:237-247, 40 B */
    float bMin[3], bMax[3];
    int32_t childL, childR;                                      /* -1 = none */
    int32_t primStart, primEnd;                                  /* inclusive range, -1 = interior */
} PtBVHNode;

/* Flattened triangle = the data members of the reference's `Triangle`
 * (include/CudaPrimitive.cuh:217-234) after Triangle::Copy, without the vptr.  352 B. */
typedef struct PtTriangle {
    float V0[3], V1[3], V2[3];
    float T0[3], T1[3], T2[3];
    float B0[3], B1[3], B2[3];
    float N0[3], N1[3], N2[3];
    float normal[3], E1[3], E2[3];
    float u0, v0, u1, v1, u2, v2;
    PtMaterial mat0, mat1, mat2;
    float area;
} PtTriangle;

typedef struct PtCamera {                                        /* what PathTracer::Render reads, srcs/pathtracer.cu:128-129,193-198 */
    float pos[3];
    float forward[3], up[3], right[3];                           /* Camera::GetForward/GetUp/GetRight */
    float fovy_deg;                                              /* Camera::fovy (degrees) */
    float aspect;                                                /* Camera::aspect */
    int32_t W, H;                                                /* Camera::Screen_W / Screen_H */
} PtCamera;

typedef struct PtParams {                                        /* the reference's compile-time tunables, include/CudaUtil.cuh:15-19 */
    int32_t passes;           /* NUM_MULTI_SAMPLE (8)   */
    int32_t spp_per_pass;     /* NUM_SAMPLE (1024)      */
    int32_t max_bounce;       /* MAX_BOUNCE (8)         */
    int32_t rr_bounce;        /* RUSSIAN_ROULETTE_BOUNCE (3) */
    float   rr_floor;         /* PROB_STOP_BOUNCE (0.5) */
    int32_t max_refract;      /* the literal 8 of `RefractCnt++>8`, include/CudaUtil.cuh:354 */
    int32_t first_pass;       /* SampleIDX of the first pass of this call (seed = offset + SampleIDX*W*H, srcs/pathtracer.cu:71) */
    /* Tile split (new; the reference is single-device).  The frame is cut into 8x8-pixel
     * tiles numbered row-major; this call renders tiles t with t % world == rank.
     * world = 1, rank = 0 renders the whole frame. */
    int32_t rank, world;
} PtParams;

PT_API void pt_params_default(PtParams* p);                      /* the reference's values */

PT_API const char* pt_last_error(void);
PT_API const char* pt_version(void);

/* ----------------------------------------------------------------------------------
 * (a1,a2) Host acceleration-structure build + flatten.
 * Replaces SAHBVH::GenBVHTree (srcs/bvh.cpp:426-511) followed by LoadFromBVH
 * (srcs/CudaPrimitive.cu:8-145) and the Triangle::Copy loop (srcs/pathtracer.cu:164-166).
 * -------------------------------------------------------------------------------- */
typedef struct PtFlatBVH PtFlatBVH;
PT_API int  pt_bvh_build_sah(const PtPrimitive* prims, int32_t n_prims, PtFlatBVH** out);
PT_API void pt_bvh_free(PtFlatBVH* b);
PT_API int32_t pt_bvh_num_nodes(const PtFlatBVH* b);
PT_API int32_t pt_bvh_num_tris(const PtFlatBVH* b);
PT_API int32_t pt_bvh_max_depth(const PtFlatBVH* b);             /* "Maximum depth of tree", CudaPrimitive.cu:144 */
PT_API const PtBVHNode*  pt_bvh_nodes(const PtFlatBVH* b);       /* == CudaBVH  */
PT_API const PtTriangle* pt_bvh_tris(const PtFlatBVH* b);        /* == CudaPrims after Copy */

/* ----------------------------------------------------------------------------------
 * (a3) Scene upload.  Replaces the cudaMallocManaged + host-fill block of
 * PathTracer::Render (srcs/pathtracer.cu:142-188).  `device` is the HIP device ordinal.
 * The scene owns its HBM allocations until pt_scene_destroy.
 * -------------------------------------------------------------------------------- */
typedef struct PtScene PtScene;
PT_API int  pt_scene_create(const PtBVHNode* nodes, int32_t n_nodes,
                            const PtTriangle* tris, int32_t n_tris,
                            const PtSphere* spheres, int32_t n_spheres,
                            int32_t device, PtScene** out);
PT_API void pt_scene_destroy(PtScene* s);
PT_API int32_t pt_scene_num_lights(const PtScene* s);            /* "ADD light" count, pathtracer.cu:167-173 */
PT_API int64_t pt_scene_device_bytes(const PtScene* s);

/* ----------------------------------------------------------------------------------
 * (a3-a11) Render.  Replaces the StartRender launch loop (srcs/pathtracer.cu:236-246).
 *
 * pt_render_tiles: device-resident; all GPU work is enqueued on `hip_stream` (a hipStream_t, may be NULL) and ordered after
 *   whatever the caller enqueued there before.  Writes this rank's tiles, tile-major, into d_tiles:
 *     d_tiles[((lt * 64) + (ty*8+tx)) * 3 + c],  lt = local tile index (global tile
 *     t = lt*world + rank), float32, size pt_tiles_floats().  Value = sum over the call's
 *     passes of the per-pass mean radiance (the reference's `image[offset] += mean`,
 *     pathtracer.cu:81), starting from 0.  d_work is scratch of pt_work_bytes() bytes.
 *   BLOCKING in the default render path (mode 1, the queue-driven pipeline): the number of bounce iterations is
 *   data-dependent, so the call polls the live-stream count (a 4-byte read-back + hipStreamSynchronize every 16-64
 *   iterations) and returns only when the render has drained; on return only the final pass-sum kernel may still be
 *   running on `hip_stream`.  As blocking as the reference's own Render (cudaDeviceSynchronize after every launch,
 *   srcs/pathtracer.cu:236-246).  A caller that drives several scenes or GPUs from one process gives each its own host
 *   thread.  Mode 0 (pt_set_mode, the one-kernel state machine) is fully asynchronous.  One render at a time per PtScene
 *   (the scene owns the pinned poll word, events and counters the render uses); different scenes are independent.
 * pt_untile: scatter gathered tile buffers (rank-major: world buffers of
 *   pt_tiles_floats() each) into a row-major W*H*3 frame, on the device.
 * pt_render: convenience, whole frame (world=1) into a host buffer, synchronous;
 *   h_accum_rgb[W*H*3] is overwritten.
 * -------------------------------------------------------------------------------- */
PT_API int64_t pt_tiles_floats(const PtCamera* cam, const PtParams* prm);
PT_API int64_t pt_work_bytes(const PtCamera* cam, const PtParams* prm);
PT_API int  pt_render_tiles(PtScene* s, const PtCamera* cam, const PtParams* prm,
                            float* d_tiles, void* d_work, void* hip_stream);
PT_API int  pt_untile(const float* d_gathered, const PtCamera* cam, int32_t world,
                      float* d_frame_rgb, void* hip_stream);
PT_API int  pt_render(PtScene* s, const PtCamera* cam, const PtParams* prm, float* h_accum_rgb);
/* Duration of the most recent pt_render_tiles launch sequence on this scene, measured with
 * HIP events on the stream it was launched on (ms), and the kernel's own work counters. */
PT_API int  pt_last_render_ms(PtScene* s, float* ms);
/* Durations (ms) of the most recent render_units launches on this scene (up to 64, oldest
 * first), each measured with a HIP event pair recorded on the launch stream around that
 * kernel only.  Returns the count written (<= cap) or a negative PtStatus; reset != 0 clears
 * the history.  Blocks until those launches have finished. */
PT_API int  pt_render_timings(PtScene* s, float* ms_out, int32_t cap, int32_t reset);

/* ----------------------------------------------------------------------------------
 * (e) Multi-GPU exchange (new: the reference is single-device, srcs/pathtracer.cu:124-259).  One process per GPU; rank r
 * renders tiles t % world == r with pt_render_tiles, then ONE collective — a gather of the tile buffers to rank 0,
 * RCCL ncclGather (rccl.h:745) over xGMI — and pt_untile on rank 0 assemble the frame.  No torch, no MPI needed:
 *   rank 0:  pt_comm_unique_id(id) and hand the 128 bytes to the other processes (any channel), or let
 *            pt_comm_create_from_file do it through a file for the processes of one node;
 *   all:     pt_comm_create(id, rank, world, device, &comm); ... pt_render_tiles(...);
 *            pt_gather_frame(comm, d_tiles, &cam, &prm, d_gathered, d_frame, stream);   (both asynchronous on `stream`)
 * world == 1 never loads RCCL (the gather is a device copy).  INTEGRATION.md shows the 8-process C++ use.
 * -------------------------------------------------------------------------------- */
#define PT_COMM_ID_BYTES 128
typedef struct PtComm PtComm;
PT_API int  pt_comm_unique_id(uint8_t id[PT_COMM_ID_BYTES]);
PT_API int  pt_comm_create(const uint8_t id[PT_COMM_ID_BYTES], int32_t rank, int32_t world, int32_t device, PtComm** out);
/* Rendezvous through a file for the processes of one node: rank 0 removes any file at `path`, writes { world, job_tag, id },
 * joins and removes the file again; ranks > 0 wait (timeout_s) for a file carrying THEIR world and job_tag — the file of an
 * earlier job is ignored.  job_tag = any value the ranks of one job share and other jobs do not (launcher pid, job id).
 * pt_comm_create_from_file = the same with job_tag 0.  ncclCommInitRank itself has no timeout: a rank that never arrives
 * leaves the others waiting in it, as with any RCCL program. */
PT_API int  pt_comm_create_from_file_tagged(const char* path, uint64_t job_tag, int32_t rank, int32_t world, int32_t device, int32_t timeout_s, PtComm** out);
PT_API int  pt_comm_create_from_file(const char* path, int32_t rank, int32_t world, int32_t device, int32_t timeout_s, PtComm** out);
PT_API void pt_comm_destroy(PtComm* c);
PT_API int32_t pt_comm_rank(const PtComm* c);
PT_API int32_t pt_comm_world(const PtComm* c);
/* d_gathered: rank 0 only (NULL elsewhere), world x n_floats floats, rank-major — what pt_untile reads. */
PT_API int  pt_gather_tiles(PtComm* c, const float* d_tiles, int64_t n_floats, float* d_gathered, void* hip_stream);
PT_API int  pt_gather_frame(PtComm* c, const float* d_tiles, const PtCamera* cam, const PtParams* prm,
                            float* d_gathered, float* d_frame_rgb, void* hip_stream);
/* The multi-GPU sibling of pt_render for hosts without HIP code of their own: this rank's tiles, the gather, and on rank 0 the
 * assembled frame in h_accum_rgb[W*H*3] (ignored elsewhere).  Rank and world come from the communicator; the scene must live on the
 * communicator's device.  Synchronous.  Before the gather the ranks exchange a 4-byte status (max all-reduce): if ANY rank failed to
 * render its tiles, EVERY rank returns an error and nobody waits in the gather. */
PT_API int  pt_render_split(PtScene* s, const PtCamera* cam, const PtParams* prm, PtComm* c, float* h_accum_rgb);

/* ----------------------------------------------------------------------------------
 * (a12,a13) Output + camera helpers (host).
 * pt_tonemap_u8 = exportImage (srcs/pathtracer.cu:94-112): /SampleCnt, ACESFilm
 *   (include/CudaUtil.cuh:383-391), ConverToUint8 (include/image.h:5-8).
 * pt_write_png  = Image::WriteTo (srcs/image.cpp:22-25), RGB8.
 * pt_camera_basis = Camera::SetRotation + GetRight (srcs/camera.cpp:32-66).
 * -------------------------------------------------------------------------------- */
PT_API int  pt_tonemap_u8(const float* raw_rgb, int64_t n_pixels, int32_t sample_cnt, uint8_t* rgb8);
PT_API int  pt_convert_u8(const float* values, int64_t n, uint8_t* out);      /* ConverToUint8, include/image.h:5-8, element-wise */
PT_API int  pt_write_png(const char* path, const uint8_t* data, int32_t W, int32_t H, int32_t channels);
PT_API void pt_camera_basis(const float rot_deg[3], float forward[3], float up[3], float right[3]);

/* ----------------------------------------------------------------------------------
 * Synthetic scenes (the reference ships none: .gitignore:365, renderer.cpp:102-115 load
 * absolute Windows paths).  Geometry exactly as SURVEY.md Appendix A.
 *   kind 0: Cornell box (12 tris)                      kind 1: + 1 stand-in mesh (69,564 tris)
 *   kind 2: + 4 instanced stand-ins (278,256 tris)
 * lat_lon: tessellation of the stand-in (187 in the configs; smaller for tests).
 * Returns the number of primitives; writes at most `cap` of them when prims != NULL.
 * -------------------------------------------------------------------------------- */
PT_API int32_t pt_scene_gen(int32_t kind, int32_t lat_lon, PtPrimitive* prims, int32_t cap);
/* Minimal Wavefront OBJ(+MTL) reader producing the Vertex data Model::processMesh would
 * (include/model.h:120-207) with BVH::AddModel's matrix bake (srcs/bvh.cpp:153-189):
 * uniform `scale` then `translate`.  Same count/cap convention as pt_scene_gen. */
PT_API int32_t pt_load_obj(const char* path, float scale, const float translate[3],
                           PtPrimitive* prims, int32_t cap);

/* ----------------------------------------------------------------------------------
 * Parity hooks: run single device functions of the integrator on the GPU so tests can
 * compare them with the oracle record by record (host pointers in and out).
 * Record layouts are those of oracle/pt_oracle.h (RAY8, HIT(29 f), BXDF in 28 f / out 12 f).
 * -------------------------------------------------------------------------------- */
PT_API int  pt_dbg_raycast(PtScene* s, const float* rays8, int32_t n, float* out_hits29, int32_t* out_prim);
PT_API int  pt_dbg_bxdf(int32_t device, int32_t lobe, const float* in28, int32_t n, float* out12);
PT_API int  pt_dbg_rng(int32_t device, uint64_t seed, int32_t n, uint32_t* raw_out, float* uniform_out);
PT_API int  pt_dbg_math(int32_t device, const float* in, int32_t n, float* out8);
/* The sin / cos pair of the BxDF samplers (pathtrace-on-cuda_amd/csrc/pt_sincos.h; angles in [0, 2 pi], include/Bxdf.cuh:23-41,140-150):
 * out2 = sin cos per input, to be compared with (float)sin((double)x), (float)cos((double)x) bit for bit. */
PT_API int  pt_dbg_sincos(int32_t device, const float* in, int32_t n, float* out2);
/* The per-ray set-up of the traversal kernel (csrc/pt_trace.h: ray_setup) on n directions (3 floats each): out5 = the reference's
 * Normalize(inv(dir)) (include/CudaUtil.cuh:60-63, :70) x, y, z | the kernel's cull scale | 1.0 for a degenerate direction; the
 * test compares it with IEEE arithmetic bit for bit. */
PT_API int  pt_dbg_ray_setup(int32_t device, const float* dir3, int32_t n, float* out5);
/* StartRender's prologue + GetPixelDirection (srcs/pathtracer.cu:33-40,70-74) for n rows (px, py, pass) of int32:
 * out8 = u1 u2 | dir.xyz (after the Ray constructor's second normalisation) | the RNG's next uniform draw | 0 0. */
PT_API int  pt_dbg_pixel_dir(int32_t device, const PtCamera* cam, const int32_t* pxpypass, int32_t n, float* out8);
/* One NEE sample + its visibility (include/CudaUtil.cuh:235-245, SamplePrimitive :38-48, GetLightColor :150-166) per row.
 * in5 = surface point p.xyz | RNG seed low, high (uint32 bits); out12 = light index (int bits) | sampled point xyz | pdfLight |
 * cosA | shadow ray t_max | closest-hit primitive of the shadow ray (int bits) | GetLightColor rgb | the RNG's next draw. */
PT_API int  pt_dbg_nee(PtScene* s, const float* in5, int32_t n, float* out12);
/* Measurement aid: stream triad a = b + s*c over three float4 arrays of `bytes_per_array`
 * each, `iters` times; *gb_per_s = bytes moved (2 reads + 1 write per element) / time of the timed launches. */
PT_API int  pt_dbg_triad(int32_t device, int64_t bytes_per_array, int32_t iters, double* gb_per_s);
/* Measurement aid: the chip's vector-ALU issue rate, measured — every wave runs a long stream of independent
 * instructions of one kind out of registers (op: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_max3_f32, 3 v_cvt_f32_ubyte1,
 * 4 v_add_u32, 5 v_fma_f64, 6 v_cndmask_b32, 7 v_pk_mul_f32, 8 v_fma_mix_f32, 9 v_cvt_f32_f16, 10 v_perm_b32, 11 v_min_f32,
 * 12 v_cvt_f32_u32, 13 v_ldexp_f32, 14 v_cmp_le_f32, 15 v_bfe_u32; +16 = lanes 32..63 masked off) at `waves_per_simd`
 * waves per SIMD (1..8).  *wave_insts_per_s = wave-instructions retired per second chip-wide; *clock_ghz = shader
 * clock during the run.  This is the denominator of bench.py's VALU roofline for the traversal kernel. */
PT_API int  pt_dbg_valu_rate(int32_t device, int32_t op, int32_t waves_per_simd, int32_t iters,
                             double* wave_insts_per_s, double* clock_ghz);
/* Work counters of the last pt_render_tiles on this scene (int64 x 8):
 * [0] rays [1] node records fetched [2] triangle tests [3] sphere tests [4] rays with hit
 * [5] camera paths [6] loop trips of the wave scheduler [7] lane-trips with an active ray */
PT_API int  pt_last_counters(PtScene* s, int64_t* out8);
/* Diagnostic (environment PTAMD_TSTAT=1 only): per wf_trace launch of the last render, in 100 MHz ticks:
 * ~(earliest wave start), ~(earliest time a wave found the ray queue empty; 0 = never), latest wave exit.
 * With PTAMD_TSTAT=1 pt_last_counters returns wf_trace's trip counters instead of the work counters.
 * n_launches = 0: out3n receives 32 int64 instead — the histogram of wave lifetimes in 32-microsecond bins.
 * n_launches = -n: out3n receives n int64 — the number of rays each of the first n launches traced.
 * n_launches = -3000 / -3001 / -3002 (PTAMD_TSTAT=1): 64 int64 histogram of node steps per ray (bins of 4) / 32 int64 histogram of
 * the stack depth after a node step / 5 int64 shader clocks summed over waves per section of the loop (refill, vote, node step,
 * triangle step, ray epilogue). */
PT_API int  pt_dbg_trace_timeline(PtScene* s, int64_t* out3n, int32_t n_launches);
/* Render path: 1 = queue-driven wavefront pipeline (default: traversal and shading are
 * separate kernels, lanes refill from a ray queue), 0 = the one-kernel state machine.
 * Both produce bit-identical frames.  Environment PTAMD_MODE overrides the default.
 * pt_last_iterations: bounce iterations the pipeline needed for the last render. */
PT_API int  pt_set_mode(PtScene* s, int32_t mode);
/* Per-launch timing of the two pipeline kernels (mode 1): pt_enable_trace_timing makes every following render record HIP
 * events, on the launch stream, before and after each of its first max_launches wf_trace launches and after the wf_shade
 * launch that follows it (0 = off); pt_trace_timing / pt_shade_timing return the summed and maximum duration in ms of the
 * wf_trace / wf_shade launches so bracketed, and how many were timed in the last render. */
PT_API int  pt_enable_trace_timing(PtScene* s, int32_t max_launches);
PT_API int  pt_trace_timing(PtScene* s, double* sum_ms, int32_t* launches, double* max_ms);
PT_API int  pt_shade_timing(PtScene* s, double* sum_ms, int32_t* launches, double* max_ms);
PT_API int  pt_last_iterations(PtScene* s);
/* Mode 1 hands the last streams of a render to one run-to-completion launch (wf_drain) once at
 * most `live_streams` are still alive (0 = never; default 80,000: the launch, its streams spread over every SIMD,
 * replaces the last ~200 latency-bound iterations of a render — +5 % for one rank of an 8-way split, +1 % on one GPU).
 * Result-neutral. */
PT_API int  pt_set_drain_threshold(PtScene* s, int32_t live_streams);
/* Mode 1, early shade: in a render call of at most `max_streams` streams (pixels of this rank x passes of the call) the shade step of
 * every iteration starts on a second HIP stream beside the draining traversal kernel (streams whose rays are all back are shaded at
 * once, the others — and all list appends — follow when the traversal has finished); renders of fewer than max_streams / 8 streams are
 * left alone too (launch-latency bound); 0 = never, default 2,500,000: on for one rank of
 * an 8-way tile split of 1080p x 8 passes (-6.5 % time), off for a full frame (whose launches are large next to the traversal's
 * ~0.3 ms launch tail).  Result-neutral: every stream goes through the same step (pathtrace-on-cuda_amd/csrc/pt_wavefront.hip:
 * wf_shade PHASE 1 / 2).  Environment PTAMD_EARLY overrides the default. */
PT_API int  pt_set_early_shade(PtScene* s, int32_t max_streams);
/* Mode 1, shading schedule: 1 = a stream whose path ends starts its next sample in the same step (bounces - 1 steps per sample,
 * two bounce evaluations per step), 0 = one bounce evaluation per step (bounces steps per sample, a shorter step), -1 = 0 while
 * more than PTAMD_TRS (4 M) streams are alive, 1 below.  Default: -1 for scenes whose surface table fits in L2 (<= 2 MB: +10 % on
 * the Cornell room), 1 otherwise (with the bunny one bounce per step is neutral on a full frame and costs 6 % for one rank of an
 * 8-way split).  Result-neutral: a stream goes through the same operations in the same order either way
 * (pathtrace-on-cuda_amd/csrc/pt_stream.h: shade_step_t). */
PT_API int  pt_set_shade_rounds(PtScene* s, int32_t mode);
/* Run the counting build of the kernel on the next pt_render_tiles calls (slower; mode 0). */
PT_API int  pt_enable_counters(PtScene* s, int32_t on);

#ifdef __cplusplus
}
#endif
#endif /* PT_API_H */
