/* oracle/pt_oracle.h — TEST INFRASTRUCTURE ONLY.
 *
 * C interface of the CPU restatement (oracle/pt_oracle.cpp) of the reference's
 * radiance-integrator path.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library, and only as the checker / the timed CPU
 * baseline.  The product (pathtrace-on-cuda_amd/) never includes, links or calls it.
 *
 * All arrays are raw little-endian float32 / int32.  Record layouts shared with
 * oracle/ref_driver.cpp (the partial build of the real reference) so outputs can be
 * compared byte for byte:
 *   TRI record  (88 f): V0 V1 V2 | T0 T1 T2 | B0 B1 B2 | N0 N1 N2 | normal | E1 | E2 |
 *                       u0 v0 u1 v1 u2 v2 | mat0 mat1 mat2 (12 f each) | area
 *   MAT record  (12 f): emittance(3) albedo(3) specular(3) opacity roughness metallic
 *   HIT record  (29 f): hit t u v frontface | p | normal | tangent | bitangent | MAT
 *   TRI48 input (48 f): V0 V1 V2 N0 N1 N2 T0 T1 T2 B0 B1 B2 | MAT
 *   SPH record  (16 f): center(3) rad | MAT
 *   RAY10 input (10 f): index org(3) dir(3) tmin tmax normalise(0/1)
 *   RAY8 input  ( 8 f): org(3) dir(3) tmin tmax           (dir used as given)
 *   NODE        (40 B): bMin(3f) bMax(3f) childL childR primStart primEnd (int32)
 */
#ifndef PT_ORACLE_H
#define PT_ORACLE_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define O_TRI_FLOATS 88
#define O_HIT_FLOATS 29

/* Counters filled by o_raycast / o_render (all int64):
 * [0] rays cast  [1] BVH nodes fetched  [2] triangle tests  [3] sphere tests
 * [4] rays with an accepted closest hit  [5] camera paths (samples)  [6] max stack depth */
#define O_NUM_COUNTERS 8

typedef struct {
    float pos[3];
    float forward[3];
    float up[3];
    float right[3];
    float fovy_deg;      /* Camera::fovy, degrees (include/camera.h:20) */
    float aspect;        /* Camera::aspect */
    int   W, H;          /* Camera::Screen_W / Screen_H */
} OCamera;

typedef struct {
    int passes;          /* NUM_MULTI_SAMPLE        (CudaUtil.cuh:18) */
    int spp_per_pass;    /* NUM_SAMPLE              (CudaUtil.cuh:19) */
    int max_bounce;      /* MAX_BOUNCE              (CudaUtil.cuh:15) */
    int rr_bounce;       /* RUSSIAN_ROULETTE_BOUNCE (CudaUtil.cuh:16) */
    float rr_floor;      /* PROB_STOP_BOUNCE        (CudaUtil.cuh:17) */
    int max_refract;     /* the literal 8 in `RefractCnt++>8` (CudaUtil.cuh:354) */
    int first_pass;      /* SampleIDX of the first pass rendered by this call */
    /* pixel window rendered (whole frame: 0,0,W,H); seeds always use the full-frame offset */
    int x0, y0, x1, y1;
} OParams;

/* libm contract: 0 = glibc float functions (what the reference gets when its source is
 * compiled for the host; used to reproduce SURVEY.md's probe anchors), 1 = correctly
 * rounded float results obtained through double precision (the pinned contract the HIP
 * kernel is held to).  Returns the previous mode. */
int  o_set_libm(int mode);
/* Dev aid (tools/trav_lab): log every ray RayCast sees (org dir tmax, 7 f each) during
 * single-threaded o_render calls.  Pass NULL to stop. */
void o_set_raylog(float* buf7, int cap);
int  o_raylog_count(void);

void o_rng(uint64_t seed, int n, uint32_t* raw_out, float* uniform_out);

int  o_bvh_build(const void* prims336, int n, void** nodes_out, int* n_nodes,
                 float** tris_out, int* n_tris, int* max_depth);
void o_free(void* p);

void o_tri_hit(const float* tris48, int M, const float* rays10, int R, float* out_hits);
void o_sphere_hit(const float* sph16, int S, const float* rays10, int R, float* out_hits);
void o_vecmath(const float* in7, int K, float* out21);

void* o_scene_create(const void* nodes, int n_nodes, const float* tris88, int n_tris,
                     const float* sph16, int n_spheres);
void  o_scene_destroy(void* scene);
int   o_scene_num_lights(void* scene);

void o_raycast(void* scene, const float* rays8, int R, float* out_hits, int* out_prim,
               int64_t* counters);

void o_camera_basis(const float rot_deg[3], float forward[3], float up[3], float right[3]);

int  o_render(void* scene, const OCamera* cam, const OParams* prm, float* accum_rgb,
              int64_t* counters, int nthreads);

void o_tonemap(const float* raw_rgb, int n_pixels, int sample_cnt, unsigned char* rgb8);
void o_u8(const float* v, int n, unsigned char* out);
void o_pixel_dir(const OCamera* cam, const int* pxpypass, int n, float* out8);   /* StartRender prologue + GetPixelDirection, srcs/pathtracer.cu:33-40,70-74 */
void o_nee(void* scene, const float* in5, int n, float* out12);                  /* NEE sample + GetLightColor, include/CudaUtil.cuh:38-48,150-166,235-245 */      /* ConverToUint8, include/image.h:5-8 */

/* BxDF known-answer table.  lobe: 0 gltfpbr, 1 reflective, 2 refractive, 3 pure_refractive.
 * in  (28 f/row): normal(3) tangent(3) bitangent(3) frontface | albedo(3) specular(3)
 *                 roughness metallic | wo(3) wi(3) | seed_lo seed_hi(as uint32 bit patterns)
 * out (12 f/row): eval(wo,wi)(3) pdf(wo,wi) | sampled wi(3) | eval(wo,sampled)(3) pdf(wo,sampled) draws */
void o_bxdf(int lobe, const float* in28, int n, float* out12);

#ifdef __cplusplus
}
#endif
#endif
