// pt_probe.hip — machine probes for the measurement side of the C-ABI (include/pt_api.h: pt_dbg_valu_rate).
//
// The traversal kernel is bound by vector-ALU issue, not by HBM, so its roofline needs the
// chip's real VALU issue rate.  This probe measures it instead of assuming it: every wave runs a long
// unrolled stream of INDEPENDENT vector instructions of one kind (no memory traffic, no dependent chains shorter
// than 16 instructions), with a chosen number of waves per SIMD, and reports wave-instructions per second
// chip-wide together with the shader clock it ran at (s_memtime ticks per s_memrealtime tick).
// Nothing of the render path depends on this file.
#include <hip/hip_runtime.h>
#include <cstdint>
#include "../../include/pt_api.h"

void pt_set_error(const char* fmt, ...);   // pt_host.cpp

namespace {

// 16 independent accumulators; `OP` picks the instruction.  Inline asm keeps the compiler from folding or
// re-associating anything; operands never leave registers.
#define PT_REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ __launch_bounds__(256) void valu_stream(float* __restrict__ out, int iters, unsigned long long* __restrict__ clk, int halfMask)
{
    float a[16];
    typedef float f2v __attribute__((ext_vector_type(2)));
    f2v p[8];
    double d[8];
#pragma unroll
    for (int k = 0; k < 16; k++) a[k] = (float)(threadIdx.x + k) * 1e-3f;
#pragma unroll
    for (int k = 0; k < 8; k++) { p[k] = (f2v){a[2 * k], a[2 * k + 1]}; d[k] = (double)a[k]; }
    const float m = 0.999f, c = 1e-6f;
    const f2v pm = {m, m}, pc = {c, c};
    const double dm = 0.999, dc = 1e-6;
    uint32_t u[16];
#pragma unroll
    for (int k = 0; k < 16; k++) u[k] = threadIdx.x * 2654435761u + k;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    // lanes 32..63 switched off for the `halfMask` variant: a masked-off lane costs the same issue slot
    if (!halfMask || (threadIdx.x & 63) < 32) {
        for (int i = 0; i < iters; i++) {
            if (OP == 0) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 1) {
#define X(k) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[k & 7]) : "v"(pm), "v"(pc));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 2) {
#define X(k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 3) {
#define X(k) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(a[k]) : "v"(u[k]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 4) {
#define X(k) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 5) {
#define X(k) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(d[k & 7]) : "v"(dm), "v"(dc));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 6) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]) : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k & 7]) : "v"(pm));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 16; k++) s += a[k] + (float)u[k];
#pragma unroll
    for (int k = 0; k < 8; k++) s += p[k].x + p[k].y + (float)d[k];
    if (s == 123.456f) out[0] = s;       // never true: keeps the registers live
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int OP>
hipError_t run(int blocks, int iters, int halfMask, float* d_out, unsigned long long* d_clk, hipEvent_t e0, hipEvent_t e1)
{
    hipLaunchKernelGGL(valu_stream<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters / 8, d_clk, halfMask);      // warm-up
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(valu_stream<OP>, dim3(blocks), dim3(256), 0, 0, d_out, iters, d_clk, halfMask);
    (void)hipEventRecord(e1, 0);
    return hipEventSynchronize(e1);
}

}  // namespace

extern "C" {

// op: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_max3_f32, 3 v_cvt_f32_ubyte1, 4 v_add_u32, 5 v_fma_f64, 6 v_cndmask_b32, 7 v_pk_mul_f32;
// op + 16: the same with lanes 32..63 masked off.  waves_per_simd 1..8 (256-thread workgroups, one wave per SIMD each).
PT_API int pt_dbg_valu_rate(int32_t device, int32_t op, int32_t waves_per_simd, int32_t iters, double* wave_insts_per_s, double* clock_ghz)
{
    const int half = (op & 16) ? 1 : 0;
    op &= 15;
    if (!wave_insts_per_s || op < 0 || op > 7 || waves_per_simd < 1 || waves_per_simd > 8 || iters < 8 || iters > (1 << 22)) {
        pt_set_error("pt_dbg_valu_rate: bad argument");
        return PT_ERR_INVALID;
    }
    if (hipSetDevice(device) != hipSuccess) { pt_set_error("pt_dbg_valu_rate: hipSetDevice failed"); return PT_ERR_DEVICE; }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) { pt_set_error("pt_dbg_valu_rate: no device properties"); return PT_ERR_DEVICE; }
    const int cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    const int blocks = cus * waves_per_simd;
    float* d_out = nullptr; unsigned long long* d_clk = nullptr; hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = PT_OK;
    do {
        if (hipMalloc((void**)&d_out, 64) != hipSuccess || hipMalloc((void**)&d_clk, 64) != hipSuccess) { pt_set_error("pt_dbg_valu_rate: hipMalloc failed"); rc = PT_ERR_DEVICE; break; }
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        hipError_t e;
        switch (op) {
        case 0: e = run<0>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 1: e = run<1>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 2: e = run<2>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 3: e = run<3>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 4: e = run<4>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 5: e = run<5>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 6: e = run<6>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        default: e = run<7>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        }
        if (e != hipSuccess) { pt_set_error("pt_dbg_valu_rate: kernel failed: %s", hipGetErrorString(e)); rc = PT_ERR_DEVICE; break; }
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long clk[2] = {0, 0};
        (void)hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
        const double waves = (double)blocks * 4.0;
        *wave_insts_per_s = waves * (double)iters * 64.0 / ((double)ms * 1e-3);
        if (clock_ghz) *clock_ghz = clk[1] ? (double)clk[0] / (double)clk[1] * 0.1 : 0.0;      // s_memrealtime ticks at 100 MHz
    } while (0);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d_out) (void)hipFree(d_out);
    if (d_clk) (void)hipFree(d_clk);
    return rc;
}

}  // extern "C"
