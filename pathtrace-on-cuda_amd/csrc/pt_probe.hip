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
    const unsigned long long smask = 0x5555aaaa5555aaaaull + (unsigned long long)iters;
    uint32_t sdummy = 0;
    unsigned long long smaskw = smask;
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
            } else if (OP == 7) {
#define X(k) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[k & 7]) : "v"(pm));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 8) {      // f16 (high half of the first operand) x f32 + f32 -> f32
#define X(k) asm volatile("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(a[k]) : "v"(u[k]), "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 9) {
#define X(k) asm volatile("v_cvt_f32_f16 %0, %1" : "=v"(a[k]) : "v"(u[k]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 10) {
#define X(k) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "v"(u[(k + 2) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 11) {
#define X(k) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[k]) : "v"(a[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 12) {
#define X(k) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(a[k]) : "v"(u[k]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 13) {
#define X(k) asm volatile("v_ldexp_f32 %0, %0, %1" : "+v"(a[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 14) {
#define X(k) asm volatile("v_cmp_le_f32 vcc, %0, %1" : : "v"(a[k]), "v"(a[(k + 1) & 15]) : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 15) {
#define X(k) asm volatile("v_bfe_u32 %0, %0, %1, 8" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 17) {
#define X(k) asm volatile("v_mov_b32 %0, %1" : "=v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 18) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "s"(smask));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 19) {
#define X(k) asm volatile("v_and_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 20) {
#define X(k) asm volatile("v_or_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 21) {
#define X(k) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(u[k]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 22) {
#define X(k) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 23) {
#define X(k) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[k]) : "v"(m));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 24) {
#define X(k) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 25) {
#define X(k) asm volatile("v_max_i32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 26) {
#define X(k) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 27) {
#define X(k) asm volatile("v_rcp_f32 %0, %1" : "=v"(a[k]) : "v"(a[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 28) {
#define X(k) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "v"(u[(k + 2) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 29) {
#define X(k) asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(u[k]), "v"(u[(k + 1) & 15]) : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 30) {
#define X(k) asm volatile("v_div_fixup_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 31) {
#define X(k) asm volatile("v_div_fmas_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c) : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 32) {
#define X(k) asm volatile("v_div_scale_f32 %0, vcc, %1, %2, %1" : "=v"(a[k]) : "v"(m), "v"(c) : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 33) {
#define X(k) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 34) {
#define X(k) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 35) {
#define X(k) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 36) {
#define X(k) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "v"(u[(k + 2) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 37) {
#define X(k) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "v"(u[(k + 2) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 38) {
#define X(k) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[k]) : "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 39) {
#define X(k) asm volatile("v_cvt_f32_ubyte0 %0, %1" : "=v"(a[k]) : "v"(u[k]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 40) {
#define X(k) asm volatile("v_bfi_b32 %0, %0, %1, %2" : "+v"(u[k]) : "v"(u[(k + 1) & 15]), "v"(u[(k + 2) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 41) {
#define X(k) asm volatile("v_alignbit_b32 %0, %0, %1, 8" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 42) {
#define X(k) asm volatile("v_lshl_or_b32 %0, %0, 1, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 43) {
#define X(k) asm volatile("v_sqrt_f32 %0, %1" : "=v"(a[k]) : "v"(a[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 44) {
#define X(k) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(m), "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 45) {
#define X(k) asm volatile("v_min_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 46) {
#define X(k) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(u[k]), "v"(u[(k + 1) & 15]) : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 47) {
#define X(k) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(u[k]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 48) {
#define X(k) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 49) {
#define X(k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[k]) : "v"(a[(k + 1) & 15]), "v"(a[(k + 2) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 50) {
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 51) {
#define X(k) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 52) {
#define X(k) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(d[k & 7]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 53) {
#define X(k) asm volatile("v_lshl_add_u64 %0, %0, 1, %1" : "+v"(d[k & 7]) : "v"(d[(k + 1) & 7]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 55) {      // compare into vcc + select on vcc: one pair per count
#define X(k) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]) : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 56) {      // compare into an SGPR pair + select on it: one pair per count
#define X(k) asm volatile("v_cmp_gt_i32 %1, %0, %2\n\tv_cndmask_b32 %0, %0, %2, %1" : "+v"(u[k]), "+s"(smaskw) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 57) {      // vcc written by one vector compare, then 64 selects on it
                asm volatile("v_cmp_gt_i32 vcc, %0, %1" : : "v"(u[0]), "v"(u[1]) : "vcc");
#define X(k) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 58) {      // compare into vcc, two independent fmas, select on vcc
#define X(k) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_fma_f32 %2, %2, %3, %4\n\tv_fma_f32 %5, %5, %3, %4\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]), "+v"(a[k]), "+v"(a[(k + 8) & 15]) : "v"(u[(k + 1) & 15]), "v"(m), "v"(c) : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 59) {      // one compare into vcc, four selects on it (a compare-exchange of two (key, ref) pairs): 5 instructions per count
#define X(k) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %2, %0, %1, vcc\n\tv_cndmask_b32 %3, %1, %0, vcc\n\tv_cndmask_b32 %4, %4, %5, vcc\n\tv_cndmask_b32 %5, %5, %4, vcc" : "+v"(u[k & 3]), "+v"(u[4 + (k & 3)]), "+v"(u[8 + (k & 3)]), "+v"(u[12 + (k & 3)]), "+v"(p[k & 7].x), "+v"(p[k & 7].y) : : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 60) {      // the same through an SGPR pair
#define X(k) asm volatile("v_cmp_gt_i32 %6, %0, %1\n\tv_cndmask_b32 %2, %0, %1, %6\n\tv_cndmask_b32 %3, %1, %0, %6\n\tv_cndmask_b32 %4, %4, %5, %6\n\tv_cndmask_b32 %5, %5, %4, %6" : "+v"(u[k & 3]), "+v"(u[4 + (k & 3)]), "+v"(u[8 + (k & 3)]), "+v"(u[12 + (k & 3)]), "+v"(p[k & 7].x), "+v"(p[k & 7].y), "+s"(smaskw));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 61) {      // compare into vcc, select, select: 3 instructions per count
#define X(k) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32 %2, %0, %1, vcc\n\tv_cndmask_b32 %3, %1, %0, vcc" : "+v"(u[k & 3]), "+v"(u[4 + (k & 3)]), "+v"(u[8 + (k & 3)]), "+v"(u[12 + (k & 3)]) : : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 62) {      // min + max of the same two operands (what the compiler makes of a key-only compare-exchange): 2 instructions per count
#define X(k) asm volatile("v_min_i32 %2, %0, %1\n\tv_max_i32 %3, %0, %1" : "+v"(u[k & 3]), "+v"(u[4 + (k & 3)]), "+v"(u[8 + (k & 3)]), "+v"(u[12 + (k & 3)]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 63) {      // select on vcc in the 64-bit (VOP3) encoding, vcc not written in the loop
#define X(k) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 64) {      // one compare into vcc, four VOP3-encoded selects on it: 5 instructions per count
#define X(k) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\tv_cndmask_b32_e64 %2, %0, %1, vcc\n\tv_cndmask_b32_e64 %3, %1, %0, vcc\n\tv_cndmask_b32_e64 %4, %4, %5, vcc\n\tv_cndmask_b32_e64 %5, %5, %4, vcc" : "+v"(u[k & 3]), "+v"(u[4 + (k & 3)]), "+v"(u[8 + (k & 3)]), "+v"(u[12 + (k & 3)]), "+v"(p[k & 7].x), "+v"(p[k & 7].y) : : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 65) {      // vcc written by the scalar unit, then one select on it: 2 instructions per count
#define X(k) asm volatile("s_not_b64 vcc, vcc\n\tv_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[k]) : "v"(u[(k + 1) & 15]) : "vcc", "scc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 66) {      // compare into vcc, scalar copy to an SGPR pair, four selects on the copy: 6 instructions per count
#define X(k) asm volatile("v_cmp_gt_i32 vcc, %0, %1\n\ts_mov_b64 %6, vcc\n\tv_cndmask_b32 %2, %0, %1, vcc\n\tv_cndmask_b32 %3, %1, %0, %6\n\tv_cndmask_b32 %4, %4, %5, %6\n\tv_cndmask_b32 %5, %5, %4, %6" : "+v"(u[k & 3]), "+v"(u[4 + (k & 3)]), "+v"(u[8 + (k & 3)]), "+v"(u[12 + (k & 3)]), "+v"(p[k & 7].x), "+v"(p[k & 7].y), "+s"(smaskw) : : "vcc");
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 67) {      // a mixed stream: fma, min, fma, max (fast and slow class alternating): 4 instructions per count
#define X(k) asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_min_f32 %1, %1, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_max_f32 %1, %1, %2" : "+v"(a[k]), "+v"(a[(k + 8) & 15]) : "v"(m), "v"(c));
                PT_REP16(X) PT_REP16(X) PT_REP16(X) PT_REP16(X)
#undef X
            } else if (OP == 54) {
#define X(k) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(sdummy) : "v"(u[k]));
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
    if (s == 123.456f) out[0] = s + (float)sdummy + (float)smaskw;       // never true: keeps the registers live
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

// op: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_max3_f32, 3 v_cvt_f32_ubyte1, 4 v_add_u32, 5 v_fma_f64, 6 v_cndmask_b32, 7 v_pk_mul_f32,
// 8 v_fma_mix_f32 (f16 x f32 + f32), 9 v_cvt_f32_f16, 10 v_perm_b32, 11 v_min_f32, 12 v_cvt_f32_u32, 13 v_ldexp_f32, 14 v_cmp_le_f32, 15 v_bfe_u32,
// 17..67: the table in tools/valu_probe.py;
// op 16 = op 0 with lanes 32..63 masked off; op + 256: any op with those lanes masked off.  waves_per_simd 1..8 (256-thread workgroups, one wave per SIMD each).
PT_API int pt_dbg_valu_rate(int32_t device, int32_t op, int32_t waves_per_simd, int32_t iters, double* wave_insts_per_s, double* clock_ghz)
{
    // op 16 is the original spelling of "v_fma_f32, half-masked"; any op + 256 masks lanes 32..63 off
    const int half = (op == 16 || (op & 256)) ? 1 : 0;
    op = op == 16 ? 0 : (op & 255);
    if (!wave_insts_per_s || op < 0 || op > 67 || waves_per_simd < 1 || waves_per_simd > 8 || iters < 8 || iters > (1 << 22)) {
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
        case 7: e = run<7>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 8: e = run<8>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 9: e = run<9>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 10: e = run<10>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 11: e = run<11>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 12: e = run<12>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 13: e = run<13>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 14: e = run<14>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 15: e = run<15>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 17: e = run<17>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 18: e = run<18>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 19: e = run<19>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 20: e = run<20>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 21: e = run<21>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 22: e = run<22>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 23: e = run<23>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 24: e = run<24>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 25: e = run<25>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 26: e = run<26>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 27: e = run<27>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 28: e = run<28>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 29: e = run<29>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 30: e = run<30>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 31: e = run<31>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 32: e = run<32>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 33: e = run<33>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 34: e = run<34>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 35: e = run<35>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 36: e = run<36>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 37: e = run<37>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 38: e = run<38>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 39: e = run<39>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 40: e = run<40>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 41: e = run<41>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 42: e = run<42>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 43: e = run<43>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 44: e = run<44>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 45: e = run<45>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 46: e = run<46>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 47: e = run<47>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 48: e = run<48>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 49: e = run<49>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 50: e = run<50>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 51: e = run<51>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 52: e = run<52>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 53: e = run<53>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 54: e = run<54>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 55: e = run<55>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 56: e = run<56>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 57: e = run<57>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 58: e = run<58>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 59: e = run<59>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 60: e = run<60>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 61: e = run<61>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 62: e = run<62>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 63: e = run<63>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 64: e = run<64>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 65: e = run<65>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 66: e = run<66>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        case 67: e = run<67>(blocks, iters, half, d_out, d_clk, e0, e1); break;
        default: e = hipErrorInvalidValue; break;
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
