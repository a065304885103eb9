#!/usr/bin/env python3
"""Measured VALU issue rates of this GPU (pt_dbg_valu_rate) -> JSON on stdout.  Run on the GPU box."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pathtrace-on-cuda_amd"))
import ptamd
names = {0: "v_fma_f32", 1: "v_pk_fma_f32", 2: "v_max3_f32", 3: "v_cvt_f32_ubyte1", 4: "v_add_u32", 5: "v_fma_f64", 6: "v_cndmask_b32 (vcc, asm clobber)", 7: "v_pk_mul_f32",
         8: "v_fma_mix_f32", 9: "v_cvt_f32_f16", 10: "v_perm_b32", 11: "v_min_f32", 12: "v_cvt_f32_u32", 13: "v_ldexp_f32", 14: "v_cmp_le_f32", 15: "v_bfe_u32",
         17: "v_mov_b32", 18: "v_cndmask_b32 (sgpr pair)", 19: "v_and_b32", 20: "v_or_b32", 21: "v_lshlrev_b32", 22: "v_lshl_add_u32", 23: "v_mul_f32", 24: "v_add_f32",
         25: "v_max_i32", 26: "v_mul_lo_u32", 27: "v_rcp_f32", 28: "v_and_or_b32", 29: "v_cmp_gt_i32", 30: "v_div_fixup_f32", 31: "v_div_fmas_f32", 32: "v_div_scale_f32",
         33: "v_xor_b32", 34: "v_sub_f32", 35: "v_fmac_f32", 36: "v_add3_u32", 37: "v_mad_u32_u24", 38: "v_max_f32", 39: "v_cvt_f32_ubyte0", 40: "v_bfi_b32",
         41: "v_alignbit_b32", 42: "v_lshl_or_b32", 43: "v_sqrt_f32", 44: "v_med3_f32", 45: "v_min_u32", 46: "v_cmp_eq_u32", 47: "v_ashrrev_i32", 48: "v_sub_u32",
         49: "v_fma_f32 (three VGPR sources)", 50: "v_cndmask_b32 (vcc)", 51: "v_mul_u32_u24", 52: "v_lshlrev_b64", 53: "v_lshl_add_u64", 54: "v_readlane_b32",
         55: "pair: v_cmp_gt_i32 vcc + v_cndmask_b32 vcc", 56: "pair: v_cmp_gt_i32 sgpr + v_cndmask_b32 sgpr", 57: "v_cndmask_b32 (vcc, set by one v_cmp per 64)",
         58: "quad: v_cmp vcc, 2 x v_fma_f32, v_cndmask vcc",
         59: "five: v_cmp vcc + 4 x v_cndmask vcc", 60: "five: v_cmp sgpr + 4 x v_cndmask sgpr", 61: "three: v_cmp vcc + 2 x v_cndmask vcc", 62: "two: v_min_i32 + v_max_i32",
         63: "v_cndmask_b32_e64 (vcc, not written in the loop)", 64: "five: v_cmp vcc + 4 x v_cndmask_b32_e64 vcc", 65: "two: s_not_b64 vcc + v_cndmask vcc",
         66: "six: v_cmp vcc, s_mov_b64 copy, v_cndmask vcc, 3 x v_cndmask copy", 67: "four: fma, min, fma, max"}
out = {}
for op, nm in names.items():
    for w in ((1, 2, 4, 8) if op < 8 else (4,)):
        r, g = ptamd.valu_rate(op, w, 20000)
        out["%s@%dw" % (nm, w)] = {"wave_insts_per_s": r, "clock_ghz": g, "cycles_per_inst_per_simd": 1024.0 * g * 1e9 / r}
r, g = ptamd.valu_rate(16, 4, 20000)
out["v_fma_f32@4w,half-masked"] = {"wave_insts_per_s": r, "clock_ghz": g, "cycles_per_inst_per_simd": 1024.0 * g * 1e9 / r}
print(json.dumps(out, indent=1))
