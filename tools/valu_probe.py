#!/usr/bin/env python3
"""Measured VALU issue rates of this GPU (pt_dbg_valu_rate) -> JSON on stdout.  Run on the GPU box."""
import json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pathtrace-on-cuda_amd"))
import ptamd
names = ["v_fma_f32", "v_pk_fma_f32", "v_max3_f32", "v_cvt_f32_ubyte1", "v_add_u32", "v_fma_f64", "v_cndmask_b32", "v_pk_mul_f32"]
out = {}
for op, nm in enumerate(names):
    for w in (1, 2, 4, 8):
        r, g = ptamd.valu_rate(op, w, 20000)
        out["%s@%dw" % (nm, w)] = {"wave_insts_per_s": r, "clock_ghz": g, "cycles_per_inst_per_simd": 1024.0 * g * 1e9 / r}
r, g = ptamd.valu_rate(16, 4, 20000)
out["v_fma_f32@4w,half-masked"] = {"wave_insts_per_s": r, "clock_ghz": g, "cycles_per_inst_per_simd": 1024.0 * g * 1e9 / r}
print(json.dumps(out, indent=1))
