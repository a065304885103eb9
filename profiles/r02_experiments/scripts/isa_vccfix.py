#!/usr/bin/env python3
"""ISA pass between the compiler and the assembler (tools/hipcc_isa.sh): selects that RE-read vcc get the VOP3 encoding.

Measured on MI355X (tools/valu_probe.py ops 50, 55-66; profiles/r02_valu_probe.json): `v_cndmask_b32_e32 d, a, b, vcc` issues
in ~2 clocks when it is the first vector read of vcc after a vector compare wrote it, and blocks the SIMD for ~23 clocks
otherwise (a second select on the same compare; vcc written by the scalar unit; vcc from another basic block).  The same
instruction in the VOP3 encoding, `v_cndmask_b32_e64 d, a, b, vcc`, costs ~4.3 clocks in every case.  The compiler shrinks
every vcc select to the 32-bit encoding, so this pass widens the ones that are not first reads.  Same opcode, same operands,
same result bits: only the encoding changes (4 -> 8 bytes; branch targets are labels, so the assembler re-lays the code).

Left alone: first reads; selects with a 32-bit literal operand (VOP3 has no literal slot on gfx9); SDWA / DPP forms.
usage: isa_vccfix.py in.s out.s     (prints the number of widened selects per kernel)"""
import re, sys

def is_literal(tok):
    tok = tok.strip()
    if re.fullmatch(r"v\d+|s\d+|vcc_lo|vcc_hi|m0|exec_lo|exec_hi|-?\d+|-?\d+\.\d+|0x[0-9a-fA-F]+", tok):
        if tok.startswith("0x"):
            v = int(tok, 16)
            return not (v <= 64 or v >= 0xfffffff0)      # inline integers -16..64
        if re.fullmatch(r"-?\d+", tok):
            return not (-16 <= int(tok) <= 64)
        if re.fullmatch(r"-?\d+\.\d+", tok):
            return float(tok) not in (0.0, 0.5, -0.5, 1.0, -1.0, 2.0, -2.0, 4.0, -4.0)
        return False
    return True      # anything unusual (symbols, modifiers): leave the instruction alone

def run(inp, outp):
    lines = open(inp).read().split("\n")
    out = []; fresh = False; kernel = None; counts = {}
    for l in lines:
        s = l.strip()
        m = re.match(r"^(_Z\S*|[A-Za-z_]\w*):\s*(;.*)?$", s)
        if m and not s.startswith(".L"):
            kernel = m.group(1); fresh = False
        elif re.match(r"^\.L\S*:", s):
            fresh = False                                     # vcc from another block: not a known-first read
        elif s and not s.startswith(";") and not s.startswith("."):
            code = s.split(";")[0].rstrip()
            op = code.split()[0]
            args = code[len(op):]
            if op == "v_cndmask_b32_e32" and re.search(r",\s*vcc\s*$", args):
                if fresh:
                    fresh = False
                else:
                    a = [t.strip() for t in args.split(",")]
                    if len(a) == 4 and not is_literal(a[1]) and not is_literal(a[2]):
                        l = l.replace("v_cndmask_b32_e32", "v_cndmask_b32_e64", 1)
                        counts[kernel] = counts.get(kernel, 0) + 1
            else:
                # other vector readers of vcc take the first-read slot as well
                if op.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co", "v_div_fmas")) or (op.startswith("v_") and not op.startswith("v_cmp") and re.search(r",\s*vcc\s*$", args)):
                    fresh = False
                # writers
                if op.startswith("v_cmp") and (op.endswith("_e32") or re.match(r"^\s*vcc\s*,", args)): fresh = True
                elif op.startswith("v_") and re.match(r"^\s*v\[?\S+,\s*vcc\s*,", args): fresh = True      # v_add_co_u32, v_div_scale ...
                elif op.startswith("s_") and re.match(r"^\s*vcc(_lo|_hi)?\s*,", args): fresh = False
                elif op.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")): fresh = False
        out.append(l)
    open(outp, "w").write("\n".join(out))
    for k, v in counts.items(): print("  isa_vccfix: %4d selects widened in %s" % (v, k[:90]))

if __name__ == "__main__":
    run(sys.argv[1], sys.argv[2])
