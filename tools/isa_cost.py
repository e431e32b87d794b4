#!/usr/bin/env python
"""Static VALU cost of a kernel from a gfx950 assembly listing, weighted by the issue cost MEASURED on MI355X
(tools/microbench_valu.hip -> profiles/r06_microbench_valu.txt): fp32 fma / mul / add / sub / mov issue in ~2.4-3.0 cycles per wave
instruction and SIMD, and / xor / add_u32 in ~3.2, everything else (conversions, floor, min / max / med3, bit-field and shift forms,
24-bit multiplies, selects, packed-u16, DPP moves, packed fp32) in ~4.2.

    python tools/isa_cost.py /tmp/fused.s 'b5126k_voteILi1ELb0ELb0' [--ops] [--blocks]

--ops: histogram per opcode; --blocks: per basic block (label) totals, to find the straight-line event code."""
import collections
import re
import sys

FAST = {"v_fma_f32": 2.5, "v_fmac_f32": 2.7, "v_mul_f32": 3.0, "v_add_f32": 2.9, "v_sub_f32": 2.85, "v_subrev_f32": 2.85, "v_mov_b32": 2.4,
        "v_and_b32": 3.2, "v_or_b32": 3.2, "v_xor_b32": 3.2, "v_add_u32": 3.2, "v_sub_u32": 3.2, "v_subrev_u32": 3.2, "v_not_b32": 3.2,
        "v_pk_fma_f32": 4.8, "v_pk_mul_f32": 4.4, "v_pk_add_f32": 4.3}
DEFAULT = 4.2


def base_op(op: str) -> str:
    return re.sub(r"_(e32|e64|dpp|sdwa)$", "", op)


def cost(op: str) -> float:
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return DEFAULT
    b = base_op(op)
    if b.startswith(("v_mad_u64", "v_mul_lo_u32", "v_mul_hi", "v_add_f64", "v_mul_f64", "v_fma_f64", "v_cvt_f64", "v_cvt_f32_f64", "v_rcp", "v_rsq", "v_sqrt", "v_exp",
                     "v_log", "v_ldexp_f64", "v_floor_f64", "v_max_f64", "v_min_f64", "v_cmp_lt_f64", "v_cmp_gt_f64")):
        return 16.0  # quarter-rate / fp64 forms (not measured here; only the rare paths hold them)
    return FAST.get(b, DEFAULT)


def main():
    txt = open(sys.argv[1]).read().split("\n")
    pat = re.compile(sys.argv[2])
    want_ops, want_blocks = "--ops" in sys.argv, "--blocks" in sys.argv
    i = 0
    while i < len(txt):
        m = re.match(r"^(_ZN4cmax\S+):\s*; @", txt[i])
        if m and pat.search(m.group(1)):
            j = i + 1
            ops = collections.Counter()
            kinds = collections.Counter()
            blocks = []  # (label, n_valu, cycles, n_other)
            cur = ["entry", 0, 0.0, 0]
            while j < len(txt) and not txt[j].startswith(".Lfunc_end"):
                line = txt[j].strip()
                if line.endswith(":") and not line.startswith(";"):
                    blocks.append(tuple(cur))
                    cur = [line[:-1], 0, 0.0, 0]
                elif line and not line.startswith((".", ";")):
                    op = line.split()[0]
                    if op.startswith("v_"):
                        ops[op] += 1
                        kinds["valu"] += 1
                        cur[1] += 1
                        cur[2] += cost(op)
                    else:
                        kinds["salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other"] += 1
                        cur[3] += 1
                j += 1
            blocks.append(tuple(cur))
            total = sum(cost(o) * n for o, n in ops.items())
            print(m.group(1)[:90])
            print("  ", dict(kinds), " VALU issue cycles (static, loops once): %.0f = %.2f per instruction" % (total, total / max(1, kinds["valu"])))
            if want_ops:
                for o, n in sorted(ops.items(), key=lambda kv: -cost(kv[0]) * kv[1]):
                    print("     %-28s %4d x %4.1f = %6.0f" % (o, n, cost(o), n * cost(o)))
            if want_blocks:
                for lab, nv, cyc, no in blocks:
                    if nv + no >= 12:
                        print("     %-14s valu %4d  cycles %6.0f  other %3d" % (lab, nv, cyc, no))
            i = j
        i += 1


if __name__ == "__main__":
    main()
