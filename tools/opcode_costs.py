#!/usr/bin/env python3
"""tools/opcode_costs.py <probe_issue.txt> <out.json> — condense the output of tools/probe_issue.bin into the cost table
bench.py prices the decoder with: shader cycles per wave-instruction per SIMD at 8 waves / SIMD, wall-clock column
(event time x measured shader clock / instructions per SIMD; the per-wave median column under-counts when the waves of a
SIMD do not all run at the same time)."""
import json
import re
import sys

src, dst = sys.argv[1], sys.argv[2]
rows = {}
for line in open(src):
    m = re.match(r"(p_\w+)\s+(.*)", line)
    if not m:
        continue
    cols = re.findall(r"([\d.]+) \|\s+([\d.]+) @([\d.]+)GHz", m.group(2))
    if len(cols) != 4:
        continue
    what = m.group(2).split("GHz")[-1].strip()
    rows[m.group(1)] = {"what": what, "cyc_W1": float(cols[0][1]), "cyc_W2": float(cols[1][1]), "cyc_W4": float(cols[2][1]),
                        "cyc_W8": float(cols[3][1]), "clock_GHz_W8": float(cols[3][2])}
g = lambda k: rows[k]["cyc_W8"]
# the opcodes of the lean Viterbi step (csrc/c8p_viterbi_lean.h), priced at saturation (8 waves / SIMD)
table = {
    "source": src,
    "note": "cycles per wave64 instruction per SIMD, 8 waves / SIMD, independent chains; every VALU opcode of the step costs one "
            "~4.15-cycle issue slot (packed f32, DPP, VOPC, carry ops, any SGPR operand), the permlane swaps two; only plain "
            "VGPR-only v_add/sub/mul/mov/and/or/xor/lshrrev/fma issue at ~2.4 cycles, and only in dense runs (mixed into a "
            "4-cycle stream they fill a whole slot: p_step_sign = p_step_cur)",
    "lean_step_opcodes": {
        "v_pk_fma_f32": g("p_pk_fma_sgpr"), "v_pk_mul_f32": g("p_pk_mul"), "v_pk_add_f32": g("p_pk_add"),
        "v_sub_f32_dpp": g("p_sub_dpp_quad"), "v_max_f32_dpp": g("p_max_dpp_quad"), "v_mov_b32_dpp": g("p_mov_dpp_quad"),
        "v_cmp_class_f32": g("p_cmpclass"), "v_cmp_gt_f32": g("p_cmpgt"), "v_addc_co_u32": g("p_addc"),
        "v_permlane16_swap_b32": g("p_swap16"), "v_permlane32_swap_b32": g("p_swap32"), "v_max_f32": g("p_max"),
        "v_or_b32": g("p_or"), "v_cmpx_class_f32": g("p_cmpclass"), "v_cmpx_gt_f32": g("p_cmpgt"),
        # the integer-compare block loop (round 2 on): borrow / max through DPP, the swap phases' plain max / borrow, the carry shift's xor
        "v_subb_co_u32_dpp": g("p_subb_dpp"), "v_max_u32_dpp": g("p_max_u32_dpp"), "v_max_u32": g("p_max_u32"),
        "v_sub_co_u32": g("p_add_co"), "v_xor_b32": g("p_xor"),
    },
    "whole_step_cycles_W8": {k: g(k) for k in rows if k.startswith("p_step") or k.startswith("p_chain")},
    "all": rows,
}
json.dump(table, open(dst, "w"), indent=1)
print(json.dumps(table["lean_step_opcodes"], indent=1))
