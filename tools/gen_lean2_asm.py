#!/usr/bin/env python3
"""tools/gen_lean2_asm.py — writes gr-ieee80211_amd/csrc/c8p_lean2_asm.h: the integer-compare block loop of the lean Viterbi forward
pass for TWO packets per wave, LLRs from LDS (lean2_lds<CR>), one inline-asm statement per code rate.

Layout.  A packet owns 32 lanes (two DPP rows) and two registers per lane: index D = l + 32 r (l = lane & 31, r = register) plays the
role the lane index plays in c8p_viterbi_lean.h — state bit k lives in bit (k + t) mod 6 of D at time t.  Exchange phases q = 0..3
cross a lane bit inside a DPP row, q = 4 crosses the row bit (v_permlane16_swap, which swaps rows 0 <-> 1 and 2 <-> 3, i.e. serves
both packets at once), q = 5 crosses the REGISTER bit: both candidates are already in the lane, four plain VALU instructions per
register pair instead of the six issue slots of the permlane32 swap step of the one-packet layout.  The branch-metric pair of a
step with both LLRs is computed once per wave-step and shared by the two registers (tools/gen_lean4_asm.py has the same idea with
16 lanes x 4 registers; this is the half-way layout that still gives a 4096-packet batch two waves per SIMD).

What differs per register is a compile-time fact: the class of a transition is GF(2)-linear in D, so register 1 sees class
c XOR delta(q).  delta = 3 swaps the keep / send halves of the lane-constant coefficient pair (an op_sel), delta = 1 or 2 swaps only
the t1 or only the t0 coefficient.

LLRs: every lane reads ITS packet's LDS buffer (two addresses per instruction, a broadcast inside each half of the wave) with
ds_read_b64 into rotating register pairs, scheduled exactly as in lean_blocks_lds (tools/gen_lean_asm.py, gen_rate_lds): the read of
group g + NSETS right after the last reader of group g, exact lgkmcnt waits.

The step arithmetic is the one of lean_blocks_int (same opcodes, operand order and rounding).  The borrow chain runs through the
registers of a step in order (subb r0, addc r0, subb r1, addc r1): register 0's decisions start a block as `initx` (bit 31-k =
borrow-in of register 1 in the same step k), register 1's as `inity` (borrow-in of register 0 in step k + 1).

Registers: v[60:61] / v[62:63] = {metric, decisions} of register 0 / 1, v[52:55] = candidate pairs, v[56:59] = two branch-metric
pairs, v51 = unused difference, v[49:50] = q=2 intermediates, v48 = survivor byte offset, v47 = LDS byte address of the row's
current block, v46 = minimum of the two metrics, v45 / v44 = hint temporaries, below that the LLR register sets.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from gen_lean4_asm import idx, keep_class          # noqa: E402  (the same puncture patterns and transition classes)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("C8P_GEN_OUT2") or os.path.join(ROOT, "gr-ieee80211_amd", "csrc", "c8p_lean2_asm.h")

BLK = 30
RATES = {"12": 0, "23": 1, "34": 2, "56": 3}
NLB = {"12": 60, "23": 45, "34": 40, "56": 36}
LDS_GS = {"12": 2, "23": 3, "34": 2, "56": 2}
LDS_NSETS = {"12": 3, "23": 3, "34": 4, "56": 3}
M = (60, 62)                  # metric of register r (its decisions in M[r] + 1)
C = (52, 54)                  # candidate pair of register r: keep, send
TAB = (56, 58)
DJ, TQ, VOFF, LADDR, VMIN, H0, H1 = 51, (49, 50), 48, 47, 46, 45, 44
PTR = 56                      # s56 blocks left, s[58:59] saved vcc, s[62:63] check mask
DPP = {0: "quad_perm:[1,0,3,2]", 1: "quad_perm:[2,3,0,1]", 3: "row_ror:8"}
RM = " row_mask:0xf bank_mask:0xf"


def delta(q):                 # class offset of register 1 (GF(2)-linear in D: checked)
    d = keep_class(q, 32) ^ keep_class(q, 0)
    for l in range(32):
        assert keep_class(q, l | 32) ^ keep_class(q, l) == d
    return d


def gen_rate(cr):
    n = NLB[cr]
    GS, NS = LDS_GS[cr], LDS_NSETS[cr]
    SW = 2 if GS == 2 else 4
    G = n // GS
    assert n % GS == 0 and G % NS == 0
    VB = 44 - SW * NS            # first register of the LLR sets (even), below the hint temporaries
    steps = [idx(cr, j) for j in range(BLK)]

    def grp(i):
        return i // GS

    def lreg(i):
        return VB + SW * (grp(i) % NS) + (i % GS)

    def lpair(i):
        r = lreg(i)
        return "v[%d:%d]" % (r & ~1, (r & ~1) + 1), r & 1

    def load_group(g, wrapped):
        off = 4 * GS * g + (4 * n if wrapped else 0)
        r = VB + SW * (g % NS)
        if GS == 2:
            return ["ds_read_b64 v[%d:%d], v%d offset:%d" % (r, r + 1, LADDR, off)]
        o0 = off // 4
        assert o0 + 1 < 256
        return ["ds_read2_b32 v[%d:%d], v%d offset0:%d offset1:%d" % (r, r + 1, LADDR, o0, o0 + 1),
                "ds_read_b32 v%d, v%d offset:%d" % (r + 2, LADDR, off + 8)]

    body = []

    def emit(x, reads=()):
        body.append((x, tuple(reads)))
    acc = lambda r: "v_addc_co_u32_e32 v%d, vcc, v%d, v%d, vcc" % (M[r] + 1, M[r] + 1, M[r] + 1)
    for j in range(BLK):
        q = j % 6
        i0, i1 = steps[j]
        dl = (0, delta(q))
        # ---- candidates of both registers
        if i0 is not None and i1 is not None:
            p0, s0 = lpair(i0)
            p1, s1 = lpair(i1)
            tabs = {}
            for kind in sorted(set(0 if d in (0, 3) else 1 for d in dl)):
                t = TAB[len(tabs)]
                tabs[kind] = t
                # kind 0: {ka t0 + kb t1, sa t0 + sb t1};  kind 1: the t1 coefficients swapped {ka t0 + sb t1, sa t0 + kb t1}
                sb = 1 if kind == 1 else 0
                emit("v_pk_mul_f32 v[%d:%d], %%[kb%d], %s op_sel:[%d,%d] op_sel_hi:[%d,%d]" % (t, t + 1, q, p1, sb, s1, 1 - sb, s1), [grp(i1)])
                emit("v_pk_fma_f32 v[%d:%d], %%[ka%d], %s, v[%d:%d] op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (t, t + 1, q, p0, t, t + 1, s0, s0), [grp(i0)])
            for r in range(2):
                d = dl[r]
                t = tabs[0 if d in (0, 3) else 1]
                sw = 1 if d in (3, 2) else 0
                emit("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[0,%d] op_sel_hi:[0,%d]" % (C[r], C[r] + 1, M[r], M[r] + 1, t, t + 1, sw, 1 - sw))
        else:
            for r in range(2):
                d = dl[r]
                if i0 is not None:
                    (p, s), sw, kk, ii = lpair(i0), (d >> 1) & 1, "ka", i0
                else:
                    (p, s), sw, kk, ii = lpair(i1), d & 1, "kb", i1
                emit("v_pk_fma_f32 v[%d:%d], %%[%s%d], %s, v[%d:%d] op_sel:[%d,%d,0] op_sel_hi:[%d,%d,0]" % (C[r], C[r] + 1, kk, q, p, M[r], M[r] + 1, sw, s, 1 - sw, s), [grp(ii)])
        # ---- exchange, decision, record (a DPP / permlane read needs two wait states after the VALU write of its source: register 1's
        #      candidates are one instruction away from register 0's exchange)
        if q in DPP:
            emit("s_nop 0")
            for r in range(2):
                emit("v_max_u32_dpp v%d, v%d, v%d %s%s" % (M[r], C[r] + 1, C[r], DPP[q], RM))
                emit("v_subb_co_u32_dpp v%d, vcc, v%d, v%d, vcc %s%s" % (DJ, C[r] + 1, C[r], DPP[q], RM))
                emit(acc(r))
        elif q == 2:
            emit("s_nop 0")
            for r in range(2):
                emit("v_mov_b32_dpp v%d, v%d quad_perm:[3,2,1,0]%s" % (TQ[r], C[r] + 1, RM))
            emit("s_nop 0")
            for r in range(2):
                emit("v_max_u32_dpp v%d, v%d, v%d row_half_mirror%s" % (M[r], TQ[r], C[r], RM))
                emit("v_subb_co_u32_dpp v%d, vcc, v%d, v%d, vcc row_half_mirror%s" % (DJ, TQ[r], C[r], RM))
                emit(acc(r))
        elif q == 4:
            emit("s_nop 1")
            for r in range(2):
                emit("v_permlane16_swap_b32 v%d, v%d" % (C[r], C[r] + 1))
            for r in range(2):
                emit("v_max_u32 v%d, v%d, v%d" % (M[r], C[r], C[r] + 1))
                emit("v_sub_co_u32_e32 v%d, vcc, v%d, v%d" % (DJ, C[r], C[r] + 1))
                emit(acc(r))
        else:
            a, b = 0, 1
            # a holds the even predecessor: [send_b > keep_a] = borrow(keep_a - send_b); b the odd one: [keep_b > send_a] = borrow(send_a - keep_b)
            emit("v_max_u32 v%d, v%d, v%d" % (M[a], C[a], C[b] + 1))
            emit("v_sub_co_u32_e32 v%d, vcc, v%d, v%d" % (DJ, C[a], C[b] + 1))
            emit(acc(a))
            emit("v_max_u32 v%d, v%d, v%d" % (M[b], C[b], C[a] + 1))
            emit("v_sub_co_u32_e32 v%d, vcc, v%d, v%d" % (DJ, C[a] + 1, C[b]))
            emit(acc(b))
    # ---- loads after the last reader of a group, exact waits before the first reader (as gen_rate_lds)
    first, last = {}, {}
    for k, (x, rd) in enumerate(body):
        for g in rd:
            first.setdefault(g, k)
            last[g] = k
    assert sorted(first) == list(range(G)) and all(last[g] <= last[g + 1] and first[g] <= first[g + 1] for g in range(G - 1))
    assert all(last[g] < first[g + NS] for g in range(G - NS))
    after = {}
    for g in range(G):
        after.setdefault(last[g], []).append(g)
    stream = []
    for k, (x, rd) in enumerate(body):
        for g in sorted(set(rd)):
            if first[g] == k:
                stream.append(("wait", g))
        stream.append(("x", x))
        for g in sorted(after.get(k, [])):
            tgt = g + NS
            for ins in load_group(tgt % G, tgt >= G):
                stream.append(("ds", ins, tgt))
    issued = []
    for g in range(NS):
        issued += [g] * len(load_group(g, False))
    waits = []
    for blk in range(2):
        w = []
        for it in stream:
            if it[0] == "ds":
                issued.append(blk * G + it[2])
            elif it[0] == "wait":
                gid = blk * G + it[1]
                pos = max(k for k, v in enumerate(issued) if v == gid)
                w.append(len(issued) - 1 - pos)
        waits.append(w)
    assert waits[0] == waits[1] and max(waits[0]) <= 15, waits
    L = []
    add = L.append
    cnt, chk = "s%d" % PTR, "s[%d:%d]" % (PTR + 6, PTR + 7)
    add("s_waitcnt lgkmcnt(0)")
    add("v_mov_b32 v%d, %%[initx]" % (M[0] + 1))
    add("v_mov_b32 v%d, %%[inity]" % (M[1] + 1))
    add("s_mov_b32 %s, %%[nblk]" % cnt)
    add("s_mov_b32 vcc_lo, 0x55555555")
    add("s_mov_b32 vcc_hi, 0x55555555")
    for g in range(NS):
        L.extend(load_group(g, False))
    add("1:")
    wi = iter(waits[0])
    for it in stream:
        add("s_waitcnt lgkmcnt(%d)" % next(wi) if it[0] == "wait" else it[1])
    n_slots = sum(1 for x in L[L.index("1:"):] if x.startswith("v_"))
    # ---- block end
    add("s_sub_u32 %s, %s, 1" % (cnt, cnt))
    add("v_add_u32 v%d, %d, v%d" % (LADDR, 4 * n, LADDR))
    for r in range(2):
        add("v_xor_b32 v%d, %%[flip], v%d" % (M[r] + 1, M[r] + 1))
    # traceback hint (tools/gen_lean_asm.py): in the last block of a segment mark the index(es) of the packet's largest metric in bit 30
    add("s_add_u32 %[segc], %[segc], 1")
    add("s_cmp_lt_u32 %[segc], %[segn]")
    add("s_cbranch_scc1 5f")
    add("s_mov_b64 s[%d:%d], vcc" % (PTR + 2, PTR + 3))
    add("v_max_u32 v%d, v%d, v%d" % (H0, M[0], M[1]))
    for rot in (1, 2, 4, 8):
        add("s_nop 1")
        add("v_max_u32_dpp v%d, v%d, v%d row_ror:%d%s" % (H0, H0, H0, rot, RM))
    add("v_mov_b32 v%d, v%d" % (H1, H0))                       # the other row of the packet: rows 0 <-> 1, 2 <-> 3
    add("s_nop 1")
    add("v_permlane16_swap_b32 v%d, v%d" % (H0, H1))
    add("v_max_u32 v%d, v%d, v%d" % (H0, H0, H1))
    add("v_mov_b32 v%d, 0x40000000" % H1)
    for r in range(2):
        add("v_cmp_eq_u32_e32 vcc, v%d, v%d" % (H0, M[r]))
        add("v_cndmask_b32_e32 v%d, 0, v%d, vcc" % (DJ, H1))
        add("v_or_b32 v%d, v%d, v%d" % (M[r] + 1, DJ, M[r] + 1))
    add("s_mov_b64 vcc, s[%d:%d]" % (PTR + 2, PTR + 3))
    add("s_mov_b32 %[segc], 0")
    add("5:")
    for r in range(2):
        add("global_store_dword v%d, v%d, %%[sv] offset:%d" % (VOFF, M[r] + 1, 128 * r))
    add("v_add_u32 v%d, 0x100, v%d" % (VOFF, VOFF))
    add("v_mov_b32 v%d, %%[initx]" % (M[0] + 1))
    add("v_mov_b32 v%d, %%[inity]" % (M[1] + 1))
    add("v_min_u32 v%d, v%d, v%d" % (VMIN, M[0], M[1]))
    add("v_cmp_gt_f32_e64 %s, %%[bound], v%d" % (chk, VMIN))     # a packet whose smallest metric is below its bound: leave
    add("s_cmp_eq_u32 %s, 0" % cnt)
    add("s_cbranch_scc1 2f")
    add("s_cmp_eq_u64 %s, 0" % chk)
    add("s_cbranch_scc1 1b")
    add("2:")
    add("s_waitcnt lgkmcnt(0)")
    add("s_mov_b32 %%[nblk], %s" % cnt)
    text = "\n".join('      "%s\\n\\t"' % x for x in L)
    fixed = (M[0], M[1], VOFF, LADDR)
    clob = ['"vcc"', '"scc"', '"memory"'] + ['"s%d"' % (PTR + i) for i in range(8)] + ['"v%d"' % r for r in range(VB, 64) if r not in fixed]
    ins = ['[ka%d] "v"(K[%d].a)' % (q, q) for q in range(6)] + ['[kb%d] "v"(K[%d].b)' % (q, q) for q in range(6)] + \
          ['[initx] "v"(initx)', '[inity] "v"(inity)', '[flip] "v"(flip)', '[bound] "v"(bound)', '[segn] "s"(segn)', '[sv] "s"(sv)']
    return """
// code rate %s: %d LLRs per block in %d groups of %d, %d register sets (v%d..v43); %d VALU instructions per 30 wave-steps = %.1f per packet
template <> __device__ __forceinline__ void lean2_lds<%d>(float& m0, float& m1, const LeanK (&K)[6], unsigned initx, unsigned inity, unsigned flip, float bound,
                                                          int& segc, int segn, unsigned& voff, unsigned& laddr, unsigned* sv, int& nblk)
{
  unsigned a0, a1;
  asm volatile(
%s
      : "+{v%d}"(m0), "+{v%d}"(m1), "=&{v%d}"(a0), "=&{v%d}"(a1), "+{v%d}"(voff), "+{v%d}"(laddr), [nblk] "+s"(nblk), [segc] "+s"(segc)
      : %s
      : %s);
}
""" % (cr, n, G, GS, NS, VB, n_slots, n_slots / 2.0, RATES[cr], text, M[0], M[1], M[0] + 1, M[1] + 1, VOFF, LADDR, ", ".join(ins), ", ".join(clob)), n_slots


def main():
    parts, counts = [], {}
    for cr in ("12", "23", "34", "56"):
        txt, nv = gen_rate(cr)
        parts.append(txt)
        counts[cr] = nv / 2.0
    hdr = """// c8p_lean2_asm.h — GENERATED by tools/gen_lean2_asm.py (do not edit; edit the generator, which documents the layout and the step).
//
// The integer-compare block loop of the lean Viterbi forward pass for TWO packets per wave (32 lanes x 2 registers per packet), LLRs
// from the packets' LDS buffers.  VALU instructions per packet and 30-step block, rate 1/2: %.1f   2/3: %.1f   3/4: %.1f   5/6: %.1f
// In / out: m0, m1 (metrics of registers 0, 1), voff (this lane's byte offset into the survivor arena), laddr (LDS byte address of
// the LLRs of its packet's current block), nblk (full blocks left; 0 when all were done), segc.  `bound` is a per-packet value;
// K, initx, inity, flip are functions of lane & 31 (lean2_k_consts, lean4_int_consts).
#pragma once

template <int CR> __device__ __forceinline__ void lean2_lds(float& m0, float& m1, const LeanK (&K)[6], unsigned initx, unsigned inity, unsigned flip, float bound,
                                                            int& segc, int segn, unsigned& voff, unsigned& laddr, unsigned* sv, int& nblk);
""" % tuple(counts[cr] for cr in ("12", "23", "34", "56"))
    open(OUT, "w").write(hdr + "".join(parts))
    print("wrote", OUT, counts)


if __name__ == "__main__":
    main()
