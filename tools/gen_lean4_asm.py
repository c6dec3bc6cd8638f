#!/usr/bin/env python3
"""tools/gen_lean4_asm.py — writes gr-ieee80211_amd/csrc/c8p_lean4_asm.h: the integer-compare block loop of the lean Viterbi forward
pass for FOUR packets per wave (lean4_int<CR>), one inline-asm statement per code rate.

Layout.  A packet owns one row of 16 lanes and four registers per lane: index D = l + 16 r (l = lane & 15, r = register) plays the
role the lane index plays in c8p_viterbi_lean.h — state bit k lives in bit (k + t) mod 6 of D at time t.  Exchange phases q = 0..3
cross a lane bit (DPP inside the row, as before); q = 4, 5 cross a REGISTER bit: both candidates are already in this lane, the step
is four plain VALU instructions (candidates, max, borrow, record) instead of the six issue slots of a permlane swap step.  Per
30 steps at rate 5/6: 128 issue slots per packet instead of 157 (tools/gen_lean_asm.py), because in addition the branch-metric pair of
a step with both LLRs is computed once per wave-step and shared by the four registers.

What differs per register is a compile-time fact: the class of a transition is GF(2)-linear in D, so register r sees class c XOR
delta(q, r).  delta = 3 swaps the keep / send halves of the lane-constant coefficient pair (an op_sel), delta = 1 or 2 swaps only
the t1 or only the t0 coefficient.

LLRs are per-row values: each lane reads its packet's stream with 16-byte vector loads (the 16 lanes of a row hit the same
address), three register sets of 12 used in rotation, one group (10 steps at rate 5/6) ahead.

The step arithmetic is the one of lean_blocks_int (same opcodes, operand order and rounding); the borrow chain runs through the
registers of a step in order (subb r0, addc r0, subb r1, ...), so the accumulators of registers 0..2 start a block as `initx` (bit
31-k = borrow-in of the same step's next register) and register 3 as `inity` (borrow-in of the next step).

Registers: v[20:21] .. v[26:27] = {metric, decisions} of register 0..3, v[28:35] = candidate pairs, v[36:39] = two branch-metric
pairs, v[40:75] = three LLR sets, v76 = unused difference, v[77:80] = q=2 intermediates, v81 / v82 = hint temporaries, v83 = survivor byte
offset, v84 = LLR byte offset, v85 = minimum of the four metrics, v86 = LLR byte offset of the next block (of this one in the last
iteration: the refill then re-reads valid addresses).
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("C8P_GEN_OUT4") or os.path.join(ROOT, "gr-ieee80211_amd", "csrc", "c8p_lean4_asm.h")

BLK = 30
RATES = {"12": 0, "23": 1, "34": 2, "56": 3}
NLB = {"12": 60, "23": 45, "34": 40, "56": 36}
NGRP = {"12": 5, "23": 5, "34": 5, "56": 3}          # groups of 12 / 9 / 8 / 12 LLRs
SET = (40, 52, 64)        # three LLR register sets: with an odd number of groups per block two would collide at the block boundary
M = (20, 22, 24, 26)          # metric of register r (its decisions in M[r] + 1)
C = (28, 30, 32, 34)          # candidate pair of register r: keep, send
TAB = (36, 38)
DJ, TQ, H0, H1, VOFF, LADDR, VMIN, LNEXT = 76, (77, 78, 79, 80), 81, 82, 83, 84, 85, 86
PTR = 56                      # s56 blocks left, s[58:59] saved vcc, s60 scratch, s[62:63] check mask
DPP = {0: "quad_perm:[1,0,3,2]", 1: "quad_perm:[2,3,0,1]", 3: "row_ror:8"}
RM = " row_mask:0xf bank_mask:0xf"


def idx(cr, j):
    if cr == "12":
        return 2 * j, 2 * j + 1
    if cr == "23":
        b, r = divmod(j, 2)
        return (3 * b + (2 if r else 0)), (None if r else 3 * b + 1)
    if cr == "34":
        b, r = divmod(j, 3)
        i0 = None if r == 2 else 4 * b + (2 if r == 1 else 0)
        i1 = 4 * b + 1 if r == 0 else (4 * b + 3 if r == 2 else None)
        return i0, i1
    b, r = divmod(j, 5)
    i0 = {0: 6 * b, 1: 6 * b + 2, 3: 6 * b + 4}.get(r)
    i1 = {0: 6 * b + 1, 2: 6 * b + 3, 4: 6 * b + 5}.get(r)
    return i0, i1


def sv_out(i, b):             # c8p_dev.h sv_out (SV_STATE_OUTPUT, cloud80211phy.cc:1877)
    x0, x1, x2, x3, x5, x6 = b, (i >> 5) & 1, (i >> 4) & 1, (i >> 3) & 1, (i >> 1) & 1, i & 1
    return 2 * (x0 ^ x2 ^ x3 ^ x5 ^ x6) + (x0 ^ x1 ^ x2 ^ x3 ^ x6)


def keep_class(q, D):         # c8p_viterbi_lean.h lean_k_consts: class of the candidate that stays at index D in phase q
    r = (q + 1) % 6
    nst = ((D >> r) | (D << (6 - r))) & 63
    m, b = nst & 31, nst >> 5
    return sv_out(2 * m + 1, 1) if b else sv_out(2 * m, 0)


def delta(q, r):              # class offset of register r (GF(2)-linear in D: checked below)
    d = keep_class(q, r << 4) ^ keep_class(q, 0)
    for l in range(16):
        assert keep_class(q, l | (r << 4)) ^ keep_class(q, l) == d
    return d


def gen_rate(cr):
    n, G = NLB[cr], NGRP[cr]
    steps = [idx(cr, j) for j in range(BLK)]
    bounds = [round(g * BLK / G) for g in range(G + 1)]
    gstart = [min(x for j in range(bounds[g], bounds[g + 1]) for x in steps[j] if x is not None) for g in range(G)] + [n]
    for g in range(G):
        used = [x for j in range(bounds[g], bounds[g + 1]) for x in steps[j] if x is not None]
        assert min(used) >= gstart[g] and max(used) < gstart[g + 1] and gstart[g + 1] - gstart[g] <= 12

    def lreg(i):
        g = max(k for k in range(G) if gstart[k] <= i)
        return SET[g % 3] + (i - gstart[g])

    def lpair(i):
        r = lreg(i)
        return "v[%d:%d]" % (r & ~1, (r & ~1) + 1), r & 1

    L = []
    emit = L.append
    cnt, chk = "s%d" % PTR, "s[%d:%d]" % (PTR + 6, PTR + 7)

    def load_group(g, areg):
        i, hi = gstart[g], gstart[g + 1]
        off = 0
        while i < hi:
            for k in (4, 2, 1):
                if i + k <= hi and (k == 1 or lreg(i) % 2 == 0):
                    sz = {1: "dword", 2: "dwordx2", 4: "dwordx4"}[k]
                    dst = "v%d" % lreg(i) if k == 1 else "v[%d:%d]" % (lreg(i), lreg(i) + k - 1)
                    emit("global_load_%s %s, v%d, %%[llr] offset:%d" % (sz, dst, areg, off + 4 * i))
                    i += k
                    break

    for r in range(4):
        emit("v_mov_b32 v%d, %%[m%d]" % (M[r], r))
        emit("v_mov_b32 v%d, %s" % (M[r] + 1, "%[inity]" if r == 3 else "%[initx]"))
    emit("v_mov_b32 v%d, %%[voff]" % VOFF)
    emit("v_mov_b32 v%d, %%[laddr]" % LADDR)
    emit("s_mov_b32 %s, %%[nblk]" % cnt)
    emit("s_mov_b32 vcc_lo, 0x55555555")
    emit("s_mov_b32 vcc_hi, 0x55555555")
    load_group(0, LADDR)
    emit("1:")
    emit("s_cmp_gt_u32 %s, 1" % cnt)
    emit("s_cselect_b32 s%d, %d, 0" % (PTR + 4, 4 * n))
    emit("v_add_u32 v%d, s%d, v%d" % (LNEXT, PTR + 4, LADDR))
    for j in range(BLK):
        q = j % 6
        i0, i1 = steps[j]
        if j in bounds[:-1]:
            g = bounds.index(j)
            emit("s_waitcnt vmcnt(0)")
            if g + 1 < G:
                load_group(g + 1, LADDR)
            else:
                load_group(0, LNEXT)          # next block's first group
        dl = [delta(q, r) for r in range(4)]
        # ---- candidates
        if i0 is not None and i1 is not None:
            p0, s0 = lpair(i0)
            p1, s1 = lpair(i1)
            tabs = {}
            for kind in sorted(set(0 if d in (0, 3) else 1 for d in dl)):
                t = TAB[len(tabs)]
                tabs[kind] = t
                # kind 0: {ka t0 + kb t1, sa t0 + sb t1};  kind 1: the t1 coefficients swapped {ka t0 + sb t1, sa t0 + kb t1}
                sb = 1 if kind == 1 else 0
                emit("v_pk_mul_f32 v[%d:%d], %%[kb%d], %s op_sel:[%d,%d] op_sel_hi:[%d,%d]" % (t, t + 1, q, p1, sb, s1, 1 - sb, s1))
                emit("v_pk_fma_f32 v[%d:%d], %%[ka%d], %s, v[%d:%d] op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (t, t + 1, q, p0, t, t + 1, s0, s0))
            for r in range(4):
                d = dl[r]
                t = tabs[0 if d in (0, 3) else 1]
                sw = 1 if d in (3, 2) else 0      # delta 3: both coefficients swapped; delta 2 (t0 only) = kind 1 swapped
                emit("v_pk_add_f32 v[%d:%d], v[%d:%d], v[%d:%d] op_sel:[0,%d] op_sel_hi:[0,%d]" % (C[r], C[r] + 1, M[r], M[r] + 1, t, t + 1, sw, 1 - sw))
        else:
            for r in range(4):
                d = dl[r]
                if i0 is not None:
                    p, s, sw, kk = lpair(i0) + ((d >> 1) & 1, "ka")
                else:
                    p, s, sw, kk = lpair(i1) + (d & 1, "kb")
                emit("v_pk_fma_f32 v[%d:%d], %%[%s%d], %s, v[%d:%d] op_sel:[%d,%d,0] op_sel_hi:[%d,%d,0]" % (C[r], C[r] + 1, kk, q, p, M[r], M[r] + 1, sw, s, 1 - sw, s))
        # ---- exchange, decision, record
        acc = lambda r: "v_addc_co_u32_e32 v%d, vcc, v%d, v%d, vcc" % (M[r] + 1, M[r] + 1, M[r] + 1)
        if q in DPP:
            for r in range(4):
                emit("v_max_u32_dpp v%d, v%d, v%d %s%s" % (M[r], C[r] + 1, C[r], DPP[q], RM))
                emit("v_subb_co_u32_dpp v%d, vcc, v%d, v%d, vcc %s%s" % (DJ, C[r] + 1, C[r], DPP[q], RM))
                emit(acc(r))
        elif q == 2:
            for r in range(4):
                emit("v_mov_b32_dpp v%d, v%d quad_perm:[3,2,1,0]%s" % (TQ[r], C[r] + 1, RM))
            for r in range(4):
                emit("v_max_u32_dpp v%d, v%d, v%d row_half_mirror%s" % (M[r], TQ[r], C[r], RM))
                emit("v_subb_co_u32_dpp v%d, vcc, v%d, v%d, vcc row_half_mirror%s" % (DJ, TQ[r], C[r], RM))
                emit(acc(r))
        else:
            bit = 1 if q == 4 else 2
            for a in range(4):
                if a & bit:
                    continue
                b = a | bit
                # a holds the even predecessor: [send_b > keep_a] = borrow(keep_a - send_b); b the odd one: [keep_b > send_a] = borrow(send_a - keep_b)
                emit("v_max_u32 v%d, v%d, v%d" % (M[a], C[a], C[b] + 1))
                emit("v_sub_co_u32_e32 v%d, vcc, v%d, v%d" % (DJ, C[a], C[b] + 1))
                emit(acc(a))
                emit("v_max_u32 v%d, v%d, v%d" % (M[b], C[b], C[a] + 1))
                emit("v_sub_co_u32_e32 v%d, vcc, v%d, v%d" % (DJ, C[a] + 1, C[b]))
                emit(acc(b))
    # ---- block end
    emit("s_sub_u32 %s, %s, 1" % (cnt, cnt))
    for r in range(4):
        emit("v_xor_b32 v%d, %%[flip], v%d" % (M[r] + 1, M[r] + 1))
    # traceback hint (see tools/gen_lean_asm.py): in the last block of a segment mark the index(es) of the row's largest metric in bit 30
    emit("s_add_u32 %[segc], %[segc], 1")
    emit("s_cmp_lt_u32 %[segc], %[segn]")
    emit("s_cbranch_scc1 5f")
    emit("s_mov_b64 s[%d:%d], vcc" % (PTR + 2, PTR + 3))
    emit("v_max_u32 v%d, v%d, v%d" % (H0, M[0], M[1]))
    emit("v_max3_u32 v%d, v%d, v%d, v%d" % (H0, H0, M[2], M[3]))
    for rot in (1, 2, 4, 8):
        emit("s_nop 1")
        emit("v_max_u32_dpp v%d, v%d, v%d row_ror:%d%s" % (H0, H0, H0, rot, RM))
    emit("v_mov_b32 v%d, 0x40000000" % H1)
    for r in range(4):
        emit("v_cmp_eq_u32_e32 vcc, v%d, v%d" % (H0, M[r]))
        emit("v_cndmask_b32_e32 v%d, 0, v%d, vcc" % (DJ, H1))
        emit("v_or_b32 v%d, v%d, v%d" % (M[r] + 1, DJ, M[r] + 1))
    emit("s_mov_b64 vcc, s[%d:%d]" % (PTR + 2, PTR + 3))
    emit("s_mov_b32 %[segc], 0")
    emit("5:")
    for r in range(4):
        emit("global_store_dword v%d, v%d, %%[sv] offset:%d" % (VOFF, M[r] + 1, 64 * r))
    emit("v_add_u32 v%d, 0x100, v%d" % (VOFF, VOFF))
    emit("v_add_u32 v%d, %d, v%d" % (LADDR, 4 * n, LADDR))
    for r in range(4):
        emit("v_mov_b32 v%d, %s" % (M[r] + 1, "%[inity]" if r == 3 else "%[initx]"))
    emit("v_min_u32 v%d, v%d, v%d" % (VMIN, M[0], M[1]))
    emit("v_min3_u32 v%d, v%d, v%d, v%d" % (VMIN, VMIN, M[2], M[3]))
    emit("v_cmp_gt_f32_e64 %s, %%[bound], v%d" % (chk, VMIN))     # a row whose smallest metric is below its bound: leave
    emit("s_cmp_eq_u32 %s, 0" % cnt)
    emit("s_cbranch_scc1 2f")
    emit("s_cmp_eq_u64 %s, 0" % chk)
    emit("s_cbranch_scc1 1b")
    emit("2:")
    emit("s_waitcnt vmcnt(0)")
    for r in range(4):
        emit("v_mov_b32 %%[m%d], v%d" % (r, M[r]))
    emit("v_mov_b32 %%[voff], v%d" % VOFF)
    emit("v_mov_b32 %%[laddr], v%d" % LADDR)
    emit("s_mov_b32 %%[nblk], %s" % cnt)
    body = "\n".join('      "%s\\n\\t"' % x for x in L)
    clob = ['"vcc"', '"scc"', '"memory"'] + ['"s%d"' % (PTR + i) for i in range(8)] + ['"v%d"' % r for r in range(20, 87)]
    ins = ['[ka%d] "v"(K[%d].a)' % (q, q) for q in range(6)] + ['[kb%d] "v"(K[%d].b)' % (q, q) for q in range(6)] + \
          ['[initx] "v"(initx)', '[inity] "v"(inity)', '[flip] "v"(flip)', '[bound] "v"(bound)', '[segn] "s"(segn)', '[llr] "s"(llr)', '[sv] "s"(sv)']
    n_slots = sum(1 for x in L[L.index("1:"):L.index("s_sub_u32 %s, %s, 1" % (cnt, cnt))] if x.startswith("v_"))
    return """
// code rate %s: %d LLRs per block in %d load groups; %d VALU instructions per 30 wave-steps = %.1f per packet (%.2f per packet-step)
template <> __device__ __forceinline__ void lean4_int<%d>(float (&m)[4], const LeanK (&K)[6], unsigned initx, unsigned inity, unsigned flip, float bound,
                                                          int& segc, int segn, unsigned& voff, unsigned& laddr, const float* llr, unsigned* sv, int& nblk)
{
  asm volatile(
%s
      : [m0] "+v"(m[0]), [m1] "+v"(m[1]), [m2] "+v"(m[2]), [m3] "+v"(m[3]), [voff] "+v"(voff), [laddr] "+v"(laddr), [nblk] "+s"(nblk), [segc] "+s"(segc)
      : %s
      : %s);
}
""" % (cr, n, G, n_slots, n_slots / 4.0, n_slots / 120.0, RATES[cr], body, ", ".join(ins), ", ".join(clob)), n_slots


def main():
    parts, counts = [], {}
    for cr in ("12", "23", "34", "56"):
        txt, nv = gen_rate(cr)
        parts.append(txt)
        counts[cr] = nv / 4.0
    hdr = """// c8p_lean4_asm.h — GENERATED by tools/gen_lean4_asm.py (do not edit; edit the generator, which documents the layout and the step).
//
// The integer-compare block loop of the lean Viterbi forward pass for FOUR packets per wave (16 lanes x 4 registers per packet).
// VALU instructions per packet and 30-step block, rate 1/2: %.1f   2/3: %.1f   3/4: %.1f   5/6: %.1f
// In / out: m[4] (metrics of registers 0..3), voff / laddr (this lane's byte offsets into the survivor / LLR arenas), nblk (full
// blocks left; 0 when all were done), segc.  `bound` is a per-row value; K, initx, inity, flip are functions of lane & 15.
#pragma once

template <int CR> __device__ __forceinline__ void lean4_int(float (&m)[4], const LeanK (&K)[6], unsigned initx, unsigned inity, unsigned flip, float bound,
                                                            int& segc, int segn, unsigned& voff, unsigned& laddr, const float* llr, unsigned* sv, int& nblk);
""" % tuple(counts[cr] for cr in ("12", "23", "34", "56"))
    open(OUT, "w").write(hdr + "".join(parts))
    print("wrote", OUT, counts)


if __name__ == "__main__":
    main()
