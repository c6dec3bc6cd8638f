#!/usr/bin/env python3
"""tools/gen_lean_asm.py — writes gr-ieee80211_amd/csrc/c8p_lean_asm.h: the full-block loop of the lean Viterbi forward pass
(c8p_viterbi_lean.h) as ONE inline-asm statement per code rate, in four flavours (the two LDS-fed ones are described further down,
at gen_rate_lds: lean_blocks_lds<CR> = lean_blocks_int with the LLRs as VGPR operands read from LDS, lean_block_flt_lds<CR> = one
block of the float step, same feed; k_rx_1ss_lds uses them).

  lean_blocks_int<CR>   the production fast path.  When every path metric of a block is known to stay POSITIVE, the unsigned
                        order of the float bit patterns is the float order, and the compare can be an integer subtract with
                        borrow.  That buys what the float domain cannot: a per-lane carry-IN, i.e. a per-lane tie rule, in the
                        same instruction that reads the partner lane through DPP.  A DPP step is then 4 VALU instructions
                        (candidates, borrow, record, max) instead of 5 (candidates, difference, class test, record, max).
  lean_blocks_asm<CR>   the float step of c8p_viterbi_lean.h, instruction for instruction (-DC8P_LEAN_ASM, experiments only).

Why one asm statement: the borrow chain lives in VCC across every step of the loop, and the compiler may use VCC between two
asm statements; DPP operands inside inline asm are also invisible to its hazard recogniser.

The step (both flavours share the candidates):
    cand  = {keep, send} = P + A(q) * t0 (+ B(q) * t1)          v_pk_fma_f32 (+ v_pk_mul_f32, v_pk_add_f32 when both LLRs exist)
  float:
    q <= 3:  d = recv - keep, pre = max(recv, keep)              v_sub_f32_dpp / v_max_f32_dpp   (q == 2: one v_mov_b32_dpp more)
    q >= 4:  swap, pre = max(a0, a1)                             v_permlane{16,32}_swap_b32, v_max_f32
    decision of the PREVIOUS step, in the two wait states a DPP read needs after the write of its source:
             v_cmp_class_f32 (d, per-lane class mask) or v_cmp_gt_f32 (a1, a0) -> vcc -> v_addc_co_u32 acc, acc, acc
  integer (all candidates > 0):
    q <= 3:  pre = max_u32(recv, keep)                           v_max_u32_dpp
    q >= 4:  swap, pre = max_u32(a0, a1)
    decision of the PREVIOUS step:
             q <= 3:  vcc = borrow(recv - keep - cin)            v_subb_co_u32_dpp      cin = 1 on lanes holding the EVEN predecessor
             q >= 4:  vcc = borrow(a0 - a1)                      v_sub_co_u32
             acc = 2 acc + vcc, vcc = old acc[31]                v_addc_co_u32
      The reference takes the odd predecessor only on strict '>' (decode_impl.cc:231-259).  A lane holding the odd predecessor
      (lane bit q = 1: keep = odd, recv = even) wants [keep > recv] = borrow(recv - keep).  A lane holding the even one wants
      [recv > keep] = NOT [recv <= keep] = NOT borrow(recv - keep - 1).  So with cin = 1 on the even-holders the borrow is the
      decision on odd-holders and its complement on even-holders, exact on ties; the complement is undone with one XOR per
      block (lane constant `flip`).  The carry-in costs nothing: acc starts every block as a lane constant `init` whose bit
      31 - k is the cin of step k + 1, and v_addc's carry-OUT (the bit shifted out of acc) is the next v_subb's carry-IN.
      Before every block all lanes are checked against `bound` (c8p_viterbi_lean.h); the loop leaves at the first block whose
      start metrics do not guarantee positive candidates, and the caller runs that block with the float step.

Registers: v62 = pre (read as the pair v[62:63] with both halves taken from v62), v63 = acc, v[56:57] / v[58:59] = candidate
pairs (alternating), v[54:55] / v[60:61] = branch-metric pairs, v52 / v53 = d (alternating; integer: v52 = unused difference),
v51 = q=2 intermediate, v50 = byte offset of this lane's survivor dword.  SGPRs: two sets of LLR registers filled alternately
(one load group in flight at a time), s[56:57] next block's LLRs, s[58:59] this block's, s60 blocks left, s[62:63] check mask.
"""
import os

EXP = os.environ.get("C8P_GEN_EXP", "")      # timing experiments only (results are wrong): nowait, nostore, nocheck

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.environ.get("C8P_GEN_OUT") or os.path.join(ROOT, "gr-ieee80211_amd", "csrc", "c8p_lean_asm.h")

BLK = 30
RATES = {"12": 0, "23": 1, "34": 2, "56": 3}   # name: C8P_CR_* value


def idx(cr, j):
    """positions of (t0, t1) of step j inside the block's LLRs, None = punctured (c8p_viterbi_lean.h idx0 / idx1 per period)"""
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


NLB = {"12": 60, "23": 45, "34": 40, "56": 36}
NGRP = {"12": 4, "23": 4, "34": 2, "56": 2}     # load groups per block (two register sets, used alternately)
SETBASE = (12, 36)    # first SGPR of the two LLR register sets (multiples of 4; s32 / s33 and the top of the file are reserved)
SETSZ = 20            # SGPRs per register set
PTR = 56              # s[56:57] next block, s[58:59] this block, s60 blocks left, s[62:63] positivity check
DPP = {0: "quad_perm:[1,0,3,2]", 1: "quad_perm:[2,3,0,1]", 3: "row_ror:8"}
RM = " row_mask:0xf bank_mask:0xf"
ADDC = "v_addc_co_u32_e32 v63, vcc, v63, v63, vcc"


def gen_rate(cr, integer):
    n, G = NLB[cr], NGRP[cr]
    steps = [idx(cr, j) for j in range(BLK)]
    # group g = the LLRs first used by steps [bounds[g], bounds[g+1]); consecutive LLR ranges
    bounds = [round(g * BLK / G) for g in range(G + 1)]
    gstart = []
    for g in range(G):
        gstart.append(min(x for j in range(bounds[g], bounds[g + 1]) for x in steps[j] if x is not None))
    gstart.append(n)
    for g in range(G):
        used = [x for j in range(bounds[g], bounds[g + 1]) for x in steps[j] if x is not None]
        assert min(used) >= gstart[g] and max(used) < gstart[g + 1] and gstart[g + 1] - gstart[g] <= SETSZ

    def sreg(i):            # SGPR of LLR i of the block
        g = max(k for k in range(G) if gstart[k] <= i)
        return SETBASE[g % 2] + (i - gstart[g])

    def sgpr_pair(i):
        r = sreg(i)
        return "s[%d:%d]" % (r & ~1, (r & ~1) + 1), r & 1

    L = []
    emit = L.append
    cur, nxt, cnt, chk = "s[%d:%d]" % (PTR + 2, PTR + 3), "s[%d:%d]" % (PTR, PTR + 1), "s%d" % (PTR + 4), "s[%d:%d]" % (PTR + 6, PTR + 7)

    def load_group(g, ptr):
        """s_load instructions of group g: sizes 16/8/4/2/1, an N-dword destination starts at a multiple of min(N, 4)"""
        i, hi = gstart[g], gstart[g + 1]
        while i < hi:
            for k in (16, 8, 4, 2, 1):
                if i + k <= hi and sreg(i) % min(k, 4) == 0:
                    sz = {1: "dword", 2: "dwordx2", 4: "dwordx4", 8: "dwordx8", 16: "dwordx16"}[k]
                    dst = "s%d" % sreg(i) if k == 1 else "s[%d:%d]" % (sreg(i), sreg(i) + k - 1)
                    emit("s_load_%s %s, %s, 0x%x" % (sz, dst, ptr, 4 * i))
                    i += k
                    break

    # ---- prologue: registers, the first group of the first block
    emit("v_mov_b32 v62, %[pre]")
    emit("v_mov_b32 v63, %s" % ("%[init]" if integer else "%[acc]"))
    emit("v_mov_b32 v50, %[voff]")
    emit("s_mov_b64 %s, %%[llr]" % cur)
    emit("s_mov_b32 %s, %%[nblk]" % cnt)
    if integer:     # carry-in of the first step (q = 0): the lanes holding the even predecessor = lane bit 0 clear
        emit("s_mov_b32 vcc_lo, 0x55555555")
        emit("s_mov_b32 vcc_hi, 0x55555555")
    load_group(0, cur)
    emit("1:")
    # next block's pointer, clamped to the last full block (the refills of the last iteration re-read valid addresses)
    emit("s_add_u32 s%d, s%d, %d" % (PTR, PTR + 2, 4 * n))
    emit("s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 3))
    emit("s_cmp_gt_u32 %s, 1" % cnt)
    emit("s_cselect_b32 s%d, s%d, s%d" % (PTR, PTR, PTR + 2))
    emit("s_cselect_b32 s%d, s%d, s%d" % (PTR + 1, PTR + 1, PTR + 3))
    pend = None                                              # decision pending from the previous step
    # Steps with both LLRs need tab = A*t0 + B*t1 (one rounding) before pre is added: that pair does not depend on the path
    # metrics, so it is computed one two-LLR step ahead (alternating v[54:55] / v[60:61]), off the pre -> pre critical path.
    two = [j for j in range(BLK) if steps[j][0] is not None and steps[j][1] is not None]
    tabreg = {j: (54 if k % 2 == 0 else 60) for k, j in enumerate(two)}

    def emit_tab(j):
        q, (i0, i1), t = j % 6, steps[j], tabreg[j]
        p0, s0 = sgpr_pair(i0)
        p1, s1 = sgpr_pair(i1)
        emit("v_pk_mul_f32 v[%d:%d], %%[kb%d], %s op_sel:[0,%d] op_sel_hi:[1,%d]" % (t, t + 1, q, p1, s1, s1))
        emit("v_pk_fma_f32 v[%d:%d], %%[ka%d], %s, v[%d:%d] op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (t, t + 1, q, p0, t, t + 1, s0, s0))

    def emit_pend(p):
        if p is None:
            emit("s_nop 1")
        elif p[0] == "cls":                                  # float, DPP step: class test of d
            emit("v_cmp_class_f32_e32 vcc, v%d, %%[cls%d]" % (p[1], p[2]))
            emit(ADDC)
        elif p[0] == "gt":                                   # float, swap step: a1 > a0
            emit("v_cmp_gt_f32_e32 vcc, v%d, v%d" % (p[1], p[2]))
            emit(ADDC)
        elif p[0] == "subb":                                 # integer, DPP step: borrow(recv - keep - cin)
            emit("v_subb_co_u32_dpp v52, vcc, v%d, v%d, vcc %s%s" % (p[1], p[2], p[3], RM))
            emit(ADDC)
        else:                                                # integer, swap step: borrow(a0 - a1) = [a1 > a0]
            emit("v_sub_co_u32_e32 v52, vcc, v%d, v%d" % (p[1], p[2]))
            emit(ADDC)

    def group_of_step(j):
        return max(g for g in range(G) if bounds[g] <= j)
    hoisted = set()
    vmax = "v_max_u32" if integer else "v_max_f32"
    for j in range(BLK):
        q = j % 6
        i0, i1 = steps[j]
        c = 56 if j % 2 == 0 else 58
        d = 52 if j % 2 == 0 else 53
        if j in bounds[:-1]:
            # group g was requested one group ago and is the only load in flight: wait (no stall), then request the next
            # group into the other register set (whose LLRs were used up by the previous group)
            g = bounds.index(j)
            if "nowait" not in EXP or j == 0:
                emit("s_waitcnt lgkmcnt(0)")
            if g + 1 < G:
                load_group(g + 1, cur)
            else:
                load_group(0, nxt)
        # candidates
        if i0 is not None and i1 is not None:
            if j not in hoisted:
                emit_tab(j)
            t = tabreg[j]
            emit("v_pk_add_f32 v[%d:%d], v[62:63], v[%d:%d] op_sel:[0,0] op_sel_hi:[0,1]" % (c, c + 1, t, t + 1))
            # the next two-LLR step's pair, if its LLRs are already here (same load group): issued now, used ~5 steps later
            nx = [k for k in two if k > j]
            if nx and group_of_step(nx[0]) == group_of_step(j):
                emit_tab(nx[0])
                hoisted.add(nx[0])
        elif i0 is not None:
            p0, s0 = sgpr_pair(i0)
            emit("v_pk_fma_f32 v[%d:%d], %%[ka%d], %s, v[62:63] op_sel:[0,%d,0] op_sel_hi:[1,%d,0]" % (c, c + 1, q, p0, s0, s0))
        else:
            p1, s1 = sgpr_pair(i1)
            emit("v_pk_fma_f32 v[%d:%d], %%[kb%d], %s, v[62:63] op_sel:[0,%d,0] op_sel_hi:[1,%d,0]" % (c, c + 1, q, p1, s1, s1))
        # the previous step's decision fills the two wait states the DPP / permlane read of the candidates needs
        emit_pend(pend)
        if q in DPP:
            if integer:
                emit("%s_dpp v62, v%d, v%d %s%s" % (vmax, c + 1, c, DPP[q], RM))
                pend = ("subb", c + 1, c, DPP[q])
            else:
                emit("v_sub_f32_dpp v%d, v%d, v%d %s%s" % (d, c + 1, c, DPP[q], RM))
                emit("%s_dpp v62, v%d, v%d %s%s" % (vmax, c + 1, c, DPP[q], RM))
                pend = ("cls", d, q)
        elif q == 2:      # lane^3 inside the quad, then mirror inside 8 lanes (lane^7): together lane^4
            emit("v_mov_b32_dpp v51, v%d quad_perm:[3,2,1,0]%s" % (c + 1, RM))
            emit("s_nop 1")
            if integer:
                emit("%s_dpp v62, v51, v%d row_half_mirror%s" % (vmax, c, RM))
                pend = ("subb", 51, c, "row_half_mirror")
            else:
                emit("v_sub_f32_dpp v%d, v51, v%d row_half_mirror%s" % (d, c, RM))
                emit("%s_dpp v62, v51, v%d row_half_mirror%s" % (vmax, c, RM))
                pend = ("cls", d, q)
        else:
            emit("v_permlane%d_swap_b32 v%d, v%d" % (16 if q == 4 else 32, c, c + 1))
            emit("%s v62, v%d, v%d" % (vmax, c, c + 1))
            pend = ("sub", c, c + 1) if integer else ("gt", c + 1, c)
    # flush the last decision, store the block's survivor word of every lane, next block
    emit_pend(pend)
    emit("s_mov_b64 %s, %s" % (cur, nxt))
    emit("s_sub_u32 %s, %s, 1" % (cnt, cnt))
    if integer:
        emit("v_xor_b32 v63, %[flip], v63")                  # even-holder lanes recorded the complement
        # Traceback hint (k_finish walks segments of `segn` blocks in parallel and has to assume the lane each one ends in): in the
        # last block of a segment, bit 30 of the survivor dword (unused: 30 decisions per block) marks the lane(s) holding the
        # largest path metric (all positive here: unsigned max over the wave on the DPP path).  Only a hint: k_finish verifies
        # it against the neighbouring segment and walks again where it was wrong.
        emit("s_add_u32 %[segc], %[segc], 1")
        emit("s_cmp_lt_u32 %[segc], %[segn]")
        emit("s_cbranch_scc1 5f")
        emit("s_mov_b64 s[%d:%d], vcc" % (PTR + 8, PTR + 9))   # the borrow chain survives the compare below
        emit("v_mov_b32 v52, v62")
        for ctrl in ("row_shr:1", "row_shr:2", "row_shr:4", "row_shr:8", "row_bcast:15 row_mask:0xa", "row_bcast:31 row_mask:0xc"):
            emit("s_nop 1")
            emit("v_max_u32_dpp v52, v52, v52 %s%s" % (ctrl, "" if "row_mask" in ctrl else " row_mask:0xf") + " bank_mask:0xf")
        emit("v_mov_b32 v53, 0x40000000")
        emit("v_readlane_b32 s%d, v52, 63" % (PTR + 10))
        emit("s_mov_b32 %[segc], 0")
        emit("v_cmp_eq_u32_e32 vcc, s%d, v62" % (PTR + 10))
        emit("v_cndmask_b32_e32 v52, 0, v53, vcc")
        emit("s_mov_b64 vcc, s[%d:%d]" % (PTR + 8, PTR + 9))
        emit("v_or_b32 v63, v52, v63")
        emit("5:")
    if "nostore" not in EXP:
        emit("global_store_dword v50, v63, %[sv]")
    emit("v_add_u32 v50, 0x100, v50")
    if integer:
        emit("v_mov_b32 v63, %[init]")
        emit("v_cmp_gt_f32_e64 %s, %%[bound], v62" % chk)    # a lane whose metric is below the bound: leave
        emit("s_cmp_eq_u32 %s, 0" % cnt)
        emit("s_cbranch_scc1 2f")
        if "nocheck" not in EXP:
            emit("s_cmp_eq_u64 %s, 0" % chk)
        emit("s_cbranch_scc1 1b")
        emit("2:")
    else:
        emit("s_cmp_lg_u32 %s, 0" % cnt)
        emit("s_cbranch_scc1 1b")
    emit("s_waitcnt lgkmcnt(0)")                             # the last refill lands before anything reuses its registers
    emit("v_mov_b32 %[pre], v62")
    emit("v_mov_b32 %[acc], v63")
    emit("v_mov_b32 %[voff], v50")
    emit("s_mov_b32 %%[nblk], %s" % cnt)
    body = "\n".join('      "%s\\n\\t"' % x for x in L)
    clob = ['"vcc"', '"scc"', '"memory"'] + ['"s%d"' % (PTR + i) for i in range(11)] + \
           ['"s%d"' % (b + i) for b in SETBASE for i in range(SETSZ)] + ['"v%d"' % r for r in range(50, 64)]
    ins = ['[ka%d] "v"(K[%d].a)' % (q, q) for q in range(6)] + ['[kb%d] "v"(K[%d].b)' % (q, q) for q in range(6)]
    if integer:
        ins += ['[init] "v"(init)', '[flip] "v"(flip)', '[bound] "s"(bound)', '[segn] "s"(segn)']
    else:
        ins += ['[cls%d] "v"(K[%d].cls)' % (q, q) for q in range(4)]
    ins += ['[llr] "s"(llr)', '[sv] "s"(sv)']
    n_valu = sum(1 for x in L if x.startswith("v_") and "%[pre]" not in x and "%[acc]" not in x and "v50" not in x
                 and "v_cmp_gt_f32_e64" not in x and "%[init]" not in x and "%[flip]" not in x and "v52, v52" not in x and "v52, v62" not in x
                 and "v_readlane" not in x and "v_cmp_eq_u32" not in x and "0x40000000" not in x and "v_cndmask_b32_e32 v52" not in x
                 and x != "v_or_b32 v63, v52, v63")
    name = "lean_blocks_int" if integer else "lean_blocks_asm"
    extra = "unsigned init, unsigned flip, float bound, int& segc, int segn, " if integer else ""
    outs2 = ', [segc] "+s"(segc)' if integer else ""
    return """
// code rate %s: %d LLRs per block in %d load groups; %d VALU instructions per 30 steps (%.2f per step)
template <> __device__ __forceinline__ void %s<%d>(float& pre, unsigned& acc, const LeanK (&K)[6], %sunsigned& voff,
                                                                 const float* llr, unsigned* sv, int& nblk)
{
  asm volatile(
%s
      : [pre] "+v"(pre), [acc] "+v"(acc), [voff] "+v"(voff), [nblk] "+s"(nblk)%s
      : %s
      : %s);
}
""" % (cr, n, G, n_valu, n_valu / 30.0, name, RATES[cr], extra, body, outs2, ", ".join(ins), ", ".join(clob)), n_valu


# ---------------------------------------------------------------------------------------------- LDS-fed flavour (k_rx_1ss_lds)
# lean_blocks_lds<CR>: the integer step above with the LLRs read from LDS instead of through the scalar cache.  The wave that
# demodulated the symbols left them in its own LDS buffer (de-interleaved), so the LLR stream never reaches HBM.  An LLR is a
# wave-uniform operand: every lane reads the same LDS address (a broadcast, no bank conflict) into a small rotating set of VGPRs,
# GS LLRs per read (`ds_read_b64`; rate 2/3, whose 45 LLRs per block are not a multiple of 8 bytes: `ds_read2_b32` + `ds_read_b32`,
# three LLRs = one puncturing period).  NSETS register sets; the read of group g + NSETS is issued right after the last instruction
# that reads group g (same registers), LDS returns in order, so the wait before the first use of group g is an exact
# `lgkmcnt(reads issued since)` and never drains the prefetches.  G % NSETS == 0, otherwise the next block's first groups would
# land in registers the end of this block still reads.  The sets are small on purpose: every register the statement clobbers is
# one the fused kernel cannot keep a long-lived value in (12-register sets made it spill five values per OFDM symbol to scratch).
# pre / acc / voff are bound to v62 / v63 / v50 ("{v62}" constraints): no copies in or out.
LDS_GS = {"12": 2, "23": 3, "34": 2, "56": 2}
LDS_NSETS = {"12": 3, "23": 3, "34": 4, "56": 3}


def gen_rate_lds(cr, integer=True):
    """integer: the block loop (lean_blocks_lds); not integer: ONE block with the float step (lean_block_flt_lds), for the blocks
    whose metrics are not provably positive — as asm so that its register use is as fixed as the loop's (the compiler's version of
    the 30 unrolled float steps wanted 10 more VGPRs than the rest of k_rx_1ss_lds)"""
    n = NLB[cr]
    GS, NS = LDS_GS[cr], LDS_NSETS[cr]
    SW = 2 if GS == 2 else 4      # registers per set (even, so that every LLR sits in an even-aligned pair)
    assert n % GS == 0
    G = n // GS
    assert G % NS == 0
    VB = 50 - SW * NS             # first register of the LLR sets (even)
    VA = VB - 1                   # LDS byte address of the current block's LLRs
    steps = [idx(cr, j) for j in range(BLK)]

    def grp(i):
        return i // GS

    def vreg(i):
        return VB + SW * (grp(i) % NS) + (i % GS)

    def vpair(i):
        r = vreg(i)
        return "v[%d:%d]" % (r & ~1, (r & ~1) + 1), r & 1

    def load_group(g, wrapped):
        """ds reads of group g of this block (wrapped: of the next block, whose LLRs follow at + 4 n bytes)"""
        off = 4 * GS * g + (4 * n if wrapped else 0)
        r = VB + SW * (g % NS)
        if GS == 2:
            assert (4 * n) % 8 == 0
            return ["ds_read_b64 v[%d:%d], v%d offset:%d" % (r, r + 1, VA, off)]
        o0 = off // 4
        assert o0 + 1 < 256
        return ["ds_read2_b32 v[%d:%d], v%d offset0:%d offset1:%d" % (r, r + 1, VA, o0, o0 + 1),
                "ds_read_b32 v%d, v%d offset:%d" % (r + 2, VA, off + 8)]

    body = []                     # (text, groups read)

    def emit(x, reads=()):
        body.append((x, tuple(reads)))
    cnt, chk = "s%d" % (PTR + 4), "s[%d:%d]" % (PTR + 6, PTR + 7)
    two = [j for j in range(BLK) if steps[j][0] is not None and steps[j][1] is not None]
    tabreg = {j: (54 if k % 2 == 0 else 60) for k, j in enumerate(two)}

    def emit_tab(j):
        q, (i0, i1), t = j % 6, steps[j], tabreg[j]
        p0, s0 = vpair(i0)
        p1, s1 = vpair(i1)
        emit("v_pk_mul_f32 v[%d:%d], %%[kb%d], %s op_sel:[0,%d] op_sel_hi:[1,%d]" % (t, t + 1, q, p1, s1, s1), [grp(i1)])
        emit("v_pk_fma_f32 v[%d:%d], %%[ka%d], %s, v[%d:%d] op_sel:[0,%d,0] op_sel_hi:[1,%d,1]" % (t, t + 1, q, p0, t, t + 1, s0, s0), [grp(i0)])

    def emit_pend(p):
        if p is None:
            emit("s_nop 1")
        elif p[0] == "subb":
            emit("v_subb_co_u32_dpp v52, vcc, v%d, v%d, vcc %s%s" % (p[1], p[2], p[3], RM))
            emit(ADDC)
        elif p[0] == "sub":
            emit("v_sub_co_u32_e32 v52, vcc, v%d, v%d" % (p[1], p[2]))
            emit(ADDC)
        elif p[0] == "cls":
            emit("v_cmp_class_f32_e32 vcc, v%d, %%[cls%d]" % (p[1], p[2]))
            emit(ADDC)
        else:
            emit("v_cmp_gt_f32_e32 vcc, v%d, v%d" % (p[1], p[2]))
            emit(ADDC)
    pend = None
    hoisted = set()
    vmax = "v_max_u32" if integer else "v_max_f32"
    for j in range(BLK):
        q = j % 6
        i0, i1 = steps[j]
        c = 56 if j % 2 == 0 else 58
        d = 52 if j % 2 == 0 else 53
        if i0 is not None and i1 is not None:
            if j not in hoisted:
                emit_tab(j)
            t = tabreg[j]
            emit("v_pk_add_f32 v[%d:%d], v[62:63], v[%d:%d] op_sel:[0,0] op_sel_hi:[0,1]" % (c, c + 1, t, t + 1))
            # the next two-LLR step's pair ahead of time (off the pre -> pre chain), if its LLRs are at most one group further on
            nx = [k for k in two if k > j]
            if nx and grp(steps[nx[0]][1]) <= grp(i1) + 1:
                emit_tab(nx[0])
                hoisted.add(nx[0])
        elif i0 is not None:
            p0, s0 = vpair(i0)
            emit("v_pk_fma_f32 v[%d:%d], %%[ka%d], %s, v[62:63] op_sel:[0,%d,0] op_sel_hi:[1,%d,0]" % (c, c + 1, q, p0, s0, s0), [grp(i0)])
        else:
            p1, s1 = vpair(i1)
            emit("v_pk_fma_f32 v[%d:%d], %%[kb%d], %s, v[62:63] op_sel:[0,%d,0] op_sel_hi:[1,%d,0]" % (c, c + 1, q, p1, s1, s1), [grp(i1)])
        emit_pend(pend)
        if q in DPP:
            if not integer:
                emit("v_sub_f32_dpp v%d, v%d, v%d %s%s" % (d, c + 1, c, DPP[q], RM))
            emit("%s_dpp v62, v%d, v%d %s%s" % (vmax, c + 1, c, DPP[q], RM))
            pend = ("subb", c + 1, c, DPP[q]) if integer else ("cls", d, q)
        elif q == 2:
            emit("v_mov_b32_dpp v51, v%d quad_perm:[3,2,1,0]%s" % (c + 1, RM))
            emit("s_nop 1")
            if not integer:
                emit("v_sub_f32_dpp v%d, v51, v%d row_half_mirror%s" % (d, c, RM))
            emit("%s_dpp v62, v51, v%d row_half_mirror%s" % (vmax, c, RM))
            pend = ("subb", 51, c, "row_half_mirror") if integer else ("cls", d, q)
        else:
            emit("v_permlane%d_swap_b32 v%d, v%d" % (16 if q == 4 else 32, c, c + 1))
            emit("%s v62, v%d, v%d" % (vmax, c, c + 1))
            pend = ("sub", c, c + 1) if integer else ("gt", c + 1, c)
    emit_pend(pend)
    # ---- loads after the last reader of a group, exact waits before the first reader
    first, last = {}, {}
    for k, (x, rd) in enumerate(body):
        for g in rd:
            first.setdefault(g, k)
            last[g] = k
    assert sorted(first) == list(range(G)) and all(last[g] <= last[g + 1] and first[g] <= first[g + 1] for g in range(G - 1))
    assert all(last[g] < first[g + NS] for g in range(G - NS))      # a set is free before its next group is needed
    after = {}
    for g in range(G):
        after.setdefault(last[g], []).append(g)
    stream = []                   # one block: ("ds", text, group: g or G + g' of the next block) / ("wait", group) / ("x", text)
    for k, (x, rd) in enumerate(body):
        for g in sorted(set(rd)):
            if first[g] == k:
                stream.append(("wait", g))
        stream.append(("x", x))
        for g in sorted(after.get(k, [])):
            tgt = g + NS
            if tgt >= G and not integer:
                continue                                     # one block only: nothing to prefetch for the next
            for ins in load_group(tgt % G, tgt >= G):
                stream.append(("ds", ins, tgt))
    # simulate the prologue and two blocks: reads in flight at every wait (LDS returns in order)
    issued = []
    for g in range(NS):
        issued += [g] * len(load_group(g, False))
    waits = []
    for blk in range(2 if integer else 1):
        w = []
        for it in stream:
            if it[0] == "ds":
                issued.append(blk * G + it[2])
            elif it[0] == "wait":
                gid = blk * G + it[1]
                pos = max(k for k, v in enumerate(issued) if v == gid)
                w.append(len(issued) - 1 - pos)
        waits.append(w)
    assert (not integer or waits[0] == waits[1]) and max(waits[0]) <= 15, waits
    if not integer:
        L = ["s_waitcnt lgkmcnt(0)", "v_mov_b32 v%d, %%[llr]" % VA]
        for g in range(NS):
            L.extend(load_group(g, False))
        wi = iter(waits[0])
        for it in stream:
            L.append("s_waitcnt lgkmcnt(%d)" % next(wi) if it[0] == "wait" else it[1])
        text = "\n".join('      "%s\\n\\t"' % x for x in L)
        clob = ['"vcc"', '"memory"'] + ['"v%d"' % r for r in range(VA, 62) if r != 50]
        ins = ['[ka%d] "v"(K[%d].a)' % (q, q) for q in range(6)] + ['[kb%d] "v"(K[%d].b)' % (q, q) for q in range(6)]
        ins += ['[cls%d] "v"(K[%d].cls)' % (q, q) for q in range(4)] + ['[llr] "s"(llr)']
        n_valu = sum(1 for (x, rd) in body if x.startswith("v_"))
        return """
// code rate %s, ONE block with the float step, LLRs from LDS (%d VALU instructions); acc: the 30 decisions in bits 29..0, older bits above
template <> __device__ __forceinline__ void lean_block_flt_lds<%d>(float& pre, unsigned& acc, const LeanK (&K)[6], unsigned llr)
{
  asm volatile(
%s
      : "+{v62}"(pre), "+{v63}"(acc)
      : %s
      : %s);
}
""" % (cr, n_valu, RATES[cr], text, ", ".join(ins), ", ".join(clob)), n_valu
    L = []
    add = L.append
    add("s_waitcnt lgkmcnt(0)")                              # nothing of the caller's in flight: the counts below are exact
    add("v_mov_b32 v63, %[init]")
    add("v_mov_b32 v%d, %%[llr]" % VA)
    add("s_mov_b32 %s, %%[nblk]" % cnt)
    # the survivor stores go through an SGPR base that the scalar unit advances by 256 bytes per block: v50 (this lane's offset)
    # stands still inside the loop and is brought up to date once, at the end
    add("s_mov_b64 s[%d:%d], %%[sv]" % (PTR, PTR + 1))
    add("s_mov_b32 s%d, 0" % (PTR + 2))
    add("s_mov_b32 vcc_lo, 0x55555555")
    add("s_mov_b32 vcc_hi, 0x55555555")
    for g in range(NS):
        L.extend(load_group(g, False))
    add("1:")
    wi = iter(waits[0])
    for it in stream:
        if it[0] == "wait":
            add("s_waitcnt lgkmcnt(%d)" % next(wi))
        else:
            add(it[1])
    add("s_sub_u32 %s, %s, 1" % (cnt, cnt))
    add("v_add_u32 v%d, %d, v%d" % (VA, 4 * n, VA))
    add("v_xor_b32 v63, %[flip], v63")
    add("s_add_u32 %[segc], %[segc], 1")
    add("s_cmp_lt_u32 %[segc], %[segn]")
    add("s_cbranch_scc1 5f")
    add("s_mov_b64 s[%d:%d], vcc" % (PTR + 8, PTR + 9))
    add("v_mov_b32 v52, v62")
    for ctrl in ("row_shr:1", "row_shr:2", "row_shr:4", "row_shr:8", "row_bcast:15 row_mask:0xa", "row_bcast:31 row_mask:0xc"):
        add("s_nop 1")
        add("v_max_u32_dpp v52, v52, v52 %s%s" % (ctrl, "" if "row_mask" in ctrl else " row_mask:0xf") + " bank_mask:0xf")
    add("v_mov_b32 v53, 0x40000000")
    add("v_readlane_b32 s%d, v52, 63" % (PTR + 10))
    add("s_mov_b32 %[segc], 0")
    add("v_cmp_eq_u32_e32 vcc, s%d, v62" % (PTR + 10))
    add("v_cndmask_b32_e32 v52, 0, v53, vcc")
    add("s_mov_b64 vcc, s[%d:%d]" % (PTR + 8, PTR + 9))
    add("v_or_b32 v63, v52, v63")
    add("5:")
    add("global_store_dword v50, v63, s[%d:%d]" % (PTR, PTR + 1))
    add("s_add_u32 s%d, s%d, 0x100" % (PTR, PTR))
    add("s_addc_u32 s%d, s%d, 0" % (PTR + 1, PTR + 1))
    add("s_add_u32 s%d, s%d, 0x100" % (PTR + 2, PTR + 2))
    add("v_mov_b32 v63, %[init]")
    add("v_cmp_gt_f32_e64 %s, %%[bound], v62" % chk)
    add("s_cmp_eq_u32 %s, 0" % cnt)
    add("s_cbranch_scc1 2f")
    add("s_cmp_eq_u64 %s, 0" % chk)
    add("s_cbranch_scc1 1b")
    add("2:")
    add("s_waitcnt lgkmcnt(0)")
    add("v_add_u32 v50, s%d, v50" % (PTR + 2))
    add("s_mov_b32 %%[nblk], %s" % cnt)
    n_valu = sum(1 for (x, rd) in body if x.startswith("v_"))
    text = "\n".join('      "%s\\n\\t"' % x for x in L)
    fixed = (50, 62, 63)
    clob = ['"vcc"', '"scc"', '"memory"'] + ['"s%d"' % (PTR + i) for i in range(0, 11)] + ['"v%d"' % r for r in range(VA, 64) if r not in fixed]
    ins = ['[ka%d] "v"(K[%d].a)' % (q, q) for q in range(6)] + ['[kb%d] "v"(K[%d].b)' % (q, q) for q in range(6)]
    ins += ['[init] "v"(init)', '[flip] "v"(flip)', '[bound] "s"(bound)', '[segn] "s"(segn)', '[llr] "s"(llr)', '[sv] "s"(sv)']
    return """
// code rate %s, LLRs from LDS: %d per block in %d groups of %d, %d register sets (v%d..v49), address in v%d; %d VALU instructions per 30 steps
template <> __device__ __forceinline__ void lean_blocks_lds<%d>(float& pre, unsigned& acc, const LeanK (&K)[6], unsigned init, unsigned flip, float bound,
                                                                 int& segc, int segn, unsigned& voff, unsigned llr, unsigned* sv, int& nblk)
{
  asm volatile(
%s
      : "+{v62}"(pre), "=&{v63}"(acc), "+{v50}"(voff), [nblk] "+s"(nblk), [segc] "+s"(segc)
      : %s
      : %s);
}
""" % (cr, n, G, GS, NS, VB, VA, n_valu, RATES[cr], text, ", ".join(ins), ", ".join(clob)), n_valu


def main():
    parts, counts = [], {}
    for integer in (True, False):
        if not integer:
            parts.append("\n#ifdef C8P_LEAN_ASM\n")
        for cr in ("12", "23", "34", "56"):
            txt, nv = gen_rate(cr, integer)
            parts.append(txt)
            counts[(cr, integer)] = nv
        if not integer:
            parts.append("#endif\n")
    for cr in ("12", "23", "34", "56"):
        txt, nv = gen_rate_lds(cr)
        parts.append(txt)
        counts[(cr, "lds")] = nv
    for cr in ("12", "23", "34", "56"):
        txt, nv = gen_rate_lds(cr, False)
        parts.append(txt)
        counts[(cr, "ldsf")] = nv
    hdr = """// c8p_lean_asm.h — GENERATED by tools/gen_lean_asm.py (do not edit; edit the generator, which documents the step).
//
// The full-block loop of the lean Viterbi forward pass as one inline-asm statement per code rate.
//   lean_blocks_int<CR>: integer compare with a per-lane borrow-in (all candidates positive); leaves at the first block whose
//                        start metrics are not all >= bound.  VALU instructions per 30-step block:
//                        rate 1/2: %d   2/3: %d   3/4: %d   5/6: %d
//   lean_blocks_asm<CR>: the float step of c8p_viterbi_lean.h (-DC8P_LEAN_ASM, experiments):
//                        rate 1/2: %d   2/3: %d   3/4: %d   5/6: %d
// In / out: pre, acc, voff (byte offset of this lane's survivor dword), nblk (full blocks left; 0 when all were done).
#pragma once

template <int CR> __device__ __forceinline__ void lean_blocks_int(float& pre, unsigned& acc, const LeanK (&K)[6], unsigned init, unsigned flip,
                                                                   float bound, int& segc, int segn, unsigned& voff, const float* llr, unsigned* sv, int& nblk);
template <int CR> __device__ __forceinline__ void lean_blocks_asm(float& pre, unsigned& acc, const LeanK (&K)[6], unsigned& voff,
                                                                   const float* llr, unsigned* sv, int& nblk);
// lean_blocks_lds<CR>: lean_blocks_int with the LLRs read from LDS (`llr` = LDS byte address of the first block's LLRs, 16-byte aligned
//                      except at rate 2/3) into a rotating set of VGPRs; the same instruction counts
template <int CR> __device__ __forceinline__ void lean_blocks_lds(float& pre, unsigned& acc, const LeanK (&K)[6], unsigned init, unsigned flip,
                                                                   float bound, int& segc, int segn, unsigned& voff, unsigned llr, unsigned* sv, int& nblk);
// lean_block_flt_lds<CR>: one block with the float step, LLRs from LDS (the blocks lean_blocks_lds will not take)
template <int CR> __device__ __forceinline__ void lean_block_flt_lds(float& pre, unsigned& acc, const LeanK (&K)[6], unsigned llr);
""" % tuple(counts[(cr, i)] for i in (True, False) for cr in ("12", "23", "34", "56"))
    open(OUT, "w").write(hdr + "".join(parts))
    print("wrote", OUT, {("%s%s" % (k[0], {"lds": "l", "ldsf": "lf", True: "i", False: "f"}[k[1]])): v for k, v in counts.items()})


if __name__ == "__main__":
    main()
