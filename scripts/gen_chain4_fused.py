#!/usr/bin/env python3
"""Generates papr_amd/csrc/chain4_fused.inc: the hot SLOT of chain4.hip as ONE inline-asm statement per variant -- the k-loop of
tile T and BOTH halves of the row phase of the other tile U (P1 | s_barrier | P2), three instruction streams interleaved.

Why: the two-role slots of chain4.hip (waves 0-3 multiply while waves 4-7 do row phases, then swap) cost 10k (inference) to
13.5k (training) cycles per slot for 6.1k of matrix-pipe time (scripts/probes/chain4_trace.py): a wave's own k-loop (4k), P1 (2k)
and P2 (2.5-5k) add up, and one SIMD issues about one instruction per four cycles whoever it belongs to.  A wave gets under its own
sum only by issuing its vector / LDS / store instructions in the shadow of its own matrix instructions: behind every MFMA (32
cycles of pipe, 64 with the SIMD partner's in between) come the k-loop's memory instruction and three or four instructions of
the row phase.

Streams (the arithmetic of chain4.hip's C++ row phases instruction for instruction: results are bit-identical, tests/test_hip_chain_variants.py)
  K    sixteen k-steps of six MFMAs (hi.lo, lo.hi, hi.hi for two 32-row tiles, chain.hip's order per accumulator); A fragments in ONE
       buffer, each register re-read for the next k-step right behind its last use; weights in a[0:127], refilled with the next
       step's behind their last use (LD variants).
  P1   on U's accumulators (still raw from the slot before): 1/scale and bias, activation, sign word (forward) or 1/scale and the
       activation derivative from the sign word (data-gradient); partial row maxima -> LDS.
  ---- s_barrier (every wave's partial maxima are in LDS)
  P2   row maximum, power-of-two scale, 1/scale table; the rows through the wave's own piece of the dead planes to memory, whole
       cache lines per instruction; split into hi / lo planes = the next layer's input.
All LDS traffic goes through one in-order queue; every wait is computed by simulating that queue.

Registers.  The accumulators live in v64-v127 outside the compiler's allocation (chain4.hip: tile X in v[64:95], tile Y in v[96:127]);
the statement names them -- hence an X and a Y flavour of every variant.  Temporaries live in fixed registers named as clobbers
(v28-v63); everything else that crosses the statement is an operand.

  python scripts/gen_chain4_fused.py > papr_amd/csrc/chain4_fused.inc
"""

import os
import sys
ABLATE = os.environ.get("C4F_ABLATE", "")      # timing experiments (scripts/probes/c4_variant.sh with -DC4_FUSED_INC=...)
PRIO = os.environ.get("C4F_PRIO", "")          # timing experiment: s_setprio flips inside the statement (operand [role]: 1 for the SIMD's older wave): "half" = the
                                               # younger wave ahead in the first half, the older in the second; "young" = the younger ahead throughout
STMOD = os.environ.get("C4F_STORE_MOD", "")    # timing experiment: cache-policy bits on the row stores (" sc1", " nt", " sc0 sc1", ...)
KVAR = os.environ.get("C4F_KVAR", "")          # timing experiments on the k-loop (with C4F_ABLATE=K: the row-phase temporaries are free): noread | dbuf | dbufh

F = {"l0": "v[28:31]", "l1": "v[32:35]", "h0": "v[36:39]", "h1": "v[40:43]"}
KA = "v44"                                      # k-loop address temporary
AD, AD2 = "v45", "v46"                          # row-phase address temporaries
ADW = "v47"                                     # this wave's place in the partial-maxima table
G = ["v%d" % i for i in range(48, 64)]          # general temporaries (tuples start at even registers: gfx950 wants 64-bit alignment)
GB = 48
CLOB_V = ["v%d" % i for i in range(28, 64)]     # (the compiler has v0-v63: 36 temporaries + the statement's operands leave it ~16 across the statement;
                                                # with v24-v27 taken as well -- a prefetch experiment's landing registers -- one operand was spilled and every hot slot
                                                # began with its reload from scratch and a vmcnt(0) wait behind the previous slot's row stores)
ACC = {"X": 64, "Y": 96}                        # the accumulators' registers (chain4.hip: REGISTERS)
SNH, SNL = "s[90:91]", "s[92:93]"              # LD variants: running bases of the next step's hi / lo fragments (clobbers)


def wh(ks): return "a[%d:%d]" % (8 * ks, 8 * ks + 3)
def wl(ks): return "a[%d:%d]" % (8 * ks + 4, 8 * ks + 7)
def vt(lo, n): return "v[%d:%d]" % (lo, lo + n - 1)


class Item:
    __slots__ = ("text", "lds", "need", "kind")

    def __init__(self, text, lds=None, need=(), kind="valu"):
        self.text, self.lds, self.need, self.kind = text, lds, tuple(need), kind


def k_stream(acc_t, ld, kcnt=16):
    """list of (mfma Item, [Items issued behind it]); acc_t = first register of T's accumulators"""
    a0, a1 = vt(acc_t, 16), vt(acc_t + 16, 16)
    off = {"l0": 4096, "l1": 36864, "h0": 0, "h1": 32768}

    def rd(kind, ks):
        return Item("ds_read_b128 %s, %s offset:%d" % (F[kind], KA if ks & 7 else "%[pbx]", off[kind] + (256 if ks >= 8 else 0)), lds=("f", kind, ks), kind="lds")

    def addr(ks):
        return [Item("v_xor_b32 %s, 0x%x, %%[pbx]" % (KA, 32 * (ks & 7)))] if ks & 7 else []

    pro = [rd("l0", 0), rd("l1", 0), rd("h0", 0), rd("h1", 0)]
    if ld:                                      # working copies of the next step's fragment bases (an asm operand has no sub-register syntax)
        pro += [Item("s_mov_b32 s90, %[nhlo]", kind="salu"), Item("s_mov_b32 s91, %[nhhi]", kind="salu"),
                Item("s_mov_b32 s92, %[nllo]", kind="salu"), Item("s_mov_b32 s93, %[nlhi]", kind="salu")]
        if kcnt < 16:
            # a short first layer (kcnt k-steps): the registers of the other k-steps are free from the start -- the next step's fragments for
            # them are requested right here, from a second pair of running bases (s[94:97])
            g0 = (kcnt >> 2) * 4096
            pro += [Item("s_add_u32 s94, s90, 0x%x" % g0, kind="salu"), Item("s_addc_u32 s95, s91, 0", kind="salu"),
                    Item("s_add_u32 s96, s92, 0x%x" % g0, kind="salu"), Item("s_addc_u32 s97, s93, 0", kind="salu")]
            for ks in range(kcnt, 16):
                pro.append(Item("global_load_dwordx4 %s, %%[wv], s[94:95] offset:%d" % (wh(ks), (ks & 3) * 1024), kind="vmem"))
                pro.append(Item("global_load_dwordx4 %s, %%[wv], s[96:97] offset:%d" % (wl(ks), (ks & 3) * 1024), kind="vmem"))
                if ks & 3 == 3 and ks < 15:
                    pro += [Item("s_add_u32 s94, s94, 0x1000", kind="salu"), Item("s_addc_u32 s95, s95, 0", kind="salu"),
                            Item("s_add_u32 s96, s96, 0x1000", kind="salu"), Item("s_addc_u32 s97, s97, 0", kind="salu")]
    steps = []
    for ks in range(kcnt):
        nx = ks + 1 if ks < kcnt - 1 else None
        first = ks == 0

        def mf(acc, w, kind, init=False):
            return Item("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (acc, w, F[kind], "0" if init else acc), need=[("f", kind, ks)], kind="mfma")
        g = []
        g.append((mf(a0, wh(ks), "l0", first), (addr(nx) + [rd("l0", nx)]) if nx is not None else []))
        g.append((mf(a1, wh(ks), "l1", first), [rd("l1", nx)] if nx is not None else []))
        g.append((mf(a0, wl(ks), "h0"), []))
        post = []
        if ld:
            post.append(Item("global_load_dwordx4 %s, %%[wv], %s offset:%d" % (wl(ks), SNL, (ks & 3) * 1024), kind="vmem"))
        g.append((mf(a1, wl(ks), "h1"), post))
        g.append((mf(a0, wh(ks), "h0"), [rd("h0", nx)] if nx is not None else []))
        post = [rd("h1", nx)] if nx is not None else []
        if ld:
            post.append(Item("global_load_dwordx4 %s, %%[wv], %s offset:%d" % (wh(ks), SNH, (ks & 3) * 1024), kind="vmem"))
            if ks & 3 == 3 and ks < kcnt - 1:   # the bases move on by four k-steps
                post += [Item("s_add_u32 s90, s90, 0x1000", kind="salu"), Item("s_addc_u32 s91, s91, 0", kind="salu"),
                         Item("s_add_u32 s92, s92, 0x1000", kind="salu"), Item("s_addc_u32 s93, s93, 0", kind="salu")]
        g.append((mf(a1, wh(ks), "h1"), post))
        steps += g
    return pro, steps


def k_stream_exp(acc_t, ld, kcnt=16):
    """timing experiments (C4F_KVAR, with C4F_ABLATE=K): noread = no fragment reads at all; dbuf = every fragment in two buffers (the second in
    v48-v63), read two k-steps ahead; dbufh = only the hi fragments (used twice per k-step, re-read latest) in two buffers"""
    a0, a1 = vt(acc_t, 16), vt(acc_t + 16, 16)
    off = {"l0": 4096, "l1": 36864, "h0": 0, "h1": 32768}
    F2 = {"l0": "v[48:51]", "l1": "v[52:55]", "h0": "v[56:59]", "h1": "v[60:63]"}
    two = {"dbuf": ("l0", "l1", "h0", "h1"), "dbufh": ("h0", "h1"), "noread": ()}[KVAR]
    KB = [KA, AD2]

    def reg(kind, ks):
        return F2[kind] if (kind in two and ks & 1) else F[kind]

    def rd(kind, ks):
        if KVAR == "noread":
            return []
        return [Item("ds_read_b128 %s, %s offset:%d" % (reg(kind, ks), KB[ks & 1] if ks & 7 else "%[pbx]", off[kind] + (256 if ks >= 8 else 0)), lds=("f", kind, ks), kind="lds")]

    def addr(ks):
        return [Item("v_xor_b32 %s, 0x%x, %%[pbx]" % (KB[ks & 1], 32 * (ks & 7)))] if ks & 7 else []

    pro = rd("l0", 0) + rd("l1", 0) + rd("h0", 0) + rd("h1", 0)
    if two:
        pro += addr(1) + [x for kind in two for x in rd(kind, 1)]
    if ld:
        pro += [Item("s_mov_b32 s90, %[nhlo]", kind="salu"), Item("s_mov_b32 s91, %[nhhi]", kind="salu"),
                Item("s_mov_b32 s92, %[nllo]", kind="salu"), Item("s_mov_b32 s93, %[nlhi]", kind="salu")]
    steps = []
    for ks in range(kcnt):
        first = ks == 0

        def nxt(kind):          # the k-step whose fragment `kind` is read behind this k-step's last use of its register
            n = ks + 2 if kind in two else ks + 1
            return n if n < kcnt else None

        def mf(acc, w, kind, init=False):
            need = [] if KVAR == "noread" else [("f", kind, ks)]
            return Item("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (acc, w, reg(kind, ks), "0" if init else acc), need=need, kind="mfma")

        def rr(kind):
            n = nxt(kind)
            return (addr(n) if kind in ("l0", "h0") and (kind == "l0" or "l0" in two or True) else []) + rd(kind, n) if n is not None else []
        g = []
        g.append((mf(a0, wh(ks), "l0", first), rr("l0")))
        g.append((mf(a1, wh(ks), "l1", first), rd("l1", nxt("l1")) if nxt("l1") is not None else []))
        g.append((mf(a0, wl(ks), "h0"), []))
        post = []
        if ld:
            post.append(Item("global_load_dwordx4 %s, %%[wv], %s offset:%d" % (wl(ks), SNL, (ks & 3) * 1024), kind="vmem"))
        g.append((mf(a1, wl(ks), "h1"), post))
        g.append((mf(a0, wh(ks), "h0"), rr("h0")))
        post = rd("h1", nxt("h1")) if nxt("h1") is not None else []
        if ld:
            post.append(Item("global_load_dwordx4 %s, %%[wv], %s offset:%d" % (wh(ks), SNH, (ks & 3) * 1024), kind="vmem"))
            if ks & 3 == 3 and ks < kcnt - 1:
                post += [Item("s_add_u32 s90, s90, 0x1000", kind="salu"), Item("s_addc_u32 s91, s91, 0", kind="salu"),
                         Item("s_add_u32 s92, s92, 0x1000", kind="salu"), Item("s_addc_u32 s93, s93, 0", kind="salu")]
        g.append((mf(a1, wh(ks), "h1"), post))
        steps += g
    return pro, steps


def k_stream_one(acc_t, ld, kcnt=16):
    """the same for the one-product mode (h1): two MFMAs per k-step (hi.hi for the two 32-row tiles), hi fragments only -- in TWO buffers
    (the lo fragments' registers are free), each re-read for k-step ks + 2 right behind its use; hi weights only."""
    a0, a1 = vt(acc_t, 16), vt(acc_t + 16, 16)
    buf = [(F["h0"], F["h1"]), (F["l0"], F["l1"])]
    KB = [KA, AD2]                              # one address temporary per buffer (AD2: the row phases of this mode do not need it)

    def rd(j, ks):
        b = ks & 1
        return Item("ds_read_b128 %s, %s offset:%d" % (buf[b][j], KB[b] if ks & 7 else "%[pbx]", (32768 if j else 0) + (256 if ks >= 8 else 0)), lds=("f", j, ks), kind="lds")

    def addr(ks):
        return [Item("v_xor_b32 %s, 0x%x, %%[pbx]" % (KB[ks & 1], 32 * (ks & 7)))] if ks & 7 else []

    pro = [rd(0, 0), rd(1, 0)] + (addr(1) + [rd(0, 1), rd(1, 1)] if kcnt > 1 else [])
    if ld:
        pro += [Item("s_mov_b32 s90, %[nhlo]", kind="salu"), Item("s_mov_b32 s91, %[nhhi]", kind="salu")]
        if kcnt < 16:
            pro += [Item("s_add_u32 s94, s90, 0x%x" % ((kcnt >> 2) * 4096), kind="salu"), Item("s_addc_u32 s95, s91, 0", kind="salu")]
            for ks in range(kcnt, 16):
                pro.append(Item("global_load_dwordx4 %s, %%[wv], s[94:95] offset:%d" % (wh(ks), (ks & 3) * 1024), kind="vmem"))
                if ks & 3 == 3 and ks < 15:
                    pro += [Item("s_add_u32 s94, s94, 0x1000", kind="salu"), Item("s_addc_u32 s95, s95, 0", kind="salu")]
    steps = []
    for ks in range(kcnt):
        nx = ks + 2 if ks + 2 < kcnt else None
        b = ks & 1
        m0 = Item("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (a0, wh(ks), buf[b][0], "0" if ks == 0 else a0), need=[("f", 0, ks)], kind="mfma")
        m1 = Item("v_mfma_f32_32x32x16_f16 %s, %s, %s, %s" % (a1, wh(ks), buf[b][1], "0" if ks == 0 else a1), need=[("f", 1, ks)], kind="mfma")
        steps.append((m0, (addr(nx) + [rd(0, nx)]) if nx is not None else []))
        post = [rd(1, nx)] if nx is not None else []
        if ld:
            post.append(Item("global_load_dwordx4 %s, %%[wv], %s offset:%d" % (wh(ks), SNH, (ks & 3) * 1024), kind="vmem"))
            if ks & 3 == 3 and ks < kcnt - 1:
                post += [Item("s_add_u32 s90, s90, 0x1000", kind="salu"), Item("s_addc_u32 s91, s91, 0", kind="salu")]
        steps.append((m1, post))
    return pro, steps


def pm_delta(acc_u):
    """LDS bytes from tile U's 1 / scale table to its partial-maxima table (chain4.hip: C4_OFF_INV / C4_OFF_PMAX; inv_all [2][64] floats,
    pmax_all [2][8][64] floats): both are indexed by the lane's row, so the statement takes ONE address operand (invad) and reaches the
    other table through the instructions' offset fields -- two vector operands fewer for the compiler to keep across the statement."""
    u = 0 if acc_u == ACC["X"] else 1
    return 512 + u * (8 * 64 * 4 - 64 * 4)


def p1_stream(acc_u, mode, act):
    """P1 on U's accumulators (first register acc_u).  mode: fwd (training forward), inf, dgrad; act: relu, leaky"""
    it = []
    PMD = pm_delta(acc_u)
    B = G[0:8]
    INV = [G[8], G[9]]
    LM = [G[10], G[11]]
    SWP, W, T1, T2 = G[12], [G[13], G[14]], G[15], G[13]       # (T2: data-gradient only, where no sign word is formed)
    it.append(Item("ds_read_b32 %s, %%[invad]" % INV[0], lds=("inv", 0), kind="lds"))
    it.append(Item("ds_read_b32 %s, %%[invad] offset:128" % INV[1], lds=("inv", 1), kind="lds"))
    if mode == "fwd":
        it += [Item("v_mov_b32 %s, 0" % W[0]), Item("v_mov_b32 %s, 0" % W[1])]
    for gp in range(2):
        if mode != "dgrad":
            for j in range(2):
                it.append(Item("ds_read_b128 %s, %%[biasad] offset:%d" % (vt(GB + 4 * j, 4), 32 * gp + 16 * j), lds=("b", gp, j), kind="lds"))
        for i in range(2):
            for gg in range(2):
                g = 2 * gp + gg
                regs = ["v%d" % (acc_u + 16 * i + 4 * g + c) for c in range(4)]
                for c in range(4):
                    a, n = regs[c], 16 * i + 4 * g + c
                    if mode != "dgrad":
                        it.append(Item("v_fma_f32 %s, %s, %s, %s" % (a, a, INV[i], B[4 * gg + c]), need=[("inv", i), ("b", gp, gg)]))
                        if act == "relu":
                            it.append(Item("v_max_f32 %s, 0, %s" % (a, a)))
                        else:
                            it.append(Item("v_fma_f32 %s, %s, %%[slope], 0" % (T1, a)))
                            it.append(Item("v_max_f32 %s, %s, %s" % (a, a, T1)))
                        if mode == "fwd":
                            it.append(Item("v_cmp_lt_f32 vcc, 0, %s" % a))
                            it.append(Item("v_addc_co_u32 %s, vcc, %s, %s, vcc" % (W[i], W[i], W[i])))
                    else:
                        it.append(Item("v_mul_f32 %s, %s, %s" % (a, a, INV[i]), need=[("inv", i)]))
                        it.append(Item("v_bfe_i32 %s, %%[word], %d, 1" % (T1, 31 - n)))
                        if act == "relu":
                            it.append(Item("v_and_b32 %s, %s, %s" % (a, a, T1)))
                        else:
                            it.append(Item("v_mul_f32 %s, %%[slope], %s" % (T2, a)))
                            it.append(Item("v_bfi_b32 %s, %s, %s, %s" % (a, T1, a, T2)))
                first = gp == 0 and gg == 0
                for c in (0, 2):
                    it.append(Item("v_max3_f32 %s, |%s|, |%s|, %s" % (LM[i], regs[c], regs[c + 1], "0" if first and c == 0 else LM[i])))
    it.append(Item("v_add_u32 %s, %%[wn256], %%[invad]" % ADW))      # (this wave's slice of the partial-maxima table: 256 bytes per wave)
    for i in range(2):
        it += [Item("v_mov_b32 %s, %s" % (SWP, LM[i])), Item("s_nop 1", kind="salu"),
               Item("v_permlane32_swap_b32 %s, %s" % (LM[i], SWP)), Item("v_max_f32 %s, %s, %s" % (LM[i], LM[i], SWP)),
               Item("ds_write_b32 %s, %s offset:%d" % (ADW, LM[i], PMD + 128 * i), lds=("pmw", i), kind="lds")]
    if mode == "fwd":
        it += [Item("v_lshl_or_b32 %%[word], %s, 16, %s" % (W[0], W[1])), Item("v_lshrrev_b32 %s, 2, %%[wv]" % T1),
               Item("global_store_dword %s, %%[word], %%[sgn]" % T1, kind="vmem")]
    return it


def p2_stream(acc_u, mode, one=False, hrows=False):
    """hrows (parity arithmetic only): the rows a training run keeps for its weight gradients leave as f16 rows -- the hi plane's bytes, as in the
    one-product mode (ChainLayer::c_half) -- instead of fp32 rows; the arithmetic of the run itself (three products, both planes) is untouched."""
    it = []
    PM = G[0:8]
    MX, E, T = G[8], G[9], G[10]
    SC, IN = [G[11], G[12]], [G[13], G[14]]
    train = mode != "inf"
    K140 = G[15]                                # (v_cndmask takes its constant from a register: a literal next to vcc is two constant-bus reads)
    it.append(Item("v_mov_b32 %s, 0x8c" % K140))
    PMD = pm_delta(acc_u)
    assert PMD % 256 == 0
    for i in range(2):
        src = "%[invad]"
        if i == 1:
            it.append(Item("v_add_u32 %s, 128, %%[invad]" % AD))
            src = AD
        for j in range(4):
            it.append(Item("ds_read2st64_b32 %s, %s offset0:%d offset1:%d" % (vt(GB + 2 * j, 2), src, PMD // 256 + 2 * j, PMD // 256 + 2 * j + 1), lds=("pm", i, j), kind="lds"))
        it.append(Item("v_max3_f32 %s, %s, %s, %s" % (MX, PM[0], PM[1], PM[2]), need=[("pm", i, 0), ("pm", i, 1)]))
        it.append(Item("v_max3_f32 %s, %s, %s, %s" % (MX, MX, PM[3], PM[4]), need=[("pm", i, 2)]))
        it.append(Item("v_max3_f32 %s, %s, %s, %s" % (MX, MX, PM[5], PM[6]), need=[("pm", i, 3)]))
        it.append(Item("v_max_f32 %s, %s, %s" % (MX, MX, PM[7])))
        if train:
            it.append(Item("global_store_dword %%[invad], %s, %%[rmp] offset:%d" % (MX, 128 * i), kind="vmem"))      # (rmp: the row maxima's address minus the table's LDS offset)
        # scale_from_max (chain4.hip): e = bits ? exponent : 140; scale = 2^(267 - e), 1/scale = 2^(e - 13), biased exponents clamped to [1, 254]
        it += [Item("v_bfe_u32 %s, %s, 23, 8" % (E, MX)), Item("v_cmp_ne_u32 vcc, 0, %s" % MX), Item("v_cndmask_b32 %s, %s, %s, vcc" % (E, K140, E)),
               Item("v_sub_u32 %s, 0x10b, %s" % (T, E)), Item("v_med3_i32 %s, %s, 1, %%[c254]" % (T, T)), Item("v_lshlrev_b32 %s, 23, %s" % (SC[i], T)),
               Item("v_add_u32 %s, -13, %s" % (T, E)), Item("v_med3_i32 %s, %s, 1, %%[c254]" % (T, T)), Item("v_lshlrev_b32 %s, 23, %s" % (IN[i], T)),
               Item("ds_write_b32 %%[invad], %s offset:%d" % (IN[i], 128 * i), lds=("invw", i), kind="lds")]
    if one:
        # ---- one-product mode: hi planes only; the f16 rows the weight gradient reads (ChainLayer::c_half) ARE the hi plane's bytes (each row
        # times its power-of-two scale): written to the planes, read back sixteen rows per instruction (four lanes per row's 64 bytes), stored
        H = ["v%d" % (GB + j) for j in range(4)]
        for i in range(2):
            for q in range(2):
                regs = ["v%d" % (acc_u + 16 * i + 8 * q + e) for e in range(8)]
                for j in range(4):
                    it.append(Item("v_fma_mixlo_f16 %s, %s, %s, 0" % (H[j], regs[2 * j], SC[i])))
                for j in range(4):
                    it.append(Item("v_fma_mixhi_f16 %s, %s, %s, 0" % (H[j], regs[2 * j + 1], SC[i])))
                dst = "%[plw]"
                if q:
                    it.append(Item("v_xor_b32 %s, 16, %%[plw]" % AD))
                    dst = AD
                it.append(Item("ds_write_b128 %s, %s offset:%d" % (dst, vt(GB, 4), 32768 * i), lds=("plh", i, q), kind="lds"))
                it.append(Item("s_nop 1", kind="salu"))
        if train:
            R = [vt(GB, 4), vt(GB + 4, 4)]          # (the split is done: its registers are free)
            for sx in range(4):
                src = "%[rdb]"
                if sx:
                    it.append(Item("v_xor_b32 %s, 0x%x, %%[rdb]" % (AD, 528 * sx)))
                    src = AD
                it.append(Item("ds_read_b128 %s, %s" % (R[sx & 1], src), lds=("rb", sx), kind="lds"))
                it.append(Item("global_store_dwordx4 %%[gso], %s, %%[crow0] offset:%d%s" % (R[sx & 1], 512 * sx, STMOD), need=[("rb", sx)], kind="vmem"))
        return it
    if train and not hrows:
        # ---- the rows: into the wave's own 128 bytes of every plane row (where its split goes afterwards) ...
        for i in range(2):
            for g in range(4):
                dst = "%[stw]"
                if g:
                    it.append(Item("v_xor_b32 %s, %d, %%[stw]" % (AD, 16 * g)))
                    dst = AD
                it.append(Item("ds_write_b128 %s, %s offset:%d" % (dst, vt(acc_u + 16 * i + 4 * g, 4), 32768 * i), lds=("stw", i, g), kind="lds"))
        # ... and back, eight lanes per row; row r = (sx & 3) + 4 (q & 3) + 16 (q >> 2) + 32 (sx >> 2) of the tile
        R = [vt(GB, 4), vt(GB + 4, 4)]
        for sx in range(8):
            src = "%[rdb]"
            if sx & 3:
                it.append(Item("v_xor_b32 %s, 0x%x, %%[rdb]" % (AD2, 528 * (sx & 3))))
                src = AD2
            it.append(Item("ds_read_b128 %s, %s offset:%d" % (R[sx & 1], src, 32768 * (sx >> 2)), lds=("rb", sx), kind="lds"))
            it.append(Item("global_store_dwordx4 %%[gso], %s, %%[crow%d] offset:%d%s" % (R[sx & 1], sx >> 2, 1024 * (sx & 3), STMOD), need=[("rb", sx)], kind="vmem"))
    # ---- split into the planes
    HS = [[["v%d" % (GB + 4 * s + j) for j in range(4)] for s in range(2)]] * 2      # [set][hi / lo][4] (one set: registers are scarce)
    k = 0
    for i in range(2):
        for q in range(2):
            H, L = HS[k & 1]
            k += 1
            regs = ["v%d" % (acc_u + 16 * i + 8 * q + e) for e in range(8)]
            for j in range(4):
                it.append(Item("v_fma_mixlo_f16 %s, %s, %s, 0" % (H[j], regs[2 * j], SC[i])))
            for j in range(4):
                it.append(Item("v_fma_mixhi_f16 %s, %s, %s, 0" % (H[j], regs[2 * j + 1], SC[i])))
            for j in range(4):
                it.append(Item("v_fma_mixlo_f16 %s, %s, %s, -%s op_sel_hi:[0,0,1]" % (L[j], regs[2 * j], SC[i], H[j])))
            for j in range(4):
                it.append(Item("v_fma_mixhi_f16 %s, %s, %s, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (L[j], regs[2 * j + 1], SC[i], H[j])))
            dst = "%[plw]"
            if q:
                it.append(Item("v_xor_b32 %s, 16, %%[plw]" % AD))
                dst = AD
            h0, l0 = int(H[0][1:]), int(L[0][1:])
            it.append(Item("ds_write_b128 %s, %s offset:%d" % (dst, vt(h0, 4), 32768 * i), lds=("plh", i, q), kind="lds"))
            it.append(Item("ds_write_b128 %s, %s offset:%d" % (dst, vt(l0, 4), 32768 * i + 4096), lds=("pll", i, q), kind="lds"))
            it.append(Item("s_nop 1", kind="salu"))          # (a wide store's data registers must not be rewritten right behind it)
    if train and hrows:
        # ---- f16 rows for the weight gradients: the hi plane's bytes, read back sixteen rows per instruction (four lanes per row's 64 bytes of this
        # wave's columns) -- the one-product mode's stores, character for character (rdb / gso / crow0 are the caller's in that mode's form)
        R = [vt(GB, 4), vt(GB + 4, 4)]
        for sx in range(4):
            src = "%[rdb]"
            if sx:
                it.append(Item("v_xor_b32 %s, 0x%x, %%[rdb]" % (AD, 528 * sx)))
                src = AD
            it.append(Item("ds_read_b128 %s, %s" % (R[sx & 1], src), lds=("rb", sx), kind="lds"))
            it.append(Item("global_store_dwordx4 %%[gso], %s, %%[crow0] offset:%d%s" % (R[sx & 1], 512 * sx, STMOD), need=[("rb", sx)], kind="vmem"))
    return it


SC_DELTA = 20992          # LDS bytes from a tile's 1 / scale table to its scale table (chain4.hip: C4_OFF_XMAX - C4_OFF_INV; static_assert there)


def sign_pos(c, j, h):
    """bit of chain.h's one-product sign word that holds value (chunk c = 2 q + i, pair j, half h) of a lane: the word collects one packed register
    per step, t = 4 c + j = 0 .. 15, `word = (word << 1) | (bits 0 and 16 of the step's register)`"""
    return (15 - (4 * c + j)) + 16 * h


def p_stream_one(acc_u, mode, act):
    """The row phase of the one-product mode (round 6): ONE stream, no barrier.  The run carries one power-of-two scale per ROW (chosen when the tile's
    input rows are staged: chain4.hip, one_scale_from_max) through all of its layers, so a middle layer's row phase never needs the row's maximum:
      forward   y s = acc + s b  (acc = s x W^T: one fma per value), packed pairs to f16 (v_cvt_pk_f16_f32, RTNE), ReLU on the f16 bit patterns
                (v_pk_max_i16 with 0: +0 for everything with the sign bit), sign bits from the packed halfs (v_pk_min_u16 with 1: one instruction
                per PAIR, in place behind the LDS write), hi plane = the next layer's operand = the f16 row the weight gradient reads;
      data-gradient  (g s) masked by the sign word in fp32, then the same conversion.
    96 / 80 vector instructions per statement against 220 / 184 of the per-layer-scale form it replaces, and the eight waves of the workgroup meet only at
    the end of the slot."""
    it = []
    SC = ["v45", "v47"]
    B = G[0:8]
    HS = [G[8:12], G[12:16]]
    fwd = mode != "dgrad"
    train = mode != "inf"
    T, T2 = G[0], G[1]                          # (data-gradient: no biases)
    if fwd:
        it.append(Item("ds_read_b32 %s, %%[invad] offset:%d" % (SC[0], SC_DELTA), lds=("sc", 0), kind="lds"))
        it.append(Item("ds_read_b32 %s, %%[invad] offset:%d" % (SC[1], SC_DELTA + 128), lds=("sc", 1), kind="lds"))

    def signs(c):
        H = HS[c & 1]
        out = []
        for j in range(4):
            if act == "relu":
                out.append(Item("v_pk_min_u16 %s, %s, %%[k11]" % (H[j], H[j])))      # (k11 = 0x00010001 in a scalar register: an inline 1 reaches the low half only)
            else:
                out += [Item("v_pk_min_i16 %s, %s, %%[k11]" % (H[j], H[j])), Item("v_pk_max_i16 %s, %s, 0" % (H[j], H[j]))]
            out.append(Item("v_lshl_or_b32 %%[word], %%[word], 1, %s" % H[j]))
        return out

    for q in range(2):
        if fwd:
            for gg in range(2):
                it.append(Item("ds_read_b128 %s, %%[biasad] offset:%d" % (vt(GB + 4 * gg, 4), 32 * q + 16 * gg), lds=("b", q, gg), kind="lds"))
        for i in range(2):
            c = 2 * q + i
            H = HS[c & 1]
            regs = ["v%d" % (acc_u + 16 * i + 8 * q + e) for e in range(8)]
            for e in range(8):
                a = regs[e]
                if fwd:
                    it.append(Item("v_fma_f32 %s, %s, %s, %s" % (a, B[e], SC[i], a), need=[("sc", i), ("b", q, e // 4)]))
                    if act != "relu":
                        it.append(Item("v_fma_f32 %s, %s, %%[slope], 0" % (H[e // 2], a)))
                        it.append(Item("v_max_f32 %s, %s, %s" % (a, a, H[e // 2])))
                else:
                    it.append(Item("v_bfe_i32 %s, %%[word], %d, 1" % (T, sign_pos(c, e // 2, e & 1))))
                    if act == "relu":
                        it.append(Item("v_and_b32 %s, %s, %s" % (a, a, T)))
                    else:
                        it.append(Item("v_mul_f32 %s, %%[slope], %s" % (T2, a)))
                        it.append(Item("v_bfi_b32 %s, %s, %s, %s" % (a, T, a, T2)))
            for j in range(4):
                it.append(Item("v_cvt_pk_f16_f32 %s, %s, %s" % (H[j], regs[2 * j], regs[2 * j + 1])))
            if fwd and act == "relu":
                for j in range(4):
                    it.append(Item("v_pk_max_i16 %s, %s, 0" % (H[j], H[j])))
            it.append(Item("ds_write_b128 %s, %s offset:%d" % ("%[plw2]" if q else "%[plw]", vt(int(H[0][1:]), 4), 32768 * i), lds=("plh", i, q), kind="lds"))
            if mode == "fwd" and c > 0:        # the sign bits of the chunk BEFORE, in place in its registers (its LDS write is long gone)
                it += signs(c - 1)
    if mode == "fwd":
        it += [Item("s_nop 1", kind="salu")] + signs(3)
        it += [Item("v_lshrrev_b32 %s, 2, %%[wv]" % SC[1]), Item("global_store_dword %s, %%[word], %%[sgn]" % SC[1], kind="vmem")]
    if train:
        # the f16 rows the weight gradient reads (ChainLayer::c_half) ARE the hi plane's bytes: read back sixteen rows per instruction (four lanes per
        # row's 64 bytes of this wave's columns), stored
        AD1 = SC[0]
        R = [vt(GB + 4 * sx, 4) for sx in range(4)]       # (every temporary is free by now: the four reads go out together, the stores follow as the rows arrive)
        for sx in range(4):
            src = "%[rdb]"
            if sx:
                it.append(Item("v_xor_b32 %s, 0x%x, %%[rdb]" % (AD1, 528 * sx)))
                src = AD1
            it.append(Item("ds_read_b128 %s, %s" % (R[sx], src), lds=("rb", sx), kind="lds"))
        for sx in range(4):
            it.append(Item("global_store_dwordx4 %%[gso], %s, %%[crow0] offset:%d%s" % (R[sx], 512 * sx, STMOD), need=[("rb", sx)], kind="vmem"))
    return it


class Emit:
    def __init__(self):
        self.lines, self.queue, self.retired = [], [], 0

    def put(self, item):
        need = [t for t in item.need if t in self.queue[self.retired:]]
        if need:
            last = max(self.queue.index(t) for t in need)
            after = len(self.queue) - 1 - last
            self.lines.append("s_waitcnt lgkmcnt(%d)" % min(after, 15))
            self.retired = max(self.retired, len(self.queue) - min(after, 15))
        self.lines.append(item.text)
        if item.lds is not None:
            self.queue.append(item.lds)

    def wait_for(self, tags):
        self.put(Item("", need=tags))
        self.lines.pop()


def build(tile, mode, act, ld, kcnt=16, one=False, hrows=False):
    acc_t, acc_u = (ACC["X"], ACC["Y"]) if tile == "X" else (ACC["Y"], ACC["X"])
    pro, steps = k_stream_one(acc_t, ld, kcnt) if one else (k_stream_exp(acc_t, ld, kcnt) if (KVAR and kcnt == 16) else k_stream(acc_t, ld, kcnt))
    NM = len(steps)                             # matrix instructions of the statement
    if one:
        return build_one(pro, steps, p_stream_one(acc_u, mode, act))
    p1, p2 = p1_stream(acc_u, mode, act), p2_stream(acc_u, mode, one, hrows)
    if ABLATE == "K":                           # (timing experiment, results wrong: the k-loop alone)
        p1, p2 = [], []
    elif ABLATE == "KP1":
        p2 = []
    elif ABLATE == "KP2":
        p1 = []
    if ABLATE == "NOST":                        # (timing experiment, results wrong: the hot slots' row stores never leave -- what do the stores cost, and is it them the phase shift spreads?)
        p2 = [x for x in p2 if not x.text.startswith("global_store_dwordx4")]
    e = Emit()
    for x in pro:
        e.put(x)
    head = 8 if mode != "dgrad" else 2          # the row phase's first LDS reads ride in front of the first MFMA
    for x in p1[:head]:
        e.put(x)
    p1 = p1[head:]
    n1 = max(1, min(NM - 6, round(NM * (len(p1) + 6.0) / (len(p1) + len(p2) + 6))))      # MFMAs that carry P1
    def prio(tag, older, younger):
        return ["s_cmp_lg_u32 %[role], 0", "s_cbranch_scc1 .Lp%%=_%sa" % tag, "s_setprio %d" % younger, "s_branch .Lp%%=_%sb" % tag,
                ".Lp%%=_%sa:" % tag, "s_setprio %d" % older, ".Lp%%=_%sb:" % tag]
    for m, (mf, post) in enumerate(steps):
        if PRIO and m == 0:
            e.lines += prio("s", 0, 1)
        if PRIO == "half" and m == NM // 2:
            e.lines += prio("m", 1, 0)
        e.put(mf)
        for x in post:
            e.put(x)
        if m < n1:
            left = n1 - m
            take = -(-len(p1) // left)
            for x in p1[:take]:
                e.put(x)
            p1 = p1[take:]
            if m == n1 - 1:
                assert not p1
                e.wait_for([("pmw", 0), ("pmw", 1)])
                e.lines.append("s_barrier")
        else:
            left = NM - m
            take = -(-len(p2) // left)
            for x in p2[:take]:
                e.put(x)
            p2 = p2[take:]
    assert not p2
    if PRIO:
        e.lines.append("s_setprio 0")
    e.lines += ["s_nop 15", "s_nop 7"]          # the last results leave the matrix pipe 16 passes after issue
    return e.lines


def build_one(pro, steps, p):
    """one-product statement: the k-loop's matrix instructions with the row phase's single stream spread evenly behind them"""
    NM = len(steps)
    if ABLATE == "K":
        p = []
    e = Emit()
    for x in pro:
        e.put(x)
    head = 4
    for x in p[:head]:                          # the row phase's first LDS reads ride in front of the first MFMA
        e.put(x)
    p = p[head:]
    for m, (mf, post) in enumerate(steps):
        e.put(mf)
        for x in post:
            e.put(x)
        take = -(-len(p) // (NM - m))
        for x in p[:take]:
            e.put(x)
        p = p[take:]
    assert not p
    e.lines += ["s_nop 15", "s_nop 7"]          # the last results leave the matrix pipe 16 passes after issue
    return e.lines


def build_pair(mode, act, one=False, hrows=False):
    """TWO hot slots in one statement: slot (tile X, step i) and slot (tile Y, step i) -- the two k-loops of a step, the row phases of Y's step i - 1
    and X's step i beside them.  Between two statements every wave spends ~1.7k cycles in compiled C++ (which slot is next, its descriptors, ~40
    lane-derived operands recomputed because nothing lane-derived may live across a statement: at two waves per SIMD every dependent instruction costs
    its full latency) with the matrix pipe idle; between the halves of this statement the same step costs a barrier and seven additions: the second
    half's lane-derived operands are the first half's plus constants (the other tile's planes: +- 64 KB; its tables: - 256 bytes; the next layer's
    biases: dbias), its pointers -- another layer's rows, maxima and sign words -- second operands (sgnb, rmpb, crow0b, crow1b).
    WHY THIS PAIR and not (Y, step i) + (X, step i + 1), which was built first and measured nothing: tile X's slot starts behind a vmcnt(0) for the next
    step's weight fragments, the last of which are requested at the very end of tile Y's k-loop (they refill their registers in place) -- between two
    statements the compiled C++ hides that latency, inside one statement it would stand exposed.  Tile Y multiplies with the weights tile X has just
    used: nothing to wait for.  The halves are the single statements' instruction lists, character for character."""
    a = build("X", mode, act, 0, 16, one, hrows)
    b = build("Y", mode, act, 1, 16, one, hrows)
    ren = lambda l: l.replace("%[sgn]", "%[sgnb]").replace("%[rmp]", "%[rmpb]").replace("%[crow0]", "%[crow0b]").replace("%[crow1]", "%[crow1b]")
    mid = ["s_waitcnt lgkmcnt(0)", "s_barrier",          # = lds_barrier(): X's step is in its accumulators, Y's rows of the step before are split into its planes
           "v_add_u32 %[pbx], 0x10000, %[pbx]",         # T: tile X -> tile Y
           "v_add_u32 %[stw], 0xffff0000, %[stw]", "v_add_u32 %[rdb], 0xffff0000, %[rdb]", "v_add_u32 %[plw], 0xffff0000, %[plw]",      # U: tile Y -> tile X
           "v_add_u32 %[invad], 0xffffff00, %[invad]",  # U's 1 / scale table (and, through the instructions' offsets, its partial maxima)
           "v_add_u32 %[biasad], %[dbias], %[biasad]"]  # the biases of the layer X has just multiplied
    if one:
        mid.append("v_add_u32 %[plw2], 0xffff0000, %[plw2]")
    if mode == "fwd":
        mid.append("v_mov_b32 %[word], 0")
    if mode == "dgrad":                         # the second half's sign word (tile X's, of the layer it has just multiplied) is a second operand
        ren0 = ren
        ren = lambda l: ren0(l).replace("%[word]", "%[wordb]")
    return a + mid + [ren(l) for l in b]


def emit(name, lines):
    print("#define %s \\" % name)
    for i, l in enumerate(lines):
        print('    "%s\\n\\t"%s' % (l, " \\" if i + 1 < len(lines) else ""))
    print()


def classify(line):
    op = line.split()[0]
    if op.startswith("v_mfma"): return "mfma"
    if op.startswith("ds_"): return "lds"
    if op.startswith(("global_", "buffer_", "flat_")): return "vmem"
    if op == "s_waitcnt": return "wait"
    if op == "s_barrier": return "barrier"
    if op == "s_nop": return "nop"
    if op.startswith("s_"): return "salu"
    if op.startswith("v_accvgpr"): return "acc_mov"
    return "valu"


def budget():
    """per statement: instructions by kind and by stream (k-loop alone; + first half of the row phase; + second half), non-MFMA per MFMA --
    python scripts/gen_chain4_fused.py --budget > profiles/rNN_chain4_statement_budget.txt"""
    global ABLATE
    kinds = ["valu", "acc_mov", "lds", "vmem", "salu", "wait", "nop", "barrier"]
    def count(lines):
        c = dict.fromkeys(["mfma"] + kinds, 0)
        for l in lines:
            c[classify(l)] += 1
        return c
    print("Instruction budget of the hot-slot statements of chain4.hip (scripts/gen_chain4_fused.py --budget; one wave, one slot = one statement).")
    print("Streams: K = the k-loop of tile T (MFMAs, fragment reads, their waits); P1 / P2 = first / second half of tile U's row phase, counted as")
    print("(statement built with that stream) - (statement with the k-loop alone): waits the interleaving adds are charged to the stream that needs them.")
    print("%-26s %5s | %-38s | %-38s | %-38s | %s" % ("statement", "MFMA", "K: valu accmov lds vmem salu wait nop", "P1: valu accmov lds vmem salu wait nop", "P2: valu accmov lds vmem salu wait nop", "non-MFMA per MFMA (all / valu only)"))
    for one in (False, True):
        for mode in ("fwd", "inf", "dgrad"):
            for act in ("relu", "leaky"):
                for tile, ld in (("X", 0), ("Y", 0), ("Y", 1)):
                    saved = ABLATE
                    c = {}
                    for ab in ("K", "KP1", "KP2", ""):
                        ABLATE = ab
                        c[ab] = count(build(tile, mode, act, ld, 16, one))
                    ABLATE = saved
                    k = c["K"]
                    p1 = {x: c["KP1"][x] - k[x] for x in k}
                    p2 = {x: c["KP2"][x] - k[x] for x in k}
                    full = c[""]
                    fmt = lambda d: " ".join("%5d" % d[x] for x in kinds[:7])
                    non = sum(full[x] for x in kinds)
                    vonly = full["valu"] + full["acc_mov"]
                    print("%-26s %5d | %-38s | %-38s | %-38s | %.2f / %.2f" % ("%s%s_%s_%s_%s" % ("h1:" if one else "", mode, act, tile, "LD" if ld else "NL"), full["mfma"], fmt(k), fmt(p1), fmt(p2), non / full["mfma"], vonly / full["mfma"]))


if __name__ == "__main__":
    if "--budget" in sys.argv:
        budget()
        sys.exit(0)
    print("// GENERATED by scripts/gen_chain4_fused.py -- do not edit.  The hot slots of chain4.hip, one asm statement each (see the script).")
    total = {}
    for mode in ("fwd", "inf", "dgrad"):
        for act in ("relu", "leaky"):
            for tile in ("X", "Y"):
                for ld in ((0,) if tile == "X" else (0, 1)):
                    lines = build(tile, mode, act, ld)
                    emit("C4F_%s_%s_%s_%s" % (mode.upper(), act.upper(), tile, "LD" if ld else "NL"), lines)
                    total[(mode, act, tile, ld)] = len(lines)
            for kcnt in (8, 10):                # the run's first layer on tile Y (on tile X its slot also stages the next tile: no fused form)
                emit("C4F_%s_%s_Y_LD_K%d" % (mode.upper(), act.upper(), kcnt), build("Y", mode, act, 1, kcnt))
            emit("C4F2_%s_%s" % (mode.upper(), act.upper()), build_pair(mode, act))
            emit("C4F21_%s_%s" % (mode.upper(), act.upper()), build_pair(mode, act, True))      # one-product mode
            # one-product mode (h1)
            for tile in ("X", "Y"):
                for ld in ((0,) if tile == "X" else (0, 1)):
                    emit("C4F1_%s_%s_%s_%s" % (mode.upper(), act.upper(), tile, "LD" if ld else "NL"), build(tile, mode, act, ld, 16, True))
            for kcnt in (8, 10):
                emit("C4F1_%s_%s_Y_LD_K%d" % (mode.upper(), act.upper(), kcnt), build("Y", mode, act, 1, kcnt, True))
            # parity arithmetic, f16 rows for the weight gradients (mode PAPR_MLP_H3_F16ROWS): training forward and data-gradient
            if mode != "inf":
                for tile in ("X", "Y"):
                    for ld in ((0,) if tile == "X" else (0, 1)):
                        emit("C4FH_%s_%s_%s_%s" % (mode.upper(), act.upper(), tile, "LD" if ld else "NL"), build(tile, mode, act, ld, 16, False, True))
                for kcnt in (8, 10):
                    emit("C4FH_%s_%s_Y_LD_K%d" % (mode.upper(), act.upper(), kcnt), build("Y", mode, act, 1, kcnt, False, True))
                emit("C4F2H_%s_%s" % (mode.upper(), act.upper()), build_pair(mode, act, False, True))
    print("#define C4F_CLOBBERS " + ", ".join('"%s"' % v for v in CLOB_V) + ', "vcc", "memory"')
    print('#define C4F_LD_CLOBBERS "s90", "s91", "s92", "s93", "s94", "s95", "s96", "s97"')
    print("#define C4F_AGPRS " + ", ".join('"a%d"' % i for i in range(128)))
    print("// instructions per statement: " + ", ".join("%s_%s_%s_%s %d" % (k[0], k[1], k[2], "LD" if k[3] else "NL", v) for k, v in sorted(total.items())))
