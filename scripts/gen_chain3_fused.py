#!/usr/bin/env python3
"""Generates papr_amd/csrc/chain3_fused.inc: the hot SLOT of chain3.hip -- the k-loop of one tile AND the row phases of the other
tile -- as ONE inline-asm statement per variant, the two instruction streams interleaved by hand.

Why: a wave that multiplies keeps the matrix pipe busy and its vector / scalar / LDS issue slots idle; a wave in its row phases the
reverse.  chain3.hip first overlapped the two by giving SIMD partners opposite orders (multiply | rows), 13.2k cycles per slot
for 6.1k of matrix-pipe time.  Here every wave does both at once: behind each MFMA (32 cycles of pipe) come its one memory
instruction and four or five instructions of the row phases, which are independent of the multiplication inside a slot.  The two
waves of a SIMD then share the pipe (6.1k per slot) and hide each other's waits.

Streams
  K     sixteen k-steps of six MFMAs (hi.lo, lo.hi, hi.hi for two 32-row tiles); A fragments in ONE buffer, each register re-read
        for the next k-step right behind its last use (a wave gets the pipe every other slot here: four to six MFMA slots of
        lead are 250-380 cycles); weights in a[0:127], refilled with the next layer's behind their last use (LD variants).
  rows  two batches of four rows of the other tile: read the raw accumulator image (fp32 overlay), bias / activation (forward) or
        1/scale and activation derivative from the sign word (data-gradient), store, sign bits, row maximum by DPP, power-of-two
        scale on the scalar unit, split into hi / lo planes, plane writes -- the arithmetic of chain3.hip's C++ row phases,
        instruction for instruction (bit-identical results).
All LDS traffic of both streams goes through one in-order queue; every wait is computed by simulating that queue.

Registers: operands for what crosses the statement (accumulators, sign word, addresses, pointers); everything else lives in fixed
registers named as clobbers (an asm statement has at most 30 operands): v72-v127, s[72:99].

  python scripts/gen_chain3_fused.py > papr_amd/csrc/chain3_fused.inc
"""
import sys

# ---- fixed registers
F = {"l0": "v[84:87]", "l1": "v[88:91]", "h0": "v[92:95]", "h1": "v[96:99]"}
KAD = "v100"
R = [[("v%d" % (104 + 4 * q + e)) for e in range(4)] for q in range(4)]          # rows data
RT = ["v[%d:%d]" % (104 + 4 * q, 107 + 4 * q) for q in range(4)]
MX = ["v%d" % (120 + q) for q in range(4)]
INV = ["v%d" % (124 + q) for q in range(4)]
INVT = "v[124:127]"
TS = [["v76", "v77", "v78", "v79"], ["v72", "v73", "v74", "v75"]]                  # split temporaries, two sets
TA, TB, TC, TD, TI, TL = "v80", "v81", "v82", "v83", "v101", "v102"              # plane address, read address, two scratch, inv_tab address, lane*4
SMX = ["s%d" % (72 + q) for q in range(4)]                                         # row maxima (bits)
SSC = ["s%d" % (76 + q) for q in range(4)]                                         # scales
SIN = ["s%d" % (80 + q) for q in range(4)]                                         # 1 / scales
SE, ST = "s84", "s85"
SC = ["s[%d:%d]" % (86 + 2 * q, 87 + 2 * q) for q in range(4)]                    # compare masks
SST = ("s94", "s95")                                                              # running store pointer
SBH, SBL = ("s96", "s97"), ("s98", "s99")                                         # weight bases
CLOB_V = ["v%d" % i for i in range(72, 128)]
CLOB_S = ["s%d" % i for i in range(70, 100)]
TH = "v84"                                                                        # one-product variants with f16 row stores: lane * 8 (the lo fragments' registers are free there)
SK, SR = "s70", "s71"                                                             # constants beyond the inline range (VOP3 takes no literals on gfx9)


def xad(dst, a, const, c, sreg):
    """dst = (a ^ const) + c as instruction texts"""
    if 0 <= const <= 64:
        return ["v_xad_u32 %s, %s, %d, %s" % (dst, a, const, c)]
    return ["s_movk_i32 %s, %d" % (sreg, const), "v_xad_u32 %s, %s, %s, %s" % (dst, a, sreg, c)]


class Emit:
    def __init__(self):
        self.lines = []
        self.queue = []             # ids of LDS operations issued, oldest first (the hardware retires them in order)
        self.done = set()
        self.n = 0

    def raw(self, s):
        self.lines.append(s)

    def lds(self, s):               # an LDS instruction; returns its id
        self.n += 1
        self.queue.append(self.n)
        self.lines.append(s)
        return self.n

    def need(self, ids):            # the data of these LDS operations is needed by the next instruction
        ids = [i for i in ids if i is not None and i not in self.done]
        if not ids:
            return
        last = max(self.queue.index(i) for i in ids)
        after = len(self.queue) - 1 - last
        self.lines.append("s_waitcnt lgkmcnt(%d)" % min(after, 15))
        if after <= 15:
            for i in self.queue[: last + 1]:
                self.done.add(i)
            self.queue = self.queue[last + 1:]
        else:                       # (the counter saturates at 15: at least the oldest len - 15 are retired)
            k = len(self.queue) - 15
            for i in self.queue[:k]:
                self.done.add(i)
            self.queue = self.queue[k:]
            self.need(ids)


def wh(ks): return "a[%d:%d]" % (8 * ks, 8 * ks + 3)
def wl(ks): return "a[%d:%d]" % (8 * ks + 4, 8 * ks + 7)


def k_stream(ld, one=False):
    """list of slots; a slot = (needs, mfma text, post: list of ('lds', text, key) / ('raw', text)).  one: the one-product mode (h1):
    hi . hi only, two MFMAs per k-step"""
    slots = []
    if one:
        rd_off = {"h0": 0, "h1": 32768}
        for ks in range(16):
            nx = ks + 1 if ks < 15 else None
            def rd1(kind): return ("lds", "ds_read_b128 %s, %s offset:%d" % (F[kind], KAD, rd_off[kind] + (256 if nx >= 8 else 0)), (kind, nx))
            def mf1(acc, kind): return "v_mfma_f32_32x32x16_f16 %%[%s], %s, %s, %s" % (acc, wh(ks), F[kind], "0" if ks == 0 else "%%[%s]" % acc)
            pre = [("raw", t) for t in xad(KAD, "%[axr]", (nx & 7) * 32, "%[pb]", SK)] if nx is not None else []
            slots.append(dict(pre=pre, need=[("h0", ks), ("h1", ks)], mf=mf1("a0", "h0"), post=[rd1("h0")] if nx is not None else []))
            p2 = [rd1("h1")] if nx is not None else []
            if ld:
                if ks % 4 == 0 and ks > 0:
                    p2 += [("raw", "s_add_u32 %s, %s, 4096" % (SBH[0], SBH[0])), ("raw", "s_addc_u32 %s, %s, 0" % (SBH[1], SBH[1]))]
                p2.append(("raw", "global_load_dwordx4 %s, %%[wv], s[96:97] offset:%d" % (wh(ks), (ks & 3) * 1024)))
            slots.append(dict(pre=[], need=[], mf=mf1("a1", "h1"), post=p2))
        return slots
    rd_off = {"l0": 4096, "l1": 36864, "h0": 0, "h1": 32768}
    def rd(kind, ks): return ("lds", "ds_read_b128 %s, %s offset:%d" % (F[kind], KAD, rd_off[kind] + (256 if ks >= 8 else 0)), (kind, ks))
    for ks in range(16):
        nx = ks + 1 if ks < 15 else None
        first = ks == 0
        def mf(acc, w, kind, z=False): return "v_mfma_f32_32x32x16_f16 %%[%s], %s, %s, %s" % (acc, w, F[kind], "0" if z else "%%[%s]" % acc)
        pre = []
        if nx is not None:
            pre += [("raw", t) for t in xad(KAD, "%[axr]", (nx & 7) * 32, "%[pb]", SK)]
        s = []
        # (one wait for the two lo fragments, one for the two hi ones: the second of each pair was requested one MFMA slot after the
        # first, and every s_waitcnt is an issue slot of a SIMD that is bound by instruction issue)
        s.append(dict(pre=pre, need=[("l0", ks), ("l1", ks)], mf=mf("a0", wh(ks), "l0", first), post=[rd("l0", nx)] if nx is not None else []))
        s.append(dict(pre=[], need=[], mf=mf("a1", wh(ks), "l1", first), post=[rd("l1", nx)] if nx is not None else []))
        p3 = []
        if ld and ks >= 1:
            if (ks - 1) % 4 == 0 and ks - 1 > 0:
                p3 += [("raw", "s_add_u32 %s, %s, 4096" % (SBH[0], SBH[0])), ("raw", "s_addc_u32 %s, %s, 0" % (SBH[1], SBH[1]))]
            p3.append(("raw", "global_load_dwordx4 %s, %%[wv], s[96:97] offset:%d" % (wh(ks - 1), ((ks - 1) & 3) * 1024)))
        s.append(dict(pre=[], need=[("h0", ks), ("h1", ks)], mf=mf("a0", wl(ks), "h0"), post=p3))
        p4 = []
        if ld:
            if ks % 4 == 0 and ks > 0:
                p4 += [("raw", "s_add_u32 %s, %s, 4096" % (SBL[0], SBL[0])), ("raw", "s_addc_u32 %s, %s, 0" % (SBL[1], SBL[1]))]
            p4.append(("raw", "global_load_dwordx4 %s, %%[wv], s[98:99] offset:%d" % (wl(ks), (ks & 3) * 1024)))
        s.append(dict(pre=[], need=[], mf=mf("a1", wl(ks), "h1"), post=p4))
        s.append(dict(pre=[], need=[], mf=mf("a0", wh(ks), "h0"), post=[rd("h0", nx)] if nx is not None else []))
        p6 = [rd("h1", nx)] if nx is not None else []
        if ld and ks == 15:
            p6.append(("raw", "global_load_dwordx4 %s, %%[wv], s[96:97] offset:3072" % wh(15)))
        s.append(dict(pre=[], need=[], mf=mf("a1", wh(ks), "h1"), post=p6))
        slots += s
    return slots


def rows_stream(mode, one=False, half=False):
    """mode: 'fwd' = training forward middle layer (store, sign bits, row maximum, planes), 'inf' = inference middle layer (planes only),
    'dgrad' = data-gradient middle layer.  half (one-product variants): the rows go to memory as f16 -- the hi plane of the split, 8 bytes
    per lane and row -- instead of fp32.  Returns a list of ops: ('raw', text) | ('lds', text, key) | ('need', [keys]) | ('group', [raw texts])"""
    ops = []
    A = ops.append
    A(("raw", "v_mov_b32 %s, %%[invb]" % TI))
    if mode != "inf":
        A(("raw", "v_lshrrev_b32 %s, 2, %%[wv]" % TL))
        if half:
            A(("raw", "v_lshrrev_b32 %s, 1, %%[wv]" % TH))
        A(("raw", "s_mov_b64 s[94:95], %[crow]"))
    for b in range(2):
        rows = [4 * b + q for q in range(4)]
        for q, u in enumerate(rows):
            for t in xad(TB, "%[rc]", (u & 7) * 16, "%[rdb]", SR):
                A(("raw", t))
            A(("lds", "ds_read_b128 %s, %s offset:%d" % (RT[q], TB, u * 512), ("row", u)))
        A(("lds", "ds_read_b128 %s, %s offset:%d" % (INVT, TI, b * 16), ("inv", b)))
        # ---- bias / activation (forward) or 1 / scale and derivative (data-gradient), store, sign bits
        for q, u in enumerate(rows):
            A(("need", [("row", u), ("inv", b)]))
            x = R[q]
            tm = TS[1]                  # four scratch registers (the split temporaries are free until the batch's splits)
            if mode == "dgrad":
                for e in range(4):
                    A(("raw", "v_mul_f32 %s, %s, %s" % (x[e], x[e], INV[q])))
                for e in range(4):      # keep x where the activation was positive (bit set), x * slope elsewhere; the word moves up four bits per row
                    A(("raw", "v_cmp_gt_i32 %s, 0, %%[sw]" % SC[e]))
                    A(("raw", "v_lshlrev_b32 %[sw], 1, %[sw]"))
                for e in range(4):
                    A(("raw", "v_mul_f32 %s, %%[slope], %s" % (tm[e], x[e])))
                for e in range(4):
                    A(("raw", "v_cndmask_b32 %s, %s, %s, %s" % (x[e], tm[e], x[e], SC[e])))
            else:
                be = ["%[b0]", "%[b1]", "%[b2]", "%[b3]"]
                for e in range(4):
                    A(("raw", "v_fma_f32 %s, %s, %s, %s" % (x[e], x[e], INV[q], be[e])))
                for e in range(4):
                    A(("raw", "v_fma_f32 %s, %s, %%[slope], 0" % (tm[e], x[e])))
                for e in range(4):
                    A(("raw", "v_max_f32 %s, %s, %s" % (x[e], x[e], tm[e])))
            if mode != "inf" and not half:
                A(("raw", "global_store_dwordx4 %%[wv], %s, s[94:95]" % RT[q]))
                A(("group", ["s_add_u32 s94, s94, %[ldcb]", "s_addc_u32 s95, s95, 0"]))         # (SCC between the two: never split)
            if mode == "fwd":
                for e in range(4):
                    A(("raw", "v_cmp_lt_f32 %s, 0, %s" % (SC[e], x[e])))
                for e in range(4):
                    A(("raw", "v_addc_co_u32 %%[sw], vcc, %%[sw], %%[sw], %s" % SC[e]))
            A(("raw", "v_max_f32 %s, |%s|, |%s|" % (MX[q], x[2], x[3])))
            A(("raw", "v_max3_f32 %s, |%s|, |%s|, %s" % (MX[q], x[0], x[1], MX[q])))
        # ---- row maxima: six DPP steps on the four rows at once (result in lane 63)
        A(("raw", "s_nop 1"))
        for ctrl in ("quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf", "row_half_mirror row_mask:0xf bank_mask:0xf",
                     "row_mirror row_mask:0xf bank_mask:0xf", "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"):
            A(("group", ["v_max_f32_dpp %s, %s, %s %s" % (m, m, m, ctrl) for m in MX]))
        A(("raw", "s_nop 1"))
        for q in range(4):
            A(("raw", "v_readlane_b32 %s, %s, 63" % (SMX[q], MX[q])))
        # ---- power-of-two scales on the scalar unit: ea = exponent of the maximum (140 for a zero row); scale = 2^(267 - ea - 127), 1 / scale = 2^(ea - 13 - 127)
        for q in range(4):
            A(("raw", "s_bfe_u32 %s, %s, 0x80017" % (SE, SMX[q])))
            A(("group", ["s_cmp_lg_u32 %s, 0" % SMX[q], "s_cselect_b32 %s, %s, 140" % (SE, SE)]))
            A(("raw", "s_sub_i32 %s, 267, %s" % (ST, SE)))
            A(("raw", "s_max_i32 %s, %s, 1" % (ST, ST)))
            A(("raw", "s_min_i32 %s, %s, 254" % (ST, ST)))
            A(("raw", "s_lshl_b32 %s, %s, 23" % (SSC[q], ST)))
            A(("raw", "s_add_i32 %s, %s, -13" % (ST, SE)))
            A(("raw", "s_max_i32 %s, %s, 1" % (ST, ST)))
            A(("raw", "s_min_i32 %s, %s, 254" % (ST, ST)))
            A(("raw", "s_lshl_b32 %s, %s, 23" % (SIN[q], ST)))
        if mode != "inf":               # the four maxima go out from lanes 0-3
            for q in range(4):
                A(("raw", "v_writelane_b32 %s, %s, %d" % (TC, SMX[q], q)))
            A(("group", ["s_mov_b64 exec, 15", "global_store_dword %s, %s, %%[rmp] offset:%d" % (TL, TC, 16 * b), "s_mov_b64 exec, -1"]))
        # ---- split into hi / lo planes (v_fma_mix: hi = f16(x s), lo = f16(x s - hi)), plane writes
        for q, u in enumerate(rows):
            t = TS[q & 1]
            x = R[q]
            s = SSC[q]
            A(("raw", "v_fma_mixlo_f16 %s, %s, %s, 0" % (t[0], x[0], s)))
            A(("raw", "v_fma_mixlo_f16 %s, %s, %s, 0" % (t[1], x[2], s)))
            A(("raw", "v_fma_mixhi_f16 %s, %s, %s, 0" % (t[0], x[1], s)))
            A(("raw", "v_fma_mixhi_f16 %s, %s, %s, 0" % (t[1], x[3], s)))
            if not one:
                A(("raw", "v_fma_mixlo_f16 %s, %s, %s, -%s op_sel_hi:[0,0,1]" % (t[2], x[0], s, t[0])))
                A(("raw", "v_fma_mixlo_f16 %s, %s, %s, -%s op_sel_hi:[0,0,1]" % (t[3], x[2], s, t[1])))
                A(("raw", "v_fma_mixhi_f16 %s, %s, %s, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (t[2], x[1], s, t[0])))
                A(("raw", "v_fma_mixhi_f16 %s, %s, %s, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (t[3], x[3], s, t[1])))
            A(("raw", "s_or_b32 %s, %%[wnb], %d" % (ST, u * 16)))
            A(("raw", "v_xad_u32 %s, %%[wp], %s, %%[wrb]" % (TA, ST)))
            tn = int(t[0][1:])
            if half:                    # (64 bits of store data: read at issue, the temporaries may be written again right away)
                A(("raw", "global_store_dwordx2 %s, v[%d:%d], s[94:95]" % (TH, tn, tn + 1)))
                A(("group", ["s_add_u32 s94, s94, %[ldcb]", "s_addc_u32 s95, s95, 0"]))
            if one:
                A(("lds", "ds_write_b64 %s, v[%d:%d] offset:%d" % (TA, tn, tn + 1, u * 512), ("pw", u)))
            else:
                A(("lds", "ds_write2st64_b64 %s, v[%d:%d], v[%d:%d] offset0:%d offset1:%d" % (TA, tn, tn + 1, tn + 2, tn + 3, u, u + 8), ("pw", u)))
        for q in range(4):
            A(("raw", "v_mov_b32 %s, %s" % (INV[q], SIN[q])))
        A(("lds", "ds_write_b128 %s, %s offset:%d" % (TI, INVT, b * 16), ("invw", b)))
    return ops


def fuse(mode, ld, one=False, half=False):
    em = Emit()
    ids = {}
    ks = k_stream(ld, one)
    ro = rows_stream(mode, one, half)
    # flatten groups so that they count as one unit of the rows stream
    units = []
    for o in ro:
        units.append(o)

    def emit_unit(o):
        if o[0] == "raw":
            em.raw(o[1])
        elif o[0] == "group":
            for t in o[1]:
                em.raw(t)
        elif o[0] == "lds":
            ids[o[2]] = em.lds(o[1])
        elif o[0] == "need":
            em.need([ids.get(k) for k in o[1]])

    if ld:
        em.raw("s_mov_b64 s[96:97], %[nh]")
        if not one:
            em.raw("s_mov_b64 s[98:99], %[nl]")
    # k-step 0's fragments, then the first instructions of the rows stream (its LDS reads) before the first MFMA
    em.raw("v_xad_u32 %s, %%[axr], 0, %%[pb]" % KAD)
    off = {"l0": 4096, "l1": 36864, "h0": 0, "h1": 32768}
    for kind in (("h0", "h1") if one else ("l0", "l1", "h0", "h1")):
        ids[(kind, 0)] = em.lds("ds_read_b128 %s, %s offset:%d" % (F[kind], KAD, off[kind]))
    pos = 0
    head = 0
    while head < len(units) and units[head][0] != "need":   # everything up to the first wait of the rows stream
        emit_unit(units[head]); head += 1
    pos = head
    weight = lambda o: (0.0 if o[0] == "need" else (len(o[1]) if o[0] == "group" else 1.0)) if True else 0
    total = sum(weight(o) for o in units[pos:])
    per = total / len(ks)
    budget = 0.0
    for sl in ks:
        for p in sl["pre"]:
            em.raw(p[1])
        em.need([ids.get(k) for k in sl["need"]])
        em.raw(sl["mf"])
        for p in sl["post"]:
            if p[0] == "lds":
                ids[p[2]] = em.lds(p[1])
            else:
                em.raw(p[1])
        budget += per
        while pos < len(units) and budget >= weight(units[pos]) - 1e-9:
            budget -= weight(units[pos])
            emit_unit(units[pos]); pos += 1
    while pos < len(units):
        emit_unit(units[pos]); pos += 1
    em.raw("s_waitcnt lgkmcnt(0)")
    em.raw("s_nop 15")
    em.raw("s_nop 7")
    return em.lines


def emit_macro(name, lines):
    print("#define %s \\" % name)
    for i, l in enumerate(lines):
        print('    "%s\\n\\t"%s' % (l, " \\" if i + 1 < len(lines) else ""))
    print()


if __name__ == "__main__":
    print("// GENERATED by scripts/gen_chain3_fused.py -- do not edit.  k-loop of one tile + row phases of the other, one asm statement (see the script).")
    for mode in ("fwd", "inf", "dgrad"):
        for ld in (True, False):
            emit_macro("C3_FUSED_%s_%s" % (mode.upper(), "LD" if ld else "NL"), fuse(mode, ld))
            if mode != "inf":
                emit_macro("C3_FUSED1_%s_%s" % (mode.upper(), "LD" if ld else "NL"), fuse(mode, ld, True))
                emit_macro("C3_FUSED1H_%s_%s" % (mode.upper(), "LD" if ld else "NL"), fuse(mode, ld, True, True))
    print("#define C3_FUSED_CLOBBERS " + ", ".join('"%s"' % r for r in CLOB_V + CLOB_S) + ', "vcc", "scc", "memory"')
    print("#define C3_FUSED_AGPRS " + ", ".join('"a%d"' % i for i in range(128)))
