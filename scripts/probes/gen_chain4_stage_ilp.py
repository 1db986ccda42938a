#!/usr/bin/env python3
"""EXPERIMENT (round 4, second form; measured: no gain -- not part of the build).  The second half of the staging of a tile's input rows in
chain4.hip (stage_finish) as three asm statements ordered for INSTRUCTION-LEVEL PARALLELISM: the eight rows of a wave go through each stage
together -- eight interleaved DPP chains, the rows' maxima gathered into the lanes of ONE register, the scale arithmetic once on that register as vector
instructions (lanes = rows), then per row eight independent v_fma_mix and two LDS writes.  193 instructions per wave against ~700 compiled ones and
against the 312 of the first form (gen_chain4_stage.py: the C++ loop's order, four rows after each other).

Hypothesis (after the first form measured nothing): the piece is a latency chain, so shortening the chain must shorten it.
Result: bit-identical (tests/test_hip_chain_variants.py with the statements wired in behind a switch), and again NOT faster: 11.78 / 11.77 against 11.79 /
11.84 ms per step in one box, run kernels within 1 %; in the cycle stamps the piece still takes 4.4-6.6k cycles per wave (slot 7: 23.1k against 23.9k;
mean of slots 7-8: inference 20.2k against 21.4k, training forward 24.0k against 22.3k -- the spread of two runs).  Whatever holds a staging slot at
~20-24k cycles, it is neither the number nor the dependency depth of the staging's own instructions.
Register note: with two or four rows in flight in the split statements the compiler ran out of its 64 registers beside the 32 of the rows and spilled
into a0 -- a weight fragment's register; papr_amd/build.py refuses such a listing.  One row at a time fits.
Wiring used: `#include "chain4_stage.inc"`; in stage_finish behind the LayerNorm core: mx[q] = max |.| of the lane's four values of row q (C++), then
    asm volatile(C4_STAGE_MAX : [m0] "+v"(mx[0]) ... [m7] "+v"(mx[7]), [smx] "=&v"(smxv), [sc] "=&v"(scv) : [rmp] "s"(rowmax0 + r0), [rmask] "s"(lanes 0 .. 7 whose
                 row exists), [l4] "v"(4 * lane), [tad] "v"(lds(inv_tab + 8 wn) + 4 * lane), [xad] "v"(lds(xmax_tab + 8 wn) + 4 * lane) : C4_STAGE_MAX_CLOBBERS);
    asm volatile(C4_STAGE_SPLIT_0 : : [a0] "v"(v[0].x) ... [a15] "v"(v[3].w), [sc] "v"(scv), [xw] "v"(wp ^ ((wn & 1) * 128)), [b0] "v"(lds(planes + wn * C4_BLK_BYTES) + wq),
                 [wm] "s"(lanes whose columns are < kpad) : C4_STAGE_SPLIT_CLOBBERS);      and C4_STAGE_SPLIT_1 with rows 4 .. 7
  python scripts/probes/gen_chain4_stage_ilp.py > /tmp/chain4_stage.inc
"""
DPP = ["quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf", "row_half_mirror row_mask:0xf bank_mask:0xf",
       "row_mirror row_mask:0xf bank_mask:0xf", "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"]
S = ["s%d" % i for i in range(80, 88)]
E, T, INV = "v60", "v61", "v62"          # clobbered temporaries of C4_STAGE_MAX
MAX_CLOB = ["v60", "v61", "v62"] + S
AD = ["v59"]                             # C4_STAGE_SPLIT, one row at a time (its four hi and four lo instructions are independent of each other: enough for
HP = [("v[60:61]", "v60", "v61")]        # a 4-cycle issue); with more rows in flight the compiler ran out of its 64 registers beside the 32 of the rows
LP = [("v[62:63]", "v62", "v63")]        # and spilled into a0 -- a weight fragment's register (the build refuses such a listing)
SPLIT_CLOB = ["v%d" % i for i in range(59, 64)] + S[:4]


def build_max():
    t = ["s_nop 1"]
    for ctrl in DPP:
        for r in range(8):
            t.append("v_max_f32_dpp %%[m%d], %%[m%d], %%[m%d] %s" % (r, r, r, ctrl))
    t.append("s_nop 0")
    for r in range(8):
        t.append("v_readlane_b32 %s, %%[m%d], 63" % (S[r], r))
    for r in range(8):
        t.append("v_writelane_b32 %%[smx], %s, %d" % (S[r], r))
    # scale_from_max on lanes = rows: e = bits ? exponent : 140; scale = 2^(267 - e), 1 / scale = 2^(e - 13), biased exponents clamped to [1, 254]
    t += ["v_bfe_u32 %s, %%[smx], 23, 8" % E, "v_cmp_ne_u32 vcc, 0, %[smx]", "v_mov_b32 %s, 0x8c" % T, "v_cndmask_b32 %s, %s, %s, vcc" % (E, T, E),
          "v_sub_u32 %s, 0x10b, %s" % (T, E), "v_add_u32 %s, -13, %s" % (INV, E),
          "v_max_i32 %s, 1, %s" % (T, T), "v_max_i32 %s, 1, %s" % (INV, INV), "v_min_i32 %s, 0xfe, %s" % (T, T), "v_min_i32 %s, 0xfe, %s" % (INV, INV),
          "v_lshlrev_b32 %%[sc], 23, %s" % T, "v_lshlrev_b32 %s, 23, %s" % (INV, INV)]
    t += ["s_mov_b64 exec, 0xff", "ds_write_b32 %%[tad], %s" % INV, "ds_write_b32 %[xad], %[smx]",
          "s_mov_b64 exec, %[rmask]", "global_store_dword %[l4], %[smx], %[rmp]", "s_mov_b64 exec, -1"]
    return t


def build_split(b):
    R = lambda q, j: "%%[a%d]" % (4 * q + j)
    t = []
    for q in range(4):
        t.append("v_readlane_b32 %s, %%[sc], %d" % (S[q], 4 * b + q))
    t.append("s_mov_b64 exec, %[wm]")
    for pair in range(4):
        qs = (pair,)
        for i, q in enumerate(qs):
            t.append("v_xor_b32 %s, 0x%x, %%[xw]" % (AD[i], 16 * (4 * b + q)))
        for i, q in enumerate(qs):
            t.append("v_add_u32 %s, %%[b0], %s" % (AD[i], AD[i]))
        for step in range(8):
            for i, q in enumerate(qs):
                (hp, h0, h1), (lp, l0, l1) = HP[i], LP[i]
                t.append(["v_fma_mixlo_f16 %s, %s, %s, 0" % (h0, R(q, 0), S[q]), "v_fma_mixlo_f16 %s, %s, %s, 0" % (h1, R(q, 2), S[q]),
                          "v_fma_mixhi_f16 %s, %s, %s, 0" % (h0, R(q, 1), S[q]), "v_fma_mixhi_f16 %s, %s, %s, 0" % (h1, R(q, 3), S[q]),
                          "v_fma_mixlo_f16 %s, %s, %s, -%s op_sel_hi:[0,0,1]" % (l0, R(q, 0), S[q], h0), "v_fma_mixlo_f16 %s, %s, %s, -%s op_sel_hi:[0,0,1]" % (l1, R(q, 2), S[q], h1),
                          "v_fma_mixhi_f16 %s, %s, %s, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (l0, R(q, 1), S[q], h0),
                          "v_fma_mixhi_f16 %s, %s, %s, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (l1, R(q, 3), S[q], h1)][step])
        for i, q in enumerate(qs):
            r = 4 * b + q
            t.append("ds_write_b64 %s, %s offset:%d" % (AD[i], HP[i][0], 512 * r))
            t.append("ds_write_b64 %s, %s offset:%d" % (AD[i], LP[i][0], 512 * r + 4096))
        if pair < 3:
            t.append("s_nop 0")
    t += ["s_mov_b64 exec, -1", "s_nop 1"]
    return t


def emit(name, lines):
    print("#define %s \\" % name)
    for i, l in enumerate(lines):
        print('    "%s\\n\\t"%s' % (l, " \\" if i + 1 < len(lines) else ""))
    print()


if __name__ == "__main__":
    print("// GENERATED by scripts/gen_chain4_stage.py -- do not edit.  The second half of the staging of a tile's rows, ordered for instruction-level parallelism (see the script).")
    emit("C4_STAGE_MAX", build_max())
    for b in range(2):
        emit("C4_STAGE_SPLIT_%d" % b, build_split(b))
    print("#define C4_STAGE_MAX_CLOBBERS " + ", ".join('"%s"' % v for v in MAX_CLOB) + ', "vcc", "memory"')
    print("#define C4_STAGE_SPLIT_CLOBBERS " + ", ".join('"%s"' % v for v in SPLIT_CLOB) + ', "memory"')
    print("// instructions: %d + 2 x %d" % (len(build_max()), len(build_split(0))))
