#!/usr/bin/env python3
"""EXPERIMENT (round 4, measured: no gain -- not part of the build).  The second half of the staging of a tile's input rows in chain4.hip
(stage_finish: row maxima, power-of-two scales, split into the tile's hi / lo planes, the tile's 1 / scale and maxima tables, the rows' maxima to
memory) as two generated asm statements of 156 instructions each instead of the ~700 the compiler makes of the C++ loop.

Hypothesis: the staging of the runs' input rows (1.39 ms of an 11.4 ms step) is bound by its instruction count.
Result: bit-identical (tests/test_hip_chain_variants.py passed with the statements wired into stage_finish behind a switch, including the value
MLP's 141 -> 32 shape and a 37-row last tile), and NOT faster: 11.66 / 11.71 ms per step against 11.77 / 11.71 for the C++ form in one box, every
run kernel within 1 %; in the cycle stamps the piece takes 4.4-6.1k cycles either way (14 cycles per instruction: the chain row maximum (6 dependent
DPP steps) -> v_readlane -> 11 dependent scalar instructions -> 8 dependent v_fma_mix -> LDS write is latency, not issue, and the SIMD's other wave is
in its own chain).  Not waiting for the rows (-DC4_X_NOWAIT at the time) changed nothing either: they have long arrived.
Kept as a record and as a starting point should the heavy slots ever become generated statements that INTERLEAVE this chain with a k-loop.

Wiring used: `#include "chain4_stage.inc"`, and in stage_finish behind the LayerNorm core
    asm volatile(C4_STAGE_FINISH_0 : : [a0] "v"(v[0].x) ... [a15] "v"(v[3].w), [xw] "v"(wp ^ ((wn & 1) * 128)), [b0] "v"(lds(planes + wn * C4_BLK_BYTES) + wq),
                 [wm] "s"(lanes whose columns are < kpad), [rmp] "s"(rowmax0 + r0), [rmask] "s"(lanes 0 .. 3 whose row exists), [l4] "v"(4 * lane),
                 [tad] "v"(lds(inv_tab + 8 wn) + 4 * lane), [xad] "v"(lds(xmax_tab + 8 wn) + 4 * lane) : C4_STAGE_CLOBBERS);
and the same with rows 4 .. 7, C4_STAGE_FINISH_1 and the table / row-maximum addresses moved on by four.
  python scripts/probes/gen_chain4_stage.py > /tmp/chain4_stage.inc
"""
MX = ["v52", "v53", "v54", "v55"]       # maxima of the batch's four rows (per lane, then the wave's in lane 63)
SMXV, INVV = "v56", "v57"               # lanes 0 .. 7: the eight rows' maxima / 1 / scales
AD = "v58"
H = [("v[60:61]", "v60", "v61"), ("v[62:63]", "v62", "v63")]     # split temporaries: (pair, first, second); two sets, alternating hi / lo of a row
CLOB_V = ["v%d" % i for i in range(52, 64)]
S_MX = ["s80", "s81", "s82", "s83"]     # the batch's maxima on the scalar unit
S_E, S_T, S_SC, S_INV = "s84", "s85", "s86", "s87"
S_EX = "s[88:89]"
CLOB_S = ["s%d" % i for i in range(80, 90)]

DPP = ["quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf", "row_half_mirror row_mask:0xf bank_mask:0xf",
       "row_mirror row_mask:0xf bank_mask:0xf", "row_bcast:15 row_mask:0xa bank_mask:0xf", "row_bcast:31 row_mask:0xc bank_mask:0xf"]


def build(b):
    """the statement of batch b (rows 4 b .. 4 b + 3 of the wave's eight): operands a0 .. a15 = the rows' values (row q: a[4 q] .. a[4 q + 3])"""
    t = []
    R = lambda q, j: "%%[a%d]" % (4 * q + j)
    for q in range(4):
        t.append("v_max3_f32 %s, |%s|, |%s|, |%s|" % (MX[q], R(q, 0), R(q, 1), R(q, 2)))
    for q in range(4):
        t.append("v_max_f32 %s, %s, |%s|" % (MX[q], MX[q], R(q, 3)))
    t.append("s_nop 1")
    for ctrl in DPP:
        for q in range(4):
            t.append("v_max_f32_dpp %s, %s, %s %s" % (MX[q], MX[q], MX[q], ctrl))
    t.append("s_nop 0")
    for q in range(4):
        t.append("v_readlane_b32 %s, %s, 63" % (S_MX[q], MX[q]))
    for q in range(4):
        r = 4 * b + q
        # scale_from_max: e = bits ? exponent : 140; scale = 2^(267 - e), 1 / scale = 2^(e - 13), biased exponents clamped to [1, 254]
        t += ["s_bfe_u32 %s, %s, 0x80017" % (S_E, S_MX[q]), "s_cmp_eq_u32 %s, 0" % S_MX[q], "s_cselect_b32 %s, 0x8c, %s" % (S_E, S_E),
              "s_sub_i32 %s, 0x10b, %s" % (S_T, S_E), "s_max_i32 %s, %s, 1" % (S_T, S_T), "s_min_i32 %s, %s, 0xfe" % (S_T, S_T), "s_lshl_b32 %s, %s, 23" % (S_SC, S_T),
              "s_add_i32 %s, %s, -13" % (S_T, S_E), "s_max_i32 %s, %s, 1" % (S_T, S_T), "s_min_i32 %s, %s, 0xfe" % (S_T, S_T), "s_lshl_b32 %s, %s, 23" % (S_INV, S_T),
              "v_writelane_b32 %s, %s, %d" % (SMXV, S_MX[q], q), "v_writelane_b32 %s, %s, %d" % (INVV, S_INV, q)]
        # under the write mask: this lane's 8 bytes of the row's hi and lo plane rows
        t += ["s_mov_b64 exec, %[wm]"]
        t += ["v_xor_b32 %s, 0x%x, %%[xw]" % (AD, 16 * r), "v_add_u32 %s, %%[b0], %s" % (AD, AD)]
        (hp, h0, h1), (lp, l0, l1) = H
        t += ["v_fma_mixlo_f16 %s, %s, %s, 0" % (h0, R(q, 0), S_SC), "v_fma_mixlo_f16 %s, %s, %s, 0" % (h1, R(q, 2), S_SC),
              "v_fma_mixhi_f16 %s, %s, %s, 0" % (h0, R(q, 1), S_SC), "v_fma_mixhi_f16 %s, %s, %s, 0" % (h1, R(q, 3), S_SC),
              "v_fma_mixlo_f16 %s, %s, %s, -%s op_sel_hi:[0,0,1]" % (l0, R(q, 0), S_SC, h0), "v_fma_mixlo_f16 %s, %s, %s, -%s op_sel_hi:[0,0,1]" % (l1, R(q, 2), S_SC, h1),
              "v_fma_mixhi_f16 %s, %s, %s, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (l0, R(q, 1), S_SC, h0),
              "v_fma_mixhi_f16 %s, %s, %s, -%s op_sel:[0,0,1] op_sel_hi:[0,0,1]" % (l1, R(q, 3), S_SC, h1),
              "ds_write_b64 %s, %s offset:%d" % (AD, hp, 512 * r), "ds_write_b64 %s, %s offset:%d" % (AD, lp, 512 * r + 4096)]
        t += ["s_mov_b64 exec, -1", "s_nop 1"]
    # lanes 0 .. 3: the tile's tables (1 / scale, maxima) and the rows' maxima to memory
    t += ["s_mov_b64 exec, 15", "ds_write_b32 %%[tad], %s" % INVV, "ds_write_b32 %%[xad], %s" % SMXV,
          "s_mov_b64 exec, %[rmask]", "global_store_dword %%[l4], %s, %%[rmp]" % SMXV, "s_mov_b64 exec, -1"]
    return t


def emit(name, lines):
    print("#define %s \\" % name)
    for i, l in enumerate(lines):
        print('    "%s\\n\\t"%s' % (l, " \\" if i + 1 < len(lines) else ""))
    print()


if __name__ == "__main__":
    print("// GENERATED by scripts/gen_chain4_stage.py -- do not edit.  The second half of the staging of a tile's rows, four rows per asm statement (see the script).")
    for b in range(2):
        emit("C4_STAGE_FINISH_%d" % b, build(b))
    print("#define C4_STAGE_CLOBBERS " + ", ".join('"%s"' % v for v in CLOB_V + CLOB_S) + ', "vcc", "scc", "memory"')
    print("// instructions per statement: %d" % len(build(0)))
