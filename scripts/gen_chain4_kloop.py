#!/usr/bin/env python3
"""Generates papr_amd/csrc/chain4_kloop.inc: the hot k-loops of chain4.hip, each as ONE inline-asm statement.

Why one statement: the A fragments are read from LDS asynchronously into registers that the matrix instructions consume two
blocks later.  With one asm statement per k-step the compiler saw sixteen opaque blocks and was free to put its own code between
them -- and it did: register copies of fragment tuples whose ds_read was still in flight (the first version passed every
bit-exactness test only because ~20 copies happened to take longer than an LDS round trip; any change of register pressure
around the loop broke it, non-deterministically).  Inside one statement nothing of the compiler's runs between the request of a
fragment and its use, and every wait is counted by hand.

  C4_KLOOP3(LD)  three products per fp32 product (parity mode): per k-step six MFMAs, one memory instruction in every gap
  C4_KLOOP1(LD)  one product (h1): two MFMAs per k-step
  LD = 1: each k-step's weight registers are refilled with the next layer's fragment behind their last use.

Accumulators: v[64:95] (tile X flavour) / v[96:127] (tile Y), by name (written, not read: their first products add to the constant 0).  Operands (named): f00..f03 / f10..f13 = fragment buffers 0 / 1 (lo0 lo1 hi0 hi1; early-clobber
temporaries); ad0..ad7 = LDS byte address of k-step j's hi0 fragment (k-step j + 8: + 256); wv = lane * 16; bh0..bh3 / bl0..bl3 =
bases of the next layer's hi / lo fragments of k-steps 4 q .. 4 q + 3.   python scripts/gen_chain4_kloop.py > papr_amd/csrc/chain4_kloop.inc
"""

def wh(ks): return "a[%d:%d]" % (8 * ks, 8 * ks + 3)
def wl(ks): return "a[%d:%d]" % (8 * ks + 4, 8 * ks + 7)
ACC = {"a0": "v[64:79]", "a1": "v[80:95]"}        # tile X; emit() rewrites them for tile Y (chain4.hip keeps the accumulators in v64-v127, outside the compiler's allocation)
def mf(acc, w, x, first=False):  # (first: the accumulator's first product takes the constant 0 as its addend -- no zeroing beforehand)
    return "v_mfma_f32_32x32x16_f16 %s, %s, %%[%s], %s" % (ACC[acc], w, x, "0" if first else ACC[acc])
def rd(dst, ks, kind):          # fragment `kind` (l0 l1 h0 h1) of k-step ks
    off = {"l0": 4096, "l1": 36864, "h0": 0, "h1": 32768}[kind] + (256 if ks >= 8 else 0)
    return "ds_read_b128 %%[%s], %%[ad%d] offset:%d" % (dst, ks & 7, off)
def wt(n): return "s_waitcnt lgkmcnt(%d)" % n
def ldh(ks): return "global_load_dwordx4 %s, %%[wv], %%[bh%d] offset:%d" % (wh(ks), ks >> 2, (ks & 3) * 1024)
def ldl(ks): return "global_load_dwordx4 %s, %%[wv], %%[bl%d] offset:%d" % (wl(ks), ks >> 2, (ks & 3) * 1024)

def kloop3(ld):
    t = []
    for b in (0, 1):                # the fragments of k-steps 0 and 1
        for j, kind in enumerate(("l0", "l1", "h0", "h1")):
            t.append(rd("f%d%d" % (b, j), b, kind))
    for ks in range(16):
        f = lambda j: "f%d%d" % (ks & 1, j)
        nxt = ks + 2 if ks + 2 < 16 else None
        out = 8 if ks <= 14 else 4                       # reads in flight at the top of the block
        if ks == 14: out = 8
        # waits: at the top `out` reads are in flight, oldest first lo0 lo1 hi0 hi1 of this k-step
        w = [out - 1] * 4
        if nxt is None:
            w = [out - 1, out - 2, out - 3, out - 4]
        else:
            w = [out - 1, out - 1, out - 1, out - 2]
        t += [wt(w[0]), mf("a0", wh(ks), f(0), ks == 0)]
        if nxt is not None: t.append(rd(f(0), nxt, "l0"))
        t += [wt(w[1]), mf("a1", wh(ks), f(1), ks == 0)]
        if nxt is not None: t.append(rd(f(1), nxt, "l1"))
        t += [wt(w[2]), mf("a0", wl(ks), f(2))]
        if ld and ks >= 1: t.append(ldh(ks - 1))
        t += [wt(w[3]), mf("a1", wl(ks), f(3))]
        if ld: t.append(ldl(ks))
        t.append(mf("a0", wh(ks), f(2)))
        if nxt is not None: t.append(rd(f(2), nxt, "h0"))
        t.append(mf("a1", wh(ks), f(3)))
        if nxt is not None: t.append(rd(f(3), nxt, "h1"))
    if ld: t.append(ldh(15))
    t += ["s_nop 15", "s_nop 7"]    # the last results leave the matrix pipe 16 passes after issue
    return t

def kloop1(ld):
    t = []
    for b in (0, 1):
        t.append(rd("f%d2" % b, b, "h0")); t.append(rd("f%d3" % b, b, "h1"))
    for ks in range(16):
        f = lambda j: "f%d%d" % (ks & 1, j)
        nxt = ks + 2 if ks + 2 < 16 else None
        if nxt is not None: w = [3, 3]
        elif ks == 14: w = [3, 2]
        else: w = [1, 0]
        t += [wt(w[0]), mf("a0", wh(ks), f(2), ks == 0)]
        if nxt is not None: t.append(rd(f(2), nxt, "h0"))
        t += [wt(w[1]), mf("a1", wh(ks), f(3), ks == 0)]
        if nxt is not None: t.append(rd(f(3), nxt, "h1"))
        if ld: t.append(ldh(ks))
    t += ["s_nop 15", "s_nop 7"]
    return t

def emit(name, lines):
    for tile, sub in (("X", {}), ("Y", {"v[64:79]": "v[96:111]", "v[80:95]": "v[112:127]"})):
        print("#define %s_%s \\" % (name, tile))
        for i, l in enumerate(lines):
            for a, b in sub.items():
                l = l.replace(a, b)
            print('    "%s\\n\\t"%s' % (l, " \\" if i + 1 < len(lines) else ""))
        print()

print("// GENERATED by scripts/gen_chain4_kloop.py -- do not edit.  The hot k-loops of chain4.hip, one asm statement each (see the script).")
emit("C4_KLOOP3_LD", kloop3(True))
emit("C4_KLOOP3_NL", kloop3(False))
emit("C4_KLOOP1_LD", kloop1(True))
emit("C4_KLOOP1_NL", kloop1(False))
print("#define C4_KLOOP_AGPRS " + ", ".join('"a%d"' % i for i in range(128)))
