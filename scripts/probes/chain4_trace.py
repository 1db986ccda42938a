"""Cycle stamps of every wave of one workgroup of mlp_chain4_kernel (library built with -DPAPR_C4_TRACE):
   bash scripts/probes/c4_variant.sh trace -DPAPR_C4_TRACE; PAPR_HIP_LIB=$PWD/scripts/probes/bin/libpapr_trace.so python scripts/probes/chain4_trace.py [keep|bwd]
   per slot and wave: cycles spent in  P2 (rows-first waves) | K | P1 | P2 (k-first waves) | barrier wait."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops, hip
M = 512000
mode = sys.argv[1] if len(sys.argv) > 1 else "inf"
d = torch.device("cuda:0")
n = int(os.environ.get("LAYERS", "4"))
d_in = int(os.environ.get("D_IN", "256"))
d_out = int(os.environ.get("D_OUT", "256"))          # (D_IN=117 LAYERS=5 NORM=1: the key MLP's shape; D_IN=141 D_OUT=32 LAYERS=8: the value MLP's)
norm = os.environ.get("NORM", "") != ""
spec = ops.MlpSpec("b", d_in, dict(n_ff_layer=n, d_ff=256, d_ff_out=d_out, norm="layernorm" if norm else "none", ff_act="relu", ff_last_act="none"))
spec.one_product = os.environ.get("ONE", "") != ""          # (ONE=1: the one-product arithmetic, f16 rows)
ws = [(torch.randn(spec.layers[i]["n_out_pad"], spec.layers[i]["n_in"]) * 0.1).to(d) for i in range(n)]
bs = [torch.zeros(spec.layers[i]["n_out_pad"], device=d) for i in range(n)]
x0 = torch.randn(M, spec.ld_in, device=d)
gy = torch.randn(M, spec.ld_out[-1], device=d)
scratch = [torch.empty((M, 256), device=d) for _ in range(2)]
dots = torch.randn((M + 19) // 20, 256, device=d) if norm and os.environ.get("DOTS") else None      # (DOTS=1: the attention scores' dot products ride in the run)
for _ in range(3):
    x = x0.clone()
    outs = ops.mlp_forward(spec, ws, bs, x, M, keep=mode != "inf", out_norm=(d_out, 1e-6) if norm else None, in_norm=(d_in, 1e-6) if norm else None,
                           dot_rows=dots, rows_per_dot=20, raw_rows=bool(os.environ.get("RAW")) and mode != "inf")
    if mode == "bwd":
        ops.mlp_backward(spec, ws, bs, x, M, outs, gy.clone(), scratch, True)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 2048)()
hip.lib().papr_chain4_trace_read(buf)
t = list(buf)
NS = int(os.environ.get("STAMPS", "5"))
names = (["decide", "operands", "statement", "barrier", "loop"] if os.environ.get("HOT") else ["P1", "barrier", "first", "second", "end"]) if NS == 5 else ["P1", "barrier", "A", "B", "C", "D", "end"]
tot = [0.0] * NS
cnt = 0
s0, s1 = int(os.environ.get("S0", "4")), int(os.environ.get("S1", "12"))
if os.environ.get("SUMMARY"):                     # one line per slot: the slot's length seen by waves 0 and 4, and its pieces for wave 0
    for sl in range(s0, s1):
        row = []
        for w in (0, 4):
            tt = t[w * 256: (w + 1) * 256]
            row.append(tt[NS * (sl + 1)] - tt[NS * sl])
        tt = t[0:256]
        print("slot %2d (tile %s): %6d %6d cycles   wave 0 pieces: %s" % (sl, "XY"[sl & 1], row[0], row[1], " ".join("%6d" % (tt[NS * sl + j + 1] - tt[NS * sl + j]) for j in range(NS))))
    sys.exit(0)
for sl in range(s0, s1):
    print("slot %d (tile %s)" % (sl, "XY"[sl & 1]))
    print("  wave  " + "  ".join("%9s" % s for s in names) + "      total")
    for w in range(8):
        tt = t[w * 256: (w + 1) * 256]
        i = NS * sl
        dts = [tt[i + j + 1] - tt[i + j] for j in range(NS)]
        print("  %2d    " % w + "  ".join("%9d" % v for v in dts) + "  %9d" % sum(dts))
        for j in range(NS): tot[j] += dts[j]
        cnt += 1
print("mean per wave and slot: " + "  ".join("%s %.0f" % (n_, v / cnt) for n_, v in zip(names, tot)) + "   slot %.0f" % (sum(tot) / cnt))
