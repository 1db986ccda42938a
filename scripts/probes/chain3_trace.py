"""Cycle stamps of every wave of one workgroup of mlp_chain3_kernel (library built with -DPAPR_H3_TRACE):
   bash scripts/probes/build_variant.sh trace3 -DPAPR_H3_TRACE; PAPR_HIP_LIB=scripts/probes/bin/libpapr_trace3.so python scripts/probes/chain3_trace.py [keep]
   per slot: when each wave finished its first piece, its second piece, passed the barrier, dumped, (next slot start)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops, hip
M = 512000
keep = len(sys.argv) > 1 and sys.argv[1] == "keep"
d = torch.device("cuda:0")
n = 4
spec = ops.MlpSpec("b", 256, dict(n_ff_layer=n, d_ff=256, d_ff_out=256, norm="none", ff_act="relu", ff_last_act="none"))
ws = [(torch.randn(256, 256) * 0.1).to(d) for _ in range(n)]
bs = [torch.zeros(256, device=d) for _ in range(n)]
x = torch.randn(M, 256, device=d)
for _ in range(3):
    ops.mlp_forward(spec, ws, bs, x, M, keep=keep)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 1024)()
hip.lib().papr_chain3_trace_read(buf)
t = list(buf)
names = ["start", "K (0-3)", "rows", "K (4-7)", "barrier", "dumped", "next"]
t0 = min(t[w * 128] for w in range(8))
for sl in range(int(os.environ.get("S0", "2")), int(os.environ.get("S1", "6"))):
    print("slot %d" % sl)
    print("  wave  " + "  ".join("%9s" % s for s in names))
    for w in range(8):
        tt = t[w * 128: (w + 1) * 128]
        i = 6 * sl
        print("  %2d    " % w + "  ".join("%9d" % (tt[i + j] - t0) for j in range(7)))
