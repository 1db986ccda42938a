"""Cycle stamps of every wave of one workgroup of mlp_chain2_kernel (library built with -DPAPR_H3_TRACE -DPAPR_C2_TRACE_ALL):
   PAPR_HIP_LIB=scripts/probes/bin/libpapr_traceall.so python scripts/probes/chain2_trace_all.py [keep]
   prints, for the second layer-slot pair of the first tile, when each wave passed each stamp (cycles after the workgroup's first stamp)."""
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
hip.lib().papr_chain2_trace_read(buf)
t = list(buf)
nw = int(os.environ.get("NW", "16"))
per = 1024 // nw
names = ["k0", "k1/bar(a)", "bar(a) out", "dumped", "synced", "rows done", "staged", "bar(c) out"]
t0 = min(t[w * per] for w in range(nw))
for l in range(int(os.environ.get("L0", "1")), int(os.environ.get("L1", "3"))):
    print("layer slot %d" % l)
    print("  wave  " + "  ".join("%10s" % s for s in names))
    for w in range(nw):
        tt = t[w * per: (w + 1) * per]
        i = 7 * l
        print("  %2d    " % w + "  ".join("%10d" % (tt[i + j] - t0) for j in range(8)))
