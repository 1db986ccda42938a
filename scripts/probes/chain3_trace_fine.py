"""Fine cycle stamps of mlp_chain3_kernel's row phases (library built with -DPAPR_H3_TRACE -DPAPR_C3_TRACE_FINE):
   per slot and wave: multiply (waves 0-3) | per 4-row batch: LDS rows arrived, bias / activation / stores issued, row maxima in
   scalars, split + plane writes | multiply (waves 4-7) | barrier | dump."""
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
names = ["K(0-3)", "rd A", "math A", "max A", "split A", "rd B", "math B", "max B", "split B", "tail", "K(4-7)", "barrier", "dump", "barrier2"]
per = len(names)
sl = int(os.environ.get("S0", "3"))
print("slot %d: segment lengths in cycles" % sl)
print("  wave  " + "  ".join("%8s" % s for s in names))
for w in range(8):
    tt = t[w * 128: (w + 1) * 128]
    i = per * sl
    print("  %2d    " % w + "  ".join("%8d" % (tt[i + j + 1] - tt[i + j]) for j in range(per)))
