"""Stamps inside a last layer's row stores (library built with -DPAPR_C4_TRACE -DPAPR_C4_TRACE_ST): start | rows in LDS | after 2, 4, 6 store instructions | end."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from papr_amd import ops, hip
M = 512000
d = torch.device("cuda:0")
spec = ops.MlpSpec("b", 256, dict(n_ff_layer=4, d_ff=256, d_ff_out=256, norm="none", ff_act="relu", ff_last_act="none"))
ws = [(torch.randn(256, 256) * 0.1).to(d) for i in range(4)]
bs = [torch.zeros(256, device=d) for _ in range(4)]
x = torch.randn(M, 256, device=d)
for _ in range(3):
    ops.mlp_forward(spec, ws, bs, x, M, keep=False)
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 2048)()
hip.lib().papr_chain4_trace_read(buf)
t = list(buf)
for w in range(8):
    tt = t[w * 256: (w + 1) * 256]
    for g in range(2, 6):
        v = tt[6 * g: 6 * g + 6]
        print("wave %d visit %d: " % (w, g) + " ".join("%6d" % (v[j + 1] - v[j]) for j in range(5)) + "   total %6d" % (v[5] - v[0]))
