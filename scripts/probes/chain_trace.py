"""Cycle stamps of one persistent workgroup of the fused layer-run kernel (library built with -DPAPR_H3_TRACE:
  hipcc <flags of papr_amd/build.py> -DPAPR_H3_TRACE -shared papr_amd/csrc/*.hip -o scripts/probes/libpapr_trace.so
and run with PAPR_HIP_LIB=scripts/probes/libpapr_trace.so)."""
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
buf = (ctypes.c_longlong * 256)()
hip.lib().papr_chain_trace_read(buf)
t = list(buf)
i = 0
for tile in range(4):
    print("tile %d: staging %d" % (tile, t[i + 1] - t[i]))
    i += 1
    print("   layer: k-loop  barrier  phase1  barrier  phase2+barrier")
    for l in range(n):
        print("   %d: %8d %8d %8d %8d %8d" % (l, t[i + 1] - t[i], t[i + 2] - t[i + 1], t[i + 3] - t[i + 2], t[i + 4] - t[i + 3], t[i + 5] - t[i + 4]))
        i += 5
    i += 1
