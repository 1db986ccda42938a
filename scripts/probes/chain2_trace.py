"""Cycle stamps of one workgroup of mlp_chain2_kernel (both groups), library built with -DPAPR_H3_TRACE:
   bash scripts/probes/build_variant.sh trace -DPAPR_H3_TRACE ; PAPR_HIP_LIB=scripts/probes/bin/libpapr_trace.so python scripts/probes/chain2_trace.py [keep]"""
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
fine = os.environ.get("FINE") == "1"       # library built with -DPAPR_C2_TRACE_FINE: six more stamps inside the row phase
names = ["k-loop", "bar(a)", "dump", "gsync"] + (["rd0", "math0", "max+split0", "rd1", "math1", "max+split1", "tail"] if fine else ["rows"]) + ["stage", "bar(c)"]
for g in range(2):
    tt = t[512 * g: 512 * (g + 1)]
    print("group %d  (first stamp %d)" % (g, tt[0] - min(t[0], t[512])))
    print("   layer: " + "  ".join("%8s" % s for s in names) + "     total")
    i = 0
    for it in range(2):
        for l in range(n):
            ns = len(names)
            seg = [tt[i + 1 + j] - tt[i + j] for j in range(ns)]
            # (the last interval ends at the next k-loop's first stamp)
            print("   %d.%d:   " % (it, l) + "  ".join("%8d" % v for v in seg) + "  %8d" % (tt[i + ns] - tt[i]))
            i += ns
