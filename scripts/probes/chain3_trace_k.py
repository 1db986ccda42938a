"""Stamps inside mlp_chain3_kernel's k_run (library built with -DPAPR_H3_TRACE -DPAPR_C3_TRACE_K): per slot and wave the lengths of
setup (zero accumulators, addresses) | wait for the weight fragments | the sixteen k-step blocks | drain | park (waves 0-3) ..."""
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
# every slot: start, [4 K stamps], after K(0-3)+park, after rows, [4 K stamps for waves 4-7], after K(4-7), barrier, dump  -> waves 0-3 and 4-7 both 10 stamps
names03 = ["setup", "vm wait", "blocks", "drain", "park", "rows", "unpark", "barrier", "dump", "barrier2"]
names47 = ["(none)", "rows", "setup", "vm wait", "blocks", "drain", "tail", "barrier", "dump", "barrier2"]
sl = int(os.environ.get("S0", "4"))
for w in range(8):
    tt = t[w * 128: (w + 1) * 128]
    i = 10 * sl
    names = names03 if w < 4 else names47
    print("wave %d  " % w + "  ".join("%s %d" % (names[j], tt[i + j + 1] - tt[i + j]) for j in range(10)))
