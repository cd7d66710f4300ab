"""phase lengths of the ping-pong conv kernel from a -DIM2IM_PP_TRACE build (tools/ab_build.sh trace -DIM2IM_PP_TRACE):
    IM2IM_LIB=im2im_uq_amd/lib/libim2im_uq_trace.so python tools/pp_trace.py [h ci co batch]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from im2im_uq_amd import hip_ops, nn_ops
from im2im_uq_amd._lib import check, dptr, lib, stream_ptr

dev = "cuda:0"
h, ci, co, b = (int(v) for v in sys.argv[1:5]) if len(sys.argv) > 4 else (160, 128, 128, 78)
hip_ops.set_option("conv_pp", 3)
x = torch.randn(b, h, h, ci, device=dev).to(torch.bfloat16)
wt = torch.randn(co, ci, 3, 3, device=dev) * 0.05
wf, _ = nn_ops.pack_weight(wt, torch.bfloat16)
y = torch.empty(b, h, h, co, dtype=torch.bfloat16, device=dev)
trace = torch.zeros(2 * 2049, dtype=torch.int64, device=dev)
dummy = torch.zeros(2, co, device=dev)
for _ in range(3):
    check(lib.im2im_conv_dgrad_bn(dptr(x), dptr(wf), dptr(y), dptr(y), dptr(dummy), dptr(dummy), dptr(trace), b, h, h, ci, co, 9, 1,
                                  stream_ptr(x.device)), "trace launch")
torch.cuda.synchronize()
t = trace.cpu().view(2, 2049)
for g in range(2):
    n = int(t[g, 0])
    ts = t[g, 1:1 + min(n, 2047)]
    d = (ts[1:] - ts[:-1]).tolist()
    print(f"group {g}: {n} stamps; first 12 intervals {d[:12]}")
    body = d[2 + g:]                      # after the prologue barriers
    # intervals come in fours per tap: [L0->M0 boundary ...]; print per-tap sums for the first two chunks and the mean by position
    per = [body[i:i + 4] for i in range(0, len(body) - 3, 4)]
    for k, p in enumerate(per[:20]):
        print(f"  tap {k:3d}: {p}  sum {sum(p)}")
    if per:
        import statistics
        print("  mean by position:", [round(statistics.mean(p[i] for p in per), 1) for i in range(4)], " mean per tap:",
              round(statistics.mean(sum(p) for p in per), 1), " median per tap:", statistics.median(sum(p) for p in per))
