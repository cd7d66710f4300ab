"""fp32 gradient conditioning probe: HIP fp32 vs the oracle in fp32 (the reference's arithmetic) vs the oracle in fp64."""
import sys, os, torch
R = '/root/repo' if len(sys.argv) < 2 else sys.argv[1]
sys.path.insert(0, R); sys.path.insert(0, R + '/tests')
from test_model_gpu import build, PARAMS, rel_l2, DEV
from oracle import model as om

def oracle_grads(x, y, dtype):
    st = om.det_state(1, 1)
    st = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in st.items()}
    leaves = {k: v.clone().requires_grad_(True) for k, v in st.items() if om.is_param(k)}
    work = dict(st); work.update(leaves)
    loss = om.quantile_loss(om.model_forward(x.to(dtype), work, training=True), y.to(dtype), PARAMS); loss.backward()
    return float(loss), {k: v.grad for k, v in leaves.items()}

for kind in ("det", "noise"):
    if kind == "det":
        x, y = om.det_images(3, 1, 64, 64, salt=5)
    else:
        g = torch.Generator().manual_seed(11)
        x = torch.randn(3, 1, 64, 64, generator=g); y = torch.rand(3, 1, 64, 64, generator=g)
    model = build(1, "fp32"); model.train()
    loss = model.loss_fn(model(x.to(DEV)), y.to(DEV)); loss.backward()
    l32, g32 = oracle_grads(x, y, torch.float32)
    l64, g64 = oracle_grads(x, y, torch.float64)
    print(kind, "lib", os.environ.get("IM2IM_LIB"), "loss gpu/cpu32/cpu64", loss.item(), l32, l64)
    rows = []
    for name, p in model.named_parameters():
        if p.grad is None or "double_conv.0.bias" in name or "double_conv.3.bias" in name: continue
        rows.append((name, rel_l2(p.grad.cpu(), g64[name]), rel_l2(g32[name], g64[name]), rel_l2(p.grad.cpu(), g32[name])))
    for n, a, b, c in rows[::4]:
        print(f"   {n:60s} gpu-vs-64 {a:.1e}  cpu32-vs-64 {b:.1e}  gpu-vs-cpu32 {c:.1e}")
    print("   max gpu-vs-64 %.1e  max cpu32-vs-64 %.1e" % (max(r[1] for r in rows), max(r[2] for r in rows)))
