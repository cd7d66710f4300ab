"""which tensor of a training step is not bit-reproducible?  Same weights, same batch, forward + backward repeated; every gradient, the
loss and the prediction are compared with the first repeat's bits.  `procs` > 1 runs that many processes on the one GPU at once
(contention moves kernel timing, which is what exposes a missing stream dependency).
usage: python tools/debug_grad_determinism.py [procs] [repeats] [batch] [hw] [depth]"""
import os
import sys

import torch
import torch.multiprocessing as mp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
DEV = "cuda:0"
PARAMS = dict(uncertainty_type="quantiles", q_lo=0.05, q_hi=0.95, q_lo_weight=1, q_hi_weight=1, mse_weight=1)


def worker(rank, reps, batch, hw, depth):
    from im2im_uq_amd import nn_ops
    from im2im_uq_amd.core.models.add_uncertainty import add_uncertainty
    from im2im_uq_amd.core.models.trunks.unet import UNet
    torch.cuda.set_device(0)
    nn_ops.set_compute_dtype(os.environ.get("DT", "bf16"))
    torch.manual_seed(3)
    model = add_uncertainty(UNet(1, 1, depth=depth), dict(PARAMS)).to(DEV).train()
    g = torch.Generator().manual_seed(31)
    x = torch.randn(batch, 1, hw, hw, generator=g).to(DEV)
    y = torch.rand(batch, 1, hw, hw, generator=g).to(DEV)
    names = [n for n, _ in model.named_parameters()]
    # WATCH=1: the first conv's weight gradient is launched three times on the same dz (the third after a device synchronisation) and dz
    # itself is fingerprinted (int64 sums of its even / odd 16-bit halves): is it the input or the kernel that moves?
    watch = {"log": []}
    if os.environ.get("WATCH") == "1":
        from im2im_uq_amd import _lib
        if os.environ.get("KEEP_MB"):
            _lib.check(_lib.lib.im2im_set_option(b"bn_apply_keep_mb", int(os.environ["KEEP_MB"])), "set_option")
        orig = nn_ops.smallconv_wgrad

        def thrice(s_nchw, l_nhwc, l_major, want_bias):
            if not l_major:
                return orig(s_nchw, l_nhwc, l_major, want_bias)
            a = orig(s_nchw, l_nhwc, l_major, want_bias)[0].clone()
            b = orig(s_nchw, l_nhwc, l_major, want_bias)[0].clone()
            torch.cuda.synchronize()
            c = orig(s_nchw, l_nhwc, l_major, want_bias)
            h = l_nhwc.contiguous().view(torch.int16).to(torch.int64).view(-1, 2).sum(0).tolist() if l_nhwc.dtype == torch.bfloat16 else [0, 0]
            watch["log"].append((torch.equal(a, b), torch.equal(a, c[0]), tuple(h), a))
            watch["last"] = (s_nchw, l_nhwc.clone(), orig)
            return a, c[1]
        nn_ops.smallconv_wgrad = thrice
    first, bad = None, {}
    if os.environ.get("EVAL") == "1":                  # the eval-mode forward (folded BatchNorm, fused pool / OutConv tails) instead of the step
        model.eval()
        with torch.no_grad():
            for it in range(reps):
                out = model(x)
                outs = out if isinstance(out, (tuple, list)) else (out,)
                cur = {f"eval out {i}": o.detach().clone() for i, o in enumerate(outs)}
                torch.cuda.synchronize()
                if first is None:
                    first = cur
                    continue
                for k, v in cur.items():
                    if not torch.equal(v, first[k]):
                        e = bad.setdefault(k, [0, 0.0, 0, [], ""])
                        e[0] += 1
                        e[1] = max(e[1], float((v.float() - first[k].float()).abs().max()))
                        e[2] = max(e[2], int((v != first[k]).sum()))
        reps_done = reps
        print(f"[det rank {rank}] EVAL {reps} forwards B={batch} {hw}x{hw} depth={depth}: " +
              ("all bits equal" if not bad else "; ".join(f"{k}: {v[0]}x max|d| {v[1]:.3e} ({v[2]} elements)" for k, v in bad.items())), flush=True)
        return
    for it in range(reps):
        for p in model.parameters():
            p.grad = None
        pred = model(x)
        loss = model.loss_fn(pred, y)
        loss.backward()
        torch.cuda.synchronize()
        cur = {"loss": loss.detach().clone(), "pred": pred.detach().clone()}
        cur.update({n: p.grad.detach().clone() for n, p in zip(names, model.parameters()) if p.grad is not None})
        if first is None:
            first = cur
            continue
        for k, v in cur.items():
            if not torch.equal(v, first[k]):
                d = (v.float() - first[k].float()).abs()
                e = bad.setdefault(k, [0, 0.0, 0, [], ""])
                e[0] += 1
                e[1] = max(e[1], float(d.max()))
                e[2] = max(e[2], int((d != 0).sum()))
                if not any(torch.equal(v, u) for u in e[3]):
                    e[3].append(v)
                if v.dim() == 4 and not e[4]:                # a conv weight gradient [co][ci][3][3]: which channels / taps moved?
                    nz = (d != 0)
                    e[4] = (f" it={it} co={nz.flatten(1).any(1).nonzero().flatten().tolist()} taps={nz.any(0).any(0).flatten().nonzero().flatten().tolist()}"
                            f" |g|max={float(first[k].abs().max()):.3e}")
    if watch["log"]:
        lg = watch["log"]
        # the same launch alone (the other processes do the same at about the same time): the step's own dz, then random data of its shape
        xs, dzs, orig = watch["last"]
        for label, dzt in (("the step's dz", dzs), ("random dz", torch.randn_like(dzs.float()).to(dzs.dtype)), ("random dz, 70 % zeros", (torch.randn_like(dzs.float()) * (torch.rand_like(dzs.float()) > 0.7)).to(dzs.dtype)),
                           ("the step's dz * 1e4", (dzs.float() * 1e4).to(dzs.dtype))):
            ref = orig(xs, dzt, True, False)[0].clone()
            n_bad = sum(not torch.equal(orig(xs, dzt, True, False)[0], ref) for _ in range(100))
            print(f"[det rank {rank}] isolated launches on {label} (|dz|max {float(dzt.float().abs().max()):.2e}): {n_bad} of 100 differ from the first", flush=True)
        print(f"[det rank {rank}] first-conv wgrad launched 3x per step: 1st==2nd in {sum(l[0] for l in lg)} of {len(lg)} steps, 1st==3rd (after sync) in "
              f"{sum(l[1] for l in lg)}; dz fingerprints seen: {len({l[2] for l in lg})} distinct (even halves {len({l[2][0] for l in lg})}, odd halves {len({l[2][1] for l in lg})}); "
              f"distinct 1st results {len({tuple(l[3].flatten().tolist()) for l in lg})}", flush=True)
    print(f"[det rank {rank}] {reps} repeats B={batch} {hw}x{hw} depth={depth}: " +
          ("all bits equal" if not bad else "; ".join(f"{k}: {v[0]}x max|d| {v[1]:.3e} ({v[2]} of {first[k].numel()} elements, {len(v[3])} distinct variants){v[4]}" for k, v in bad.items())),
          flush=True)


def main():
    a = [int(v) for v in sys.argv[1:]]
    procs, reps, batch, hw, depth = (a + [1, 40, 3, 32, 2][len(a):])[:5]
    if procs == 1:
        worker(0, reps, batch, hw, depth)
    else:
        mp.spawn(worker, args=(reps, batch, hw, depth), nprocs=procs, join=True)


if __name__ == "__main__":
    main()
