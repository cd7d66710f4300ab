"""repeat tests/test_graph_ddp_gpu.py's two-rank eager-vs-graphed comparison and report WHERE the two runs part (first step whose loss
differs, state entries that differ) -- the test compares bits, one failure in a full-suite run needs a rate and a location.
usage: python tools/debug_ddp_flake.py [repeats] [mode]     mode: eg (eager vs graph, default) | ee (eager vs eager) | gg"""
import os
import sys
import tempfile

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def worker(rank, world, port, tmpdir, mode):
    import test_graph_ddp_gpu as T
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        la, sa, _ = T._run(rank, world, graph=(mode[0] == "g"))
        lb, sb, _ = T._run(rank, world, graph=(mode[1] == "g"))
        torch.save({"la": la, "lb": lb, "sa": sa, "sb": sb}, os.path.join(tmpdir, f"r{rank}.pt"))
    finally:
        dist.destroy_process_group()


def main():
    import test_graph_ddp_gpu as T
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    mode = sys.argv[2] if len(sys.argv) > 2 else "eg"
    bad = 0
    for it in range(reps):
        with tempfile.TemporaryDirectory() as tmp:
            mp.spawn(worker, args=(2, T._free_port(), tmp, mode), nprocs=2, join=True)
            r = [torch.load(os.path.join(tmp, f"r{i}.pt")) for i in range(2)]
        msgs = []
        for i, d in enumerate(r):
            ne = (d["la"] != d["lb"]).nonzero().flatten().tolist()
            if ne:
                msgs.append(f"rank {i}: losses differ from step {ne[0]} ({len(ne)} of {d['la'].numel()}), max |d| {float((d['la'] - d['lb']).abs().max()):.3e}")
            keys = [k for k in d["sa"] if not torch.equal(d["sa"][k], d["sb"][k])]
            if keys:
                msgs.append(f"rank {i}: {len(keys)} of {len(d['sa'])} state entries differ, e.g. {keys[:4]}")
        cross = [k for k in r[0]["sb"] if not k.endswith(("running_mean", "running_var")) and not torch.equal(r[0]["sb"][k], r[1]["sb"][k])]
        if cross:
            msgs.append(f"ranks disagree on {len(cross)} entries after run b, e.g. {cross[:4]}")
        bad += bool(msgs)
        print(f"[{mode}] repeat {it}: {'OK' if not msgs else ' | '.join(msgs)}", flush=True)
    print(f"[{mode}] {bad} of {reps} repeats differed (HIP_FORCE_DEV_KERNARG={os.environ.get('HIP_FORCE_DEV_KERNARG')})", flush=True)


if __name__ == "__main__":
    main()
