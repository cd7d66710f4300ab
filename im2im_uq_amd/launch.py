"""One process per GPU without an external launcher.

The reference goes multi-GPU by itself: `if torch.cuda.device_count() > 1: net = DataParallelPassthrough(net)`
(core/scripts/train.py:112-115) -- a user types `python router.py` and every GPU of the node is used.  Here data
parallelism is one process per GPU over RCCL, so the entry points (bench.py, core/scripts/router.py) re-execute
themselves under `torch.distributed.run` when they are asked for N > 1 GPUs and find no rendezvous environment:
`python bench.py --gpus 8` is then the same job as `torchrun --nproc-per-node 8 bench.py --gpus 8`.
"""
from __future__ import annotations

import datetime
import os
import socket
import subprocess
import sys


def in_rendezvous_env() -> bool:
    """True inside a torch.distributed.run / torchrun worker (RANK and WORLD_SIZE exported)."""
    return "WORLD_SIZE" in os.environ and "RANK" in os.environ


def free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_ranks(nproc: int, argv, module: str | None = None, script: str | None = None, env=None) -> int:
    """run `python -m torch.distributed.run --nnodes=1 --nproc-per-node nproc <script | -m module> argv...` on 127.0.0.1 and
    return its exit code (non-zero when any rank failed).  stdout / stderr are inherited, so the one JSON line rank 0
    prints is this process's output too."""
    if nproc < 2:
        raise ValueError("spawn_ranks is for N > 1")
    if (module is None) == (script is None):
        raise ValueError("give exactly one of module / script")
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # the host driver only supports dmabuf IPC (RCCL needs it)
    e.setdefault("MASTER_ADDR", "127.0.0.1")
    e.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port())]
    cmd += (["-m", module] if module else [script]) + list(argv)
    return subprocess.call(cmd, env=e)


def init_distributed(expected_world: int | None = None):
    """rendezvous of one rank: device = LOCAL_RANK, backend nccl (= RCCL over xGMI) unless IM2IM_DIST_BACKEND says gloo
    (several ranks sharing one GPU in tests; RCCL refuses duplicate devices).  Returns (dist or None, rank, world, device,
    backend).  Fails loudly when the world is not the one the caller asked for or the node has fewer GPUs than ranks."""
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("IM2IM_DIST_BACKEND", "nccl")
    if expected_world is not None and world != expected_world:
        raise SystemExit(f"asked for {expected_world} ranks but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("no GPU visible: the HIP path has no CPU fallback")
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks over RCCL need {world} GPUs, this node shows {ndev} "
                         f"(IM2IM_DIST_BACKEND=gloo lets ranks share a GPU for functional tests)")
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world == 1:
        return None, 0, 1, dev, None
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    # a rank that never shows up must fail the job, not hang it until the driver's clock runs out.  The same timeout is the
    # watchdog of every later collective on the nccl backend, and on a fresh box the ranks' first `import torch` alone can be
    # a minute or two apart: 300 s, not 60 (a WRONG world -- too few devices or ranks -- is refused by verify_world within
    # seconds of the rendezvous, which is the check that has to be fast)
    timeout = datetime.timedelta(seconds=int(os.environ.get("IM2IM_RENDEZVOUS_TIMEOUT_S", "300")))
    if backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=dev, timeout=timeout)
    else:
        dist.init_process_group(backend=backend, timeout=timeout)
    if dist.get_world_size() != world:
        raise SystemExit(f"process group has {dist.get_world_size()} ranks, expected {world}")
    verify_world(dist, rank, world, dev, backend)
    # the short timeout above is for the rendezvous and the world check only.  Training has rank-0-only phases (a checkpoint on a slow
    # filesystem, validation images, a wandb upload) during which the other ranks wait in a collective: those get the backends' usual
    # patience back (nccl's default is 10 min, gloo's 30; IM2IM_COLLECTIVE_TIMEOUT_S overrides)
    relax_collective_timeout()
    return dist, rank, world, dev, backend


def relax_collective_timeout() -> bool:
    """the default process group's collective timeout -> IM2IM_COLLECTIVE_TIMEOUT_S (default 30 min); False when this torch cannot"""
    try:
        from torch.distributed.distributed_c10d import _set_pg_timeout
        _set_pg_timeout(datetime.timedelta(seconds=int(os.environ.get("IM2IM_COLLECTIVE_TIMEOUT_S", "1800"))))
        return True
    except Exception:  # noqa: BLE001  (a torch without the hook keeps the rendezvous timeout)
        return False


def device_identity(dev, rank: int) -> dict:
    """what tells two ranks' GPUs apart: host, index, name, uuid, PCI bus id (+ the pid that holds it).  `uuid` / `pci_bus_id` are None
    when this torch build does not expose them (then only the host and the local index are left to go by)."""
    import torch
    props = torch.cuda.get_device_properties(dev)
    uuid = str(getattr(props, "uuid", "") or "") or None
    pci = None
    if all(hasattr(props, k) for k in ("pci_domain_id", "pci_bus_id", "pci_device_id")):
        pci = "%04x:%02x:%02x" % (props.pci_domain_id, props.pci_bus_id, props.pci_device_id)
    return {"rank": rank, "host": socket.gethostname(), "local_device_index": dev.index, "name": props.name, "uuid": uuid, "pci_bus_id": pci,
            "pid": os.getpid()}


def identifiable(ranks) -> bool:
    """True when every rank reported a hardware identity (a uuid or a PCI address); without one, two GPUs cannot be told apart beyond
    (host, local index) and the distinct-device check says nothing."""
    return all(r.get("uuid") or r.get("pci_bus_id") for r in ranks)


def distinct_devices(ranks) -> int:
    """devices behind the ranks, counted PER HOST: two nodes with the same topology hold different GPUs at the same PCI address"""
    return len({(r.get("host"), r.get("uuid"), r.get("pci_bus_id"), r["local_device_index"]) for r in ranks})


def verify_world(dist, rank: int, world: int, dev, backend: str):
    """[r5] first collective of the job, right after the rendezvous: every rank's device identity is gathered over the job's own
    process group and a one-element all-reduce counts the ranks that answer.  N ranks over RCCL on fewer than N distinct devices,
    or a communicator that sums to anything but N, is a non-zero exit on EVERY rank here -- before any benchmark leg has run --
    instead of a line that says n_gpus = N about a job that was not.  (gloo with IM2IM_DIST_BACKEND=gloo is the functional-test
    path: ranks may share a GPU there and only the rank count is checked.)  Returns the gathered identities."""
    import torch
    ranks = [None] * world
    dist.all_gather_object(ranks, device_identity(dev, rank))
    one = torch.ones(1, device=dev if backend == "nccl" else "cpu")
    dist.all_reduce(one)
    n = int(round(float(one.item())))
    if n != world:
        raise SystemExit(f"[rank {rank}] the {backend} communicator summed {n} ranks, expected {world}")
    nd = distinct_devices(ranks)
    if backend == "nccl" and nd != world and identifiable(ranks):
        raise SystemExit(f"[rank {rank}] {world} RCCL ranks sit on {nd} distinct device(s): "
                         + ", ".join(f"rank {r['rank']} -> {r['pci_bus_id']}" for r in ranks))
    return ranks
