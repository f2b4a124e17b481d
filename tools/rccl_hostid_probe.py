"""Probe: several RCCL ranks on ONE GPU by giving every rank its own NCCL_HOSTID.

RCCL refuses two ranks on one device ("Duplicate GPU detected", profiles/r02_rccl_two_ranks_one_gpu.txt) — but only when
both ranks report the same host hash. With a distinct NCCL_HOSTID per rank the ranks look like different hosts: the
duplicate check passes and the data path is RCCL's NET transport over the loopback socket (NCCL_SOCKET_IFNAME=lo) instead
of xGMI P2P. Slow, but it is the real library: ncclCommInitRank with world > 1, ncclAllGather in place, grouped
ncclSend/ncclRecv — the calls of the sharded step that a 1-GPU box could not execute before.

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/rccl_hostid_probe.py
"""
import datetime
import os
import sys

rank = int(os.environ["RANK"])
world = int(os.environ["WORLD_SIZE"])
os.environ["NCCL_HOSTID"] = f"nbody-fake-host-{rank}"
os.environ.setdefault("NCCL_SOCKET_IFNAME", "lo")
os.environ.setdefault("NCCL_IB_DISABLE", "1")
os.environ.setdefault("NCCL_DEBUG", "WARN")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
try:
    dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=120))
    n = 1024
    full = torch.zeros((world * n, 4), device=dev)
    full[rank * n:(rank + 1) * n] = float(rank + 1)
    dist.all_gather_into_tensor(full, full[rank * n:(rank + 1) * n])      # in place, as the sharded step does
    torch.cuda.synchronize()
    want = torch.arange(1, world + 1, device=dev, dtype=torch.float32).repeat_interleave(n)
    ok_gather = bool(torch.equal(full[:, 0], want))
    # grouped send/recv: everybody sends its rank to the next rank, receives from the previous one
    out = torch.full((n, 4), float(rank), device=dev)
    inp = torch.empty((n, 4), device=dev)
    ops = [dist.P2POp(dist.isend, out, (rank + 1) % world), dist.P2POp(dist.irecv, inp, (rank - 1) % world)]
    for req in dist.batch_isend_irecv(ops):
        req.wait()
    torch.cuda.synchronize()
    ok_p2p = bool((inp == float((rank - 1) % world)).all())
    t = torch.tensor([float(rank)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    print(f"rank {rank}/{world}: all_gather_in_place={ok_gather} grouped_send_recv={ok_p2p} max={t.item()}", flush=True)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if (ok_gather and ok_p2p) else 4)
except Exception as e:  # noqa: BLE001
    print("rank", rank, "FAILED:", repr(e)[:800], flush=True)
    sys.exit(3)
