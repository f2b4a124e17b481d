"""Probe: can RCCL run two ranks on ONE GPU (it would let a 1-GPU box exercise the multi-rank collectives)?
Launch with torch.distributed.run --nproc-per-node 2. Answer on this image (profiles/r02_rccl_two_ranks_one_gpu.txt):
no — ncclInvalidUsage, "Duplicate GPU detected". Multi-rank RCCL therefore needs one GPU per rank."""
import os, sys, torch, torch.distributed as dist, datetime
rank=int(os.environ["RANK"]); world=int(os.environ["WORLD_SIZE"])
dev=torch.device("cuda",0); torch.cuda.set_device(dev)
try:
    dist.init_process_group("nccl", device_id=dev, timeout=datetime.timedelta(seconds=60))
    t=torch.full((4,),float(rank),device=dev)
    out=torch.empty((4*world,),device=dev)
    dist.all_gather_into_tensor(out,t)
    torch.cuda.synchronize()
    print("rank",rank,"all_gather ok",out.tolist(),flush=True)
    dist.destroy_process_group()
except Exception as e:
    print("rank",rank,"FAILED:",repr(e)[:500],flush=True)
    sys.exit(3)
