"""Holds a GPU context with several busy-once streams, like a test runner that has already used the GPU, then sleeps.
Used in round 4's priority probe (profiles/r04a_rehearsal_priority_probe.txt; that round's visit script is gone) to see what a bystander process does to ranks that share the GPU (hardware queue slots)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kfd_queue_census  # noqa: E402

n_streams = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
x = torch.zeros(1 << 20, device="cuda")
streams = [torch.cuda.Stream() for _ in range(n_streams)] + [torch.cuda.Stream(priority=-1)]
for s in streams:
    with torch.cuda.stream(s):
        x.add_(1.0)
torch.cuda.synchronize()
print("holder", os.getpid(), "queues", kfd_queue_census(), flush=True)
time.sleep(seconds)
