"""RCCL rehearsal on a one-GPU box: a world of one rank over backend "nccl" runs the two exchanges
of the multi-GPU path (arg-min records, best descent) on device tensors."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from cameracalibration_amd.distributed import merge_argmin, merge_descents

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
t = time.time()
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
print(f"init {time.time()-t:.2f} s", flush=True)
t = time.time()
print(merge_argmin(0.25, 1234567, device=torch.device("cuda", 0)), f"{time.time()-t:.2f} s", flush=True)
f, k, x = merge_descents([3.0, 1.0, 1.0], np.arange(18.0).reshape(3, 6), 10, device=torch.device("cuda", 0))
print(f, k, x, flush=True)
assert (f, k) == (1.0, 11) and np.array_equal(x, np.arange(6.0, 12.0))
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("ok", flush=True)
