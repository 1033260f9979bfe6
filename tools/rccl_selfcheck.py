#!/usr/bin/env python3
"""The collective calls bench.py makes at N > 1, on an RCCL group of ONE rank (all a one-GPU box can host): group set-up with device_id,
gather into a list on the root from a side stream, barrier, all_reduce.  Says nothing about xGMI; it checks that the calls exist in this
torch / RCCL build and order correctly against a side stream."""
import os
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
side = torch.cuda.Stream(device=dev)
src = torch.arange(1 << 20, device=dev, dtype=torch.float32)
dst = [torch.zeros_like(src)]
torch.cuda.synchronize(dev)
with torch.cuda.stream(side):
    packed = src * 2
    dist.gather(packed, dst, dst=0)
    out = dst[0] + 1
    ev = torch.cuda.Event()
    ev.record(side)
torch.cuda.current_stream(dev).wait_event(ev)
dist.barrier()
torch.cuda.synchronize(dev)
assert torch.equal(out, src * 2 + 1)
t = torch.tensor([1.5], device=dev, dtype=torch.float64)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
r = torch.tensor([7], device=dev, dtype=torch.int64)
dist.all_reduce(r, op=dist.ReduceOp.SUM)
assert float(t.item()) == 1.5 and int(r.item()) == 7
dist.destroy_process_group()
print("rccl selfcheck ok")
