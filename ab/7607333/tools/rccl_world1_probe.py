"""developer probe: does RCCL (torch.distributed backend "nccl") initialise and run the collectives the data-parallel
path uses -- on device tensors, on the one GPU of a box (world size 1: the transport is trivial, the library, the
process-group plumbing and the in-place flat-bucket calls are the real ones)?  python tools/rccl_world1_probe.py"""
import os, sys, time
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist
t0 = time.time()
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
flat = torch.randn(36_498_112, device="cuda")          # the canonical model's gradient bytes (146 MB) as one bucket
ref = flat.clone()
torch.cuda.synchronize()
t1 = time.time()
dist.all_reduce(flat)
torch.cuda.synchronize()
t2 = time.time()
for _ in range(5):
    dist.all_reduce(flat)
torch.cuda.synchronize()
t3 = time.time()
assert torch.equal(flat, ref)
b = torch.arange(8, device="cuda", dtype=torch.float32)
dist.broadcast(b, 0)
parts = [torch.empty_like(b)]
dist.all_gather(parts, b)
assert torch.equal(parts[0], b)
flag = torch.tensor([1], device="cuda")
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.barrier()
print("backend", dist.get_backend(), "| init %.2f s | first all_reduce of 146 MB %.1f ms | steady %.2f ms each" % (
    t1 - t0, (t2 - t1) * 1e3, (t3 - t2) / 5 * 1e3))
dist.destroy_process_group()
print("ok")
