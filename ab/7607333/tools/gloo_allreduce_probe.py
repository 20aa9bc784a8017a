"""Developer probe: cost of a gloo all-reduce on CUDA tensors of the gradient size (explains the slow 2-rank gloo rehearsal)."""
import os, time, torch, torch.distributed as dist
dist.init_process_group("gloo")
r = dist.get_rank()
x = torch.randn(36_500_000, device="cuda")
for n in (1, 3):
    torch.cuda.synchronize(); t = time.time()
    for _ in range(n):
        dist.all_reduce(x)
    torch.cuda.synchronize()
    if r == 0:
        print("gloo all_reduce of %.0f MB x%d: %.3f s each" % (x.numel() * 4 / 1e6, n, (time.time() - t) / n), flush=True)
y = x.cpu()
t = time.time(); dist.all_reduce(y); 
if r == 0:
    print("gloo all_reduce CPU tensor: %.3f s" % (time.time() - t), flush=True)
