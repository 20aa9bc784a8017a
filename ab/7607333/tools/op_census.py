"""Developer tool: aten-op census of one eager training step (count, device time, shapes): python tools/op_census.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from torch.profiler import profile, ProfilerActivity

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
solver, args = bench.build_solver(B, 20, dev)
g = torch.Generator().manual_seed(1)
batches = [torch.rand(B, 20, 1, 64, 64, generator=g).to(dev) for _ in range(2)]
for i in range(3):
    solver.train_step(batches[i % 2])
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    solver.train_step(batches[0])
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True)
rows = [(e.count, e.key, getattr(e, "device_time_total", getattr(e, "cuda_time_total", 0)) / 1e3, str(e.input_shapes)[:110])
        for e in ka if getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0)) > 0]
rows.sort(key=lambda r: -r[0])
print("%6s %-38s %9s  shapes" % ("count", "op", "dev ms"))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 70]:
    print("%6d %-38s %9.3f  %s" % r)
tot = {}
for e in prof.key_averages():
    sd = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
    if sd > 0:
        tot[e.key] = (e.count, sd / 1e3)
print("---- by op (self device time)")
for k, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%6d %-50s %9.3f ms" % (v[0], k, v[1]))
if os.environ.get("STACK"):
    # second pass: where in this repo do the glue ops come from (forward ops have a Python stack; native autograd
    # nodes of the backward do not and show up under "<no python frame>")
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof2:
        solver.train_step(batches[0])
        torch.cuda.synchronize()
    by = {}
    for e in prof2.key_averages(group_by_stack_n=12):
        sd = getattr(e, "self_device_time_total", getattr(e, "self_cuda_time_total", 0))
        if sd <= 0:
            continue
        fr = [s for s in e.stack if "recurrent-flows-msc_amd" in s or "bench.py" in s]
        where = fr[0].split("recurrent-flows-msc_amd/")[-1][:70] if fr else "<no python frame>"
        k = (e.key, where)
        c = by.setdefault(k, [0, 0.0])
        c[0] += e.count
        c[1] += sd / 1e3
    print("---- glue ops by source line")
    for k, v in sorted(by.items(), key=lambda kv: -kv[1][0])[:90]:
        print("%6d %9.3f ms  %-34s %s" % (v[0], v[1], k[0][:34], k[1]))
