#!/usr/bin/env python3
"""Developer tool: which aten ops (outside librfn_hip) launch kernels in one eager training step, and from where.
B=4 python tools/torch_ops.py   -> the ops by launch count, then the Python call sites of the most frequent ones."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
import torch, bench
from torch.profiler import profile, ProfilerActivity
B, T = int(os.environ.get("B", 32)), 20
solver, args = bench.build_solver(B, T, torch.device("cuda"))
x = bench.make_batch(B, T, 5, "cuda")
for _ in range(3):
    solver.train_step(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    solver.train_step(x)
    torch.cuda.synchronize()
rows = [e for e in prof.key_averages() if e.key.startswith("aten::") and e.self_device_time_total > 0]
rows.sort(key=lambda e: -e.count)
print("%-40s %8s %12s" % ("op", "calls", "self GPU us"))
for e in rows[:25]:
    print("%-40s %8d %12.0f" % (e.key, e.count, e.self_device_time_total))
hot = {e.key for e in rows[:8]}
sites = {}
for e in prof.key_averages(group_by_stack_n=8):
    if e.key in hot and e.self_device_time_total > 0:
        st = [s for s in e.stack if "recurrent-flows-msc_amd" in s or "bench.py" in s][:3]
        k = (e.key, " <- ".join(s.split("recurrent-flows-msc_amd/")[-1] for s in st))
        a = sites.setdefault(k, [0, 0.0])
        a[0] += e.count
        a[1] += e.self_device_time_total
print()
for (op, st), (n, us) in sorted(sites.items(), key=lambda kv: -kv[1][0])[:60]:
    print("%5d %8.0f us  %-22s %s" % (n, us, op, st))
