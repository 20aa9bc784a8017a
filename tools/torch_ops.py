#!/usr/bin/env python3
"""Developer tool: which aten ops (outside librfn_hip) cost GPU time in one eager training step."""
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
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    solver.train_step(x)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=60))
