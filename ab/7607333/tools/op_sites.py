#!/usr/bin/env python3
"""Developer tool: Python call sites of the aten ops that launch glue kernels (fill / copy / add / cat ...) in one eager
training step, through a TorchDispatchMode.  B=4 python tools/op_sites.py"""
import os, sys, traceback, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
import torch, bench
from torch.utils._python_dispatch import TorchDispatchMode
B, T = int(os.environ.get("B", 4)), 20
solver, args = bench.build_solver(B, T, torch.device("cuda"))
x = bench.make_batch(B, T, 5, "cuda")
for _ in range(3):
    solver.train_step(x)
torch.cuda.synchronize()
WATCH = ("fill_", "zero_", "zeros", "copy_", "add", "add_", "cat", "sum", "mul", "clone", "contiguous", "stack", "div", "sub",
         "neg", "mean", "_to_copy", "zeros_like", "ones_like", "full")
sites = collections.Counter()


class Spy(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split(".")[0]
        if name in WATCH:
            ts = [a for a in args if isinstance(a, torch.Tensor)]
            if any(t.is_cuda for t in ts) or (not ts and "cuda" in str((kwargs or {}).get("device", ""))):
                fr = [f for f in traceback.extract_stack()[:-1] if "recurrent-flows-msc_amd" in f.filename or f.filename.endswith("bench.py")]
                where = " <- ".join("%s:%d" % (f.filename.split("recurrent-flows-msc_amd/")[-1], f.lineno) for f in fr[-3:][::-1]) or "(autograd engine)"
                sites[(name, where)] += 1
        return func(*args, **(kwargs or {}))


with Spy():
    solver.train_step(x)
torch.cuda.synchronize()
tot = collections.Counter()
for (n, w), c in sites.items():
    tot[n] += c
print(dict(tot))
for (n, w), c in sites.most_common(70):
    print("%4d  %-10s %s" % (c, n, w))
