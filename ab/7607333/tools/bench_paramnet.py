#!/usr/bin/env python3
"""Developer tool: tiny 2x2 convs of the prior/encoder nets — MIOpen (nn.Conv2d) vs rfn_hip ConvFn, fwd+bwd."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
import torch, torch.nn.functional as F
from rfn_hip import ops as K

def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for cin, cout in ((768, 256), (256, 256), (256, 112)):
    x = torch.randn(32, cin, 2, 2, device="cuda", requires_grad=True)
    w = (torch.randn(cout, cin, 3, 3, device="cuda") * 0.02).requires_grad_(True)
    b = torch.zeros(cout, device="cuda", requires_grad=True)
    z = torch.zeros(cout, device="cuda")
    def mi():
        y = F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
        y.sum().backward()
    def mine():
        y = K.conv_ep(x, None, w, b, z, 1, 2)
        y.sum().backward()
    def mi_f():
        with torch.no_grad(): F.leaky_relu(F.conv2d(x, w, b, padding=1), 0.2)
    def mine_f():
        with torch.no_grad(): K.conv_ep(x, None, w, b, z, 1, 2)
    print("%d->%d  fwd: miopen %.0f us  mine %.0f us | fwd+bwd: miopen %.0f us  mine %.0f us" % (
        cin, cout, timeit(mi_f), timeit(mine_f), timeit(mi), timeit(mine)))
