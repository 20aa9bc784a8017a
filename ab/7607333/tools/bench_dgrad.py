#!/usr/bin/env python3
"""Developer tool: the few-output-channel 3x3 data-gradient kernel (csrc/dgrad_small.hip) against the generic bf16x3
implicit-GEMM kernel on the canonical level-0 / level-1 shapes (N = 608 frames)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
import torch
from rfn_hip import ops as K

def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

for (N, Cin, Cout, S, split) in [(608, 256, 18, 32, 2), (608, 256, 36, 16, 4)]:
    x = torch.randn(N, Cin, S, S, device="cuda")
    w = torch.randn(Cin, Cout, 3, 3, device="cuda") * 0.05
    wpk = K.pack_weight(w, flip=True)
    o1 = torch.zeros(N, split, S, S, device="cuda")
    o2 = torch.zeros(N, Cout - split, S, S, device="cuda")
    t_new = timeit(lambda: K.conv3x3_smallcout(x, wpk, Cout, o1, o2, split, True, False))
    t_old = timeit(lambda: K.conv2d_raw(x, None, wpk, Cout, 3, 0, None, None, 0, out1=o1, out2=o2, cout_split=split,
                                        acc1=True, acc2=False))
    gb = 4.0 * N * S * S * (Cin + Cout) / 1e9
    print("N%d %d->%d %dx%d: dgrad_small %.3f ms (%.0f GB/s)   generic %.3f ms (%.0f GB/s)" %
          (N, Cin, Cout, S, S, t_new, gb / t_new * 1e3, t_old, gb / t_old * 1e3), flush=True)
