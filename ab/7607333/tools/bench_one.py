#!/usr/bin/env python3
"""Developer tool: run ONE conv shape repeatedly (for rocprofv3 --pmc).  args: kind cin cout S ks [N]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
import torch
from rfn_hip import ops as K
from rfn_hip import lib as _L
if os.environ.get('RFN_LIB'):
    _L.LIB_PATH = os.path.join(ROOT, os.environ['RFN_LIB'])
kind, cin, cout, S, ks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
N = int(sys.argv[6]) if len(sys.argv) > 6 else 608
x = torch.randn(N, cin, S, S, device="cuda")
w = torch.randn(cout, cin, ks, ks, device="cuda") * 0.05
g = torch.randn(N, cout, S, S, device="cuda")
wp = K.pack_weight(w)
out = torch.empty(N, cout, S, S, device="cuda")
def run():
    if kind == "fwd":
        K.conv2d_raw(x, None, wp, cout, ks, out1=out)
    else:
        K.conv2d_wgrad(x, None, g, cout, ks)
for _ in range(5):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    run()
e1.record()
torch.cuda.synchronize()
print("%s cin%d cout%d S%d k%d N%d: %.1f us per call" % (kind, cin, cout, S, ks, N, e0.elapsed_time(e1) / 20 * 1e3))
