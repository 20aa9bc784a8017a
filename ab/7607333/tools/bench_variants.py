#!/usr/bin/env python3
"""Developer tool: time one conv shape under RFN_CONV_VARIANT=0..n (each variant in a child process)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import os, sys
sys.path.insert(0, os.path.join(%r, "recurrent-flows-msc_amd"))
import torch
from rfn_hip import ops as K
cin, cout, S, ks, N = %s
x = torch.randn(N, cin, S, S, device="cuda"); w = torch.randn(cout, cin, ks, ks, device="cuda") * 0.05
wp = K.pack_weight(w); out = torch.empty(N, cout, S, S, device="cuda")
f = lambda: K.conv2d_raw(x, None, wp, cout, ks, out1=out)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
t = e0.elapsed_time(e1) / 10
ref = torch.nn.functional.conv2d(x[:2], w, padding=ks // 2)
err = float((out[:2] - ref).abs().max() / ref.abs().max())
print("variant %%s: %%.3f ms  %%.1f TFLOP/s  relerr %%.1e" %% (os.environ.get("RFN_CONV_VARIANT", "0"), t, 2.0 * N * S * S * cin * cout * ks * ks / t / 1e9, err))
'''
shape = tuple(int(a) for a in sys.argv[1:6]) if len(sys.argv) > 5 else (256, 256, 32, 1, 608)
for v in range(int(os.environ.get("NVAR", 6))):
    env = dict(os.environ, RFN_CONV_VARIANT=str(v))
    r = subprocess.run([sys.executable, "-c", code % (ROOT, repr(shape))], env=env, capture_output=True, text=True)
    print((r.stdout.strip().splitlines() or ["(no output) " + r.stderr[-300:]])[-1], flush=True)
