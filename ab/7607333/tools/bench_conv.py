#!/usr/bin/env python3
"""Developer micro-benchmark: the MFMA conv kernels (fwd / dgrad / wgrad) at the canonical SM-MNIST shapes,
timed with HIP events on the launch stream.  python tools/bench_conv.py [N_frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
import torch  # noqa: E402
from rfn_hip import ops as K  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 608
LEVELS = [(4, 16, 32), (8, 32, 16), (16, 64, 8), (32, 128, 4), (64, 256, 2)]  # (C, Cc, HW side)
Hd = 256


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    tot = {"fwd": [0, 0], "dgrad": [0, 0], "wgrad": [0, 0]}
    print("%-28s %10s %10s %10s   (TFLOP/s | ms)" % ("shape", "fwd", "dgrad", "wgrad"))
    for (C, Cc, S) in LEVELS:
        convs = [("conv1 3x3 %d->%d" % (C // 2 + Cc, Hd), C // 2 + Cc, Hd, 3),
                 ("conv2 1x1 %d->%d" % (Hd, Hd), Hd, Hd, 1),
                 ("conv3 3x3 %d->%d" % (Hd, C), Hd, C, 3)]
        for name, cin, cout, ks in convs:
            x = torch.randn(N, cin, S, S, device="cuda")
            w = torch.randn(cout, cin, ks, ks, device="cuda") * 0.05
            g = torch.randn(N, cout, S, S, device="cuda")
            wp, wpt = K.pack_weight(w), K.pack_weight(w, True)
            fl = 2.0 * N * S * S * cin * cout * ks * ks
            out = torch.empty(N, cout, S, S, device="cuda")
            gin = torch.empty(N, cin, S, S, device="cuda")
            t_f = timeit(lambda: K.conv2d_raw(x, None, wp, cout, ks, out1=out))
            t_d = timeit(lambda: K.conv2d_raw(g, None, wpt, cin, ks, out1=gin))
            t_w = timeit(lambda: K.conv2d_wgrad(x, None, g, cout, ks))
            for k, t in (("fwd", t_f), ("dgrad", t_d), ("wgrad", t_w)):
                tot[k][0] += fl
                tot[k][1] += t
            print("L%dx%-2d %-22s %5.1f|%6.3f %5.1f|%6.3f %5.1f|%6.3f" % (
                S, S, name, fl / t_f / 1e9, t_f, fl / t_d / 1e9, t_d, fl / t_w / 1e9, t_w))
    for k, (fl, t) in tot.items():
        print("total %-6s %.1f TFLOP/s  %.2f ms per GlowStep-set (x K=10 per train step)" % (k, fl / t / 1e9, t))


if __name__ == "__main__":
    main()
