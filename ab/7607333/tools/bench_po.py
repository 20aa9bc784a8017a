"""developer probe: time the fused coupling-net forward (coupling_po_fwd) against the unfused kernels on one level-0 /
level-1 shaped GlowStep (N = 608 frames).  python tools/bench_po.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recurrent-flows-msc_amd")):
    sys.path.insert(0, p)
import torch
from rfn_hip import ops as K
from rfn_hip import lib as _L
if os.environ.get("RFN_LIB"):   # A/B of two builds of the library on one box
    _L.LIB_PATH = os.path.join(ROOT, os.environ["RFN_LIB"])

def run(N, C, Cc, S, reps=10):
    g = torch.Generator().manual_seed(0)
    Ch, Cin = C // 2, C // 2 + Cc
    z = torch.randn(N, C, S, S, generator=g).cuda()
    cond = torch.randn(N, Cc, S, S, generator=g).cuda()
    w1 = (torch.randn(256, Cin, 3, 3, generator=g) * 0.05).cuda()
    w2 = (torch.randn(256, 256, 1, 1, generator=g) * 0.05).cuda()
    w3 = (torch.randn(C, 256, 3, 3, generator=g) * 0.05).cuda()
    nb = (torch.randn(256, generator=g) * 0.1).cuda()
    nl = (torch.randn(256, generator=g) * 0.1).cuda()
    plan = K.POPackPlan([(w1, w2, w3)])
    plan.run()
    def fused():
        return K.coupling_po_fwd(z, cond, plan.bufs[0], nb, nl, nb, nl, C, 1)
    pk1, pk2 = K.pack_weight(w1), K.pack_weight(w2)
    w3t = w3.permute(2, 3, 0, 1).reshape(9 * C, 256, 1, 1).contiguous()
    pk3 = K.pack_weight(w3t)
    def unfused():
        h1 = K.conv2d_raw(z[:, :Ch], cond, pk1, 256, 3, 1, nb, nl, 1)
        h2 = K.conv2d_raw(h1, None, pk2, 256, 1, 1, nb, nl, 1)
        P = K.conv2d_raw(h2, None, pk3, 9 * C, 1)
        return h1, h2, P
    for name, fn in (("fused f16x3s", fused), ("unfused x3", unfused)):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        fl = 2.0 * N * S * S * (9 * Cin * 256 + 65536 + 9 * C * 256)
        print("N%d C%d Cc%d %dx%d %-11s %.3f ms  %.1f TFLOP/s fp32-equiv" % (N, C, Cc, S, S, name, dt * 1e3, fl / dt / 1e12), flush=True)

if __name__ == "__main__":
    run(608, 4, 16, 32)
    run(608, 8, 32, 16)
    run(608, 16, 64, 8)
    run(76, 12, 64, 32)

