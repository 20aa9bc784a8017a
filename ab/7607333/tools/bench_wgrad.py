"""developer probe: the big weight-gradient GEMMs of the shallow flow levels (N = 608 frames), LDS-DMA ring kernel
against the register-staged one.  python tools/bench_wgrad.py   (RFN_WGRAD_DMA=0 for the register-staged kernel)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recurrent-flows-msc_amd")):
    sys.path.insert(0, p)
import torch
from rfn_hip import ops as K


def run(F_, M, Nc, S, reps=10):
    g = torch.Generator().manual_seed(0)
    a = torch.randn(F_, M, S, S, generator=g).cuda()
    b = torch.randn(F_, Nc, S, S, generator=g).cuda()
    K.gemm_wgrad(a, b, M, Nc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        K.gemm_wgrad(a, b, M, Nc)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    by = 4.0 * F_ * S * S * (M + Nc)
    print("F%d %dx%d %dx%d  %s  %.3f ms  %.2f TB/s  %.0f TFLOP/s" % (
        F_, M, Nc, S, S, K._gemm_wgrad_name(M, Nc, F_ * S * S, S * S).split("<")[0], dt * 1e3, by / dt / 1e12,
        2.0 * F_ * S * S * M * Nc / dt / 1e12), flush=True)


def run_implicit(F_, C1, C2, S, reps=10):
    g = torch.Generator().manual_seed(0)
    z = torch.randn(F_, C1, S, S, generator=g).cuda()
    cond = torch.randn(F_, C2, S, S, generator=g).cuda()
    ga = torch.randn(F_, 256, S, S, generator=g).cuda()
    K.conv2d_wgrad(z, cond, ga, 256, 3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        K.conv2d_wgrad(z, cond, ga, 256, 3)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    by = 4.0 * F_ * S * S * (256 + C1 + C2)
    print("F%d 256x%d %dx%d implicit3x3  %.3f ms  %.2f TB/s" % (F_, 9 * (C1 + C2), S, S, dt * 1e3, by / dt / 1e12), flush=True)


if __name__ == "__main__":
    run_implicit(608, 2, 16, 32)
    run_implicit(608, 4, 32, 16)
    run(608, 256, 256, 32)
    run(608, 256, 256, 16)
    run(608, 36, 256, 32)
    run(608, 72, 256, 16)
    run(608, 256, 256, 8)
