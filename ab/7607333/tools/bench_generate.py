#!/usr/bin/env python3
"""Developer tool: latency of autoregressive generation (RFN.predict / RFN.sample, RFN_new.py:256-360, 453-494) on the
canonical model: B sequences, 5 conditioning frames, 10 generated frames.  RFN_GEN_GRAPH=0: eager launches instead of
the per-frame hipGraph replay; RFN_GEN_CACHE=0: no per-call cache of inverse matrices and weight packs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("RFN_GRAPH_ENV_BEFORE_TORCH", "1")
import torch, bench
B = int(os.environ.get("B", 32))
solver, args = bench.build_solver(B, 20, torch.device("cuda"))
x = bench.make_batch(B, 20, 5, "cuda")
solver.train_step(x)  # ActNorm init
m = solver.model.eval()
xin = solver.preprocess(x)
mode = "graph" if os.environ.get("RFN_GEN_GRAPH", "1") != "0" else "eager"
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        m.sample(xin, 10)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("sample  B=%d (%s): %.1f ms for 10 frames = %.2f ms per frame-batch, %.0f frames/s" %
          (B, mode, 1e3 * dt, 1e2 * dt, B * 10 / dt), flush=True)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.no_grad():
        m.predict(xin, 10, 5)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("predict B=%d (%s): %.1f ms (5 conditioning + 10 generated frames)" % (B, mode, 1e3 * dt), flush=True)
print("graph builds:", getattr(m, "_gen_graph_builds", 0))
# where the time of one generated frame goes (eager step, HIP events)
from torch.profiler import profile, ProfilerActivity
with torch.no_grad():
    hp, cp, _, _, zp, _, _, _, _ = m.get_inits()
    eps = [torch.randn(sh, device="cuda") for sh in m._gen_eps_shapes(B)]
    m._gen_step(xin[:, 0], hp, cp, zp, eps, 1.0)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        m._gen_step(xin[:, 0], hp, cp, zp, eps, 1.0)
        torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
