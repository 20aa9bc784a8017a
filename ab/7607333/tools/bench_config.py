#!/usr/bin/env python3
"""Developer tool: training-step time of the OTHER BASELINE configurations on one GPU (the bench line itself is the
canonical SM-MNIST model): `bair` = RFN on 64x64x3 video with the deep decoder (K=16, L=4, Hd=256, with_skip, D=2
overshooting) at the per-GPU batch of the DP=8 setting (32 / 8 = 4 sequences of 20 frames); `kth` = the canonical
architecture at 64 / 8 = 8 sequences.  Synthetic inputs, hipGraph replay, fwd + bwd + Adam."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
os.environ.setdefault("RFN_GRAPH_ENV_BEFORE_TORCH", "1")
import torch, bench
which = sys.argv[1] if len(sys.argv) > 1 else "bair"
dev = torch.device("cuda")
T = 20
if which == "bair":
    B, C = 4, 3
    argv = ("--extractor_structure 32-32-pool-64 64-pool-128 128-pool-256 256-pool-512 "
            "--upscaler_structure 256 upsample-128-128 upsample-64-64 upsample-32-32 "
            "--prior_structure 256 256 --encoder_structure 256 256 --make_conditional --learn_prior "
            "--skip_connection_features --flow_norm actnorm --structure_scaler 2 --choose_data bair "
            "--n_units_affine 256 --n_units_prior 256 --temperature 0.7 --norm_type none --z_dim 32 --h_dim 128 "
            "--n_bits 8 --n_frames %d --image_size 64 --K 16 --L 4 --D 2 --overshot_w 0.5 "
            "--x_dim %d 3 64 64 --condition_dim %d 3 64 64 --batch_size %d "
            "--skip_connection_flow with_skip --no-upscaler_tanh --no-downscaler_tanh --synthetic_data" % (T, B, B, B)).split()
else:
    B, C, argv = 8, 1, None
solver, args = bench.build_solver(B, T, dev, argv=argv)
g = torch.Generator().manual_seed(3)
x = torch.rand(B, T, C, 64, 64, generator=g).to(dev)
for _ in range(3):
    solver.train_step(x)
ok = solver.capture_graph(x)
print("hipGraph capture:", "ok" if ok else "FAILED " + getattr(solver, "_graph_error", ""), flush=True)
solver.train_step(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 10
for _ in range(K):
    solver.train_step(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
solver.flush_log()
print("%s: B=%d T=%d  %.1f ms/step  %.0f frames/s  bits/dim %.3f" % (which, B, T, 1e3 * dt, B * T / dt, solver.bits[-1]), flush=True)
