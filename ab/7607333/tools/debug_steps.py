"""eager train steps with per-step loss terms (debug aid): python tools/debug_steps.py [B] [steps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
dev = torch.device("cuda:0")
solver, args = bench.build_solver(B, 20, dev)
g = torch.Generator().manual_seed(1)
scale = float(os.environ.get('DBG_SCALE', '1'))
batches = [(torch.randint(0, 256, (B, 20, 1, 64, 64), generator=g).float() * scale).to(dev) for _ in range(2)]
for i in range(steps):
    solver.train_step(batches[i % 2])
    torch.cuda.synchronize()
    bad = [k for k, p in solver.model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
    print(i, "bits", solver.bits[-1], "kl", solver.kl_loss[-1], "nll", solver.recon_loss[-1], "nonfinite grads:", bad[:5],
          flush=True)
if len(sys.argv) > 3:  # graph replays
    del bad
    ok = solver.capture_graph(batches[0])
    print("captured", ok, flush=True)
    for i in range(int(sys.argv[3])):
        solver.train_step(batches[i % 2])
        torch.cuda.synchronize()
        badg = [k for k, p in solver.model.named_parameters() if p.grad is not None and not torch.isfinite(p.grad).all()]
        badp = [k for k, p in solver.model.named_parameters() if not torch.isfinite(p).all()]
        gmax = max((float(p.grad.abs().max()), k) for k, p in solver.model.named_parameters() if p.grad is not None)
        print("replay", i, [float(v) for v in solver._g_out.tolist()], "bad grads", badg[:4], "bad params", badp[:4],
              "max grad", gmax, flush=True)
        from rfn_hip import debug as D
        if D.ENABLED:
            print("   non-finite:", D.report()[:40], flush=True)
