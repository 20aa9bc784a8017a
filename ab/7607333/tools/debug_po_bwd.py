"""developer probe: where does the fused coupling-net backward kernel differ from fp64 autograd?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
import torch, torch.nn.functional as F
from rfn_hip import ops as K
N, C, Cc, S, act = [int(v) for v in (sys.argv[1:6] if len(sys.argv) > 5 else (41, 4, 16, 32, 1))]
g = torch.Generator().manual_seed(23)
Ch, Cin = C // 2, C // 2 + Cc
z = torch.randn(N, C, S, S, generator=g); cond = torch.randn(N, Cc, S, S, generator=g)
w1 = torch.randn(256, Cin, 3, 3, generator=g) * 0.05; w2 = torch.randn(256, 256, 1, 1, generator=g) * 0.05
w3 = torch.randn(C, 256, 3, 3, generator=g) * 0.05
n1b, n1l = torch.randn(256, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
n2b, n2l = torch.randn(256, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
go = torch.randn(N, C, S, S, generator=g)
actf = (lambda t: F.relu(t)) if act == 1 else (lambda t: F.leaky_relu(t, 0.2))
d = lambda t: t.double().clone().requires_grad_(True)
w1d, w2d, w3d, n1bd, n1ld, n2bd, n2ld = d(w1), d(w2), d(w3), d(n1b), d(n1l), d(n2b), d(n2l)
xin = torch.cat((z[:, :Ch], cond), 1).double()
a1 = F.conv2d(xin, w1d, padding=1); a1.retain_grad()
h1r = actf((a1 + n1bd.view(1, -1, 1, 1)) * n1ld.exp().view(1, -1, 1, 1))
a2 = F.conv2d(h1r, w2d); a2.retain_grad()
h2r = actf((a2 + n2bd.view(1, -1, 1, 1)) * n2ld.exp().view(1, -1, 1, 1))
F.conv2d(h2r, w3d, padding=1).backward(go.double())
cu = lambda t: t.cuda()
w1c, w2c, w3c = cu(w1), cu(w2), cu(w3)
plan = K.POPackPlan([(w1c, w2c, w3c)]); plan.run()
n1bc, n1lc, n2bc, n2lc = cu(n1b), cu(n1l), cu(n2b), cu(n2l)
h1, h2, P, masks = K.coupling_po_fwd(cu(z), cu(cond), plan.bufs[0], n1bc, n1lc, n2bc, n2lc, C, act, want_masks=True)
for rep in range(2):
    ga2, ga1, part = K.coupling_po_bwd(cu(go), plan.bwd_bufs[0], n1lc, n2lc, masks, act)
    torch.cuda.synchronize()
    for name, got, ref in (("ga2", ga2, a2.grad), ("ga1", ga1, a1.grad)):
        e = (got.cpu().double() - ref).abs()
        print(name, "rep", rep, "max err", float(e.max()), "max ref", float(ref.abs().max()))
        thr = 1e-4 * float(ref.abs().max())
        bad = (e > thr)
        print("  bad elements", int(bad.sum()), "of", bad.numel())
        if bad.any():
            idx = bad.nonzero()
            print("  frames", sorted(set(idx[:, 0].tolist()))[:50])
            print("  channels", sorted(set(idx[:, 1].tolist()))[:64], "n", len(set(idx[:, 1].tolist())))
            print("  rows", sorted(set(idx[:, 2].tolist())))
            print("  cols", sorted(set(idx[:, 3].tolist())))
            i0 = idx[0].tolist()
            print("  first", i0, float(got[tuple(i0)]), float(ref[tuple(i0)]))
            # the forward values behind the activation mask of the worst element
            iw = (e == e.max()).nonzero()[0].tolist()
            hg, hr = (h2, h2r) if name == "ga2" else (h1, h1r)
            print("  worst", iw, "forward value gpu %.9g  ref %.9g  (max |h| %.4g)" % (
                float(hg[tuple(iw)]), float(hr[tuple(iw)]), float(hr.abs().max())))
            fe = (hg.cpu().double() - hr.detach()).abs()
            print("  forward error: max %.3g at %s" % (float(fe.max()), (fe == fe.max()).nonzero()[0].tolist()))
            # mask disagreement? compare got==0 vs ref==0
            mz = ((got.cpu() == 0) != (ref == 0))
            print("  zero-pattern mismatches", int(mz.sum()))
            rel = e[bad] / ref.abs()[bad].clamp_min(1e-30)
            print("  rel err of bad: min %.3g median %.3g max %.3g" % (float(rel.min()), float(rel.median()), float(rel.max())))
