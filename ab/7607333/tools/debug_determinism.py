"""developer probe: which sub-module of RFN.loss gives different bits on two identical evaluations?"""
import os, sys
os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "recurrent-flows-msc_amd"))
import torch
import main_rfn
from RFN import RFN
B, T = 2, 4
args = main_rfn.build_parser().parse_args(main_rfn.canonical_smmnist_argv(B, T))
torch.manual_seed(71)
m = RFN(args).cuda().train()
g = torch.Generator().manual_seed(72)
x = ((torch.rand(B, T, 1, 64, 64, generator=g) * 255).floor() / 256 - 0.5).cuda()
draws = []
for _ in range(T - 1):
    draws += [torch.randn(B, 56, 2, 2, generator=g).cuda(), torch.randn(B, 56, 2, 2, generator=g).cuda(),
              (torch.rand(B, 1, 64, 64, generator=g) / 256).cuda()]
with torch.no_grad():
    m.loss(x, 0, draws=draws)
    gp = torch.Generator().manual_seed(73)
    for prm in m.flow.parameters():
        prm.add_(0.003 * torch.randn(prm.shape, generator=gp).cuda())
rec = []
def flat(o):
    if torch.is_tensor(o):
        return [o]
    if isinstance(o, (list, tuple)):
        r = []
        for v in o:
            r += flat(v)
        return r
    return []
def hook(name):
    def f(mod, inp, out):
        rec.append((name, [t.detach().clone() for t in flat(out)]))
    return f
for name, mod in m.named_modules():
    if name and name.count(".") <= 1 and not name.startswith("flow.glow_frame"):
        mod.register_forward_hook(hook(name))
for i in range(len(m.flow.glow_frame)):
    m.flow.glow_frame[i].register_forward_hook(hook("flow.glow_frame.%d" % i))
runs = []
for rep in range(3):
    rec.clear()
    out = m.loss(x, 0, draws=draws)
    runs.append((list(rec), [o.detach().clone() for o in out]))
for rep in (1, 2):
    print("rep", rep, "loss equal:", [bool(torch.equal(a, b)) for a, b in zip(runs[rep][1], runs[0][1])])
    shown = 0
    for (n0, t0), (n1, t1) in zip(runs[0][0], runs[rep][0]):
        assert n0 == n1
        for j, (a, b) in enumerate(zip(t0, t1)):
            if a.shape == b.shape and not torch.equal(a, b):
                print("   first differing hook output:", n0, j, tuple(a.shape), float((a - b).abs().max()))
                shown += 1
                break
        if shown >= 6:
            break

print("---- extractor.forward_steps alone")
x_tm = x.transpose(0, 1).reshape(T * B, *x.shape[2:])
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    outs = []
    for rep in range(3):
        with torch.no_grad():
            f = m.extractor.forward_steps(x_tm, T)
        outs.append([t.clone() for t in (f if isinstance(f, (list, tuple)) else [f])])
    for rep in (1, 2):
        print("cudnn.deterministic", det, "rep", rep, [bool(torch.equal(a, b)) for a, b in zip(outs[rep], outs[0])],
              [float((a - b).abs().max()) for a, b in zip(outs[rep], outs[0])])
print("---- first conv of the extractor alone (MIOpen)")
conv = [mm for mm in m.extractor.modules() if isinstance(mm, torch.nn.Conv2d)][0]
for det in (False, True):
    torch.backends.cudnn.deterministic = det
    with torch.no_grad():
        ys = [conv(x_tm).clone() for _ in range(3)]
    print("det", det, [bool(torch.equal(ys[i], ys[0])) for i in (1, 2)])

print("---- upscaler.forward_steps alone (fixed random input)")
torch.backends.cudnn.deterministic = False
gi = torch.Generator().manual_seed(5)
uin = torch.randn((T - 1) * B, 256, 2, 2, generator=gi).cuda()
with torch.no_grad():
    ft = m.extractor.forward_steps(x_tm, T)
    skips = [f[:(T - 1) * B].contiguous() for f in ft]
    us = [[t.clone() for t in m.upscaler.forward_steps(uin, T - 1, skip_list=skips)] for _ in range(3)]
for rep in (1, 2):
    print("rep", rep, [bool(torch.equal(a, b)) for a, b in zip(us[rep], us[0])], [tuple(a.shape) for a in us[0]])
print("---- flow.log_prob alone (fixed conditions)")
conds = [c.contiguous() for c in us[0]]
base = torch.randn((T - 1) * B, 256, 2, 2, generator=gi).cuda()
xf = x[:, 1:].reshape((T - 1) * B, 1, 64, 64).contiguous()
noise = (torch.rand((T - 1) * B, 1, 64, 64, generator=gi) / 256).cuda()
res = []
for rep in range(3):
    with torch.no_grad():
        z, nll = m.flow.log_prob(xf, conds, base, torch.zeros((T - 1) * B, device='cuda'), noise=noise)
    res.append((z.clone(), nll.clone()))
for rep in (1, 2):
    print("rep", rep, "z equal", bool(torch.equal(res[rep][0], res[0][0])), "nll equal", bool(torch.equal(res[rep][1], res[0][1])),
          float((res[rep][1] - res[0][1]).abs().max()))
