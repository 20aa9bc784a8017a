#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE implementation on CPU.

Test infrastructure only.  Runs ONLY in the build container (needs
/root/reference, which never travels to the GPU box).  It imports the
reference's own `Flow`, `Utils` and `RFN.RFN_new` modules, feeds them seeded
inputs and writes *data* (inputs, parameters, captured RNG draws, outputs,
gradients) as small `.pt` files next to this script.  No reference source is
copied: the fixtures hold tensors and plain Python scalars only, and are
loaded with `torch.load(..., weights_only=True)`.

Usage:  python tests/golden/make_golden.py          (from the repo root)

Shims needed to run the reference on CPU under torch 2.10:
  * `torch.Tensor.cuda` -> identity (RFN_new.py hard-codes `.cuda()`).
  * RNG capture: `torch.distributions.normal._standard_normal` and
    `torch.Tensor.uniform_` are wrapped to record every draw, so the oracle and
    the HIP path can be fed the very same noise.
"""
import os
import sys
import warnings
from argparse import Namespace

REF = os.environ.get("RFN_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.dont_write_bytecode = True
warnings.filterwarnings("ignore")

import torch  # noqa: E402

torch.Tensor.cuda = lambda self, *a, **k: self  # CPU shim
torch.set_num_threads(4)
torch.use_deterministic_algorithms(False)

import torch.distributions.normal as _tdn  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

# --------------------------------------------------------------------------- RNG capture
_draws = []
_orig_std_normal = _tdn._standard_normal
_orig_uniform_ = torch.Tensor.uniform_


def _rec_std_normal(shape, dtype, device):
    e = _orig_std_normal(shape, dtype, device)
    _draws.append(("normal", e.detach().clone()))
    return e


def _rec_uniform_(self, *a, **k):
    r = _orig_uniform_(self, *a, **k)
    _draws.append(("uniform", r.detach().clone()))
    return r


_orig_normal = torch.normal


def _rec_normal(mean, std, *a, **k):
    """torch.normal(mean, std) as used by td.Normal.sample (Split2d reverse, base prior sampling):
    drawn here as mean + std*eps with eps recorded, so the draw can be replayed."""
    e = _orig_std_normal(mean.shape, mean.dtype, mean.device)
    _draws.append(("normal_eps", e.detach().clone()))
    return mean + std * e


def start_capture():
    _draws.clear()
    _tdn._standard_normal = _rec_std_normal
    torch.Tensor.uniform_ = _rec_uniform_
    torch.normal = _rec_normal


def stop_capture():
    _tdn._standard_normal = _orig_std_normal
    torch.Tensor.uniform_ = _orig_uniform_
    torch.normal = _orig_normal
    d = list(_draws)
    _draws.clear()
    return d


def sd_clone(m):
    return {k: v.detach().clone() for k, v in m.state_dict().items()}


def grads_of(m):
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


def save(name, obj):
    path = os.path.join(OUT, name)
    torch.save(obj, path)
    print("wrote %s (%.1f KB)" % (name, os.path.getsize(path) / 1024))


def randomize_(m, gen, std=0.1):
    """Perturb every parameter so that zero-initialised layers (Conv2dZeros,
    realnvp scale, ActNorm) exercise non-trivial arithmetic."""
    with torch.no_grad():
        for n, p in m.named_parameters():
            p.add_(torch.randn(p.shape, generator=gen) * std)


# --------------------------------------------------------------------------- module-level fixtures
def gen_modules():
    from Flow.glow_modules import (ActNorm, InvConv, AffineCoupling, Squeeze2d, Split2d,
                                   Conv2dNorm, Conv2dZeros)
    from Utils import split_feature
    g = torch.Generator().manual_seed(1234)
    fx = {}

    # a1 Squeeze2d (bit exact) --------------------------------------------------
    x = torch.randn(2, 3, 4, 6, generator=g)
    sq = Squeeze2d()
    y = sq(x, undo_squeeze=False)
    fx["squeeze"] = {"x": x, "y": y, "x_back": sq(y, undo_squeeze=True)}

    # a2 split_feature -----------------------------------------------------------
    t = torch.randn(2, 6, 2, 2, generator=g)
    a, b = split_feature(t, "split")
    c, d = split_feature(t, "cross")
    fx["split_feature"] = {"x": t, "split0": a.clone(), "split1": b.clone(),
                           "cross0": c.clone(), "cross1": d.clone()}

    # a3 ActNorm: data dependent init + fwd + reverse ----------------------------
    an = ActNorm(5)
    an.train()
    x = torch.randn(3, 5, 4, 4, generator=g) * 2.0 + 0.7
    y, ld = an(x, torch.zeros(3), reverse=False)
    xb, ldb = an(y, torch.zeros(3), reverse=True)
    fx["actnorm_init"] = {"x": x, "y": y.detach(), "logdet": ld.detach(), "x_back": xb.detach(),
                          "logdet_back": ldb.detach(), "sd": sd_clone(an)}
    an_e = ActNorm(5)
    an_e.eval()  # eval: init skipped but flag still set (glow_modules.py:22-24,34-36)
    y_e, _ = an_e(x, None, reverse=False)
    fx["actnorm_eval_noinit"] = {"x": x, "y": y_e.detach(), "sd": sd_clone(an_e)}

    # a4 InvConv (LU and plain) ----------------------------------------------------
    for C in (4, 8):
        torch.manual_seed(100 + C)
        ic = InvConv(C, LU_decomposed=True)
        randomize_(ic, g, 0.05)
        x = torch.randn(2, C, 3, 5, generator=g, requires_grad=True)
        z, ld = ic(x, torch.zeros(2), reverse=False)
        w, dld = ic.get_weight(x, reverse=False)
        (z.square().sum() + ld.sum()).backward()
        xb, ldb = ic(z.detach(), torch.zeros(2), reverse=True)
        fx["invconv_lu_%d" % C] = {"sd": sd_clone(ic), "x": x.detach(), "z": z.detach(), "logdet": ld.detach(),
                                   "weight": w.detach().view(C, C), "dlogdet": dld.detach(),
                                   "x_back": xb.detach(), "logdet_back": ldb.detach(),
                                   "grad_x": x.grad.clone(), "grads": grads_of(ic)}
    torch.manual_seed(7)
    ic = InvConv(4, LU_decomposed=False)
    x = torch.randn(2, 4, 3, 3, generator=g)
    z, ld = ic(x, torch.zeros(2), reverse=False)
    fx["invconv_plain_4"] = {"sd": sd_clone(ic), "x": x, "z": z.detach(), "logdet": ld.detach()}

    # a5.1 / a5.2 Conv2dNorm / Conv2dZeros -----------------------------------------
    torch.manual_seed(11)
    cn = Conv2dNorm(5, 7)
    cn.train()
    x = torch.randn(3, 5, 6, 6, generator=g)
    y = cn(x)  # first call initialises the ActNorm
    y2 = cn(x)
    fx["conv2dnorm"] = {"sd": sd_clone(cn), "x": x, "y_first": y.detach(), "y": y2.detach()}
    torch.manual_seed(12)
    cn1 = Conv2dNorm(5, 7, kernel_size=[1, 1])
    cn1.train()
    y = cn1(x)
    fx["conv2dnorm_1x1"] = {"sd": sd_clone(cn1), "x": x, "y": y.detach()}
    torch.manual_seed(13)
    cz = Conv2dZeros(5, 6)
    randomize_(cz, g, 0.1)
    y = cz(x)
    fx["conv2dzeros"] = {"sd": sd_clone(cz), "x": x, "y": y.detach()}

    # a5 AffineCoupling: every clamp type, fwd/reverse/grads -----------------------
    for clamp in ("realnvp", "glow", "softclamp", "none"):
        for non_lin in ("relu", "leakyrelu"):
            torch.manual_seed(21)
            ac = AffineCoupling([2, 6, 4, 4], [2, 5, 4, 4], hidden_units=16, non_lin=non_lin, clamp_type=clamp)
            ac.train()
            x0 = torch.randn(2, 6, 4, 4, generator=g)
            c0 = torch.randn(2, 5, 4, 4, generator=g)
            ac(x0, c0, None, False)  # data dependent init of the two inner ActNorms
            randomize_(ac, g, 0.1)
            x = x0.clone().requires_grad_(True)
            c = c0.clone().requires_grad_(True)
            y, ld = ac(x, c, torch.zeros(2), False)
            wgt = torch.randn(y.shape, generator=g)
            ((y * wgt).sum() + (ld * torch.tensor([0.3, -1.1])).sum()).backward()
            xb, ldb = ac(y.detach(), c.detach(), torch.zeros(2), True)
            fx["affine_%s_%s" % (clamp, non_lin)] = {
                "sd": sd_clone(ac), "x": x.detach(), "cond": c.detach(), "y": y.detach(), "logdet": ld.detach(),
                "wgt": wgt, "gld": torch.tensor([0.3, -1.1]), "grad_x": x.grad.clone(), "grad_cond": c.grad.clone(),
                "grads": grads_of(ac), "x_back": xb.detach(), "logdet_back": ldb.detach()}

    # a7 Split2d (conditional / unconditional, softplus / exp) --------------------
    for cond_on in (True, False):
        for clampf in ("softplus", "exp"):
            torch.manual_seed(31)
            sp = Split2d([2, 8, 4, 4], [2, 6, 4, 4], make_conditional=cond_on, clamp_function=clampf)
            sp.train()
            x0 = torch.randn(2, 8, 4, 4, generator=g)
            c0 = torch.randn(2, 6, 4, 4, generator=g)
            sp(x0, c0, torch.zeros(2), False)
            randomize_(sp, g, 0.1)
            x = x0.clone().requires_grad_(True)
            c = c0.clone().requires_grad_(True)
            z1, ld = sp(x, c, torch.zeros(2), False)
            wgt = torch.randn(z1.shape, generator=g)
            ((z1 * wgt).sum() + (ld * torch.tensor([0.5, 2.0])).sum()).backward()
            fx["split2d_%s_%s" % ("cond" if cond_on else "uncond", clampf)] = {
                "sd": sd_clone(sp), "x": x.detach(), "cond": c.detach(), "z1": z1.detach(), "logdet": ld.detach(),
                "wgt": wgt, "gld": torch.tensor([0.5, 2.0]), "grad_x": x.grad.clone(),
                "grad_cond": c.grad.clone() if c.grad is not None else torch.zeros_like(c), "grads": grads_of(sp)}
    save("modules.pt", fx)


def glow_args(**kw):
    a = dict(learn_prior=True, n_units_prior=16, make_conditional=True, base_norm="actnorm",
             non_lin_glow="relu", split2d_act="softplus", L=2, K=2, n_bits=8, LU_decomposed=True,
             n_units_affine=16, clamp_type="realnvp", flow_norm="actnorm", flow_batchnorm_momentum=0.0)
    a.update(kw)
    return Namespace(**a)


def gen_glow():
    from Flow import ListGlow
    from Flow.glow import GlowStep
    g = torch.Generator().manual_seed(4321)
    fx = {}

    # a6 GlowStep fwd + reverse + grads -------------------------------------------
    torch.manual_seed(41)
    args = glow_args()
    gs = GlowStep([2, 8, 4, 4], [2, 6, 4, 4], args)
    gs.train()
    x0 = torch.randn(2, 8, 4, 4, generator=g)
    c0 = torch.randn(2, 6, 4, 4, generator=g)
    gs(x0, c0, torch.zeros(2), False)
    randomize_(gs, g, 0.05)
    x = x0.clone().requires_grad_(True)
    c = c0.clone().requires_grad_(True)
    y, ld = gs(x, c, torch.zeros(2), False)
    wgt = torch.randn(y.shape, generator=g)
    gld = torch.tensor([0.7, -0.4])
    ((y * wgt).sum() + (ld * gld).sum()).backward()
    xb, ldb = gs(y.detach(), c.detach(), torch.zeros(2), True)
    fx["glowstep"] = {"sd": sd_clone(gs), "x": x.detach(), "cond": c.detach(), "y": y.detach(), "logdet": ld.detach(),
                      "wgt": wgt, "gld": gld, "grad_x": x.grad.clone(), "grad_cond": c.grad.clone(),
                      "grads": grads_of(gs), "x_back": xb.detach(), "logdet_back": ldb.detach()}

    # a8 ListGlow.log_prob: (i) first training call = data dependent init; (ii) steady state + grads
    for name, kw, xs, cs in (
        ("listglow_L2K2", dict(), [3, 1, 8, 8], [[3, 6, 4, 4], [3, 10, 2, 2]]),
        ("listglow_L3K2_rgb_leaky_glowclamp", dict(L=3, K=2, non_lin_glow="leakyrelu", clamp_type="glow", n_bits=5,
                                                   split2d_act="exp"),
         [2, 3, 16, 16], [[2, 4, 8, 8], [2, 6, 4, 4], [2, 8, 2, 2]]),
        ("listglow_uncond", dict(make_conditional=False, learn_prior=False), [2, 3, 8, 8],
         [[2, 0, 4, 4], [2, 0, 2, 2]]),
    ):
        torch.manual_seed(51)
        args = glow_args(**kw)
        base = (xs[0], 12, cs[-1][2], cs[-1][3])
        import Flow.glow as fg
        fg.device = torch.device("cpu")
        flow = ListGlow(xs, cs, base, args)
        flow.train()
        sd_fresh = sd_clone(flow)
        x = torch.rand(xs, generator=g) - 0.5
        conds = [torch.randn(c, generator=g) for c in cs]
        bc = torch.randn(base, generator=g)
        start_capture()
        z, nll = flow.log_prob(x, conds, bc, 0)
        d = stop_capture()
        noise0 = d[0][1]
        entry = {"args": vars(args), "x_size": xs, "cond_sizes": cs, "base_size": list(base),
                 "sd_fresh": sd_fresh, "sd_init": sd_clone(flow), "x": x, "conds": conds, "base_cond": bc,
                 "noise_init": noise0, "z_init": z.detach(), "nll_init": nll.detach()}
        randomize_(flow, g, 0.03)
        x2 = torch.rand(xs, generator=g) - 0.5
        conds2 = [torch.randn(c, generator=g).requires_grad_(True) for c in cs]
        bc2 = torch.randn(base, generator=g).requires_grad_(True)
        start_capture()
        z2, nll2 = flow.log_prob(x2, conds2, bc2, 0)
        d = stop_capture()
        nll2.mean().backward()
        entry.update({"sd": sd_clone(flow), "x2": x2, "conds2": [c.detach() for c in conds2], "base_cond2": bc2.detach(),
                      "noise2": d[0][1], "z2": z2.detach(), "nll2": nll2.detach(), "grads": grads_of(flow),
                      "grad_conds2": [c.grad.clone() if c.grad is not None else torch.zeros_like(c) for c in conds2],
                      "grad_base_cond2": bc2.grad.clone() if bc2.grad is not None else torch.zeros_like(bc2)})
        # reverse path: g(f(x)) with the last level only being exactly invertible; store x from z (sample w/ given z)
        flow.eval()
        torch.manual_seed(99)
        start_capture()
        xs_ = flow.sample(z2.detach(), [c.detach() for c in conds2], bc2.detach(), num_samples=xs[0], temperature=0.8)
        d = stop_capture()
        entry.update({"sample_from_z2": xs_.detach(), "sample_draws": [t for _, t in d]})
        fx[name] = entry
    save("glow.pt", fx)

    # canonical level-0 GlowStep slice (C=4, Cc=16, Hd=256, 32x32), B=1
    torch.manual_seed(61)
    args = glow_args(n_units_affine=256)
    gs = GlowStep([1, 4, 32, 32], [1, 16, 32, 32], args)
    gs.train()
    x0 = torch.randn(1, 4, 32, 32, generator=g)
    c0 = torch.randn(1, 16, 32, 32, generator=g)
    gs(x0, c0, torch.zeros(1), False)
    randomize_(gs, g, 0.02)
    # recompute outputs with fp16-rounded weights so the fixture is self-consistent
    with torch.no_grad():
        for k, v in gs.state_dict().items():
            if v.dtype == torch.float32 and v.numel() > 4096:
                v.copy_(v.half().float())
    y, ld = gs(x0, c0, torch.zeros(1), False)
    sd = sd_clone(gs)
    save("glowstep_canonical_l0.pt", {"sd": {k: v.half() if v.dtype == torch.float32 and v.numel() > 4096 else v
                                             for k, v in sd.items()},
                                      "x": x0, "cond": c0, "y": y.detach(), "logdet": ld.detach()})


def gen_convlstm():
    import Utils.modules as um
    um.device = torch.device("cpu")
    from Utils import ConvLSTM
    g = torch.Generator().manual_seed(777)
    fx = {}
    # sibling_8x8: the shape class of the SRNN / VRNN baselines' ConvLSTMs (8x8 maps after three stride-2 convs, input =
    # cat(frame features, phi_z), SRNN/SRNN.py:161-171, VRNN/VRNN.py:169-173), channel counts scaled down for the fixture
    for name, (cin, hc, H, W, B, S) in {"small": (5, 4, 2, 2, 3, 1), "seq3_4x4": (6, 8, 4, 4, 2, 3),
                                        "sibling_8x8": (40, 12, 8, 8, 2, 2)}.items():
        torch.manual_seed(71)
        m = ConvLSTM(cin, hc, [3, 3], bias=True, peephole=True)
        x = torch.randn(B, S, cin, H, W, generator=g, requires_grad=True)
        h0 = torch.randn(B, hc, H, W, generator=g, requires_grad=True)
        c0 = torch.randn(B, hc, H, W, generator=g, requires_grad=True)
        out, ht, ct = m(x, h0, c0)
        wh = torch.randn(ht.shape, generator=g)
        wc = torch.randn(ct.shape, generator=g)
        ((ht * wh).sum() + (ct * wc).sum() + out.sum() * 0.1).backward()
        out_n, ht_n, ct_n = m(x.detach(), None, None)  # None state -> zeros
        fx[name] = {"cfg": [cin, hc, H, W, B, S], "sd": sd_clone(m), "x": x.detach(), "h0": h0.detach(), "c0": c0.detach(),
                    "out": out.detach(), "ht": ht.detach(), "ct": ct.detach(), "wh": wh, "wc": wc,
                    "grad_x": x.grad.clone(), "grad_h0": h0.grad.clone(), "grad_c0": c0.grad.clone(),
                    "grads": grads_of(m), "ht_none": ht_n.detach(), "ct_none": ct_n.detach()}
    save("convlstm.pt", fx)


def rfn_args(**kw):
    B = 2
    a = dict(batch_size=B, x_dim=[B, 1, 16, 16], condition_dim=[B, 1, 16, 16], h_dim=8, z_dim=4,
             structure_scaler=2, L=2, K=2, norm_type="none", norm_type_features="batchnorm", temperature=0.7,
             prior_structure=[12, 12], encoder_structure=[12, 12], free_bits=-1.0,
             skip_connection_flow="without_skip", downscaler_tanh=False, skip_connection_features=True,
             upscaler_tanh=False, a_dim=6, enable_smoothing=False, res_q=False, D=0, overshot_w=1.0,
             extractor_structure=[[4, "pool", 8], [8, "pool", 16]],
             upscaler_structure=[[16], ["upsample", 8, 8]],
             learn_prior=True, n_units_prior=16, make_conditional=True, base_norm="actnorm", non_lin_glow="relu",
             split2d_act="softplus", n_bits=8, LU_decomposed=True, n_units_affine=16, clamp_type="realnvp",
             flow_norm="actnorm", flow_batchnorm_momentum=0.0)
    a.update(kw)
    return Namespace(**a)


def gen_rfn():
    import Utils.modules as um
    import Flow.glow as fg
    um.device = torch.device("cpu")
    fg.device = torch.device("cpu")
    from RFN.RFN_new import RFN
    g = torch.Generator().manual_seed(2024)
    fx = {}
    cfgs = {
        "plain": dict(),
        "smooth_resq": dict(enable_smoothing=True, res_q=True),
        "overshoot_D2": dict(D=2, overshot_w=0.5, free_bits=0.02),
        "with_skip": dict(skip_connection_flow="with_skip"),
        "no_skipfeat": dict(skip_connection_features=False, upscaler_structure=[[16], ["upsample", 8]]),
    }
    T = 4
    for name, kw in cfgs.items():
        torch.manual_seed(81)
        args = rfn_args(**kw)
        m = RFN(args)
        m.train()
        x = (torch.rand(args.batch_size, T, 1, 16, 16, generator=g) * 255).floor() / 256 - 0.5
        sd_fresh = sd_clone(m)
        start_capture()
        out0 = m.loss(x, 0)  # first call: ActNorm init + BatchNorm running stats update
        d0 = stop_capture()
        randomize_(m, g, 0.02)
        sd = sd_clone(m)
        start_capture()
        kl_fb, kl, nll = m.loss(x, 0)
        d1 = stop_capture()
        (nll + 0.3 * kl_fb).backward()
        dims = x.shape[2:]
        import numpy as np
        bpd = float((kl + nll).item() / (np.log(2.) * float(torch.prod(torch.tensor(dims))) * (T - 1)))
        fx[name] = {"args": {k: v for k, v in vars(args).items()}, "T": T, "x": x, "sd_fresh": sd_fresh,
                    "out_first": [float(o) for o in out0], "draws_first": [(k, t) for k, t in d0],
                    "sd": sd, "sd_after": sd_clone(m), "out": [float(kl_fb), float(kl), float(nll)],
                    "draws": [(k, t) for k, t in d1], "bits_per_dim": bpd,
                    "grads": grads_of(m)}
    save("rfn_loss.pt", fx)


def gen_analysis():
    """evaluation-time methods of RFN_new (RFN/RFN_new.py:496-788) on two small configurations, model in eval mode
    after one training call (ActNorm init, BatchNorm running statistics) and a parameter perturbation."""
    import Utils.modules as um
    import Flow.glow as fg
    um.device = torch.device("cpu")
    fg.device = torch.device("cpu")
    from RFN.RFN_new import RFN
    g = torch.Generator().manual_seed(4242)
    fx = {}
    cfgs = {"plain": dict(), "smooth_resq_skip": dict(enable_smoothing=True, res_q=True, skip_connection_flow="with_skip"),
            "bair_like": dict(x_dim=[2, 3, 16, 16], condition_dim=[2, 3, 16, 16], D=2, skip_connection_flow="with_skip")}
    T = 5
    for name, kw in cfgs.items():
        torch.manual_seed(91)
        args = rfn_args(**kw)
        C = args.x_dim[1]
        m = RFN(args)
        m.train()
        x = (torch.rand(args.batch_size, T, C, 16, 16, generator=g) * 255).floor() / 256 - 0.5
        m.loss(x, 0)
        randomize_(m, g, 0.02)
        m.eval()
        sd = sd_clone(m)
        e = {"args": {k: v for k, v in vars(args).items()}, "T": T, "x": x, "sd": sd}
        with torch.no_grad():
            start_capture()
            _, _, kld, nlls = m.reconstruct_elbo_gap(x, sample=False)
            e["elbo_gap"] = {"draws": [(k, t) for k, t in stop_capture()], "kld": kld, "nlls": nlls}
            start_capture()
            pf = m.probability_future(x, 3)
            e["prob_future"] = {"draws": [(k, t) for k, t in stop_capture()], "n_conditions": 3, "out": pf}
            start_capture()
            pa = m.param_analysis(x, 2, 3)
            e["param_analysis"] = {"draws": [(k, t) for k, t in stop_capture()], "n_predictions": 2, "n_conditions": 3,
                                   "out": [t.detach() for t in pa]}
            start_capture()
            tx, pr = m.predict(x, 2, 3)
            e["predict"] = {"draws": [(k, t) for k, t in stop_capture()], "n_predictions": 2, "n_conditions": 3,
                            "true_x": tx.detach(), "predictions": pr.detach()}
            start_capture()
            rc, rcf = m.reconstruct(x)
            e["reconstruct"] = {"draws": [(k, t) for k, t in stop_capture()], "recons": rc.detach(), "recons_flow": rcf.detach()}
            start_capture()
            sm = m.sample(x, 3)
            e["sample"] = {"draws": [(k, t) for k, t in stop_capture()], "n_samples": 3, "samples": sm.detach()}
            # the training loss of the same (eval-mode) model: one more end-to-end point, C = 3 + overshooting included
            start_capture()
            kl_fb, kl, nll = m.loss(x, 0)
            e["loss_eval"] = {"draws": [(k, t) for k, t in stop_capture()], "out": [float(kl_fb), float(kl), float(nll)]}
        fx[name] = e
    save("rfn_analysis.pt", fx)


def gen_trainer():
    """Solver.preprocess / compute_loss (RFN/trainer.py:165-175,206-219).  `RFN.trainer` imports
    data_generators (needs torchvision) so a stub module is registered first."""
    import types
    stub = types.ModuleType("data_generators")
    for n in ("MovingMNIST", "PushDataset", "KTH"):
        setattr(stub, n, object)
    sys.modules["data_generators"] = stub
    import matplotlib
    matplotlib.use("Agg")
    from RFN.trainer import Solver
    g = torch.Generator().manual_seed(5)
    base = dict(n_bits=8, n_epochs=1, learning_rate=1e-4, verbose=False, path="/tmp/", batch_size=2,
                patience_lr=1, factor_lr=0.5, min_lr=1e-5, patience_es=1, beta_max=1.0, beta_min=1e-4,
                beta_steps=100, choose_data="mnist", n_frames=4, digit_size=28, step_length=4, num_digits=2,
                image_size=64, preprocess_range="0.5", preprocess_scale=255, num_workers=0, multigpu=False,
                n_predictions=2, n_conditions=2, scheduler_type="linear", use_validation_set=False)
    fx = {}
    x = torch.rand(2, 3, 1, 8, 8, generator=g)
    for nb in (5, 8):
        for rng in ("0.5", "1.0"):
            a = dict(base)
            a.update(n_bits=nb, preprocess_range=rng)
            s = Solver(Namespace(**a))
            y = s.preprocess(x)
            yb = s.preprocess(y, reverse=True)
            fx["preprocess_%d_%s" % (nb, rng)] = {"x": x, "y": y, "y_back": yb}
    s = Solver(Namespace(**base))
    s.beta = 0.25
    nll, klfb, kl = torch.tensor(1234.5), torch.tensor(3.25), torch.tensor(4.5)
    loss = s.compute_loss(nll, klfb, kl, torch.Size([1, 64, 64]), t=9)
    fx["compute_loss"] = {"nll": nll, "kl_fb": klfb, "kl": kl, "beta": 0.25, "dims": [1, 64, 64], "t": 9,
                          "loss": loss, "bits": s.bits[-1], "losses": s.losses[-1], "kl_loss": s.kl_loss[-1],
                          "recon_loss": s.recon_loss[-1]}
    save("trainer.pt", fx)


if __name__ == "__main__":
    which = sys.argv[1:] or ["modules", "glow", "convlstm", "rfn", "analysis", "trainer"]
    for w in which:
        {"modules": gen_modules, "glow": gen_glow, "convlstm": gen_convlstm, "rfn": gen_rfn, "analysis": gen_analysis,
         "trainer": gen_trainer}[w]()
