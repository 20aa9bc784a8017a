"""The oracle (oracle/rfn_oracle.py) pinned against golden vectors produced by the reference itself
(tests/golden/make_golden.py).  CPU only.  Tolerances: bit-exact for indexing; 1e-6 abs for
elementwise maths; 1e-4 relative for log-det / nll / bits-per-dim (north_star)."""
import copy

import pytest
import torch

from oracle import rfn_oracle as O


def clone_sd(sd, grad=False):
    out = {}
    for k, v in sd.items():
        v = v.clone().float() if v.dtype == torch.float16 else v.clone()
        if grad and v.is_floating_point() and "running_" not in k:
            v.requires_grad_(True)
        out[k] = v
    return out


def close(a, b, rtol=1e-4, atol=1e-5):
    torch.testing.assert_close(a, b, rtol=rtol, atol=atol)


def test_squeeze_bit_exact(golden):
    f = golden("modules.pt")["squeeze"]
    assert torch.equal(O.squeeze2d(f["x"]), f["y"])
    assert torch.equal(O.squeeze2d(f["y"], undo=True), f["x_back"])
    assert torch.equal(f["x_back"], f["x"])


def test_split_feature_bit_exact(golden):
    f = golden("modules.pt")["split_feature"]
    a, b = O.split_feature(f["x"], "split")
    c, d = O.split_feature(f["x"], "cross")
    assert torch.equal(a, f["split0"]) and torch.equal(b, f["split1"])
    assert torch.equal(c, f["cross0"]) and torch.equal(d, f["cross1"])


def test_actnorm_init_and_reverse(golden):
    f = golden("modules.pt")["actnorm_init"]
    sd = {"bias": torch.zeros(1, 5, 1, 1), "logs": torch.zeros(1, 5, 1, 1), "initialized": torch.tensor(0, dtype=torch.uint8)}
    y, ld = O.actnorm(sd, "", f["x"], torch.zeros(3), False, True)
    close(sd["bias"], f["sd"]["bias"], 1e-5, 1e-6)
    close(sd["logs"], f["sd"]["logs"], 1e-5, 1e-6)
    assert int(sd["initialized"]) == 1
    close(y, f["y"], 1e-5, 1e-5)
    close(ld, f["logdet"], 1e-5, 1e-5)
    xb, ldb = O.actnorm(sd, "", y, torch.zeros(3), True, True)
    close(xb, f["x_back"], 1e-5, 1e-5)
    close(ldb, f["logdet_back"], 1e-5, 1e-5)
    # eval: no init, flag still set
    f = golden("modules.pt")["actnorm_eval_noinit"]
    sd = {"bias": torch.zeros(1, 5, 1, 1), "logs": torch.zeros(1, 5, 1, 1), "initialized": torch.tensor(0, dtype=torch.uint8)}
    y, _ = O.actnorm(sd, "", f["x"], None, False, False)
    assert torch.equal(y, f["y"]) and int(sd["initialized"]) == int(f["sd"]["initialized"]) == 1


@pytest.mark.parametrize("C", [4, 8])
def test_invconv_lu(golden, C):
    f = golden("modules.pt")["invconv_lu_%d" % C]
    sd = clone_sd(f["sd"], grad=True)
    w, ls = O.invconv_weight(sd, "", False)
    close(w, f["weight"], 1e-5, 1e-6)
    x = f["x"].clone().requires_grad_(True)
    z, ld = O.invconv(sd, "", x, torch.zeros(2), False)
    close(z, f["z"], 1e-5, 1e-5)
    close(ld, f["logdet"], 1e-5, 1e-5)
    (z.square().sum() + ld.sum()).backward()
    close(x.grad, f["grad_x"], 1e-4, 1e-5)
    for k, g in f["grads"].items():
        close(sd[k].grad, g, 1e-4, 1e-5)
    xb, ldb = O.invconv(clone_sd(f["sd"]), "", f["z"], torch.zeros(2), True)
    close(xb, f["x_back"], 1e-4, 1e-4)
    close(ldb, f["logdet_back"], 1e-5, 1e-5)


def test_invconv_plain(golden):
    f = golden("modules.pt")["invconv_plain_4"]
    z, ld = O.invconv(clone_sd(f["sd"]), "", f["x"], torch.zeros(2), False)
    close(z, f["z"], 1e-5, 1e-5)
    close(ld, f["logdet"], 1e-5, 1e-5)


def test_conv2dnorm_and_zeros(golden):
    m = golden("modules.pt")
    f = m["conv2dnorm"]
    sd = clone_sd(f["sd"])
    close(O.conv2dnorm(sd, "", f["x"], True), f["y"], 1e-5, 1e-5)
    # first (initialising) call: start from an uninitialised ActNorm
    sd0 = clone_sd(f["sd"])
    sd0["norm_type.bias"].zero_(); sd0["norm_type.logs"].zero_(); sd0["norm_type.initialized"].zero_()
    close(O.conv2dnorm(sd0, "", f["x"], True), f["y_first"], 1e-5, 1e-5)
    close(sd0["norm_type.logs"], f["sd"]["norm_type.logs"], 1e-5, 1e-6)
    f = m["conv2dnorm_1x1"]
    close(O.conv2dnorm(clone_sd(f["sd"]), "", f["x"], True), f["y"], 1e-5, 1e-5)
    f = m["conv2dzeros"]
    close(O.conv2dzeros(clone_sd(f["sd"]), "", f["x"]), f["y"], 1e-5, 1e-5)


@pytest.mark.parametrize("clamp", ["realnvp", "glow", "softclamp", "none"])
@pytest.mark.parametrize("non_lin", ["relu", "leakyrelu"])
def test_affine_coupling(golden, clamp, non_lin):
    f = golden("modules.pt")["affine_%s_%s" % (clamp, non_lin)]
    sd = clone_sd(f["sd"], grad=True)
    x = f["x"].clone().requires_grad_(True)
    c = f["cond"].clone().requires_grad_(True)
    y, ld = O.affine_coupling(sd, "", x, c, torch.zeros(2), False, True, non_lin, clamp)
    close(y, f["y"], 1e-5, 1e-5)
    close(ld, f["logdet"], 1e-4, 1e-5)
    ((y * f["wgt"]).sum() + (ld * f["gld"]).sum()).backward()
    close(x.grad, f["grad_x"], 1e-4, 1e-5)
    close(c.grad, f["grad_cond"], 1e-4, 1e-5)
    for k, g in f["grads"].items():
        close(sd[k].grad, g, 1e-3, 1e-4)
    xb, ldb = O.affine_coupling(clone_sd(f["sd"]), "", f["y"], f["cond"], torch.zeros(2), True, True, non_lin, clamp)
    close(xb, f["x_back"], 1e-4, 1e-5)
    close(ldb, f["logdet_back"], 1e-4, 1e-5)


@pytest.mark.parametrize("cond_on", [True, False])
@pytest.mark.parametrize("clampf", ["softplus", "exp"])
def test_split2d(golden, cond_on, clampf):
    f = golden("modules.pt")["split2d_%s_%s" % ("cond" if cond_on else "uncond", clampf)]
    sd = clone_sd(f["sd"], grad=True)
    x = f["x"].clone().requires_grad_(True)
    c = f["cond"].clone().requires_grad_(True)
    z1, ld = O.split2d(sd, "", x, c, torch.zeros(2), False, True, cond_on, clampf)
    assert torch.equal(z1, f["z1"])
    close(ld, f["logdet"], 1e-4, 1e-5)
    ((z1 * f["wgt"]).sum() + (ld * f["gld"]).sum()).backward()
    close(x.grad, f["grad_x"], 1e-4, 1e-5)
    if cond_on:
        close(c.grad, f["grad_cond"], 1e-4, 1e-5)
    for k, g in f["grads"].items():
        close(sd[k].grad, g, 1e-3, 1e-4)


def test_glowstep(golden):
    f = golden("glow.pt")["glowstep"]
    sd = clone_sd(f["sd"], grad=True)
    x = f["x"].clone().requires_grad_(True)
    c = f["cond"].clone().requires_grad_(True)
    y, ld = O.glowstep(sd, "", x, c, torch.zeros(2), False, True)
    close(y, f["y"], 1e-5, 1e-5)
    close(ld, f["logdet"], 1e-4, 1e-5)
    ((y * f["wgt"]).sum() + (ld * f["gld"]).sum()).backward()
    close(x.grad, f["grad_x"], 1e-4, 1e-5)
    close(c.grad, f["grad_cond"], 1e-4, 1e-5)
    for k, g in f["grads"].items():
        close(sd[k].grad, g, 1e-3, 1e-4)
    xb, ldb = O.glowstep(clone_sd(f["sd"]), "", f["y"], f["cond"], torch.zeros(2), True, True)
    close(xb, f["x_back"], 1e-4, 1e-5)
    close(ldb, f["logdet_back"], 1e-4, 1e-5)
    close(xb, f["x"], 1e-3, 1e-4)  # bijection


def test_glowstep_canonical_level0(golden):
    f = golden("glowstep_canonical_l0.pt")
    y, ld = O.glowstep(clone_sd(f["sd"]), "", f["x"], f["cond"], torch.zeros(1), False, True)
    close(y, f["y"], 1e-4, 1e-4)
    close(ld, f["logdet"], 1e-4, 1e-4)


@pytest.mark.parametrize("name", ["listglow_L2K2", "listglow_L3K2_rgb_leaky_glowclamp", "listglow_uncond"])
def test_listglow_log_prob(golden, name):
    f = golden("glow.pt")[name]
    cfg = f["args"]
    # (i) data dependent init on the first training call
    sd = clone_sd(f["sd_fresh"])
    z, nll = O.listglow_log_prob(sd, "", cfg, f["x"], f["conds"], f["base_cond"], 0, f["noise_init"], True)
    close(z, f["z_init"], 1e-4, 1e-4)
    close(nll, f["nll_init"], 1e-4, 1e-3)
    for k, v in f["sd_init"].items():
        if v.is_floating_point():
            close(sd[k], v, 1e-4, 1e-5)
        else:
            assert torch.equal(sd[k], v), k
    # (ii) steady state + gradients
    sd = clone_sd(f["sd"], grad=True)
    conds = [c.clone().requires_grad_(True) for c in f["conds2"]]
    bc = f["base_cond2"].clone().requires_grad_(True)
    z, nll = O.listglow_log_prob(sd, "", cfg, f["x2"], conds, bc, 0, f["noise2"], True)
    close(z, f["z2"], 1e-4, 1e-4)
    close(nll, f["nll2"], 1e-4, 1e-3)
    nll.mean().backward()
    for k, g in f["grads"].items():
        close(sd[k].grad, g, 2e-3, 2e-4)
    for c, g in zip(conds, f["grad_conds2"]):
        if c.numel():
            close(c.grad if c.grad is not None else torch.zeros_like(c), g, 1e-3, 1e-5)
    if cfg["learn_prior"]:
        close(bc.grad, f["grad_base_cond2"], 1e-3, 1e-5)
    # (iii) reverse path from a given z with the Split2d draws pinned
    xs = O.listglow_sample(clone_sd(f["sd"]), "", cfg, f["z2"], f["conds2"], f["base_cond2"],
                           temperature=0.8, eps_list=f["sample_draws"])
    close(xs, f["sample_from_z2"], 1e-3, 1e-4)


@pytest.mark.parametrize("name", ["small", "seq3_4x4", "sibling_8x8"])
def test_convlstm(golden, name):
    f = golden("convlstm.pt")[name]
    sd = clone_sd(f["sd"], grad=True)
    x = f["x"].clone().requires_grad_(True)
    h0 = f["h0"].clone().requires_grad_(True)
    c0 = f["c0"].clone().requires_grad_(True)
    out, ht, ct = O.convlstm(sd, "", x, h0, c0)
    close(out, f["out"], 1e-5, 1e-6)
    close(ht, f["ht"], 1e-5, 1e-6)
    close(ct, f["ct"], 1e-5, 1e-6)
    ((ht * f["wh"]).sum() + (ct * f["wc"]).sum() + out.sum() * 0.1).backward()
    close(x.grad, f["grad_x"], 1e-4, 1e-6)
    close(h0.grad, f["grad_h0"], 1e-4, 1e-6)
    close(c0.grad, f["grad_c0"], 1e-4, 1e-6)
    for k, g in f["grads"].items():
        close(sd[k].grad, g, 1e-4, 1e-5)
    _, htn, ctn = O.convlstm(clone_sd(f["sd"]), "", f["x"], None, None)
    close(htn, f["ht_none"], 1e-5, 1e-6)
    close(ctn, f["ct_none"], 1e-5, 1e-6)


@pytest.mark.parametrize("name", ["plain", "smooth_resq", "overshoot_D2", "with_skip", "no_skipfeat"])
def test_rfn_loss_end_to_end(golden, name):
    f = golden("rfn_loss.pt")[name]
    cfg = f["args"]
    # first call from fresh weights (ActNorm init + BatchNorm running stats)
    sd = clone_sd(f["sd_fresh"])
    out = O.rfn_loss(sd, cfg, f["x"], [t for _, t in f["draws_first"]], True)
    for a, b in zip(out, f["out_first"]):
        assert abs(float(a) - b) <= 1e-4 * abs(b) + 1e-4, (float(a), b)
    # steady state with gradients
    sd = clone_sd(f["sd"], grad=True)
    kl_fb, kl, nll = O.rfn_loss(sd, cfg, f["x"], [t for _, t in f["draws"]], True)
    for a, b in zip((kl_fb, kl, nll), f["out"]):
        assert abs(float(a.detach()) - b) <= 1e-4 * abs(b) + 1e-5, (float(a.detach()), b)
    bpd = O.bits_per_dim(kl.detach(), nll.detach(), f["x"].shape[2:], f["T"] - 1)
    assert abs(bpd - f["bits_per_dim"]) <= 1e-4 * abs(f["bits_per_dim"])
    (nll + 0.3 * kl_fb).backward()
    for k, g in f["grads"].items():
        got = sd[k].grad if sd[k].grad is not None else torch.zeros_like(g)
        close(got, g, 5e-3, 5e-4 * float(g.abs().max()) + 1e-6)
    # BatchNorm running statistics after the call
    for k, v in f["sd_after"].items():
        if k.startswith(("extractor.net.", "upscaler.net.")):
            continue  # aliases of the last block (Utils/modules.py:86-87,194-195): same tensors in the reference
        if "running_" in k or "num_batches" in k:
            close(sd[k].detach().to(v.dtype), v, 1e-5, 1e-6)


def test_trainer_arithmetic(golden):
    f = golden("trainer.pt")
    for nb in (5, 8):
        for rng in ("0.5", "1.0"):
            g = f["preprocess_%d_%s" % (nb, rng)]
            assert torch.equal(O.preprocess(g["x"], nb, rng), g["y"])
            assert torch.equal(O.preprocess(g["y"], nb, rng, reverse=True), g["y_back"])
    g = f["compute_loss"]
    bpd = O.bits_per_dim(g["kl"], g["nll"], g["dims"], g["t"])
    assert abs(bpd - g["bits"]) <= 1e-6 * abs(g["bits"])
    close(g["nll"] + g["beta"] * g["kl_fb"], g["loss"])


@pytest.mark.parametrize("name", ["plain", "smooth_resq_skip", "bair_like"])
def test_rfn_analysis_methods(golden, name):
    """RFN.reconstruct_elbo_gap / probability_future / param_analysis (RFN/RFN_new.py:496-788) and the eval-mode loss:
    the oracle's restatements against the reference's outputs, same weights and captured draws.  'bair_like' is the
    C = 3 configuration with overshooting (D = 2) and skip conditions (BASELINE config 5 in miniature)."""
    f = golden("rfn_analysis.pt")[name]
    cfg, x = f["args"], f["x"]
    e = f["elbo_gap"]
    kld, nlls = O.rfn_reconstruct_elbo_gap(clone_sd(f["sd"]), cfg, x, [t for _, t in e["draws"]], False)
    close(kld, e["kld"], 1e-4, 1e-5)
    close(nlls, e["nlls"], 1e-4, 1e-4)
    e = f["prob_future"]
    out = O.rfn_probability_future(clone_sd(f["sd"]), cfg, x, e["n_conditions"], [t for _, t in e["draws"]], False)
    close(out, e["out"], 1e-4, 1e-4)
    e = f["param_analysis"]
    out = O.rfn_param_analysis(clone_sd(f["sd"]), cfg, x, e["n_predictions"], e["n_conditions"], [t for _, t in e["draws"]], False)
    for a, b in zip(out, e["out"]):
        close(a, b, 1e-4, 1e-5)
    e = f["loss_eval"]
    out = O.rfn_loss(clone_sd(f["sd"]), cfg, x, [t for _, t in e["draws"]], False)
    for a, b in zip(out, e["out"]):
        assert abs(float(a) - b) <= 1e-4 * abs(b) + 1e-5, (float(a), b)


def test_evaluator_bits_per_dim(golden):
    """evaluation_metrics/error_metrics.py:358-368 is the trainer's bits/dim figure (RFN/trainer.py:214): the restatement
    against the reference's own number in the trainer fixture (the evaluator module itself needs lpips / skimage /
    tensorflow and cannot be imported in the build container)."""
    g = golden("trainer.pt")["compute_loss"]
    bpd, kl_t, nll_t = O.evaluator_compute_loss(g["nll"], g["kl"], g["dims"], g["t"])
    assert abs(bpd - g["bits"]) <= 1e-6 * abs(g["bits"])
    assert abs(kl_t - g["kl_loss"]) <= 1e-6 * abs(g["kl_loss"]) and abs(nll_t - g["recon_loss"]) <= 1e-6 * abs(g["recon_loss"])


@pytest.mark.parametrize("name", ["plain", "smooth_resq_skip", "bair_like"])
def test_rfn_generation_methods(golden, name):
    """RFN.predict / reconstruct / sample (RFN/RFN_new.py:256-494) restated in the oracle against the reference's
    outputs with the captured draws (eval mode)."""
    f = golden("rfn_analysis.pt")[name]
    cfg, x = f["args"], f["x"]
    e = f["predict"]
    tx, pr = O.rfn_predict(clone_sd(f["sd"]), cfg, x, e["n_predictions"], e["n_conditions"], [t for _, t in e["draws"]])
    assert torch.equal(tx, e["true_x"])
    close(pr, e["predictions"], 1e-4, 1e-5)
    e = f["reconstruct"]
    rc, rcf = O.rfn_reconstruct(clone_sd(f["sd"]), cfg, x, [t for _, t in e["draws"]])
    close(rc, e["recons"], 1e-4, 1e-5)
    close(rcf, e["recons_flow"], 1e-4, 1e-5)
    e = f["sample"]
    sm = O.rfn_sample(clone_sd(f["sd"]), cfg, x, e["n_samples"], [t for _, t in e["draws"]])
    close(sm, e["samples"], 1e-4, 1e-5)
