"""Multi-scale conditional Glow with the reference's surface (Flow/glow.py of the reference):
`ListGlow(x_size, condition_size, base_dist_size, args)`, `.log_prob(x, condition, base_condition, logdet)`,
`.sample(z, condition, base_condition, num_samples, temperature, eval_params)`, and the same `glow_frame` /
`prior` module layout (hence the same state_dict keys).

MI355X-first differences from the reference implementation (results are the same):
  * a GlowStep is ONE autograd node (rfn_hip.ops.GlowStepFn) launching hand-written gfx950 kernels;
  * `log_prob` accepts any number of frames N in dim 0 — the RFN driver time-batches all B·(T−1) frames into one call
    so the MFMA convolutions see GEMM-N = N·H·W even at the 2×2 / 4×4 levels;
  * no per-ActNorm `.item()` host syncs: initialisation state is mirrored on the host;
  * parameter-only log-det terms of all K steps of a level are summed once, not once per step.
"""
import os

import numpy as np
import torch
import torch.nn as nn

from Utils import split_feature, ActFun
from rfn_hip import ops as K
from rfn_hip import debug as D
from .glow_modules import (ActNorm, Conv2dZeros, Conv2dNorm, InvConv, AffineCoupling, Squeeze2d, Split2d,
                           BatchNormFlow, add_logdet)


class GlowStep(nn.Module):
    """Flow/glow.py:10-41 — ActNorm → InvConv(LU) → AffineCoupling (reverse: inverted order)."""

    def __init__(self, x_size, condition_size, args):
        super().__init__()
        b, c, h, w = x_size
        self.flow_norm = args.flow_norm
        if args.flow_norm == "batchnorm":
            self.norm = BatchNormFlow(x_size, momentum=args.flow_batchnorm_momentum)
        else:
            self.norm = ActNorm(c)
        self.invconv = InvConv(c, LU_decomposed=args.LU_decomposed)
        self.affine = AffineCoupling(x_size, condition_size, hidden_units=args.n_units_affine,
                                     non_lin=args.non_lin_glow, clamp_type=args.clamp_type)

    def _param_logdet(self, x):
        """(Σ actnorm.logs + Σ log_s)·H·W — the log-det terms that do not depend on the data."""
        hw = x.shape[2] * x.shape[3]
        if self.invconv.LU_decomposed:
            ld = torch.sum(self.invconv.log_s)
        else:
            ld = torch.slogdet(self.invconv.weight)[1]
        return (ld + torch.sum(self.norm.logs)) * hw

    def forward(self, x, condition, logdet, reverse, Wm=None, defer_logdet=None, packs=None):
        """`Wm` (optional): this step's C×C matrix when the caller built all matrices of a level in one batched
        computation; `defer_logdet` (optional list): receives the data dependent log-det [N] instead of adding it (the
        caller sums all steps at once and adds the parameter-only terms per level)."""
        if self.flow_norm == "batchnorm":  # non-default variant: unfused module chain
            if not reverse:
                x, logdet = self.norm(x, logdet, reverse=False)
                x, logdet = self.invconv(x, logdet, reverse=False)
                return self.affine(x, condition, logdet, reverse=False)
            x, logdet = self.affine(x, condition, logdet, reverse=True)
            x, logdet = self.invconv(x, logdet, reverse=True)
            return self.norm(x, logdet, reverse=True)
        aff, an = self.affine, self.norm
        act, clamp = K.ACT[aff.non_lin], K.CLAMP[aff.clamp_type]
        x = x if x.stride(-1) == 1 and x.stride(1) == x.shape[2] * x.shape[3] else x.contiguous()
        condition = condition.contiguous()
        if not reverse:
            if an.needs_init():  # data dependent init, first training call (glow_modules.py:22-36)
                an.initialize(x)
                an.mark_initialized()
            if Wm is None:
                Wm, _ = self.invconv.get_weight(x, reverse=False)
            if aff.net[0].norm_type.needs_init() or aff.net[2].norm_type.needs_init():
                with torch.no_grad():
                    z = K.actnorm_invconv_fwd(x.detach(), an.bias.detach().reshape(-1), an.logs.detach().reshape(-1),
                                              Wm.detach())
                    aff.maybe_init(z[:, : z.shape[1] // 2], condition)
            out, dl = K.GlowStepFn.apply(x, condition, Wm, an.bias, an.logs, *aff.nn_params(), act, clamp, packs)
            if defer_logdet is not None:
                defer_logdet.append(dl)
            elif logdet is not None:
                logdet = logdet + dl + self._param_logdet(x)
            return out, logdet
        if an.needs_init():
            an.mark_initialized()  # the reference flips the flag on any first call (glow_modules.py:34-36)
        for m in (aff.net[0].norm_type, aff.net[2].norm_type):
            if m.needs_init():
                m.mark_initialized()
        # `Wm` (optional, reverse): the inverse matrix from the caller's generation cache; `packs`: its packed-weight dict
        Winv = Wm if Wm is not None else self.invconv.get_weight(x, reverse=True)[0]
        out, dl = K.GlowStepRevFn.apply(x, condition, Winv, an.bias, an.logs, *aff.nn_params(), act, clamp, packs)
        if logdet is not None:
            logdet = logdet + dl - self._param_logdet(x)
        return out, logdet


class ListGlow(nn.Module):
    """Flow/glow.py:43-160."""

    def __init__(self, x_size, condition_size, base_dist_size, args):
        super().__init__()
        assert isinstance(condition_size, list), "condition_size is not a list, make sure it fits L"
        self.learn_prior = args.learn_prior
        self.n_units_prior = args.n_units_prior
        self.make_conditional = args.make_conditional
        self.base_norm = args.base_norm
        self.non_lin_glow = args.non_lin_glow
        self.conditional_clamp_function = args.split2d_act
        self.L, self.K, self.n_bits = args.L, args.K, args.n_bits
        Bx, Cx, Hx, Wx = x_size
        Bc, Cc, Hc, Wc = base_dist_size
        layers = []
        for l in range(self.L):
            layers.append(Squeeze2d())
            Cx, Hx, Wx = Cx * 4, Hx // 2, Wx // 2
            size = [Bx, Cx, Hx, Wx]
            for _ in range(self.K):
                layers.append(GlowStep(size, condition_size[l], args))
            if l < self.L - 1:
                layers.append(Split2d(size, condition_size[l], self.make_conditional, self.conditional_clamp_function))
                Cx = Cx // 2
        self.glow_frame = nn.ModuleList(layers)
        self.z_shape = (Cx, Hx, Wx)
        if self.learn_prior:
            self.prior = nn.Sequential(
                Conv2dNorm(Cc, self.n_units_prior, norm=self.base_norm),
                ActFun(self.non_lin_glow),
                Conv2dNorm(self.n_units_prior, self.n_units_prior // 2, norm=self.base_norm),
                ActFun(self.non_lin_glow),
                Conv2dZeros(in_channel=self.n_units_prior // 2, out_channel=2 * Cx),
            )

    # ---- x -> z ------------------------------------------------------------------------------------------
    def _level_steps(self):
        """glow_frame grouped per level: [(squeeze, [GlowStep]*K, split|None)] (built once)."""
        if getattr(self, "_levels", None) is None:
            levels, cur = [], None
            for m in self.glow_frame:
                if isinstance(m, Squeeze2d):
                    cur = [m, [], None]
                    levels.append(cur)
                elif isinstance(m, Split2d):
                    cur[2] = m
                else:
                    cur[1].append(m)
            self._levels = levels
        return self._levels

    @staticmethod
    def _batched_invconv(steps, hw):
        """All K matrices W = P·L·U of a level in ONE batched computation (glow_modules.py:188-205) plus the level's
        InvConv part of the parameter-only log-det, Σ_k Σ log_s·H·W: a handful of launches per level, not per step."""
        ics = [s.invconv for s in steps]
        if not all(ic.LU_decomposed for ic in ics) or any(s.flow_norm == "batchnorm" for s in steps):
            return None, None
        if K.invconv_weights_ok(ics):  # one launch each way instead of ~17 + ~20 (rfn_invconv_weights_*_f32)
            return K.InvConvWeightsFn.apply(int(hw), len(ics), *[ic.p for ic in ics], *[ic.sign_s for ic in ics],
                                            *[ic.lower for ic in ics], *[ic.upper for ic in ics],
                                            *[ic.log_s for ic in ics])
        lower = torch.stack([ic.lower for ic in ics])
        upper = torch.stack([ic.upper for ic in ics])
        log_s = torch.stack([ic.log_s for ic in ics])
        sign_s = torch.stack([ic.sign_s for ic in ics])
        p = torch.stack([ic.p for ic in ics])
        l_mask, eye = ics[0].l_mask, ics[0].eye
        L = lower * l_mask + eye
        U = upper * l_mask.t() + torch.diag_embed(sign_s * torch.exp(log_s))
        W = torch.matmul(p, torch.matmul(L, U))
        return W, log_s.sum() * hw

    @staticmethod
    def _level_node_ok(steps, z):
        """the one-node-per-level path: device tensors, every ActNorm of the level already initialised (the first,
        data-initialising call walks the steps one by one), one activation / clamp for the whole level"""
        if not z.is_cuda or os.environ.get("RFN_LEVEL_NODE") == "0":
            return False
        a0 = steps[0].affine
        for s_ in steps:
            a = s_.affine
            if s_.norm.needs_init() or a.net[0].norm_type.needs_init() or a.net[2].norm_type.needs_init():
                return False
            if a.non_lin != a0.non_lin or a.clamp_type != a0.clamp_type:
                return False
        return True

    def _packed_weights(self, x_shape, condition):
        """two launches re-pack every coupling-net weight of the flow: the split-precision (bf16x3) packs of the
        data-gradient convolutions -- and of the forward convolutions where bf16x3 is the forward arithmetic -- and the
        fragment streams of the fused forward kernel (shallow levels, 'mixed' arithmetic).
        Returns {GlowStep: (w1 f, w1 d, w2 f, w2 d, w3 f, w3 d, fused forward stream, fused backward stream)} (entries
        None where unused) or None."""
        if not K.bwd_b3():
            return None
        levels = self._level_steps()
        if not levels or not levels[0][1] or not levels[0][1][0].affine.net[0].conv.weight.is_cuda:
            return None
        N, C, H, W = (int(v) for v in x_shape)
        items, nets, slots = [], [], {}
        for l, (_, steps, split) in enumerate(levels):
            C, H, W = C * 4, H // 2, W // 2
            Cc = int(condition[l].shape[1])
            for s in steps:
                n0, n2, n4 = s.affine.net[0], s.affine.net[2], s.affine.net[4]
                w1, w2, w3 = n0.conv.weight, n2.conv.weight, n4.conv.weight
                fused = s.flow_norm != "batchnorm" and int(w2.shape[2]) == 1 and \
                    K.coupling_po_ok(N, C, Cc, int(w1.shape[0]), H, W, w1, w3)
                fp = K.fwd_prec(H, W)
                b3fwd = not fused and fp in ("bf16x3", "bf16x6")
                x6 = 4 if fp == "bf16x6" else 0   # forward packs in three planes where bf16x6 is the forward arithmetic
                slot = [None] * 7
                for j, (w, mode) in enumerate(((w1, 0), (w1, 1), (w2, 0), (w2, 1),
                                               (w3, 2 if K.zeros_conv_uses_taps(w3) else 0), (w3, 1))):
                    if j % 2 == 1 or b3fwd:
                        slot[j] = len(items)
                        items.append((w, mode + (x6 if j % 2 == 0 else 0)))
                if fused:
                    slot[6] = len(nets)
                    nets.append((w1, w2, w3))
                slots[s] = slot
            if split is not None:
                C = C // 2
        # the dense packs of the small-map levels (H*W <= 16: level 4 of the canonical flow), forward and -- when a
        # gradient will be asked for -- data-gradient orientation: queued here, all of them leave in ONE launch before the
        # first kernel that reads one (rfn_hip.ops.smallmap_pack).  Re-packed on every call like the plans below: a
        # captured training step must contain the launch (a version-keyed cache would be hit during capture and the
        # replays would run on stale packs).
        dense = {}
        Cd, Hd_, Wd_ = (int(v) for v in x_shape[1:])

        def dense_pack(s_, slot, w, transpose):
            return K.smallmap_pack(w, Hd_, Wd_, transpose)

        for l, (_, steps, split) in enumerate(levels):
            Cd, Hd_, Wd_ = Cd * 4, Hd_ // 2, Wd_ // 2
            Ccd = int(condition[l].shape[1])
            for s in steps:
                w1, w3 = s.affine.net[0].conv.weight, s.affine.net[4].conv.weight
                k33 = int(w1.shape[2]) == 3 and int(w3.shape[2]) == 3
                d8 = d9 = d10 = None
                if k33 and K.smallmap_conv_ok(Hd_, Wd_, Cd // 2, Ccd, int(w1.shape[0]), N):
                    d8 = dense_pack(s, 8, w1, False)
                if k33 and K.smallmap_conv_ok(Hd_, Wd_, int(w1.shape[0]), 0, Cd, N) and not K.zeros_conv_uses_taps(w3):
                    d9 = dense_pack(s, 9, w3, False)
                if (torch.is_grad_enabled() and k33
                        and K.smallmap_conv_ok(Hd_, Wd_, int(w1.shape[0]), 0, Cd // 2 + Ccd, N, bwd=True)):
                    d10 = dense_pack(s, 10, w1, True)
                dense[s] = (d8, d9, d10)
            if split is not None:
                Cd = Cd // 2
        plan = getattr(self, "_pack_plan", None)
        if plan is None or not plan.valid_for(items):
            plan = self._pack_plan = K.PackPlan(items)
        plan.run()
        po = None
        if nets:
            po = getattr(self, "_po_plan", None)
            if po is None or not po.valid_for(nets):
                po = self._po_plan = K.POPackPlan(nets)
            po.run(bwd=torch.is_grad_enabled())
        return {s: tuple((None if i is None else plan.bufs[i]) for i in sl[:6])
                + ((None if sl[6] is None else po.bufs[sl[6]]), (None if sl[6] is None else po.bwd_bufs[sl[6]]))
                + dense[s]
                for s, sl in slots.items()}

    def f(self, x, condition, logdet):
        """Flow/glow.py:105-117 (same order of operations; per-level batching of the tiny parameter algebra)."""
        z = x
        dls, const = [], 0
        packs = self._packed_weights(x.shape, condition)
        for l, (squeeze, steps, split) in enumerate(self._level_steps()):
            z = squeeze(z, undo_squeeze=False)
            W, c = self._batched_invconv(steps, z.shape[2] * z.shape[3])
            if W is not None and self._level_node_ok(steps, z):
                # the K steps of the level as ONE autograd node (rfn_hip.ops.GlowLevelFn)
                aff = steps[0].affine
                flat = []
                for s_ in steps:
                    flat += [s_.norm.bias, s_.norm.logs, *s_.affine.nn_params()]
                zc = z if z.stride(-1) == 1 and z.stride(1) == z.shape[2] * z.shape[3] else z.contiguous()
                z, dl = K.GlowLevelFn.apply(zc, condition[l].contiguous(), W, K.ACT[aff.non_lin],
                                            K.CLAMP[aff.clamp_type],
                                            None if packs is None else [packs[s_] for s_ in steps], *flat)
                dls.append(dl)
                D.check("f.l%d.z" % l, z); D.check("f.l%d.dl" % l, dl)
                steps_run = ()
            else:
                steps_run = steps
            Wk = W.unbind(0) if (W is not None and steps_run) else None  # one stack-backward instead of K select-backward + K adds
            for k, step in enumerate(steps_run):
                if W is not None:
                    z, _ = step(z, condition[l], logdet=logdet, reverse=False, Wm=Wk[k], defer_logdet=dls,
                                packs=None if packs is None else packs[step])
                else:
                    z, logdet = step(z, condition[l], logdet=logdet, reverse=False)
                D.check("f.l%d.k%d.z" % (l, k), z)
                if dls:
                    D.check("f.l%d.k%d.dl" % (l, k), dls[-1])
            if W is not None and not steps_run:
                const = const + c  # (the level node adds the ActNorm terms H*W * sum logs itself)
            elif W is not None:
                # ActNorm logs are read AFTER the steps ran: the first training call initialises them in place
                logs = torch.stack([s.norm.logs.reshape(-1) for s in steps])
                const = const + c + logs.sum() * (z.shape[2] * z.shape[3])
            if split is not None:
                z, logdet = split(z, condition[l], logdet=logdet, reverse=False)
                D.check("f.l%d.split.z" % l, z); D.check("f.l%d.split.logdet" % l, logdet)
        D.check("f.const", const if torch.is_tensor(const) else None)
        if logdet is not None and dls:
            logdet = logdet + torch.stack(dls).sum(0) + const
        return z, logdet

    # ---- z -> x ------------------------------------------------------------------------------------------
    def _reverse_cache(self):
        """{GlowStep: (U^-1 L^-1 P^T, packed-weight dict)} for generation, valid while no parameter changes (keyed by
        the parameters' version counters).  Autoregressive generation calls g() once per frame with unchanged weights:
        the reference rebuilds three matrix inverses per step and frame (glow_modules.py:198-203); here the inverses of
        a level are two batched triangular solves, done once, and the weight packs are kept."""
        key = tuple((p._version, p.data_ptr()) for p in self.parameters())
        cache = getattr(self, "_rev_cache", None)
        if cache is not None and cache[0] == key:
            return cache[1]
        table = {}
        with torch.no_grad():
            for _, steps, _ in self._level_steps():
                ics = [s.invconv for s in steps]
                if all(ic.LU_decomposed for ic in ics) and not any(s.flow_norm == "batchnorm" for s in steps):
                    l_mask, eye = ics[0].l_mask, ics[0].eye
                    Lm = torch.stack([ic.lower for ic in ics]) * l_mask + eye
                    U = torch.stack([ic.upper for ic in ics]) * l_mask.t() + torch.diag_embed(
                        torch.stack([ic.sign_s for ic in ics]) * torch.exp(torch.stack([ic.log_s for ic in ics])))
                    Pt = torch.stack([ic.p for ic in ics]).transpose(1, 2)
                    Winv = torch.linalg.solve_triangular(
                        U, torch.linalg.solve_triangular(Lm, Pt, upper=False, unitriangular=True), upper=True)
                    for i, s_ in enumerate(steps):
                        table[s_] = (Winv[i].contiguous(), {})
        self._rev_cache = (key, table)
        return table

    def g(self, z, condition, logdet, temperature, eps_list=None):
        """Flow/glow.py:90-102.  `eps_list` (optional) pins the N(0,1) draws of the Split2d layers, coarsest first."""
        x, l = z, len(condition) - 1
        eps_list = list(eps_list) if eps_list is not None else None
        cache = (self._reverse_cache()
                 if (x.is_cuda and not torch.is_grad_enabled() and os.environ.get("RFN_GEN_CACHE") != "0") else {})
        for step in reversed(self.glow_frame):
            if isinstance(step, Squeeze2d):
                x = step(x, undo_squeeze=True)
            elif isinstance(step, Split2d):
                l -= 1
                e = eps_list.pop(0) if eps_list else None
                x, logdet = step(x, condition[l], logdet=logdet, reverse=True, temperature=temperature, eps=e)
            else:
                Winv, pk = cache.get(step, (None, None))
                x, logdet = step(x, condition[l], logdet=logdet, reverse=True, Wm=Winv, packs=pk)
        return x, logdet

    def uniform_binning_correction(self, x, noise=None):
        """Flow/glow.py:119-126 — dequantisation: x + U(0, 1/2^n_bits), objective −ln(2^n_bits)·C·H·W."""
        b, c, h, w = x.size()
        n_bins = 2 ** self.n_bits
        if noise is None:
            noise = torch.rand(x.shape, device=x.device, dtype=x.dtype) * (1.0 / n_bins)
        objective = torch.full((b,), -np.log(n_bins) * (c * h * w), device=x.device, dtype=torch.float32)
        return x + noise, objective

    def _base_params(self, base_condition, n, device):
        if self.learn_prior:
            h = self.prior[0](base_condition.contiguous(), act=self.non_lin_glow)
            h = self.prior[2](h, act=self.non_lin_glow)
            return self.prior[4](h)  # [n, 2*Cz, h, w]: "split" halves = (mean, log_scale)
        return torch.zeros((n, 2 * self.z_shape[0]) + self.z_shape[1:], device=device)

    def log_prob(self, x, condition, base_condition, logdet=0, noise=None):
        """Flow/glow.py:128-141.  `noise` optionally pins the dequantisation draw (tests / parity runs)."""
        x, obj_unif = self.uniform_binning_correction(x, noise)
        assert isinstance(condition, list), "Condition is not a list, make sure it fits L"
        z, obj = self.f(x, condition, logdet)
        D.check("f.obj", obj)
        obj = obj + obj_unif
        params = self._base_params(base_condition, x.shape[0], x.device)
        D.check("base_params", params)
        obj = obj + K.GaussLogpFn.apply(z.contiguous(), params, 1, 1)
        return z, -obj

    def sample(self, z, condition, base_condition, num_samples=32, temperature=0.8, eval_params=False, eps_base=None,
               eps_list=None):
        """Flow/glow.py:143-160."""
        with torch.no_grad():
            mean = log_scale = None
            if z is None:
                n = base_condition.shape[0] if self.learn_prior else num_samples
                dev = base_condition.device if base_condition is not None else condition[0].device
                params = self._base_params(base_condition, n, dev)
                mean, log_scale = split_feature(params, "split")
                if eps_base is None:
                    eps_base = torch.randn(mean.shape, device=dev)
                z = K.gauss_sample(params, eps_base, 1, 1, temperature)
            x, _ = self.g(z, condition, logdet=None, temperature=temperature, eps_list=eps_list)
        if eval_params:
            return x, (mean, torch.exp(log_scale))
        return x
