"""RFN — recurrent flow network (SRNN latent dynamics + conditional Glow decoder) with the reference's surface:
`RFN(args)`, `.loss(x, logdet) -> (kl_free_bit, kl, nll)` (also exposed as `.forward`), `.predict`, `.reconstruct`,
`.sample`; same sub-module names, hence the same state_dict keys (RFN/RFN_new.py of the reference).

MI355X-first restructuring of `loss` (same mathematics, RFN/RFN_new.py:116-247):
  1. extractor over the T frames, ConvLSTM over t (HIP cell), optional backward smoothing LSTM;
  2. the latent recurrence (prior / encoder at the coarsest resolution) and the upscaler run per t — they are the only
     parts that are sequential through z_{t-1};
  3. the Glow decoder — ≈84 % of the reference's time — is evaluated ONCE on all B·(T−1) frames, t-major
     (`frame = (t-1)·B + b`), so every kernel launch carries the whole sequence batch.
Data dependent ActNorm initialisation uses the first B frames (t = 1) exactly like the reference's first call.
"""
import os

import torch
import torch.nn as nn

from Flow import ListGlow
from Flow.glow_modules import ActNorm
from rfn_hip import ops as K
from rfn_hip import debug as DBG
from Utils import VGG_upscaler, VGG_downscaler, SimpleParamNet, ConvLSTM, free_bits_kl, batch_reduce
from Utils.modules import recurrent_pair, recurrent_pair_split


def kl_normal(q_mean, q_std, p_mean, p_std):
    """KL(N(q) || N(p)) element-wise — the closed form torch.distributions.kl_divergence evaluates for two Normals
    (RFN_new.py:206-207,236), written out so that no distribution object validates its arguments with a host sync
    (which would also make the step impossible to capture into a hipGraph)."""
    var_ratio = (q_std / p_std) ** 2
    t1 = ((q_mean - p_mean) / p_std) ** 2
    return 0.5 * (var_ratio + t1 - 1 - var_ratio.log())


class RFN(nn.Module):
    def __init__(self, args):
        super().__init__()
        self.params = args
        batch_size = args.batch_size
        self.u_dim = args.x_dim
        self.x_dim = args.condition_dim
        self.h_dim, self.z_dim = args.h_dim, args.z_dim
        self.beta = 1
        self.L, self.K = args.L, args.K
        self.temperature = args.temperature
        self.free_bits = args.free_bits
        self.skip_connection_flow = args.skip_connection_flow
        self.skip_connection_features = args.skip_connection_features
        self.kl_temperature = 1
        self.a_dim = args.a_dim
        self.enable_smoothing = args.enable_smoothing
        self.res_q = args.res_q
        self.D = args.D + 1
        self.overshot_w = args.overshot_w
        down_structure, up_structure = args.extractor_structure, args.upscaler_structure
        nf = args.norm_type_features

        self.single_feature = self.skip_connection_flow == "without_skip" and not self.skip_connection_features
        self.extractor = VGG_downscaler(down_structure, L=self.L, in_channels=self.x_dim[1], norm_type=nf,
                                        non_lin="relu", scale=args.structure_scaler,
                                        skip_con=not self.single_feature, tanh=args.downscaler_tanh)
        channel_dims = [i[-1] for i in up_structure][::-1]
        dims_skip = self.extractor.get_layer_size(down_structure, self.x_dim)
        hu, wu = self.u_dim[2], self.u_dim[3]
        condition_size_list = []
        for i in range(self.L):
            hu, wu = hu // 2, hu // 2
            if self.skip_connection_flow == "with_skip":
                cc = channel_dims[i] + dims_skip[i][1]
            elif self.skip_connection_flow == "without_skip":
                cc = channel_dims[i]
            elif self.skip_connection_flow == "only_skip":
                cc = dims_skip[i][1]
            else:
                raise ValueError("choose skip setting")
            condition_size_list.append([batch_size, cc, hu, wu])
        c_features = dims_skip[-1][1]

        # learnable initial states: batch-shaped, exactly as in the reference (RFN_new.py:69-76)
        self.z_0 = nn.Parameter(torch.zeros(batch_size, self.z_dim, hu, wu))
        self.z_0x = nn.Parameter(torch.zeros(batch_size, self.z_dim, hu, wu))
        self.h_0 = nn.Parameter(torch.zeros(batch_size, self.h_dim, hu, wu))
        self.c_0 = nn.Parameter(torch.zeros(batch_size, self.h_dim, hu, wu))
        self.a_0 = nn.Parameter(torch.zeros(batch_size, self.a_dim, hu, wu))
        self.ca_0 = nn.Parameter(torch.zeros(batch_size, self.a_dim, hu, wu))

        self.upscaler = VGG_upscaler(up_structure, L=self.L, in_channels=self.h_dim + self.z_dim, norm_type=nf,
                                     non_lin="leakyrelu", scale=args.structure_scaler,
                                     skips=self.skip_connection_features, size_skips=dims_skip, tanh=args.upscaler_tanh)
        self.lstm = ConvLSTM(in_channels=c_features, hidden_channels=self.h_dim, kernel_size=[3, 3], bias=True,
                             peephole=True)
        if self.enable_smoothing:
            self.a_lstm = ConvLSTM(in_channels=c_features + self.h_dim, hidden_channels=self.a_dim,
                                   kernel_size=[3, 3], bias=True, peephole=True)
        self.prior = SimpleParamNet(args.prior_structure, in_channels=self.h_dim + self.z_dim,
                                    out_channels=self.z_dim, norm_type=args.norm_type, non_lin="leakyrelu")
        base_dim = (batch_size, self.h_dim + self.z_dim, hu, wu)
        self.flow = ListGlow(self.x_dim, condition_size_list, base_dim, args=self.params)
        enc_in = (self.a_dim + self.z_dim) if self.enable_smoothing else (c_features + self.h_dim + self.z_dim)
        self.encoder = SimpleParamNet(args.encoder_structure, in_channels=enc_in, out_channels=self.z_dim,
                                      norm_type=args.norm_type, non_lin="leakyrelu")

    # ------------------------------------------------------------------------------------------------ helpers
    def get_inits(self):
        return self.h_0, self.c_0, self.a_0, self.ca_0, self.z_0, self.z_0x, 0, 0, 0

    def _last(self, feats):
        return feats if self.single_feature else feats[-1]

    def combineconditions(self, flow_conditions, skip_conditions):
        return [torch.cat((a, b), dim=1) for a, b in zip(flow_conditions, skip_conditions)]

    def _flow_conditions(self, hz, feats_prev):
        if self.skip_connection_features:
            fc = self.upscaler(hz, skip_list=feats_prev)
        else:
            fc = self.upscaler(hz)
        if self.skip_connection_flow == "with_skip":
            fc = self.combineconditions(fc, feats_prev)
        elif self.skip_connection_flow == "only_skip":
            fc = feats_prev
        return fc

    def _deterministic_states(self, feats, n_steps, hprev, cprev, aprev, caprev, x_all=None):
        """h_t for t = 1..n_steps-1 (RFN_new.py:131-139) and, with smoothing, the backward a_t (:142-153).
        x_all (optional): the deepest features of frames 0..n_steps-2 as one step-major [n_steps-1, B, ...] tensor."""
        if n_steps < 2:
            return [], [], hprev, cprev
        if x_all is None:
            x_all = torch.stack([self._last(feats[i - 1]) for i in range(1, n_steps)], 0)  # frames 0..n-2, step-major
        store_ht, hprev, cprev = self.lstm.forward_steps(x_all, hprev, cprev)
        store_at = [None] * (n_steps - 1)
        if self.enable_smoothing:  # runs backward in time over (h_t, features of frame t+1)
            inp = torch.stack([torch.cat([store_ht[n_steps - i - 1], self._last(feats[n_steps - i])], 1)
                               for i in range(1, n_steps)], 0)
            a_rev, aprev, caprev = self.a_lstm.forward_steps(inp, aprev, caprev)
            for i in range(1, n_steps):
                store_at[n_steps - i - 1] = a_rev[i - 1]
        return store_ht, store_at, hprev, cprev

    def _flow_needs_init(self):
        return any(m.needs_init() for m in self.flow.modules() if isinstance(m, ActNorm))

    # ------------------------------------------------------------------------------------------------ training
    def loss(self, x, logdet=0, draws=None):
        """RFN/RFN_new.py:116-247.  `draws` (optional) = list of noise tensors in the reference's draw order
        (per t: prior ε, encoder ε, dequantisation U; then the overshooting prior ε's) for parity runs."""
        assert len(x.shape) == 5, "x must be [bs, t, c, h, w]"
        B, T = x.shape[0], x.shape[1]
        dev = x.device
        draws = list(draws) if draws is not None else None

        def eps_like(ref):
            return draws.pop(0).to(dev) if draws is not None else torch.randn(ref.shape, device=dev)

        hprev, cprev, aprev, caprev, zprev, zxprev, _, _, _ = self.get_inits()
        # extractor on all T frames at once (step-major), BatchNorm statistics per timestep as in the reference
        x_tm = x.transpose(0, 1).reshape(T * B, *x.shape[2:])
        feats_tb = self.extractor.forward_steps(x_tm, T)
        # the deepest features per step: a step-major view for the ConvLSTM and one unbind for the encoder inputs, so
        # that their gradients come back as one slice and one stack instead of T zero-fill + copy + add triples
        last_tb = feats_tb if self.single_feature else feats_tb[-1]
        last_steps = last_tb.view(T, B, *last_tb.shape[1:])
        last_t = last_steps.unbind(0)
        if self.single_feature:
            feats = list(last_t)
        else:
            feats = [[f[i * B:(i + 1) * B] for f in feats_tb[:-1]] + [last_t[i]] for i in range(T)]
        x_all = None if self.enable_smoothing else last_steps[:T - 1]
        store_ht, store_at, _, _ = self._deterministic_states(feats, T, hprev, cprev, aprev, caprev, x_all)
        for j, f in enumerate(feats_tb if not self.single_feature else [feats_tb]):
            DBG.check("feat%d" % j, f)
        DBG.check("ht_last", store_ht[T - 1] if len(store_ht) >= T else store_ht[-1])

        kl_loss = 0
        st_mean, st_std, st_zx = [], [], []
        base_t, noise_t = [], []
        # all 2(T-1) reparameterisation draws of the loop in one launch
        eps_all = None if draws is not None else torch.randn((T - 1, 2) + tuple(zprev.shape), device=dev)
        # The first conv of the encoder / prior sees cat(static, z) channels, and h_t, a_t and the frame features of every t
        # are known here: their share of that conv is one time-batched product per net, the per-step launches multiply
        # the z channels only (Utils.modules.recurrent_pair_split; None -> full inputs through recurrent_pair).
        h_all = torch.cat(store_ht[:T - 1], dim=0)
        Ch, Cz = int(h_all.shape[1]), int(zprev.shape[1])
        if self.enable_smoothing:
            s_enc = torch.cat(store_at[:T - 1], dim=0)
            rng_enc = (int(s_enc.shape[1]), int(s_enc.shape[1]) + Cz)
        else:
            s_enc = torch.cat((h_all, last_steps[1:T].reshape((T - 1) * B, *last_tb.shape[1:])), dim=1)
            rng_enc = (Ch, Ch + Cz)
        split_nets = recurrent_pair_split(self.encoder, s_enc, rng_enc, self.prior, h_all, (Ch, Ch + Cz), T - 1) \
            if x.is_cuda and torch.is_grad_enabled() else None
        both_nets = None if split_nets is not None else recurrent_pair(self.encoder, self.prior)
        for i in range(1, T):
            ht = store_ht[i - 1]
            if split_nets is not None:
                enc_raw, pri_raw = split_nets(i - 1, zxprev, zxprev if self.res_q else zprev)
            else:  # weight gradients time-batched, layer pairs co-launched
                if self.enable_smoothing:
                    enc_in = torch.cat((store_at[i - 1], zxprev), dim=1)
                else:
                    enc_in = torch.cat((ht, zxprev, self._last(feats[i])), dim=1)
                enc_raw, pri_raw = both_nets(enc_in, torch.cat((ht, zxprev if self.res_q else zprev), dim=1))
            # chunk + softplus, res_q shift, both reparameterised draws and the KL in one kernel (RNG order: prior first)
            eps_p = eps_like(zprev) if eps_all is None else eps_all[i - 1, 0]
            eps_q = eps_like(zprev) if eps_all is None else eps_all[i - 1, 1]
            zt, zxt, kl_t, enc_mean, enc_std = K.LatentStepFn.apply(enc_raw, pri_raw, eps_p, eps_q, self.res_q)
            DBG.check("enc_raw%d" % i, enc_raw); DBG.check("pri_raw%d" % i, pri_raw); DBG.check("zxt%d" % i, zxt)
            st_mean.append(enc_mean); st_std.append(enc_std); st_zx.append(zxprev)
            base_t.append(zxt)
            if draws is not None:
                noise_t.append(draws.pop(0).to(dev))
            if self.D == 1:
                kl_loss = kl_loss + kl_t
            zprev, zxprev = zt, zxt

        # ---- the decoder: all B*(T-1) frames in one call, t-major
        xs = x_tm[B:]
        # base condition cat(h_t, z^x_t) of every step: two time-batched stacks and one channel cat
        base = torch.cat((h_all, torch.cat(base_t, dim=0)), dim=1)
        # upscaler for all T-1 steps at once; its skip maps are the extractor features of frames 0..T-2
        n1 = (T - 1) * B
        skips = None if self.single_feature else [f[:n1] for f in feats_tb]
        if self.skip_connection_features:
            conds = self.upscaler.forward_steps(base, T - 1, skip_list=skips)
        else:
            conds = self.upscaler.forward_steps(base, T - 1)
        if self.skip_connection_flow == "with_skip":
            conds = self.combineconditions(conds, skips)
        elif self.skip_connection_flow == "only_skip":
            conds = skips
        noise = torch.cat(noise_t, dim=0) if noise_t else None
        if self.training and self._flow_needs_init():
            with torch.no_grad():  # data dependent init on the t = 1 batch, as the reference's first call does
                self.flow.log_prob(xs[:B], [c[:B] for c in conds], base[:B], 0,
                                   None if noise is None else noise[:B])
        for j, c in enumerate(conds):
            DBG.check("cond%d" % j, c)
        _, nll = self.flow.log_prob(xs, conds, base, logdet, noise)
        DBG.check("nll", nll)
        nll_loss = nll.view(T - 1, B).sum(0)

        if self.D > 1:  # overshooting (RFN_new.py:213-240)
            kl_loss = 0
            for i in range(1, T):
                overshot_loss, idt, zp = 0, i - 1, st_zx[i - 1]
                D = min(T - i, self.D)
                for d in range(D):
                    pm, ps = self.prior(torch.cat((store_ht[idt + d], zp), dim=1))
                    zp = pm + ps * eps_like(pm)
                    em, es = st_mean[idt + d], st_std[idt + d]
                    if d > 0:
                        em, es = em.detach().clone(), es.detach().clone()
                    overshot_loss = overshot_loss + self.overshot_w * kl_normal(em, es, pm, ps)
                kl_loss = kl_loss + 1 / D * overshot_loss

        kl_free_bit = free_bits_kl(kl_loss, free_bits=self.free_bits) if self.free_bits > 0 else kl_loss
        return batch_reduce(kl_free_bit).mean(), batch_reduce(kl_loss).mean(), nll_loss.mean()

    def forward(self, x, logdet=0):
        """Alias of `loss` so wrappers that hook `forward` (gradient all-reduce) see the training call."""
        return self.loss(x, logdet)

    # ------------------------------------------------------------------------------------------------ generation
    # `draws` (optional, tests): the noise in the reference's draw order, see each method.
    def _gen_step(self, prediction, hprev, cprev, zprev, eps, kl_temp):
        """one autoregressive generation step (RFN_new.py:331-356 / :480-491): frame t-1 -> frame t.
        eps: [prior eps, flow base eps, Split2d eps (coarsest first) ...].  Returns (frame, ht, ct, zt)."""
        eps = list(eps)
        take = lambda ref=None: eps.pop(0)
        condition_list = self.extractor(prediction)
        _, ht, ct = self.lstm(self._last(condition_list).unsqueeze(1), hprev, cprev)
        pm, ps = self.prior(torch.cat((ht, zprev), dim=1))
        zt = pm + ps * kl_temp * take(pm)
        hz = torch.cat((ht, zt), dim=1)
        fc = self._flow_conditions(hz, condition_list)
        frame = self._flow_sample(fc, hz, take, True)
        return frame, ht, ct, zt

    def _gen_eps_shapes(self, B):
        """shapes of the draws of one generation step: prior eps, base eps, Split2d eps list (coarsest first)"""
        hu, wu = self.z_0.shape[2], self.z_0.shape[3]
        shapes = [(B, self.z_dim, hu, wu)]
        c, h, w = self.x_dim[1], self.x_dim[2], self.x_dim[3]
        split = []
        for l in range(self.L):
            c, h, w = c * 4, h // 2, w // 2
            if l < self.L - 1:
                c = c // 2
                split.append((B, c, h, w))
        shapes.append((B, c, h, w))          # base distribution = what is left after the last level
        return shapes + split[::-1]

    def _gen_step_graphed(self, prediction, hprev, cprev, zprev, kl_temp):
        """_gen_step with fresh N(0,1) draws, replayed from a hipGraph: generation is one frame at a time, a few hundred
        launches of a few microseconds each, i.e. bound by the host's launch rate when launched eagerly (38 ms per frame at
        B = 32 against ~6 ms of GPU work).  The graph is rebuilt whenever a parameter changed (the inverse matrices and
        weight packs of ListGlow._reverse_cache are baked into it) or the shapes do; the draws are inputs of the graph."""
        import rfn_hip
        dev = prediction.device
        eps = [torch.randn(sh, device=dev) for sh in self._gen_eps_shapes(prediction.shape[0])]
        args = [prediction, hprev, cprev, zprev] + eps
        ok = (dev.type == "cuda" and rfn_hip.graph_capture_safe() and not self.training
              and os.environ.get("RFN_GEN_GRAPH", "1") != "0")
        if not ok:
            return self._gen_step(prediction, hprev, cprev, zprev, eps, kl_temp)
        key = (tuple((p._version, p.data_ptr()) for p in self.parameters()), tuple(tuple(a.shape) for a in args),
               float(kl_temp), float(self.temperature))
        g = getattr(self, "_gen_graph", None)
        if g is None or g[0] != key:
            static_in = [a.clone() for a in args]
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):  # warm-up on the capture path (lazy per-stream state, generation cache, MIOpen)
                    self._gen_step(*static_in[:4], static_in[4:], kl_temp)
            torch.cuda.current_stream().wait_stream(side)
            K.flush_packs()   # (nothing queued outside may be launched -- and replayed -- inside the capture)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self._gen_step(*static_in[:4], static_in[4:], kl_temp)
            g = self._gen_graph = (key, graph, static_in, static_out)
            self._gen_graph_builds = getattr(self, "_gen_graph_builds", 0) + 1
        _, graph, static_in, static_out = g
        for s_, a in zip(static_in, args):
            s_.copy_(a)
        graph.replay()
        return tuple(o.clone() for o in static_out)

    def _flow_sample(self, fc, hz, take, pinned, z=None):
        """flow.sample with pinned draws: base eps (only when z is None), then the Split2d eps list, coarsest first"""
        eb = take() if (pinned and z is None) else None
        el = [take() for _ in range(self.L - 1)] if pinned else None
        return self.flow.sample(z, fc, hz, temperature=self.temperature, eps_base=eb, eps_list=el)

    def predict(self, x, n_predictions, n_conditions, draws=None):
        """RFN/RFN_new.py:256-360 — condition on n_conditions frames, roll the prior forward n_predictions frames.
        draws: per warm-up step prior eps, encoder eps; per prediction prior eps, base eps, Split2d eps list."""
        assert len(x.shape) == 5, "x must be [bs, t, c, h, w]"
        with torch.no_grad():
            take, dr = self._taker(draws, x.device)
            hprev, cprev, aprev, caprev, zprev, zxprev, _, _, _ = self.get_inits()
            feats = [self.extractor(x[:, i]) for i in range(n_conditions)]
            store_ht, store_at, hprev, cprev = self._deterministic_states(feats, n_conditions, hprev, cprev, aprev, caprev)
            for i in range(1, n_conditions):
                ht = store_ht[i - 1]
                if self.enable_smoothing:
                    enc_mean, enc_std = self.encoder(torch.cat((store_at[i - 1], zxprev), dim=1))
                else:
                    enc_mean, enc_std = self.encoder(torch.cat((ht, zxprev, self._last(feats[i])), dim=1))
                if self.res_q:
                    prior_mean, prior_std = self.prior(torch.cat((ht, zxprev), dim=1))
                    enc_mean = prior_mean + enc_mean
                else:
                    prior_mean, prior_std = self.prior(torch.cat((ht, zprev), dim=1))
                zprev = prior_mean + prior_std * self.kl_temperature * take(prior_mean)
                zxprev = enc_mean + enc_std * take(enc_mean)
            true_x = x[:, :n_conditions].transpose(0, 1).detach().cpu().clone()
            frames = []  # kept on the device: ONE device-to-host copy at the end instead of a sync per frame
            prediction = x[:, n_conditions - 1]
            for i in range(n_predictions):
                if dr is None:
                    prediction, ht, ct, zt = self._gen_step_graphed(prediction, hprev, cprev, zprev, self.kl_temperature)
                else:
                    eps = [take() for _ in range(self.L + 1)]
                    prediction, ht, ct, zt = self._gen_step(prediction, hprev, cprev, zprev, eps, self.kl_temperature)
                frames.append(prediction.detach())
                hprev, cprev, zprev = ht, ct, zt
            predictions = (torch.stack(frames, 0).cpu() if frames else torch.zeros((0, *x[:, 0].shape)))
        return true_x, predictions

    def reconstruct(self, x, draws=None):
        """RFN/RFN_new.py:362-450 — posterior reconstructions and the flow bijection check g(f(x)).
        draws: per frame encoder eps, dequantisation noise, Split2d eps list of g(f(x)), base eps + Split2d eps list of
        the fresh sample."""
        assert len(x.shape) == 5, "x must be [bs, t, c, h, w]"
        with torch.no_grad():
            take, dr = self._taker(draws, x.device)
            T = x.shape[1]
            hprev, cprev, aprev, caprev, _, zxprev, _, _, _ = self.get_inits()
            # results stay on the device until the end: a pageable device-to-host copy per frame stalls the launch stream
            recons = torch.zeros((T, *x[:, 0].shape), device=x.device)
            recons_flow = torch.zeros((T, *x[:, 0].shape), device=x.device)
            feats = [self.extractor(x[:, i]) for i in range(T)]
            store_ht, store_at, _, _ = self._deterministic_states(feats, T, hprev, cprev, aprev, caprev)
            for i in range(1, T):
                ht = store_ht[i - 1]
                if self.enable_smoothing:
                    enc_mean, enc_std = self.encoder(torch.cat((store_at[i - 1], zxprev), dim=1))
                else:
                    enc_mean, enc_std = self.encoder(torch.cat((ht, zxprev, self._last(feats[i])), dim=1))
                if self.res_q:
                    prior_mean, _ = self.prior(torch.cat((ht, zxprev), dim=1))
                    enc_mean = prior_mean + enc_mean
                zxt = enc_mean + enc_std * take(enc_mean)
                hz = torch.cat((ht, zxt), dim=1)
                fc = self._flow_conditions(hz, feats[i - 1])
                z, _ = self.flow.log_prob(x[:, i], fc, hz, 0.0, take() if dr is not None else None)
                recons_flow[i] = self._flow_sample(fc, hz, take, dr is not None, z=z)
                recons[i] = self._flow_sample(fc, hz, take, dr is not None)
                zxprev = zxt
        return recons.cpu(), recons_flow.cpu()

    def sample(self, x, n_samples, draws=None):
        """RFN/RFN_new.py:453-494 — unconditional roll-out from the first frame.
        draws: per sample prior eps, base eps, Split2d eps list."""
        assert len(x.shape) == 5, "x must be [bs, t, c, h, w]"
        with torch.no_grad():
            take, dr = self._taker(draws, x.device)
            hprev, cprev, _, _, zprev, _, _, _, _ = self.get_inits()
            frames = []
            sample = x[:, 0]
            for i in range(n_samples):
                if dr is None:
                    sample, ht, ct, zt = self._gen_step_graphed(sample, hprev, cprev, zprev, 1.0)
                else:
                    eps = [take() for _ in range(self.L + 1)]
                    sample, ht, ct, zt = self._gen_step(sample, hprev, cprev, zprev, eps, 1.0)
                frames.append(sample)
                zprev, hprev, cprev = zt, ht, ct
            samples = torch.stack(frames, 0).cpu() if frames else torch.zeros((0, *x[:, 0].shape))
        return samples

    # ------------------------------------------------------------------------------------------------ analyses
    # The evaluation-time methods of the reference (RFN/RFN_new.py:496-788), built on the same kernels.  They run under
    # torch.no_grad() like the reference's evaluator does; `draws` (optional) pins the noise in the reference's draw
    # order so that parity tests can feed the GPU path, the oracle and the reference the same numbers.
    def _post_prior_step(self, i, store_ht, store_at, feats, zprev, zxprev, take):
        """one latent step (RFN_new.py:548-566 and its copies): (zt, zxt, prior_mean, prior_std, enc_mean, enc_std)"""
        ht = store_ht[i - 1]
        if self.enable_smoothing:
            enc_mean, enc_std = self.encoder(torch.cat((store_at[i - 1], zxprev), dim=1))
        else:
            enc_mean, enc_std = self.encoder(torch.cat((ht, zxprev, self._last(feats[i])), dim=1))
        if self.res_q:
            prior_mean, prior_std = self.prior(torch.cat((ht, zxprev), dim=1))
            enc_mean = prior_mean + enc_mean
        else:
            prior_mean, prior_std = self.prior(torch.cat((ht, zprev), dim=1))
        zt = prior_mean + prior_std * take(prior_mean)        # draw order: prior first, then encoder
        zxt = enc_mean + enc_std * take(enc_mean)
        return zt, zxt, prior_mean, prior_std, enc_mean, enc_std

    @staticmethod
    def _taker(draws, dev):
        draws = list(draws) if draws is not None else None

        def take(ref=None, uniform=False):
            if draws is not None:
                return draws.pop(0).to(dev)
            if ref is None:
                return None
            return torch.randn(ref.shape, device=dev)
        return take, draws

    def reconstruct_elbo_gap(self, x, sample=True, draws=None):
        """RFN/RFN_new.py:687-788 -- per-frame KL and the flow NLL under z ~ prior (index 0) and z ~ posterior (index 1),
        plus (sample=True) reconstructions.  Returns (recons, recons_flow, averageKLDseq [T,B], averageNLLseq [2,T,B]).
        draws: per t: prior eps, encoder eps, then per z in (prior, posterior): dequantisation noise [, with sample:
        the Split2d eps list of g(f(x)), the base eps and the Split2d eps list of the fresh sample]."""
        assert len(x.shape) == 5, "x must be [bs, t, c, h, w]"
        with torch.no_grad():
            B, T = x.shape[0], x.shape[1]
            take, dr = self._taker(draws, x.device)
            hprev, cprev, aprev, caprev, zprev, zxprev, _, _, _ = self.get_inits()
            feats = [self.extractor(x[:, i]) for i in range(T)]
            store_ht, store_at, _, _ = self._deterministic_states(feats, T, hprev, cprev, aprev, caprev)
            kld = torch.zeros((T, B), device=x.device)
            nlls = torch.zeros((2, T, B), device=x.device)
            recons = torch.zeros((2, T, *x[:, 0].shape), device=x.device) if sample else 0
            recons_flow = torch.zeros((2, T, *x[:, 0].shape), device=x.device) if sample else 0
            nsplit = self.L - 1
            for i in range(1, T):
                zt, zxt, pm, ps, em, es = self._post_prior_step(i, store_ht, store_at, feats, zprev, zxprev, take)
                ht = store_ht[i - 1]
                for count, zk in enumerate((zt, zxt)):
                    hz = torch.cat((ht, zk), dim=1)
                    fc = self._flow_conditions(hz, feats[i - 1])
                    b, nll = self.flow.log_prob(x[:, i], fc, hz, 0.0, take() if dr is not None else None)
                    nlls[count, i] = nll
                    if sample:
                        e1 = [take() for _ in range(nsplit)] if dr is not None else None
                        rf = self.flow.sample(b, fc, hz, temperature=self.temperature, eps_list=e1)
                        eb = take() if dr is not None else None
                        e2 = [take() for _ in range(nsplit)] if dr is not None else None
                        rs = self.flow.sample(None, fc, hz, temperature=self.temperature, eps_base=eb, eps_list=e2)
                        recons[count, i] = rs
                        recons_flow[count, i] = rf
                zprev, zxprev = zt, zxt
                kld[i] = kl_normal(em, es, pm, ps).sum([1, 2, 3])
        if sample:
            recons, recons_flow = recons.cpu(), recons_flow.cpu()
        return recons, recons_flow, kld.cpu(), nlls.cpu()

    def probability_future(self, x, n_conditions, draws=None):
        """RFN/RFN_new.py:590-685 -- NLL of the frames after n_conditions conditioning frames under the LAST conditioned
        state, with z ~ prior (index 0) and z ~ posterior (index 1).  The reference writes frame i's value to column
        i - n_conditions - 1 of a [B, 2, T - n_conditions - 1] tensor, i.e. frame n_conditions lands in the last column
        and is overwritten by the last frame; that indexing is kept.  draws: per warm-up step prior eps, encoder eps;
        then per frame and z the dequantisation noise."""
        assert len(x.shape) == 5, "x must be [bs, t, c, h, w]"
        with torch.no_grad():
            B, T = x.shape[0], x.shape[1]
            take, dr = self._taker(draws, x.device)
            hprev, cprev, aprev, caprev, zprev, zxprev, _, _, _ = self.get_inits()
            out = torch.zeros((B, 2, T - n_conditions - 1), device=x.device)
            feats = [self.extractor(x[:, i]) for i in range(n_conditions)]
            store_ht, store_at, _, _ = self._deterministic_states(feats, n_conditions, hprev, cprev, aprev, caprev)
            zt = zxt = None
            for i in range(1, n_conditions):
                zt, zxt, _, _, _, _ = self._post_prior_step(i, store_ht, store_at, feats, zprev, zxprev, take)
                zprev, zxprev = zt, zxt
            ht = store_ht[n_conditions - 2]
            for i in range(n_conditions, T):
                for count, zk in enumerate((zt, zxt)):
                    hz = torch.cat((ht, zk), dim=1)
                    fc = self._flow_conditions(hz, feats[n_conditions - 2])
                    _, nll = self.flow.log_prob(x[:, i], fc, hz, 0.0, take() if dr is not None else None)
                    out[:, count, i - n_conditions - 1] = nll
        return out.cpu()

    def param_analysis(self, x, n_predictions, n_conditions, draws=None):
        """RFN/RFN_new.py:496-588 -- prior / posterior / flow-base parameters along the sequence and one flow sample per
        step.  (The reference passes 1.0 in the num_samples slot of flow.sample, so the sampling temperature is the
        default 0.8; frame 0 of `predictions` stays zero.)  draws: per step prior eps, encoder eps, base eps, the
        Split2d eps list."""
        assert len(x.shape) == 5, "x must be [bs, t, c, h, w]"
        with torch.no_grad():
            B = x.shape[0]
            T = n_conditions + n_predictions
            take, dr = self._taker(draws, x.device)
            hprev, cprev, aprev, caprev, zprev, zxprev, _, _, _ = self.get_inits()
            feats = [self.extractor(x[:, i]) for i in range(T)]
            store_ht, store_at, _, _ = self._deterministic_states(feats, T, hprev, cprev, aprev, caprev)
            zs = tuple(zprev.shape[1:])
            dv = x.device
            mu_p, std_p = torch.zeros((T - 1, B) + zs, device=dv), torch.zeros((T - 1, B) + zs, device=dv)
            mu_q, std_q = torch.zeros((T - 1, B) + zs, device=dv), torch.zeros((T - 1, B) + zs, device=dv)
            mu_flow, std_flow = [], []
            predictions = torch.zeros((B, T, *x.shape[2:]), device=dv)
            nsplit = self.L - 1
            for i in range(1, T):
                zt, zxt, pm, ps, em, es = self._post_prior_step(i, store_ht, store_at, feats, zprev, zxprev, take)
                mu_p[i - 1], std_p[i - 1], mu_q[i - 1], std_q[i - 1] = pm, ps, em, es
                ht = store_ht[i - 1]
                fc = self._flow_conditions(torch.cat((ht, zxt), dim=1), feats[i - 1])
                base = torch.cat((ht, zt), dim=1)
                eb = take() if dr is not None else None
                el = [take() for _ in range(nsplit)] if dr is not None else None
                pred, params = self.flow.sample(None, fc, base, 1.0, eval_params=True, eps_base=eb, eps_list=el)
                mu_flow.append(params[0])
                std_flow.append(params[1])
                predictions[:, i] = pred
                zprev, zxprev = zt, zxt
        return (mu_p.cpu(), std_p.cpu(), mu_q.cpu(), std_q.cpu(), torch.stack(mu_flow).cpu(), torch.stack(std_flow).cpu(),
                predictions.cpu())
