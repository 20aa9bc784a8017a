"""Host-side modules with the reference's class names and state_dict keys (Utils/modules.py of the reference).

* ConvLSTM / ConvLSTMLayer — ON the hot path: the cell runs on the gfx950 kernels (MFMA conv over cat(x,h) without
  materialising the cat + fused gate update), see rfn_hip.ops.ConvLSTMCellFn.
* VGG_downscaler / VGG_upscaler / SimpleParamNet / NormLayer / ActFun — callers of the hot path:
  ordinary conv/BN stacks kept as PyTorch-ROCm modules, same constructor arguments and parameter names.
"""
import os
from types import SimpleNamespace

import torch
import torch.nn as nn
import torch.nn.functional as F

from rfn_hip import ops as K


class ActFun(nn.Module):
    """Utils/modules.py:8-19."""

    def __init__(self, non_lin, in_place=False):
        super().__init__()
        if non_lin == "relu":
            self.net = nn.ReLU(inplace=in_place)
        elif non_lin == "leakyrelu":
            self.net = nn.LeakyReLU(negative_slope=0.20, inplace=in_place)
        else:
            assert False, "Please specify a activation type from the set {relu,leakyrelu}"
        self.non_lin = non_lin

    def forward(self, x):
        return self.net(x)


class NoNorm(nn.Module):
    def forward(self, x):
        return x


class NormLayer(nn.Module):
    """Utils/modules.py:28-41."""

    def __init__(self, in_channels, norm_type):
        super().__init__()
        if norm_type == "batchnorm":
            self.norm = nn.BatchNorm2d(in_channels)
        elif norm_type == "instancenorm":
            self.norm = nn.InstanceNorm2d(in_channels)
        elif norm_type == "none":
            self.norm = NoNorm()
        else:
            assert False, "Please specify a norm type from the set {batchnorm, instancenorm, none}"

    def forward(self, x):
        return self.norm(x)


def per_step_batchnorm(bn, x, steps):
    """`bn` (nn.BatchNorm2d, training mode) applied to a step-major time-batched tensor [steps*B, C, H, W] with the
    statistics of EACH step's B samples — what `steps` separate calls bn(x_t) compute (the reference runs its extractor
    and upscaler once per timestep, RFN_new.py:126-128,191-194), including the running-statistics EMA applied in step
    order.  One set of reductions / elementwise launches instead of `steps` of them."""
    if not (bn.training or not bn.track_running_stats):
        return bn(x)
    SB, C, H, W = x.shape
    B = SB // steps
    # steps become channel groups: [S*B,C,H,W] -> [B, S*C, H, W]; one fused batch-norm launch then normalises every
    # (step, channel) pair over its own (B,H,W) samples.  Temporary running buffers with momentum 1 receive the batch
    # mean / unbiased variance of each pair for the EMA below.
    xt = x.view(steps, B, C, H, W).transpose(0, 1).reshape(B, steps * C, H, W)
    track = bn.track_running_stats and bn.training
    tmp_m = torch.zeros(steps * C, device=x.device, dtype=x.dtype) if track else None
    tmp_v = torch.ones(steps * C, device=x.device, dtype=x.dtype) if track else None
    w = bn.weight.repeat(steps) if bn.affine else None
    b = bn.bias.repeat(steps) if bn.affine else None
    yt = torch.nn.functional.batch_norm(xt, tmp_m, tmp_v, w, b, True, 1.0, bn.eps)
    y = yt.view(B, steps, C, H, W).transpose(0, 1).reshape(SB, C, H, W)
    if track:
        with torch.no_grad():
            m = bn.momentum if bn.momentum is not None else 0.1
            # r <- (1-m) r + m s_t for t = 0..steps-1  ==  (1-m)^S r + Σ_t m (1-m)^(S-1-t) s_t
            coef = m * (1.0 - m) ** torch.arange(steps - 1, -1, -1, device=x.device, dtype=x.dtype)
            decay = (1.0 - m) ** steps
            bn.running_mean.mul_(decay).add_((coef.view(steps, 1) * tmp_m.view(steps, C)).sum(0))
            bn.running_var.mul_(decay).add_((coef.view(steps, 1) * tmp_v.view(steps, C)).sum(0))
            bn.num_batches_tracked += steps
    return y


def _act_code(m):
    """(code, slope) of an activation module the fused per-step BatchNorm kernels implement, else None"""
    if isinstance(m, ActFun):
        m = m.net
    if isinstance(m, nn.LeakyReLU):
        return 2, float(m.negative_slope)
    if isinstance(m, nn.ReLU):
        return 1, 0.0
    if isinstance(m, nn.Tanh):
        return 3, 0.0
    return None


_EMA_COEF = {}


def _ema_coef(steps, m, n, device):
    """m (1-m)^(S-1-t) for t = 0..S-1 (and the same times n/(n-1)), cached per (S, momentum, n, device)"""
    key = (steps, float(m), int(n), str(device))
    if key not in _EMA_COEF:
        c = m * (1.0 - m) ** torch.arange(steps - 1, -1, -1, dtype=torch.float64)
        _EMA_COEF[key] = (c.to(device=device, dtype=torch.float32),
                          (c * (n / max(n - 1, 1))).to(device=device, dtype=torch.float32))
    return _EMA_COEF[key]


def per_step_batchnorm_act(bn, x, steps, act_code, slope):
    """per_step_batchnorm fused with the following activation on the HIP kernels (rfn_stepbn_*): no permute copies, one
    statistics pass and one normalise+activate pass forward, one reduction and one apply pass backward."""
    running = None
    if bn.track_running_stats:
        n = (x.shape[0] // steps) * x.shape[2] * x.shape[3]
        m = bn.momentum if bn.momentum is not None else 0.1
        # r <- (1-m) r + m s_t for t = 0..steps-1  ==  (1-m)^S r + Σ_t m (1-m)^(S-1-t) s_t, applied by the kernel
        # (biased batch variance -> unbiased: the factor is in coef_u)
        coef, coef_u = _ema_coef(steps, m, n, x.device)
        running = (bn.running_mean, bn.running_var, coef, coef_u, (1.0 - m) ** steps, bn.num_batches_tracked)
    y, _, _ = K.StepBatchNormActFn.apply(x, bn.weight if bn.affine else None, bn.bias if bn.affine else None, steps,
                                         float(bn.eps), act_code, slope, running)
    return y


def queue_conv_packs(seqs, x):
    """{conv module: (forward pack, data-gradient pack or None)} for every convolution of the nn.Sequentials `seqs` that
    runs on the package's kernels (forward arithmetic; data gradient when one will be asked for).  The packs are QUEUED
    (rfn_hip.ops.pack_weight): asked for up front, all of a network's leave in one launch before its first convolution."""
    packs = {}
    if x.is_cuda and x.dtype == torch.float32 and K.CONV_PRECISION != "f32" and K.bwd_b3():
        for seq in seqs:
            for m in seq:
                if _own_conv(m, x) and not K.fewcin_ok(x, None, m.weight, 0):
                    packs[m] = (K.pack_weight(m.weight, prec="bf16x6"),
                                K.pack_weight(m.weight, flip=True) if torch.is_grad_enabled() else None)
    return packs


def run_time_batched(seq, x, steps, packs=None):
    """run an nn.Sequential on a step-major time-batched tensor, BatchNorm statistics per step.
    packs: queue_conv_packs of (at least) this stack, when the caller has asked for a whole network's at once."""
    mods = list(seq)
    if packs is None:
        packs = queue_conv_packs([seq], x)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, NormLayer) and isinstance(m.norm, nn.BatchNorm2d):
            bn = m.norm
            fused = x.is_cuda and bn.training and x.dtype == torch.float32 and os.environ.get("RFN_STEPBN") != "0"
            if fused:
                act = _act_code(mods[i + 1]) if i + 1 < len(mods) else None
                x = per_step_batchnorm_act(bn, x, steps, act[0] if act else 0, act[1] if act else 0.0)
                i += 2 if act else 1
                continue
            x = per_step_batchnorm(bn, x, steps)
        elif _own_conv(m, x):
            # The 3x3 convolutions of the extractor / upscaler run on the package's own kernels, not on MIOpen: on
            # few-pixel problems (the 8x8 .. 2x2 blocks, or a small batch at any size) MIOpen picks split-K kernels whose
            # float atomics make the features differ in the last bits from call to call (measured 2.5e-6), and the flow
            # amplifies that.  Here split-K slices are added in a fixed order and the arithmetic is fp32-grade where the
            # mode has one: the whole forward pass is bit-reproducible
            # (tests/test_hip_modules.py::test_forward_pass_is_bit_reproducible).  RFN_VGG_CONV=miopen: the old route.
            # (these layers were fp32 on MIOpen: fp32-grade here too, also in the all-bf16x3 test arithmetic)
            x = K.conv_ep(x.contiguous(), None, m.weight, None, None, 0, 0,
                          prec="f32" if K.CONV_PRECISION == "f32" else "bf16x6", packs=packs.get(m))
        else:
            x = m(x)
        i += 1
    return x


def _own_conv(m, x):
    return (isinstance(m, nn.Conv2d) and x.is_cuda and x.dtype == torch.float32
            and tuple(m.kernel_size) == (3, 3) and tuple(m.stride) == (1, 1) and tuple(m.padding) == (1, 1)
            and tuple(m.dilation) == (1, 1) and m.groups == 1 and m.bias is None
            and os.environ.get("RFN_VGG_CONV", "own") != "miopen")


class Squeeze2dDecoder(nn.Module):
    """Utils/modules.py:122-138 — space-to-depth inside the extractor/upscaler (same index map as Flow.Squeeze2d)."""

    def __init__(self, undo_squeeze=False):
        super().__init__()
        self.undo_squeeze = undo_squeeze

    def forward(self, x):
        return K.Squeeze2dFn.apply(x.contiguous(), self.undo_squeeze)


class _NearestUp2xFn(torch.autograd.Function):
    """nn.Upsample(scale_factor=2, mode='nearest') with a pooling backward: the gradient of a 2x nearest up-sampling is
    the sum over each 2x2 block = 4 * avg_pool2d — torch's generic upsample_nearest2d_backward kernel took 0.5 ms per
    call on the [608, 16..128, H, W] condition maps."""

    @staticmethod
    def forward(ctx, x):
        return torch.nn.functional.interpolate(x, scale_factor=2, mode="nearest")

    @staticmethod
    def backward(ctx, g):
        return torch.nn.functional.avg_pool2d(g, 2) * 4.0


class NearestUp2x(nn.Upsample):
    """drop-in for nn.Upsample(scale_factor=2, mode='nearest') (same class hierarchy, no parameters)."""

    def __init__(self):
        super().__init__(scale_factor=2, mode="nearest")

    def forward(self, x):
        return _NearestUp2xFn.apply(x)


class tanh0_5(nn.Module):
    def forward(self, x):
        return 0.5 * torch.tanh(x)


def _conv_block(cin, cout, norm_type, act, stride=1):
    return [nn.Conv2d(cin, cout, kernel_size=3, stride=stride, padding=1, bias=False),
            NormLayer(cout, norm_type=norm_type), act]


class VGG_downscaler(nn.Module):
    """Utils/modules.py:43-120 — L blocks; entries: int = conv3x3 to that width, 'pool', 'conv' (stride 2, x scale),
    'squeeze'.  The last layer of the last block uses Tanh; `self.net` aliases the last block as in the reference."""

    def __init__(self, structures, L, in_channels, norm_type="batchnorm", non_lin="relu", scale=2, skip_con=False,
                 tanh=False):
        super().__init__()
        assert len(structures) == L, "Please specify number of blocks = L"
        self.l_nets = nn.ModuleList([])
        self.L, self.skip_con, self.scale = L, skip_con, scale
        for l, structure in enumerate(structures):
            layers = []
            for count, item in enumerate(structure, 1):
                last = count == len(structure)
                if l == L - 1 and last:
                    act = nn.Tanh()
                elif last and tanh:
                    act = tanh0_5()
                else:
                    act = ActFun(non_lin, in_place=True)
                if item == "pool":
                    layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
                elif item == "conv":
                    cc = int(in_channels * scale)
                    layers += _conv_block(in_channels, cc, norm_type, act, stride=2)
                    in_channels = cc
                elif item == "squeeze":
                    cc = in_channels * 4
                    layers += [Squeeze2dDecoder(undo_squeeze=False), NormLayer(cc, norm_type=norm_type), act]
                    in_channels = cc
                else:
                    layers += _conv_block(in_channels, item, norm_type, act)
                    in_channels = item
            self.net = nn.Sequential(*layers)
            self.l_nets.append(self.net)

    def get_layer_size(self, structures, x_size):
        bs, c, hx, wx = x_size
        dims = []
        for structure in structures:
            for item in structure:
                if item in ("pool", "conv", "squeeze"):
                    hx, wx = hx // 2, wx // 2
                    c = c if item == "pool" else (int(c * self.scale) if item == "conv" else c * 4)
                else:
                    c = item
            dims.append([bs, c, hx, wx])
        return dims

    def forward(self, x, block_size=None):
        outputs = []
        for i in range(self.L):
            x = self.l_nets[i](x)
            if self.skip_con:
                outputs.append(x)
            else:
                outputs = x
        return outputs

    def forward_steps(self, x, steps):
        """all `steps` per-timestep calls of forward() at once on a step-major [steps*B, C, H, W] tensor."""
        outputs = []
        packs = queue_conv_packs(self.l_nets, x)
        for i in range(self.L):
            x = run_time_batched(self.l_nets[i], x, steps, packs)
            if self.skip_con:
                outputs.append(x)
            else:
                outputs = x
        return outputs


class VGG_upscaler(nn.Module):
    """Utils/modules.py:147-214 — L blocks; block l>0 begins with an up-sampling stage kept in `upscales_nets[l-1]`;
    with `skips` the matching extractor map is concatenated before the block's first conv."""

    def __init__(self, structures, L, in_channels, norm_type="batchnorm", non_lin="relu", scale=2, skips=False,
                 size_skips=None, tanh=False):
        super().__init__()
        assert len(structures) == L, "Please specify number of blocks = L"
        self.l_nets = nn.ModuleList([])
        self.upscales_nets = nn.ModuleList([])
        self.L, self.skips = L, skips
        size_skips.reverse()  # the reference mutates the caller's list too (Utils/modules.py:155)
        for l, structure in enumerate(structures):
            layers, layer_up = [], None
            for count, item in enumerate(structure, 1):
                last = count == len(structure)
                act = tanh0_5() if (last and tanh) else ActFun(non_lin, in_place=True)
                first_conv = (count == 1 and l == 0) or (count == 2 and l != 0)
                skip_channels = size_skips[l][1] if (skips and first_conv) else 0
                if item == "upsample":
                    layer_up = [NearestUp2x()]
                elif item == "deconv":
                    dc = in_channels // scale
                    layer_up = [nn.ConvTranspose2d(in_channels, dc, kernel_size=4, stride=2, padding=1, bias=False),
                                NormLayer(dc, norm_type=norm_type), act]
                    in_channels = dc
                elif item == "squeeze":
                    dc = in_channels // 4
                    layer_up = [Squeeze2dDecoder(undo_squeeze=True), NormLayer(dc, norm_type=norm_type), act]
                    in_channels = dc
                else:
                    layers += [nn.Conv2d(in_channels + skip_channels, item, kernel_size=3, stride=1, padding=1,
                                         bias=False), NormLayer(item, norm_type=norm_type), act]
                    in_channels = item
            if l > 0:
                self.upscales_nets.append(nn.Sequential(*layer_up))
            self.net = nn.Sequential(*layers)
            self.l_nets.append(self.net)

    def forward(self, x, skip_list=None):
        outputs = []
        rev = list(reversed(skip_list)) if self.skips else None
        for i in range(self.L):
            if i > 0:
                x = self.upscales_nets[i - 1](x)
            if self.skips:
                x = self.l_nets[i](torch.cat((x, rev[i]), dim=1))
            else:
                x = self.l_nets[i](x)
            outputs.append(x)
        outputs.reverse()
        return outputs

    def forward_steps(self, x, steps, skip_list=None):
        """all `steps` per-timestep calls of forward() at once (step-major time-batched x and skip maps)."""
        outputs = []
        rev = list(reversed(skip_list)) if self.skips else None
        packs = queue_conv_packs(list(self.upscales_nets) + list(self.l_nets), x)
        for i in range(self.L):
            if i > 0:
                x = run_time_batched(self.upscales_nets[i - 1], x, steps, packs)
            if self.skips:
                x = torch.cat((x, rev[i]), dim=1)
            x = run_time_batched(self.l_nets[i], x, steps, packs)
            outputs.append(x)
        outputs.reverse()
        return outputs


class _StepStash:
    """inputs / pre-activation gradients of one conv layer collected over the timesteps of a recurrence"""

    def __init__(self):
        self.xs, self.gs = [], []


class _WeightPort(torch.autograd.Function):
    """Entry of a conv layer's (weight, bias) into a recurrence that applies the layer once per timestep.  The per-step
    functions (_StepConvAct) return no weight gradient; autograd runs this node's backward after all of them, and it
    computes the weight and bias gradients of ALL steps as one convolution-backward over the time-batched stash — one
    launch sequence and one accumulation into .grad instead of one per timestep."""

    @staticmethod
    def forward(ctx, w, b, stash):
        ctx.set_materialize_grads(False)
        ctx.stash, ctx.has_bias = stash, b is not None
        ctx.save_for_backward(w)
        return w.view_as(w), (b.view_as(b) if b is not None else None)

    @staticmethod
    def backward(ctx, _gw, _gb):
        (w,) = ctx.saved_tensors
        st = ctx.stash
        if not st.gs:
            return None, None, None
        X = st.xs[0] if len(st.xs) == 1 else torch.cat(st.xs, 0)
        G = st.gs[0] if len(st.gs) == 1 else torch.cat(st.gs, 0)
        st.xs, st.gs = [], []
        _, gw, gb = torch.ops.aten.convolution_backward(G, X, w, [w.shape[0]] if ctx.has_bias else None, [1, 1], [1, 1],
                                                        [1, 1], False, [0, 0], 1, [False, True, ctx.has_bias])
        return gw, (gb if ctx.has_bias else None), None


class _SplitColumns(torch.autograd.Function):
    """w [Cout, Cin, k, k] -> (w[:, outside r0:r1], w[:, r0:r1]) as two dense tensors; backward = one cat.  The first conv
    of the encoder / prior sees cat(static, recurrent) channels: the two column blocks multiply different operands
    (SimpleParamNet.recurrent_split)."""

    @staticmethod
    def forward(ctx, w, r0, r1):
        ctx.rng = (r0, r1)
        return torch.cat((w[:, :r0], w[:, r1:]), 1), w[:, r0:r1].contiguous()

    @staticmethod
    def backward(ctx, gs, gr):
        r0, r1 = ctx.rng
        return torch.cat((gs[:, :r0], gr, gs[:, r0:]), 1), None, None


class _StepConvAct(torch.autograd.Function):
    """one timestep of conv3x3(pad 1) + bias [+ leaky_relu]; backward = activation backward + data gradient only,
    the weight gradient is deferred to _WeightPort."""

    @staticmethod
    def forward(ctx, x, w, b, stash, slope):
        if x.is_cuda and x.dtype == torch.float32 and tuple(w.shape[2:]) == (3, 3):
            # own convolution kernel (fixed summation order), not MIOpen: see run_time_batched
            fp = K.fwd_prec(int(x.shape[2]), int(x.shape[3]))
            wd = w.detach()
            y = K.conv2d_raw(x.contiguous(), None, K.pack_weight(wd, prec=fp), int(w.shape[0]), 3,
                             0 if b is None else 3, None if b is None else b.detach().reshape(-1).contiguous(), None, 0,
                             prec=fp)
        else:
            y = F.conv2d(x, w, b, padding=1)
        if slope is not None:
            F.leaky_relu_(y, slope)
        ctx.stash, ctx.slope = stash, slope
        ctx.save_for_backward(x, w, y if slope is not None else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, w, y = ctx.saved_tensors
        if ctx.slope is not None:
            g = torch.ops.aten.leaky_relu_backward(g, y, ctx.slope, True)
        else:
            g = g.contiguous()
        ctx.stash.xs.append(x)
        ctx.stash.gs.append(g)
        gx = None
        if ctx.needs_input_grad[0]:
            gx = torch.ops.aten.convolution_backward(g, x, w, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1,
                                                     [True, False, False])[0]
        return gx, None, None, None, None


class _StepDenseAct(torch.autograd.Function):
    """_StepConvAct on a small map (H*W <= 16) through the dense split-precision kernels (rfn_smallmap_dense_bf16x3):
    one launch forward (bias + leaky_relu fused), one launch backward (leaky_relu backward fused, pre-activation
    gradient written out for the deferred weight gradient)."""

    @staticmethod
    def forward(ctx, x, w, b, stash, slope, packs):
        x = x.contiguous()
        y = K.smallmap_dense(x, packs[0], int(w.shape[0]), bias=b, slope_out=slope)
        ctx.stash, ctx.slope, ctx.packs, ctx.cin = stash, slope, packs, int(w.shape[1])
        ctx.save_for_backward(x, y if slope is not None else None)
        return y

    @staticmethod
    def backward(ctx, g):
        x, y = ctx.saved_tensors
        g = g.contiguous()
        if ctx.slope is not None:
            gx, gpre = K.smallmap_dense(g, ctx.packs[1], ctx.cin, y=y, slope_in=ctx.slope, want_a_out=True)
        else:
            gx, gpre = K.smallmap_dense(g, ctx.packs[1], ctx.cin), g
        ctx.stash.xs.append(x)
        ctx.stash.gs.append(gpre)
        return gx, None, None, None, None, None


class _StepDenseActPair(torch.autograd.Function):
    """_StepDenseAct for TWO independent layers (the encoder's and the prior's i-th layer of a timestep) in one launch
    each way (rfn_smallmap_dense_pair_bf16x3)."""

    @staticmethod
    def forward(ctx, x0, w0, b0, st0, slope0, packs0, x1, w1, b1, st1, slope1, packs1, add0=None, add1=None):
        """add0 / add1 (optional, [B, Cout, H, W]): pre-activation addends -- the time-batched projection of the static
        input channels of a split first layer; their gradient is the pre-activation gradient"""
        x0, x1 = x0.contiguous(), x1.contiguous()
        y0, y1 = K.smallmap_dense_pair(x0, packs0[0], int(w0.shape[0]), x1, packs1[0], int(w1.shape[0]), bias0=b0, bias1=b1,
                                       slope_out0=slope0, slope_out1=slope1, add0=add0, add1=add1)
        ctx.cfg = (st0, slope0, packs0, int(w0.shape[1]), st1, slope1, packs1, int(w1.shape[1]), add0 is not None,
                   add1 is not None)
        ctx.save_for_backward(x0, y0 if slope0 is not None else None, x1, y1 if slope1 is not None else None)
        return y0, y1

    @staticmethod
    def backward(ctx, g0, g1):
        x0, y0, x1, y1 = ctx.saved_tensors
        st0, slope0, packs0, cin0, st1, slope1, packs1, cin1, has0, has1 = ctx.cfg
        g0, g1 = g0.contiguous(), g1.contiguous()
        gx0, gp0, gx1, gp1 = K.smallmap_dense_pair(g0, packs0[1], cin0, g1, packs1[1], cin1, y0=y0, y1=y1,
                                                   slope_in0=slope0 or 0.0, slope_in1=slope1 or 0.0, want_a_out=True)
        st0.xs.append(x0); st0.gs.append(gp0)
        st1.xs.append(x1); st1.gs.append(gp1)
        return (gx0, None, None, None, None, None, gx1, None, None, None, None, None, gp0 if has0 else None,
                gp1 if has1 else None)


def recurrent_pair(net0, net1):
    """`run(x0, x1) -> (net0.raw(x0), net1.raw(x1))` for two SimpleParamNets applied side by side once per timestep
    (encoder and prior, RFN_new.py:167-179): when both take the dense small-map path and have the same number of
    layers, layer i of both runs in ONE launch each way; otherwise the two `recurrent()` callables run one after the
    other.  Weight gradients are time-batched either way (_WeightPort)."""
    f0, f1 = net0.recurrent(), net1.recurrent()
    p0, p1 = getattr(f0, "ports", None), getattr(f1, "ports", None)
    if p0 is None or p1 is None or len(p0) != len(p1) or os.environ.get("RFN_PAIR_LAUNCH") == "0":
        return lambda x0, x1: (f0(x0), f1(x1))

    def run(x0, x1):
        H, W = int(x0.shape[2]), int(x0.shape[3])
        k0, k1 = f0.dense_packs(x0), f1.dense_packs(x1)
        if k0 is None or k1 is None or tuple(x1.shape[2:]) != (H, W) or x0.shape[0] != x1.shape[0]:
            return f0(x0), f1(x1)
        for i in range(len(p0)):
            w0, b0, st0, s0 = p0[i]
            w1, b1, st1, s1 = p1[i]
            x0, x1 = _StepDenseActPair.apply(x0, w0, b0, st0, s0, k0[i], x1, w1, b1, st1, s1, k1[i])
        return x0, x1
    return run


def recurrent_pair_split(net0, static0, rng0, net1, static1, rng1, steps):
    """recurrent_pair for first-layer inputs of the form cat(static channels, recurrent channels) (RFN_new.py:167-179:
    h_t and the frame features are known for every t before the latent loop, only z is recurrent).  `static` = the static
    channels of ALL steps, step-major [steps*B, Cs, H, W]; `rng` = (start, stop) of the recurrent channels in the layer's
    input order.  The static block of the first conv is ONE time-batched product per net before the loop; the per-step
    launch multiplies the recurrent channels only and adds that projection.  Returns run(t, xr0, xr1) -> (raw0, raw1), or
    None when the dense small-map path cannot take these nets (the caller then uses recurrent_pair on the full inputs)."""
    if os.environ.get("RFN_SPLIT_FIRST_LAYER") == "0":
        return None
    f0, f1 = net0.recurrent_split(static0, rng0, steps), net1.recurrent_split(static1, rng1, steps)
    if f0 is None or f1 is None or len(f0.ports) != len(f1.ports):
        return None

    def run(t, x0, x1):
        for i in range(len(f0.ports)):
            w0, b0, st0, s0 = f0.ports[i]
            w1, b1, st1, s1 = f1.ports[i]
            if i == 0:
                x0, x1 = _StepDenseActPair.apply(x0, w0, b0, st0, s0, f0.packs[0], x1, w1, b1, st1, s1, f1.packs[0],
                                                 f0.proj[t], f1.proj[t])
            else:
                x0, x1 = _StepDenseActPair.apply(x0, w0, b0, st0, s0, f0.packs[i], x1, w1, b1, st1, s1, f1.packs[i])
        return x0, x1
    return run


class SimpleParamNet(nn.Module):
    """Utils/modules.py:216-244 — conv stack then a conv producing (loc, softplus(raw scale))."""

    def __init__(self, structure, in_channels, out_channels, norm_type="batchnorm", non_lin="leakyrelu", scale=2):
        super().__init__()
        layers = []
        for item in structure:
            if item == "pool":
                layers.append(nn.MaxPool2d(kernel_size=2, stride=2))
            elif item == "conv":
                cc = int(scale * in_channels)
                layers += [nn.Conv2d(in_channels, cc, kernel_size=3, stride=2, padding=1),
                           NormLayer(cc, norm_type=norm_type), ActFun(non_lin, in_place=True)]
                in_channels = cc
            else:
                layers += [nn.Conv2d(in_channels, item, kernel_size=3, padding=1), NormLayer(item, norm_type=norm_type),
                           ActFun(non_lin, in_place=True)]
                in_channels = item
        self.net = nn.Sequential(*layers)
        self.param_net = nn.Conv2d(in_channels, 2 * out_channels, kernel_size=3, stride=1, padding=1)
        self.softplus = nn.Softplus()

    def raw(self, x):
        """[B, 2*out, h, w] = (loc | raw scale) before the chunk + softplus of forward()."""
        return self.param_net(self.net(x))

    def _recurrent_convs(self, force=False):
        """[(conv, leaky slope | None)] when the stack is [conv3x3 s1, no norm, leaky_relu]* + param_net, else None"""
        layers = list(self.net)
        # (also under no_grad: the forward-only loss -- bench parity check, evaluator -- must take the same launches as the
        # training forward, not MIOpen's split-K kernels with their order-dependent float atomics)
        ok = len(layers) % 3 == 0 and (self.param_net.weight.is_cuda or force)
        convs = []
        for i in range(0, len(layers), 3):
            if not ok:
                break
            c, n, a = layers[i:i + 3]
            ok = (isinstance(c, nn.Conv2d) and c.kernel_size == (3, 3) and c.stride == (1, 1) and c.padding == (1, 1)
                  and isinstance(n, NormLayer) and isinstance(n.norm, NoNorm) and isinstance(a, ActFun)
                  and isinstance(a.net, nn.LeakyReLU))
            if ok:
                convs.append((c, a.net.negative_slope))
        if not ok:
            return None
        return convs + [(self.param_net, None)]

    def recurrent_split(self, static_all, rng, steps):
        """see recurrent_pair_split: an object with .ports [(w, b, stash, slope)] (layer 0 = the recurrent column block),
        .packs [(forward, data-gradient)] per layer and .proj = the `steps` per-step projections of the static channels
        through the first conv's static column block; or None when the dense small-map kernels do not apply."""
        convs = self._recurrent_convs()
        if convs is None or len(convs) < 2 or not static_all.is_cuda:
            return None
        H, W = int(static_all.shape[2]), int(static_all.shape[3])
        c0 = convs[0][0]
        r0, r1 = rng
        Cs, Cr = int(static_all.shape[1]), r1 - r0
        if (not all(K.smallmap_supported(c, H, W) for c, _ in convs) or Cs + Cr != c0.in_channels or Cs == 0 or Cr <= 0
                or (Cs * H * W) % 8 or (Cr * H * W) % 8 or static_all.shape[0] % steps):
            return None
        B = static_all.shape[0] // steps
        ws, wr = _SplitColumns.apply(c0.weight, r0, r1)
        st_s, st_r = _StepStash(), _StepStash()
        ws_p, _ = _WeightPort.apply(ws, None, st_s)
        wr_p, b_p = _WeightPort.apply(wr, c0.bias, st_r)
        ports = [(wr_p, b_p, st_r, convs[0][1])]
        packs = [(K.smallmap_pack(wr, H, W, False), K.smallmap_pack(wr, H, W, True))]
        for c, slope in convs[1:]:
            st = _StepStash()
            w, b = _WeightPort.apply(c.weight, c.bias, st)
            ports.append((w, b, st, slope))
            packs.append((K.smallmap_pack(c.weight, H, W, False), K.smallmap_pack(c.weight, H, W, True)))
        # all steps' static projection in one product (no bias, no activation: both belong to the per-step launch)
        proj = _StepDenseAct.apply(static_all, ws_p, None, st_s, None,
                                   (K.smallmap_pack(ws, H, W, False), K.smallmap_pack(ws, H, W, True)))
        return SimpleNamespace(ports=ports, packs=packs, proj=proj.view(steps, B, *proj.shape[1:]).unbind(0))

    def recurrent(self, force=False):
        """A callable equal to `raw` for use once per timestep inside ONE loss evaluation: the weight / bias gradients
        of all its calls are computed time-batched when the backward pass leaves the recurrence (_WeightPort).  Falls
        back to `raw` for layer stacks other than [conv3x3 s1, no norm, leaky_relu / relu-free]* (e.g. batchnorm)."""
        convs = self._recurrent_convs(force)
        if convs is None:
            return self.raw
        ports = []
        for c, slope in convs:
            st = _StepStash()
            w, b = _WeightPort.apply(c.weight, c.bias, st)
            ports.append((w, b, st, slope))

        packs = {}

        def dense_packs(x):
            """packed (forward, data-gradient) matrices of every layer for x's map size, or None (dense path unusable)"""
            H, W = int(x.shape[2]), int(x.shape[3])
            if not (x.is_cuda and all(K.smallmap_supported(c, H, W) for c, _ in convs)):
                return None
            if (H, W) not in packs:  # once per loss evaluation: both products of every layer
                packs[(H, W)] = [(K.smallmap_pack(c.weight, H, W, False), K.smallmap_pack(c.weight, H, W, True))
                                 for c, _ in convs]
            return packs[(H, W)]

        def run(x):
            pk = dense_packs(x)
            for i, (w, b, st, slope) in enumerate(ports):
                if pk is not None:
                    x = _StepDenseAct.apply(x, w, b, st, slope, pk[i])
                else:
                    x = _StepConvAct.apply(x, w, b, st, slope)
            return x
        run.ports, run.dense_packs = ports, dense_packs
        return run

    def forward(self, x):
        loc, log_scale = self.raw(x).chunk(2, 1)
        return loc, self.softplus(log_scale)


class ConvLSTMLayer(nn.Module):
    """Utils/modules.py:326-393.  Parameters live in `self.conv[0]` (weight [4Hc, Cin+Hc, k, k], bias U(0,1), weights
    xavier-normal).  The peephole tensors Wci/Wcf/Wco are identically zero in every reference run.  They
    exist as registered (non-trained) parameters only when a checkpoint that contains them was loaded (reference CPU
    checkpoints do, GPU ones do not); otherwise they are not materialised, so `state_dict()` has the same keys before and
    after a forward pass and matches a reference GPU checkpoint."""

    def __init__(self, in_channels, hidden_channels, kernel_size, bias, dropout=0, peephole=True, norm=False):
        super().__init__()
        assert not norm and dropout == 0, "GroupNorm / Dropout2d variants are not on the RFN path (defaults only)"
        self.in_channels, self.hidden_channels = in_channels, hidden_channels
        self.kernel_size, self.peephole, self.bias = kernel_size, peephole, bias
        self.padding = ((kernel_size[0] - 1) // 2, (kernel_size[1] - 1) // 2)
        assert kernel_size[0] == kernel_size[1] and kernel_size[0] in (1, 3), "kernels: 1x1 or 3x3"
        self.conv = nn.Sequential(nn.Conv2d(in_channels + hidden_channels, 4 * hidden_channels, kernel_size, 1,
                                            self.padding, bias=bias))
        nn.init.xavier_normal_(self.conv[0].weight)
        if bias:
            nn.init.uniform_(self.conv[0].bias)
        self.init_done = False
        self._pe_nonzero = None

    def initialize_peephole(self, height, width, device):
        """the reference creates zero peephole tensors here (Utils/modules.py:385-393); zeros contribute nothing, so
        nothing is materialised unless a checkpoint brought them (see _load_from_state_dict)"""
        return

    def _load_from_state_dict(self, state_dict, prefix, *args, **kw):
        # reference CPU checkpoints carry lstm.LSTMlayer.{Wci,Wcf,Wco}; GPU ones do not
        for n in ("Wci", "Wcf", "Wco"):
            k = prefix + n
            if k in state_dict and not hasattr(self, n):
                self.register_parameter(n, nn.Parameter(torch.zeros_like(state_dict[k]), requires_grad=False))
        self._pe_nonzero = None
        super()._load_from_state_dict(state_dict, prefix, *args, **kw)

    def _peephole_tensors(self, h, w, device):
        if not self.init_done:
            self.initialize_peephole(h, w, device)
            self.init_done = True
        pe = [getattr(self, n, None) for n in ("Wci", "Wcf", "Wco")]
        if pe[0] is not None:
            if self._pe_nonzero is None:  # one host sync per (re)load, not per step; they are never trained
                self._pe_nonzero = any(bool(t.any()) for t in pe)
            if not self._pe_nonzero:
                pe = [None, None, None]  # identically zero in every reference run: skip the reads
        return pe

    def forward(self, input_tensor, cur_state):
        b, c, h, w = input_tensor.shape
        conv = self.conv[0]
        if cur_state[0] is None:
            h_cur = torch.zeros(b, self.hidden_channels, h, w, device=input_tensor.device)
            c_cur = torch.zeros(b, self.hidden_channels, h, w, device=input_tensor.device)
        else:
            h_cur, c_cur = cur_state
        pe = self._peephole_tensors(h, w, input_tensor.device)
        return K.ConvLSTMCellFn.apply(input_tensor.contiguous(), h_cur.contiguous(), c_cur.contiguous(), conv.weight,
                                      conv.bias, pe[0], pe[1], pe[2])


class ConvLSTM(nn.Module):
    """Utils/modules.py:396-414 — x [B,S,C,H,W] -> (stack [B,S,Hc,H,W], h_S, c_S)."""

    def __init__(self, in_channels, hidden_channels, kernel_size, bias=True, dropout=0, peephole=True, norm=False):
        super().__init__()
        self.hidden_channels = hidden_channels
        self.LSTMlayer = ConvLSTMLayer(in_channels=in_channels, hidden_channels=hidden_channels,
                                       kernel_size=kernel_size, bias=bias, dropout=dropout, peephole=peephole,
                                       norm=norm)

    def forward(self, x, ht=None, ct=None):
        output = []
        for t in range(x.size(1)):
            ht, ct = self.LSTMlayer(input_tensor=x[:, t], cur_state=[ht, ct])
            output.append(ht)
        return torch.stack(output, 1), ht, ct

    def forward_steps(self, x_all, ht, ct):
        """x_all [S,B,C,H,W] (step-major) with given initial states -> ([h_1..h_S], h_S, c_S): what S calls
        forward(x_t.unsqueeze(1), h, c) return (RFN_new.py:131-139 drives the layer one frame at a time).  On small maps
        the whole sequence is one autograd node with the input projection / input and weight gradients time-batched
        (rfn_hip.ops.ConvLSTMSeqFn)."""
        layer, conv = self.LSTMlayer, self.LSTMlayer.conv[0]
        S, B, C, H, W = x_all.shape
        pe = layer._peephole_tensors(H, W, x_all.device)
        if (x_all.is_cuda and ht is not None and ct is not None and pe[0] is None and layer.kernel_size[0] == 3
                and K.convlstm_seq_supported(conv.weight, C, H, W)):
            h_all, c_last = K.ConvLSTMSeqFn.apply(x_all, ht, ct, conv.weight, conv.bias)
            hs = list(h_all.unbind(0))
            return hs, hs[-1], c_last
        hs = []
        for t in range(S):
            ht, ct = layer(input_tensor=x_all[t], cur_state=[ht, ct])
            hs.append(ht)
        return hs, ht, ct
