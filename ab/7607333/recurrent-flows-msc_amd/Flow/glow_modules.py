"""Glow building blocks with the reference's class names, call signatures and state_dict keys
(Flow/glow_modules.py of the reference), computed by the gfx950 kernels of librfn_hip.so.

Every module keeps the reference's `forward(input, [condition,] logdet, reverse)` contract, where `logdet` may be
None (left alone), a Python number, a 0-d tensor or an [N] tensor.  GlowStep (Flow/glow.py) does not call these
forwards one by one: it hands the parameters of its three sub-modules to one fused autograd node.
"""
import torch
import torch.nn as nn

from Utils import split_feature, ActFun
from rfn_hip import ops as K


def add_logdet(logdet, d, sign=1.0):
    if logdet is None:
        return None
    return logdet + d if sign > 0 else logdet - d


class ActNorm(nn.Module):
    """Flow/glow_modules.py:10-54 — per-channel affine with data dependent init on the first TRAINING call.
    `initialized` is a uint8 buffer (part of checkpoints); the host mirrors it in `_init_done` to avoid the
    reference's per-call `.item()` device sync."""

    def __init__(self, num_channels):
        super().__init__()
        size = [1, num_channels, 1, 1]
        self.bias = nn.Parameter(torch.zeros(*size))
        self.logs = nn.Parameter(torch.zeros(*size))
        self.register_buffer("initialized", torch.tensor(0, dtype=torch.uint8))
        self._init_done = None  # unknown until first use / after load_state_dict

    def _load_from_state_dict(self, *a, **k):
        super()._load_from_state_dict(*a, **k)
        self._init_done = None

    def needs_init(self):
        if self._init_done is None:
            self._init_done = bool(self.initialized.item() != 0)
        return not self._init_done

    def set_from_stats(self, mean, var):
        """glow_modules.py:26-31: bias = -mean, logs = log(1/(std+1e-6)) with the unbiased std."""
        with torch.no_grad():
            self.bias.copy_((-mean).view_as(self.bias))
            self.logs.copy_((1.0 / (var.sqrt() + 1e-6)).log().view_as(self.logs))

    def mark_initialized(self):
        self.initialized.fill_(1)
        self._init_done = True

    def initialize(self, input):
        if not self.training:
            return
        mean, var = K.channel_stats(input.detach().contiguous())
        self.set_from_stats(mean, var)

    def forward(self, input, logdet, reverse):
        if self.needs_init():
            self.initialize(input)
            self.mark_initialized()
        C = input.shape[1]
        dims = input.size(2) * input.size(3)
        eye = torch.eye(C, device=input.device)
        if not reverse:
            out = _ShellFn.apply(input.contiguous(), eye, self.bias, self.logs)
            return out, add_logdet(logdet, torch.sum(self.logs) * dims)
        out = K.invconv_actnorm_rev(input.detach().contiguous(), self.bias.detach().reshape(-1),
                                    self.logs.detach().reshape(-1), eye)
        return out, add_logdet(logdet, torch.sum(self.logs) * dims, -1.0)


class _ShellFn(torch.autograd.Function):
    """z = Wm · ((x + bias) * exp(logs)) with gradients (the HBM-bound shell of a Glow step)."""

    @staticmethod
    def forward(ctx, x, Wm, bias, logs):
        ctx.save_for_backward(x, Wm, bias, logs)
        return K.actnorm_invconv_fwd(x, bias.detach().reshape(-1), logs.detach().reshape(-1), Wm.detach())

    @staticmethod
    def backward(ctx, gz):
        x, Wm, bias, logs = ctx.saved_tensors
        gx, gW, gb, gl = K.actnorm_invconv_bwd(x, bias.detach().reshape(-1), logs.detach().reshape(-1), Wm.detach(),
                                               gz.contiguous())
        return gx, gW, gb.view_as(bias), gl.view_as(logs)


class Conv2dZeros(nn.Module):
    """Flow/glow_modules.py:106-121 — zero-initialised conv, output scaled by exp(3·logs)."""

    def __init__(self, in_channel, out_channel, kernel_size=[3, 3], stride=[1, 1]):
        super().__init__()
        assert list(stride) == [1, 1] and kernel_size[0] == kernel_size[1] and kernel_size[0] in (1, 3)
        padding = (kernel_size[0] - 1) // 2
        self.conv = nn.Conv2d(in_channel, out_channel, kernel_size=kernel_size, stride=stride, padding=padding)
        self.logscale_factor = 3
        self.logs = nn.Parameter(torch.zeros(out_channel, 1, 1))
        self.conv.weight.data.zero_()
        self.conv.bias.data.zero_()

    def forward(self, input, input2=None):
        return K.conv_ep(input.contiguous() if input.stride(-1) != 1 else input, input2, self.conv.weight,
                         self.conv.bias, self.logs, 2, 0)


class Conv2dNorm(nn.Module):
    """Flow/glow_modules.py:123-147 — conv (N(0,0.05) weights, no bias under actnorm) + ActNorm (log-det dropped).
    `forward(x, act=...)` lets the caller fuse the following ActFun into the conv epilogue."""

    def __init__(self, in_channels, out_channels, kernel_size=[3, 3], stride=[1, 1], norm="actnorm"):
        super().__init__()
        assert list(stride) == [1, 1] and kernel_size[0] == kernel_size[1] and kernel_size[0] in (1, 3)
        padding = [(kernel_size[0] - 1) // 2, (kernel_size[1] - 1) // 2]
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, bias=(norm != "actnorm"))
        self.conv.weight.data.normal_(mean=0.0, std=0.05)
        self.norm = norm
        if norm == "actnorm":
            self.norm_type = ActNorm(out_channels)
        elif norm == "batchnorm":
            self.conv.bias.data.zero_()
            self.norm_type = nn.BatchNorm2d(out_channels)

    def maybe_init(self, input, input2=None):
        """data dependent init of the inner ActNorm from the raw conv output (first training call only)."""
        an = self.norm_type
        if self.norm != "actnorm" or not an.needs_init():
            return
        if self.training:
            w = self.conv.weight
            u = K.conv2d_raw(input.detach(), None if input2 is None else input2.detach(), K.pack_weight(w),
                             int(w.shape[0]), int(w.shape[2]))
            an.set_from_stats(*K.channel_stats(u))
        an.mark_initialized()

    def forward(self, input, input2=None, act="none"):
        if self.norm == "actnorm":
            self.maybe_init(input, input2)
            return K.conv_ep(input, input2, self.conv.weight, self.norm_type.bias, self.norm_type.logs, 1, K.ACT[act])
        out = K.conv_ep(input, input2, self.conv.weight, self.conv.bias, None, 3, 0)
        if self.norm == "batchnorm":
            out = self.norm_type(out)
        if act == "relu":
            out = torch.relu(out)
        elif act == "leakyrelu":
            out = torch.nn.functional.leaky_relu(out, 0.2)
        return out


class InvConv(nn.Module):
    """Flow/glow_modules.py:150-221 — invertible 1x1 convolution, optionally LU-parameterised
    W = P (L∘tril₋₁ + I)(U∘triu₊₁ + diag(sign_s·exp(log_s))).  The tiny C×C algebra stays in torch (autograd takes
    the kernel's gW back to lower/upper/log_s); the per-pixel mat-vec runs in the fused shell kernel."""

    def __init__(self, num_channels, LU_decomposed):
        super().__init__()
        w_shape = [num_channels, num_channels]
        w_init = torch.linalg.qr(torch.randn(*w_shape))[0]
        if not LU_decomposed:
            self.weight = nn.Parameter(w_init.clone())
        else:
            p, lower, upper = torch.linalg.lu(w_init)
            s = torch.diag(upper)
            self.register_buffer("p", p)
            self.register_buffer("sign_s", torch.sign(s))
            self.lower = nn.Parameter(lower)
            self.log_s = nn.Parameter(torch.log(torch.abs(s)))
            self.upper = nn.Parameter(torch.triu(upper, 1))
            self.register_buffer("l_mask", torch.tril(torch.ones(w_shape), -1), persistent=False)
            self.register_buffer("eye", torch.eye(*w_shape), persistent=False)
        self.w_shape = w_shape
        self.LU_decomposed = LU_decomposed

    def matrices(self):
        lower = self.lower * self.l_mask + self.eye
        u = self.upper * self.l_mask.t() + torch.diag(self.sign_s * torch.exp(self.log_s))
        return lower, u

    def get_weight(self, input, reverse):
        """returns (C×C matrix, dlogdet) — glow_modules.py:178-207."""
        h, w = input.shape[2], input.shape[3]
        if not self.LU_decomposed:
            dlogdet = torch.slogdet(self.weight)[1] * h * w
            weight = torch.inverse(self.weight) if reverse else self.weight
        else:
            lower, u = self.matrices()
            dlogdet = torch.sum(self.log_s) * h * w
            if reverse:
                # U⁻¹ L⁻¹ P⁻¹ (glow_modules.py:198-203) by two triangular solves; P is a permutation, P⁻¹ = Pᵀ
                weight = torch.linalg.solve_triangular(
                    u, torch.linalg.solve_triangular(lower, self.p.t(), upper=False, unitriangular=True), upper=True)
            else:
                weight = torch.matmul(self.p, torch.matmul(lower, u))
        return weight, dlogdet

    def forward(self, input, logdet, reverse):
        weight, dlogdet = self.get_weight(input, reverse)
        C = input.shape[1]
        zero = torch.zeros(C, device=input.device)
        if not reverse:
            z = _ShellFn.apply(input.contiguous(), weight, zero, zero)
            return z, add_logdet(logdet, dlogdet)
        z = K.invconv_actnorm_rev(input.detach().contiguous(), zero, zero, weight.detach())
        return z, add_logdet(logdet, dlogdet, -1.0)


class AffineCoupling(nn.Module):
    """Flow/glow_modules.py:223-291 — conditional affine coupling; `net` = Conv2dNorm 3x3, act, Conv2dNorm 1x1, act,
    Conv2dZeros (same Sequential indices, hence the same state_dict keys)."""

    def __init__(self, x_size, condition_size, hidden_units=256, non_lin="relu", clamp_type="realnvp"):
        super().__init__()
        Bx, Cx, Hx, Wx = x_size
        B, C, H, W = condition_size
        channels = Cx // 2 + C
        self.net = nn.Sequential(
            Conv2dNorm(channels, hidden_units),
            ActFun(non_lin),
            Conv2dNorm(hidden_units, hidden_units, kernel_size=[1, 1]),
            ActFun(non_lin),
            Conv2dZeros(hidden_units, Cx),
        )
        self.non_lin = non_lin
        self.clamp_type = clamp_type if clamp_type in ("glow", "softclamp", "realnvp") else "none"
        if clamp_type == "realnvp":
            self.scale = nn.Parameter(torch.zeros(Cx // 2, 1, 1))
            self.scale_shift = nn.Parameter(torch.zeros(Cx // 2, 1, 1))

    def clamper(self, s):
        if self.clamp_type == "realnvp":
            return self.scale * torch.tanh(s) + self.scale_shift
        if self.clamp_type == "glow":
            return torch.log(torch.sigmoid(s + 2.0))
        if self.clamp_type == "softclamp":
            return 2.5 * 0.636 * torch.atan(s / 2.5)
        return s

    def nn_params(self):
        n0, n2, n4 = self.net[0], self.net[2], self.net[4]
        sc = getattr(self, "scale", None)
        sh = getattr(self, "scale_shift", None)
        return (n0.conv.weight, n0.norm_type.bias, n0.norm_type.logs, n2.conv.weight, n2.norm_type.bias,
                n2.norm_type.logs, n4.conv.weight, n4.conv.bias, n4.logs, sc, sh)

    def maybe_init(self, z1, cond):
        """first training call: initialise the two inner ActNorms layer by layer (glow_modules.py:139-142)."""
        n0, n2 = self.net[0], self.net[2]
        if not (n0.norm_type.needs_init() or n2.norm_type.needs_init()):
            return
        with torch.no_grad():
            c2 = cond if cond.shape[1] > 0 else None
            n0.maybe_init(z1, c2)
            h1 = n0(z1, c2, act=self.non_lin)
            n2.maybe_init(h1)

    def forward(self, x, condition, logdet, reverse):
        assert condition.shape[2:4] == x.shape[2:4], "condition and x in affine needs to match"
        N, C = x.shape[0], x.shape[1]
        eye = torch.eye(C, device=x.device)
        zero = torch.zeros(C, device=x.device)
        z1, _ = split_feature(x, "split")
        self.maybe_init(z1, condition)
        fn = K.GlowStepRevFn if reverse else K.GlowStepFn
        # identity shell (ActNorm(0,0), W = I) + coupling == the coupling layer alone
        out, dl = fn.apply(x.contiguous(), condition.contiguous(), eye, zero, zero, *self.nn_params(),
                           K.ACT[self.non_lin], K.CLAMP[self.clamp_type])
        return out, add_logdet(logdet, dl)


class Squeeze2d(nn.Module):
    """Flow/glow_modules.py:294-310 — out[b,4c+2i+j,h,w] = in[b,c,2h+i,2w+j] (bit exact copy kernel)."""

    def forward(self, x, undo_squeeze):
        return K.Squeeze2dFn.apply(x if x.stride(-1) == 1 else x.contiguous(), bool(undo_squeeze))


class Split2d(nn.Module):
    """Flow/glow_modules.py:312-369 — factor out half the channels under a conditional Gaussian."""

    def __init__(self, x_size, condition_size, make_conditional=True, clamp_function="softplus"):
        super().__init__()
        self.make_conditional = make_conditional
        Bx, Cx, Hx, Wx = x_size
        non_lin = "relu"
        if make_conditional:
            B, C, H, W = condition_size
            channels = Cx // 2 + C
            self.convcond = nn.Sequential(
                Conv2dNorm(C, C),
                ActFun(non_lin),
                Conv2dNorm(C, C, kernel_size=[1, 1]),
                ActFun(non_lin),
            )
        else:
            channels = Cx // 2
        self.conv = nn.Sequential(Conv2dZeros(channels, Cx), )
        assert clamp_function in ("softplus", "exp"), \
            "Please specify a clamp function for the split2d from the set {softplus, exp}"
        self.clamp_function = clamp_function
        self.std_mode = 0 if clamp_function == "softplus" else 1

    def _params(self, z1, condition):
        if self.make_conditional:
            c = self.convcond[0](condition, act="relu")
            c = self.convcond[2](c, act="relu")
            return self.conv[0](z1, c)
        return self.conv[0](z1)

    def forward(self, x, condition, logdet, reverse, temperature=None, eps=None):
        if not reverse:
            z1, z2 = split_feature(x, "split")
        else:
            z1 = x
        out = self._params(z1, condition)
        if not reverse:
            if logdet is not None:
                logdet = logdet + K.GaussLogpFn.apply(z2, out, 0, self.std_mode)
            return z1, logdet
        if eps is None:
            eps = torch.randn((z1.shape[0], out.shape[1] // 2) + tuple(z1.shape[2:]), device=z1.device)
        z2 = K.gauss_sample(out.detach(), eps, 0, self.std_mode, temperature)
        return torch.cat((z1, z2), dim=1), logdet


class BatchNormFlow(nn.Module):
    """Flow/glow_modules.py:56-104 — optional (non-default) flow normalisation; plain torch ops as scoped in
    SURVEY.md §2 row 1."""

    def __init__(self, x_size, momentum=0.1, eps=1e-5):
        super().__init__()
        Bx, Cx, Hx, Wx = x_size
        size = [1, Cx, Hx, Wx]
        self.log_gamma = nn.Parameter(torch.zeros(size))
        self.beta = nn.Parameter(torch.zeros(size))
        self.momentum, self.eps = momentum, eps
        self.register_buffer("running_mean", torch.zeros(size))
        self.register_buffer("running_var", torch.ones(size))

    def forward(self, input, logdet, reverse):
        if self.training and not reverse:
            mean = input.mean(0)
            var = (input - mean).pow(2).mean(0) + self.eps
            self.running_mean.mul_(self.momentum).add_(mean.data * (1 - self.momentum))
            self.running_var.mul_(self.momentum).add_(var.data * (1 - self.momentum))
        else:
            mean, var = self.running_mean, self.running_var
        dlogdet = torch.sum(self.log_gamma - 0.5 * torch.log(var))
        if not reverse:
            z = torch.exp(self.log_gamma) * ((input - mean) / var.sqrt()) + self.beta
            return z, add_logdet(logdet, dlogdet)
        z = ((input - self.beta) / torch.exp(self.log_gamma)) * var.sqrt() + mean
        return z, add_logdet(logdet, dlogdet, -1.0)
