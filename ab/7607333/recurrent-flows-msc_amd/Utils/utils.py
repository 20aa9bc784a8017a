"""Helpers with the reference's names and indexing conventions (Utils/utils.py of the reference).

Pure host-side logic: device selection, channel splitting views, per-sample reductions, free bits.
"""
import torch
import torch.nn as nn

use_gpu = False
device = None


def set_gpu(mode, verbose=False):
    """Utils/utils.py:16-23 — pick the accelerator when `mode` and one is present."""
    global use_gpu, device
    use_gpu = bool(mode) and torch.cuda.is_available()
    device = torch.device("cuda" if use_gpu else "cpu")
    if verbose:
        print("Device set to: ", device)
    return device


def batch_reduce(x, reduce=torch.sum, batch_dim=0):
    """Utils/utils.py:32-35 — reduce every non-batch dimension."""
    return reduce(x.reshape(x.size(batch_dim), -1), dim=-1)


def split_feature(tensor, type="split"):
    """Utils/utils.py:93-98 — "split": channel halves; "cross": even / odd channels (views, no copy)."""
    C = tensor.size(1)
    if type == "split":
        return tensor[:, : C // 2, ...], tensor[:, C // 2:, ...]
    if type == "cross":
        return tensor[:, 0::2, ...], tensor[:, 1::2, ...]
    raise ValueError("split_feature type must be 'split' or 'cross'")


def free_bits_kl(kl, free_bits=0.0, eps=1e-6):
    """Utils/utils.py:100-105."""
    return kl if free_bits < eps else kl.clamp(min=free_bits)


def get_numpy(t):
    return t.detach().to("cpu").numpy()


class Flatten(nn.Module):
    def forward(self, x):
        return x.view(x.size(0), -1)


class UnFlatten(nn.Module):
    def __init__(self, C_x, H_x, W_x):
        super().__init__()
        self.dims = (C_x, H_x, W_x)

    def forward(self, x):
        return x.view(x.size(0), *self.dims)


def get_layer_size(dims, kernels, paddings, strides, dilations, output_paddings=(), uneven_format=False,
                   transpose=False):
    """Utils/utils.py:70-91 — output size of a conv / transposed-conv stack."""
    h, w = dims
    pair = (lambda v: v) if uneven_format else (lambda v: (v, v))
    if not transpose:
        for k, p, s, d in zip(kernels, paddings, strides, dilations):
            k, p, s, d = pair(k), pair(p), pair(s), pair(d)
            h = (h + 2 * p[0] - d[0] * (k[0] - 1) - 1) // s[0] + 1
            w = (w + 2 * p[1] - d[1] * (k[1] - 1) - 1) // s[1] + 1
        return h, w
    assert len(output_paddings) == len(paddings), "Please specify output_padding when using transpose"
    for k, p, s, d, op in zip(kernels, paddings, strides, dilations, output_paddings):
        k, p, s, d, op = pair(k), pair(p), pair(s), pair(d), pair(op)
        h = (h - 1) * s[0] - 2 * p[0] + d[0] * (k[0] - 1) + op[0] + 1
        w = (w - 1) * s[1] - 2 * p[1] + d[1] * (k[1] - 1) + op[1] + 1
    return h, w
