"""CPU study of split-precision convolution arithmetics (developer tool; imports the oracle = test infrastructure).

Emulates, inside every convolution of the oracle's flow / recurrent nets, the operand splitting the gfx950 kernels use
and reports the bits/dim deviation from the plain fp32 oracle on the canonical architecture (B=2, T frames, pinned
draws, flow perturbed by N(0, s^2)):
    x3   a_hi b_hi + a_hi b_lo + a_lo b_hi                (bf16x3, 2 x 8-bit pieces per operand)
    x4   x3 + a_lo b_lo
    x6   3 pieces per operand (hi, mid, lo = 24 bits): hh + hm + mh + mm + hl + lh
Products are formed exactly (float64) so only representation + dropped-term errors show; the fp32 accumulation error of
the MFMA path is the same as the fp32 oracle's order of magnitude.

    python tools/precision_study.py [T] [scale ...]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "recurrent-flows-msc_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from oracle import rfn_oracle as O  # noqa: E402

_conv = F.conv2d


def pieces(t, n):
    out, r = [], t.float()
    for _ in range(n):
        h = r.to(torch.bfloat16).float()
        out.append(h.double())
        r = r - h
    return out


ZEROS_CTX = [False]   # True while the oracle is inside a Conv2dZeros


def pieces_h(t, n):
    """fp16 pieces after a per-tensor power-of-two scaling that puts max|t| in [2^14, 2^15)"""
    t = t.float()
    m = float(t.abs().max())
    if m == 0 or m != m or m == float("inf"):
        return [t.double()] + [torch.zeros_like(t).double() for _ in range(n - 1)], 1.0
    import math
    e = 14 - math.floor(math.log2(m))
    sc = 2.0 ** e
    out, r = [], t * sc
    for _ in range(n):
        h = r.to(torch.float16).float()
        out.append(h.double())
        r = r - h
    return out, sc


def make_conv(scheme_):
    def conv(x, w, b=None, stride=1, padding=0, dilation=1, groups=1):
        scheme = scheme_
        if scheme_ == "x3z6":    # bf16x3 everywhere, 3-piece arithmetic only in the Conv2dZeros layers
            scheme = "x6" if ZEROS_CTX[0] else "x3"
        if scheme_ == "x6z3":
            scheme = "x3" if ZEROS_CTX[0] else "x6"
        if scheme_ == "mixA":    # fp32-grade on the big maps (flow levels 0-1), bf16x3 elsewhere
            scheme = "h3" if x.shape[2] * x.shape[3] >= 256 else "x3"
        if scheme_ == "mixB":    # the opposite
            scheme = "x3" if x.shape[2] * x.shape[3] >= 256 else "h3"
        if scheme_ == "mixC":    # bf16x3 only on the 2x2 maps (latent nets, level 4)
            scheme = "h3" if x.shape[2] * x.shape[3] > 4 else "x3"
        if scheme == "f32":
            return _conv(x, w, b, stride, padding, dilation, groups)
        if scheme in ("h3", "h4"):
            (xs, sx), (ws, sw) = pieces_h(x, 2), pieces_h(w, 2)
            terms = [(0, 0), (0, 1), (1, 0)] + ([(1, 1)] if scheme == "h4" else [])
            acc = None
            for wi, xi in terms:
                y = _conv(xs[xi], ws[wi], None, stride, padding, dilation, groups)
                acc = y if acc is None else acc + y
            acc = acc / (sx * sw)
            if b is not None:
                acc = acc + b.double().view(1, -1, 1, 1)
            return acc.float()
        np_ = 3 if scheme == "x6" else 2
        xs, ws = pieces(x, np_), pieces(w, np_)
        if scheme == "x3":
            terms = [(0, 0), (0, 1), (1, 0)]
        elif scheme == "x4":
            terms = [(0, 0), (0, 1), (1, 0), (1, 1)]
        elif scheme == "x5w":  # weights 3 pieces, activations 2
            ws = pieces(w, 3)
            terms = [(0, 0), (0, 1), (1, 0), (1, 1), (2, 0)]  # (w piece, x piece)
        else:
            terms = [(0, 0), (0, 1), (1, 0), (1, 1), (0, 2), (2, 0)]
        acc = None
        for wi, xi in terms:
            y = _conv(xs[xi], ws[wi], None, stride, padding, dilation, groups)
            acc = y if acc is None else acc + y
        if b is not None:
            acc = acc + b.double().view(1, -1, 1, 1)
        return acc.float()
    return conv


_zeros = O.conv2dzeros


def _zeros_tagged(*a, **k):
    ZEROS_CTX[0] = True
    try:
        return _zeros(*a, **k)
    finally:
        ZEROS_CTX[0] = False


O.conv2dzeros = _zeros_tagged


def main():
    import main_rfn
    from RFN import RFN
    import bench
    T = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    scales = [float(v) for v in sys.argv[2:]] or [0.01]
    B = 2
    args = main_rfn.build_parser().parse_args(main_rfn.canonical_smmnist_argv(B, T))
    g = torch.Generator().manual_seed(2)
    x = bench.make_batch(B, T, 11, "cpu") * 255 / 256 - 0.5
    draws = []
    for _ in range(T - 1):
        draws += [torch.randn(B, 56, 2, 2, generator=g), torch.randn(B, 56, 2, 2, generator=g),
                  torch.rand(B, 1, 64, 64, generator=g) / 256]
    torch.set_num_threads(8)
    for s_ in scales:
        torch.manual_seed(1)
        sd = {k: v.detach().clone() for k, v in RFN(args).state_dict().items()}
        with torch.no_grad():
            O.rfn_loss(sd, vars(args), x, draws, True)  # data dependent init (in place on sd)
            gp = torch.Generator().manual_seed(int(os.environ.get("PSEED", 3)))
            for k, v in sd.items():
                if k.startswith("flow.") and v.is_floating_point() and "initialized" not in k and k.split(".")[-1] not in ("p", "sign_s"):
                    v.add_(s_ * torch.randn(v.shape, generator=gp))
            res = {}
            for scheme in ("f32", "x3", "h3", "mixA", "mixB", "mixC"):
                F.conv2d = make_conv(scheme)
                O.F.conv2d = F.conv2d
                r = O.rfn_loss({k: v.clone() for k, v in sd.items()}, vars(args), x, draws, os.environ.get("PTRAIN", "1") == "1")
                res[scheme] = O.bits_per_dim(r[1], r[2], x.shape[2:], T - 1)
                F.conv2d = _conv
                O.F.conv2d = _conv
            ref = res["f32"]
            print("scale %g T %d: bits/dim f32 %.6g | " % (s_, T, ref) +
                  "  ".join("%s %.2e" % (k, abs(v - ref) / abs(ref)) for k, v in res.items() if k != "f32"), flush=True)


if __name__ == "__main__":
    main()
