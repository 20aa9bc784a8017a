"""Tensor-level wrappers and autograd Functions over the C ABI (include/rfn_hip.h).

Everything here runs on the GPU through librfn_hip.so; torch is used for memory, streams and autograd plumbing.
Frame-batched layout: tensors are [N, C, H, W] fp32 where N = all frames handed over by the caller (the RFN driver
time-batches B*(T-1) frames into one call).
"""
import ctypes

import torch

from . import lib as L

_i = ctypes.c_int
_l = ctypes.c_long

import os

# Arithmetic of the MFMA convolutions (RFN_CONV_PRECISION):
#   "mixed"  (default) forward pass fp32-grade, gradients split precision:
#              forward  coupling nets of the shallow levels: fused kernel, two scaled fp16 pieces per operand ("f16x3s",
#                       22 significant bits, csrc/coupling_po.hip); every other forward convolution on a map larger than
#                       2x2: fp32 MFMA (v_mfma_f32_32x32x2_f32); 2x2 maps (latent nets, ConvLSTM, deepest flow level):
#                       bf16x3 -- tools/precision_study.py: bits/dim error 1.3e-5 vs 1.0e-5 all-fp32-grade, 4.9e-4 all-bf16x3
#              backward bf16x3 (data and weight gradients)
#   "bf16x3" every convolution on two bf16 pieces per operand, three v_mfma_f32_32x32x16_bf16 per product (16 bits)
#   "f32"    every convolution on v_mfma_f32_32x32x2_f32
CONV_PRECISION = os.environ.get("RFN_CONV_PRECISION", "mixed")


def bwd_b3():
    """gradient convolutions run split precision (bf16x3)"""
    return CONV_PRECISION in ("bf16x3", "mixed")


MIXED_FWD = os.environ.get("RFN_MIXED_FWD", "bf16x6")  # fp32-grade arithmetic of the unfused forward convs: bf16x6 | f32


def fwd_prec(H, W):
    """arithmetic of an (unfused) forward convolution on an H x W map: 'bf16x3', 'bf16x6' (three bf16 pieces per operand,
    six MFMAs per product: fp32-grade at a third of the fp32-MFMA cost) or 'f32'"""
    if CONV_PRECISION == "mixed":
        return "bf16x3" if H * W <= 4 else MIXED_FWD
    return CONV_PRECISION

ACT = {"none": 0, "relu": 1, "leakyrelu": 2}
CLAMP = {"realnvp": 0, "glow": 1, "softclamp": 2, "none": 3}


def _hw(t):
    return int(t.shape[2]) * int(t.shape[3])


# ----------------------------------------------------------------------------------------------- raw kernels
def squeeze2d_raw(x, undo=False):
    N, C, H, W = x.shape
    xp, xns = L.frames(x, "x")
    if not undo:
        y = torch.empty((N, C * 4, H // 2, W // 2), device=x.device, dtype=x.dtype)
    else:
        y = torch.empty((N, C // 4, H * 2, W * 2), device=x.device, dtype=x.dtype)
    yp, yns = L.frames(y, "y")
    L.call("rfn_squeeze2d_f32", xp, _l(xns), yp, _l(yns), _i(N), _i(C), _i(H), _i(W), _i(1 if undo else 0),
           meta=_shell("squeeze2d", x, 2))
    return y


def _shell(name, t, n_tensors):
    """profiling metadata of a memory-bound shell launch: algorithmic bytes = n_tensors x the tensor's fp32 size"""
    return ("shell", name, 0.0, "x".join(str(int(d)) for d in t.shape), 4.0 * n_tensors * t.numel())


def channel_stats(x):
    """per-channel (mean, unbiased variance) over (N,H,W) — ActNorm data dependent init."""
    N, C = x.shape[0], x.shape[1]
    xp, xns = L.frames(x, "x")
    mean = torch.empty(C, device=x.device, dtype=torch.float32)
    var = torch.empty(C, device=x.device, dtype=torch.float32)
    L.call("rfn_channel_stats_f32", xp, _l(xns), L.dev(mean), L.dev(var), _i(N), _i(C), _i(_hw(x)))
    return mean, var


def actnorm_invconv_fwd(x, bias, logs, Wm):
    N, C = x.shape[0], x.shape[1]
    xp, xns = L.frames(x, "x")
    z = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    zp, zns = L.frames(z, "z")
    bias, logs, Wm = bias.contiguous(), logs.contiguous(), Wm.contiguous()  # held until the launch is enqueued
    L.call("rfn_actnorm_invconv_fwd_f32", xp, _l(xns), L.dev(bias), L.dev(logs), L.dev(Wm), zp, _l(zns), _i(N), _i(C),
           _i(_hw(x)), meta=_shell("actnorm_invconv_fwd", x, 2))
    return z


class ZeroArena:
    """one zero-filled allocation handed out in slices: the accumulate-into outputs of a backward node (weight
    gradients, per-channel sums) share ONE fill launch instead of one torch.zeros each."""

    def __init__(self, numel, device):
        self.buf = torch.zeros(int(numel), device=device, dtype=torch.float32)
        self.off = 0

    def take(self, *shape):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 3) // 4 * 4  # keep every slice 16-byte aligned
        assert self.off + n <= self.buf.numel(), "ZeroArena exhausted"
        t = self.buf[self.off:self.off + n].view(*shape)
        self.off += n_al
        return t


def _zeros(arena, *shape, device=None):
    return arena.take(*shape) if arena is not None else torch.zeros(shape, device=device, dtype=torch.float32)


def actnorm_invconv_bwd(x, bias, logs, Wm, gz, arena=None):
    N, C = x.shape[0], x.shape[1]
    xp, xns = L.frames(x, "x")
    gzp, gzns = L.frames(gz, "gz")
    gx = torch.empty(x.shape, device=x.device, dtype=x.dtype)
    gxp, gxns = L.frames(gx, "gx")
    gW = _zeros(arena, C, C, device=x.device)
    gb = _zeros(arena, C, device=x.device)
    gl = _zeros(arena, C, device=x.device)
    bias, logs, Wm = bias.contiguous(), logs.contiguous(), Wm.contiguous()  # held until the launch is enqueued
    L.call("rfn_actnorm_invconv_bwd_f32", xp, _l(xns), L.dev(bias), L.dev(logs), L.dev(Wm), gzp, _l(gzns), gxp, _l(gxns), L.dev(gW), L.dev(gb), L.dev(gl), _i(N), _i(C),
           _i(_hw(x)), meta=_shell("actnorm_invconv_bwd", x, 3))
    return gx, gW, gb, gl


def invconv_actnorm_rev(z, bias, logs, Winv):
    N, C = z.shape[0], z.shape[1]
    zp, zns = L.frames(z, "z")
    x = torch.empty(z.shape, device=z.device, dtype=z.dtype)
    xp, xns = L.frames(x, "x")
    bias, logs, Winv = bias.contiguous(), logs.contiguous(), Winv.contiguous()  # held until the launch is enqueued
    L.call("rfn_invconv_actnorm_rev_f32", zp, _l(zns), L.dev(bias), L.dev(logs), L.dev(Winv), xp, _l(xns), _i(N), _i(C),
           _i(_hw(z)))
    return x


class _ConvPackDesc(ctypes.Structure):   # rfn_pack_desc (include/rfn_hip.h)
    _fields_ = [("w", ctypes.c_void_p), ("wpk", ctypes.c_void_p), ("Cout", ctypes.c_int), ("Cin", ctypes.c_int),
                ("ks", ctypes.c_int), ("mode", ctypes.c_int)]


# split-precision conv-weight packs asked for but not launched yet; they leave in ONE launch
# (rfn_pack_conv_weights_hostdescs_bf16x3) before the next kernel of the library (rfn_hip.lib.PENDING_FLUSH)
_CONV_PACK_QUEUE = []


def pack_weight(w, flip=False, prec=None):
    """Pack a torch-layout conv weight [Cout,Cin,k,k] for the MFMA conv kernel (flip=True: data-gradient conv).
    Done per call (weights change every optimizer step).  The split-precision packs are QUEUED: the buffer is returned at
    once, its contents exist when the next kernel of the library is launched (every pack queued until then shares one
    launch: a module that knows its convolutions up front -- the extractor / upscaler -- asks for all of them first).
    prec: 'bf16x3' | 'bf16x6' | 'f32' (default: the gradient arithmetic, which is what un-annotated callers are)."""
    Cout, Cin, ks = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
    prec = prec if prec is not None else ("bf16x3" if bwd_b3() else "f32")
    sfx = {"bf16x3": "_bf16x3", "bf16x6": "_bf16x6", "f32": ""}[prec]
    size = getattr(L.load(), "rfn_packed_weight_size" + sfx)(Cout, Cin, ks)
    wpk = torch.empty(size, device=w.device, dtype=torch.float32)
    wc = w.detach().contiguous()
    if prec == "f32":
        L.call("rfn_pack_conv_weight_f32", L.dev(wc, "w"), L.dev(wpk), _i(Cout), _i(Cin), _i(ks), _i(1 if flip else 0))
        return wpk
    L.dev(wc, "w")
    _CONV_PACK_QUEUE.append(((wc.data_ptr(), wpk.data_ptr(), Cout, Cin, ks, (1 if flip else 0) + (4 if prec == "bf16x6" else 0)),
                             wc, wpk))
    L.PENDING_FLUSH = flush_packs
    return wpk


def flush_packs():
    """launch every queued weight pack (conv packs and small-map dense packs), one launch per kind and 64 matrices"""
    if _CONV_PACK_QUEUE:
        q = list(_CONV_PACK_QUEUE)
        del _CONV_PACK_QUEUE[:]
        arr = (_ConvPackDesc * len(q))(*[_ConvPackDesc(*f) for f, _, _ in q])
        L.call("rfn_pack_conv_weights_hostdescs_bf16x3", ctypes.cast(arr, ctypes.c_void_p), _i(len(q)),
               meta=_shell("rfn_pack_conv_weights_batched_bf16x3", q[0][2], sum(b.numel() for _, _, b in q) / max(q[0][2].numel(), 1)))
    smallmap_pack_flush()


def conv2d_dgrad_act(gin, wpk_flip, y, logs, act, Cout, ks, arena=None):
    """data-gradient conv + backward through the producer's ActNorm/activation in one kernel
    (rfn_conv2d_dgrad_act_bf16x3).  Returns (gu, grad_bias[Cout], grad_logs[Cout]); the two per-channel sums are
    accumulated by the kernel (float atomics) into a zeroed [2, Cout] slice of `arena`."""
    N, Cin, H, W = gin.shape
    gp, gns = L.frames(gin, "gin")
    yp, yns = L.frames(y, "y")
    gu = torch.empty((N, Cout, H, W), device=gin.device, dtype=torch.float32)
    up, uns = L.frames(gu, "gu")
    sums = _zeros(arena, 2, Cout, device=gin.device)
    L.call("rfn_conv2d_dgrad_act_bf16x3", gp, _l(gns), _i(Cin), L.dev(wpk_flip), yp, _l(yns), L.dev(logs), _i(act), up,
           _l(uns), L.dev(sums), _i(Cout), _i(N), _i(H), _i(W), _i(ks),
           meta=("conv", conv_b3_kernel_name(Cout, ks, N * H * W, Cin, (H, W), True, True) + "+actbwd", 2.0 * N * H * W * Cin * Cout * ks * ks,
                 "N%d %d->%d %dx%d k%d dgrad+actbwd" % (N, Cin, Cout, H, W, ks),
                 4.0 * (N * H * W * (Cin + 2 * Cout) + Cin * Cout * ks * ks)))
    return gu, sums[0], sums[1]


class PackPlan:
    """Persistent packed-weight buffers for a list of (weight, mode) and ONE launch that refreshes all of them
    (rfn_pack_conv_weights_batched_bf16x3).  mode: 0 forward, 1 data-gradient, 2 tap-expanded 1x1 (tiny-Cout 3x3);
    + 4: three planes per operand (bf16x6)."""

    def __init__(self, items):
        import numpy as np
        self.items = list(items)
        self.ptrs = [w.data_ptr() for w, _ in self.items]
        dev = self.items[0][0].device
        lib = L.load()
        self.bufs = []
        rec = np.zeros(len(self.items), dtype=np.dtype([("w", "<u8"), ("wpk", "<u8"), ("Cout", "<i4"), ("Cin", "<i4"),
                                                          ("ks", "<i4"), ("mode", "<i4")]))
        for i, (w, mode) in enumerate(self.items):
            Cout, Cin, ks = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
            assert w.is_contiguous() and w.dtype == torch.float32
            lc, lk = (9 * Cout, 1) if (mode & 3) == 2 else (Cout, ks)
            size_fn = lib.rfn_packed_weight_size_bf16x6 if (mode & 4) else lib.rfn_packed_weight_size_bf16x3
            buf = torch.empty(size_fn(lc, Cin, lk), device=dev, dtype=torch.float32)
            self.bufs.append(buf)
            rec[i] = (w.data_ptr(), buf.data_ptr(), Cout, Cin, ks, mode)
        self.table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)

    def valid_for(self, items):
        return (len(items) == len(self.items) and all(w.data_ptr() == p for (w, _), p in zip(items, self.ptrs))
                and all(m == m0 for (_, m), (_, m0) in zip(items, self.items)))

    def run(self):
        L.call("rfn_pack_conv_weights_batched_bf16x3", L._c_f(self.table.data_ptr()), _i(len(self.items)))


# ------------------------------------------------------------------------------------------------ fused coupling net
def coupling_po_ok(N, C, Cc, Hd, H, W, w1, w3, any_size=False):
    """shapes the fused forward kernel (csrc/coupling_po.hip) takes; both 3x3 convs must really be 3x3.
    Policy on top of the capability (skipped with any_size): the wide instantiations (more than 40 input channels: level
    2 of the canonical flow) take ~65 us per 128-pixel round -- with fewer rounds than half the CUs (a local batch of 4
    or 8) the three unfused launches, which spread the same work over the whole chip, are faster."""
    if os.environ.get("RFN_COUPLING_PO") == "0" or CONV_PRECISION != "mixed":
        return False
    if tuple(w1.shape[2:]) != (3, 3) or tuple(w3.shape[2:]) != (3, 3):
        return False
    if not any_size and C // 2 + Cc > 40 and N * H * W < 128 * 128:
        return False
    return bool(L.load().rfn_coupling_po_supported(int(N), int(C), int(Cc), int(Hd), int(H), int(W)))


class POPackPlan:
    """Persistent fragment-ordered weight streams of a list of coupling nets [(w1, w2, w3)] and ONE launch that
    refreshes all of them (rfn_coupling_po_pack): weights change every optimizer step."""

    def __init__(self, nets):
        import numpy as np
        self.nets = [tuple(n) for n in nets]
        self.ptrs = [tuple(w.data_ptr() for w in n) for n in self.nets]
        dev = self.nets[0][0].device
        lib = L.load()
        rec = np.zeros(len(self.nets), dtype=np.dtype([("w1", "<u8"), ("w2", "<u8"), ("w3", "<u8"), ("dst", "<u8"),
                                                       ("Cin", "<i4"), ("C", "<i4")]))
        self.bufs = []
        for i, (w1, w2, w3) in enumerate(self.nets):
            Cin, C = int(w1.shape[1]), int(w3.shape[0])
            for w in (w1, w2, w3):
                assert w.is_contiguous() and w.dtype == torch.float32
            assert tuple(w2.shape) == (256, 256, 1, 1) and int(w1.shape[0]) == 256 and int(w3.shape[1]) == 256
            buf = torch.empty(int(lib.rfn_coupling_po_packed_bytes(Cin, C)) // 4, device=dev, dtype=torch.float32)
            self.bufs.append(buf)
            rec[i] = (w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), buf.data_ptr(), Cin, C)
        self.table = torch.from_numpy(rec.view(np.uint8).copy()).to(dev)

        # the backward kernel's streams (w3 transposed + mirrored, w2 transposed) for the nets whose gradient image has
        # at most 16 channels (two 8-channel groups: rfn_coupling_po_bwd_supported)
        self.bwd_bufs = [None] * len(self.nets)
        recb = []
        for i, (w1, w2, w3) in enumerate(self.nets):
            C = int(w3.shape[0])
            if C <= 16:
                buf = torch.empty(int(lib.rfn_coupling_po_bwd_packed_bytes(C)) // 4, device=dev, dtype=torch.float32)
                self.bwd_bufs[i] = buf
                recb.append((w1.data_ptr(), w2.data_ptr(), w3.data_ptr(), buf.data_ptr(), int(w1.shape[1]), C))
        self.n_bwd = len(recb)
        if recb:
            rb = np.zeros(len(recb), dtype=rec.dtype)
            for i, r in enumerate(recb):
                rb[i] = r
            self.table_bwd = torch.from_numpy(rb.view(np.uint8).copy()).to(dev)

    def valid_for(self, nets):
        return len(nets) == len(self.nets) and all(tuple(w.data_ptr() for w in n) == p for n, p in zip(nets, self.ptrs))

    def run(self, bwd=True):
        """`bwd`: also refresh the backward streams (not needed when no gradient will be asked for)"""
        L.call("rfn_coupling_po_pack", L._c_f(self.table.data_ptr()), _i(len(self.nets)))
        if bwd and self.n_bwd:
            L.call("rfn_coupling_po_pack_bwd", L._c_f(self.table_bwd.data_ptr()), _i(self.n_bwd))


def coupling_po_fwd(z, cond, wpk, n1b, n1l, n2b, n2l, C, act, want_masks=False):
    """h1, h2, P, masks = fused coupling net on z[:, :C/2] | cond (rfn_coupling_po_fwd); P is the tap-expanded conv3
    output; masks = (m1, m2): the 1-bit "h <= 0" masks of h1 / h2 in the backward kernel's order (None unless asked)."""
    N, _, H, W = z.shape
    Cc = 0 if cond is None else int(cond.shape[1])
    zp, zns = L.frames(z, "z")
    cp, cns = (None, 0) if cond is None else L.frames(cond, "cond")
    h1 = torch.empty((N, 256, H, W), device=z.device, dtype=torch.float32)
    h2 = torch.empty((N, 256, H, W), device=z.device, dtype=torch.float32)
    P = torch.empty((N, 9 * C, H, W), device=z.device, dtype=torch.float32)
    Cin = C // 2 + Cc
    npx = float(N * H * W)
    masks = None
    if want_masks and act != 0:
        nm = int(L.load().rfn_coupling_po_mask_floats(N, H, W))
        masks = (torch.empty(nm, device=z.device, dtype=torch.float32), torch.empty(nm, device=z.device, dtype=torch.float32))
    m1, m2 = masks if masks is not None else (None, None)
    L.call("rfn_coupling_po_fwd", zp, _l(zns), cp, _l(cns), L.dev(wpk), L.dev(n1b), L.dev(n1l), L.dev(n2b), L.dev(n2l),
           L.dev(h1), _l(256 * H * W), L.dev(h2), _l(256 * H * W), L.dev(P), _l(9 * C * H * W), L.dev(m1), L.dev(m2),
           _i(N), _i(C), _i(Cc), _i(H), _i(W), _i(act),
           meta=("conv", "coupling_po_fwd_kernel", 2.0 * npx * (9 * Cin * 256 + 256 * 256 + 9 * C * 256),
                 "N%d %d+%d->256->256->9x%d %dx%d" % (N, C // 2, Cc, C, H, W),
                 4.0 * npx * (Cin + 512 + 9 * C + (16 if masks is not None else 0))
                 + 4.0 * (9 * Cin * 256 + 65536 + 9 * C * 256)))
    return h1, h2, P, masks


def coupling_po_bwd_ok(N, C, H, W):
    return (os.environ.get("RFN_COUPLING_PO_BWD") != "0" and CONV_PRECISION == "mixed"
            and bool(L.load().rfn_coupling_po_bwd_supported(int(N), int(C), int(H), int(W))))


def coupling_po_bwd(go, wpk_bwd, n1l, n2l, masks, act):
    """ga2, ga1, part = fused data-gradient chain of the coupling net from `go` (gradient at conv3's output):
    rfn_coupling_po_bwd.  ga2 / ga1 [N,256,H,W]: gradients at the outputs of conv2 / conv1; part: per-workgroup sums."""
    N, C, H, W = go.shape
    gp, gns = L.frames(go, "go")
    ga2 = torch.empty((N, 256, H, W), device=go.device, dtype=torch.float32)
    ga1 = torch.empty((N, 256, H, W), device=go.device, dtype=torch.float32)
    part = torch.empty(int(L.load().rfn_coupling_po_bwd_part_floats(N, H, W)), device=go.device, dtype=torch.float32)
    m1, m2 = masks if masks is not None else (None, None)
    npx = float(N * H * W)
    L.call("rfn_coupling_po_bwd", gp, _l(gns), L.dev(wpk_bwd), L.dev(n1l), L.dev(n2l), L.dev(m1), L.dev(m2),
           L.dev(ga2), _l(256 * H * W), L.dev(ga1), _l(256 * H * W), L.dev(part), _i(N), _i(C), _i(H), _i(W), _i(act),
           meta=("conv", "coupling_po_bwd_kernel", 2.0 * npx * (9 * C * 256 + 256 * 256),
                 "N%d %d->256->256 %dx%d dgrad chain" % (N, C, H, W),
                 4.0 * npx * (C + 512 + (16 if masks is not None else 0)) + 4.0 * (9 * C * 256 + 65536)))
    return ga2, ga1, part


def coupling_po_bwd_finish(tickets):
    """ActNorm gradients of the hidden layers of up to 16 nets per launch (rfn_coupling_po_bwd_finish); a ticket is
    (part, w1, gw1, n1b, w2, gw2, n2b, out[4,256]); `out` is written."""
    for i0 in range(0, len(tickets), 16):
        tk = tickets[i0:i0 + 16]
        cols = [L.ptr_array([t[j].detach() for t in tk], "finish") for j in range(8)]
        nblk = int(tk[0][0].numel()) // 512
        K1 = int(tk[0][1].numel()) // 256
        for t in tk:
            assert t[0].numel() == nblk * 512 and t[1].numel() == K1 * 256 and t[2].numel() == K1 * 256
            assert t[4].numel() == 65536 and t[5].numel() == 65536 and t[2].is_contiguous() and t[5].is_contiguous()
        L.call("rfn_coupling_po_bwd_finish", *cols, _i(len(tk)), _i(nblk), _i(K1),
               meta=_shell("po_bwd_finish", tk[0][7], len(tk) * (nblk * 512 + 512 * K1 + 131072 + 1024) / 1024.0))


def conv2d_raw(in1, in2, wpk, Cout, ks, ep_mode=0, p0=None, p1=None, act=0, out1=None, out2=None, cout_split=None,
               acc1=False, acc2=False, prec=None):
    """out = epilogue(conv(cat(in1,in2))) ; see rfn_conv2d_fwd_f32.  prec: arithmetic `wpk` was packed for ('bf16x3' |
    'f32'; default: the gradient arithmetic)."""
    N, C1, H, W = in1.shape
    C2 = 0 if in2 is None else int(in2.shape[1])
    i1p, i1ns = L.frames(in1, "in1")
    i2p, i2ns = (None, 0) if in2 is None else L.frames(in2, "in2")
    if cout_split is None:
        cout_split = Cout
    if out1 is None:
        out1 = torch.empty((N, cout_split, H, W), device=in1.device, dtype=torch.float32)
    o1p, o1ns = L.frames(out1, "out1")
    o2p, o2ns = (None, 0) if out2 is None else L.frames(out2, "out2")
    prec = prec if prec is not None else ("bf16x3" if bwd_b3() else "f32")
    b3 = prec == "bf16x3"
    fn = {"bf16x3": "rfn_conv2d_fwd_bf16x3", "bf16x6": "rfn_conv2d_fwd_bf16x6", "f32": "rfn_conv2d_fwd_f32"}[prec]
    kname = (_fwd_b3_name(Cout, ks, N, H, W, C1, C2, cout_split, acc1, acc2, ep_mode) if b3 else
             conv_b3_kernel_name(Cout, ks, N * H * W, None, None, False) + " x6" if prec == "bf16x6" else
             conv_kernel_name(Cout, ks, N * H * W))
    L.call(fn, i1p, _l(i1ns), _i(C1), i2p, _l(i2ns), _i(C2),
           L.dev(wpk), o1p, _l(o1ns), o2p,
           _l(o2ns), _i(Cout), _i(cout_split), _i(1 if acc1 else 0), _i(1 if acc2 else 0), _i(N), _i(H), _i(W), _i(ks),
           _i(ep_mode), L.dev(p0), L.dev(p1), _i(act),
           meta=("conv", kname,
                 2.0 * N * H * W * (C1 + C2) * Cout * ks * ks,
                 "N%d %d+%d->%d %dx%d k%d ep%d%s" % (N, C1, C2, Cout, H, W, ks, ep_mode,
                                                  "" if cout_split == Cout else " split"),
                 4.0 * (N * H * W * (C1 + C2 + Cout) + (C1 + C2) * Cout * ks * ks)))
    return out1


def _fwd_b3_name(Cout, ks, N, H, W, C1, C2, cout_split, acc1, acc2, ep_mode):
    plain = cout_split == Cout and not acc1 and not acc2 and ep_mode <= 3
    if ks == 1:
        return conv_b3_kernel_name(Cout, ks, N * H * W, (C1 + C2) if C2 == 0 else None)
    return conv_b3_kernel_name(Cout, ks, N * H * W, C1 + C2, (H, W), plain, False)


def conv_b3_kernel_name(Cout, ks, npix=1 << 30, Cin=None, hw=None, plain=True, actbwd=False):
    """the template instantiation rfn_conv2d_fwd_bf16x3 dispatches to (mirrors csrc/conv_bf16x3.hip) -- for profiling
    labels.  Cin = total input channels when the input is ONE tensor (else None); hw = (H, W); plain = single output
    tensor, no accumulate."""
    few = npix * ((Cout + 127) // 128) < 256 * 128
    ws_on = os.environ.get("RFN_CONV_WS") != "0"
    if ks == 3 and ws_on and hw is not None and plain and Cout % 256 == 0 and npix >= 64 * 256:
        H, W = hw
        if H & (H - 1) == 0 and W & (W - 1) == 0 and W >= 8 and H * W >= 64:
            cin = Cin if isinstance(Cin, int) else None
            if actbwd and cin is not None and cin <= 8:
                return "conv3x3_ws_kernel<1,2,0>"
            if not actbwd and Cin is not None and Cin <= 40:
                return "conv3x3_ws_kernel<%s>" % ("3,2,1" if Cin <= 24 else "5,1,1")
    if ks == 3:
        cfg = "1,4,1,1" if Cout <= 32 else ("2,2,1,1" if few else "2,2,1,2")
        return "conv_b3_kernel<3,%s,16>" % cfg
    if ks == 1 and Cin is not None and Cout % 256 == 0 and 128 < Cin <= 256 and npix >= 64 * 256 and os.environ.get("RFN_CONV_WS") != "0":
        return "conv1x1_ws_kernel<16>"
    cfg = "1,4,1,1" if Cout <= 32 else ("2,2,1,1" if (few or Cout <= 64) else ("2,2,2,2" if Cout <= 128 else "4,1,2,2"))
    return "conv_b3_kernel<1,%s,32>" % cfg


def conv_kernel_name(Cout, ks, npix=1 << 30):
    """the template instantiation rfn_conv2d_fwd_f32 dispatches to (mirrors csrc/conv.hip) — for profiling labels"""
    few = Cout > 64 and npix * ((Cout + 127) // 128) < 256 * 128
    cfg = "1,4,1,2" if Cout <= 32 else ("1,4,2,1" if Cout <= 64 else ("4,1,1,1" if few else
                                                                     ("4,1,2,2" if ks == 1 else "2,2,2,2")))
    return "conv_mfma_kernel<%d,%s,%d>" % (ks, cfg, 8 if ks == 3 else 32)


def wgrad_kernel_name(Cout, Cin, ks, HW):
    if ks == 3:
        cfg = "4,1,1,1" if Cin <= 32 else ("1,4,1,1" if Cout <= 32 else "2,2,1,1")
    else:
        cfg = "4,1,2,1" if Cin <= 32 else ("1,4,1,2" if Cout <= 32 else "2,2,2,2")
    if ks == 1 and 32 < Cout <= 64 and Cin > 32:
        cfg = "1,4,2,2"
    return "wgrad_mfma_kernel<%d,%s,64>" % (ks, cfg)


def _gemm_wgrad_name(M, Nc, total, HW=0, ans=0, bns=0, G=1):
    """mirror of the kernel / tile choice in rfn_gemm_wgrad_bf16x3 (csrc/wgrad_bf16x3.hip), for profiling labels only"""
    if (os.environ.get("RFN_WGRAD_DMA", "1") != "0" and not os.environ.get("RFN_WGRAD_VARIANT") and total * G >= 100000
            and total >= 2048 and HW % 32 == 0 and ans % 4 == 0 and bns % 4 == 0 and Nc > 128 and Nc % 256 == 0
            and (M >= 192 or M <= 64)):
        return "gemm_wgrad_dma_kernel<%s>" % ("2,4,4,2,32,2" if M > 128 else "1,8,2,1,32,3")
    if M > 128 and Nc > 128 and total >= 100000:
        cfg = "4,2,2,3,64" if -(-Nc // 192) * 192 < -(-Nc // 256) * 256 else "2,4,4,2,64"
    else:
        cfg = "1,4,2,2,32" if M <= 64 else ("4,1,2,2,32" if Nc <= 64 else "2,2,2,2,64")
    return "gemm_wgrad_b3_kernel<%s>" % cfg


def gemm_wgrad(a, b, M, Nc, arena=None):
    """gw[M][Nc] = Σ_{frames,pixels} a[f,m,p] b[f,n,p] on the split-precision MFMA GEMM (rfn_gemm_wgrad_bf16x3)."""
    F_, HW = int(a.shape[0]), _hw(a)
    ap, ans = L.frames(a, "a")
    bp, bns = L.frames(b, "b")
    gw = _zeros(arena, M, Nc, device=a.device)
    L.call("rfn_gemm_wgrad_bf16x3", ap, _l(ans), _i(M), bp, _l(bns), _i(Nc), L.dev(gw), _i(F_), _i(HW),
           meta=("wgrad", _gemm_wgrad_name(M, Nc, F_ * HW, HW, ans, bns),
                 2.0 * F_ * HW * M * Nc, "F%d %dx%d HW%d" % (F_, M, Nc, HW), 4.0 * (F_ * HW * (M + Nc) + M * Nc)))
    return gw


def conv2d_wgrad_b3(in1, in2, g, Cout, ks, arena=None):
    """weight gradient through the split-precision GEMM: 3x3 convs expand their smaller operand in HBM first
    (im2col of the input when Cin <= Cout, tap-scatter of the gradient otherwise)."""
    N, C1, H, W = in1.shape
    C2 = 0 if in2 is None else int(in2.shape[1])
    Cin = C1 + C2
    if ks == 1:
        x = in1 if in2 is None else torch.cat((in1, in2), 1)
        return gemm_wgrad(g, x, Cout, Cin, arena).view(Cout, Cin, 1, 1)
    if Cin <= Cout and W % 8 == 0 and os.environ.get("RFN_WGRAD_IMPLICIT") != "0":
        # shifted input planes are built while staging: no im2col buffer (rfn_conv3x3_wgrad_implicit_bf16x3)
        i1p, i1ns = L.frames(in1, "in1")
        i2p, i2ns = (None, 0) if in2 is None else L.frames(in2, "in2")
        gp, gns = L.frames(g, "g")
        gw = _zeros(arena, Cout, 9 * Cin, device=in1.device)
        big = Cout > 128 and N * H * W >= 100000
        dma = (big and (H * W) % 32 == 0 and gns % 4 == 0 and os.environ.get("RFN_WGRAD_DMA", "1") != "0")
        L.call("rfn_conv3x3_wgrad_implicit_bf16x3", gp, _l(gns), _i(Cout), i1p, _l(i1ns), _i(C1), i2p, _l(i2ns), _i(C2),
               L.dev(gw), _i(N), _i(H), _i(W),
               meta=("wgrad", "gemm_wgrad_dma_impl_kernel<4,2,2,3>" if dma else
                     "gemm_wgrad_b3_kernel<%s,1>" % ("4,2,2,3,64" if big else ("1,4,1,2,32" if Cout <= 32 else "2,2,2,2,64")),
                     2.0 * N * H * W * Cout * 9 * Cin, "F%d %dx%d HW%d implicit3x3" % (N, Cout, 9 * Cin, H * W),
                     4.0 * (N * H * W * (Cout + Cin) + Cout * 9 * Cin)))
        return gw.view(Cout, Cin, 3, 3)  # rows of the implicit operand are (ci, tap): already the torch layout
    if Cin <= Cout:
        i1p, i1ns = L.frames(in1, "in1")
        i2p, i2ns = (None, 0) if in2 is None else L.frames(in2, "in2")
        x9 = torch.empty((N, 9 * Cin, H, W), device=in1.device, dtype=torch.float32)
        L.call("rfn_im2col3x3_f32", i1p, _l(i1ns), _i(C1), i2p, _l(i2ns), _i(C2), L.dev(x9), _i(N), _i(H), _i(W),
               meta=_shell("im2col3x3", x9, 10.0 / 9.0))
        gw = gemm_wgrad(g, x9, Cout, 9 * Cin, arena)  # [co][tap*Cin + ci]
        return gw.view(Cout, 9, Cin).permute(0, 2, 1).reshape(Cout, Cin, 3, 3)
    x = in1 if in2 is None else torch.cat((in1, in2), 1)
    gs = torch.empty((N, 9 * Cout, H, W), device=in1.device, dtype=torch.float32)
    L.call("rfn_tap_scatter_f32", L.dev(g.contiguous()), L.dev(gs), _i(N), _i(Cout), _i(H), _i(W),
           meta=_shell("tap_scatter", gs, 10.0 / 9.0))
    gw = gemm_wgrad(gs, x, 9 * Cout, Cin, arena)  # [tap*Cout + co][ci]
    return gw.view(3, 3, Cout, Cin).permute(2, 3, 0, 1).contiguous()


def gemm_wgrad_grouped(a_list, b_list, M, Nc, arena=None):
    """G gradients gw[g][M][Nc] = Σ a_g b_g^T of one shape in ONE launch (rfn_gemm_wgrad_grouped_bf16x3)."""
    G = len(a_list)
    F_, HW = int(a_list[0].shape[0]), _hw(a_list[0])
    ans, bns = L.frames(a_list[0], "a")[1], L.frames(b_list[0], "b")[1]
    gw = _zeros(arena, G, M, Nc, device=a_list[0].device)
    pa, pb = L.ptr_array(a_list, "a"), L.ptr_array(b_list, "b")
    pg = L.ptr_array([gw[g] for g in range(G)], "gw")
    L.call("rfn_gemm_wgrad_grouped_bf16x3", pa, _l(ans), _i(M), pb, _l(bns), _i(Nc), pg, _i(G), _i(F_), _i(HW),
           meta=("wgrad", _gemm_wgrad_name(M, Nc, F_ * HW, HW, ans, bns, G).replace("<", "<grouped "),
                 2.0 * G * F_ * HW * M * Nc, "G%d F%d %dx%d HW%d" % (G, F_, M, Nc, HW),
                 4.0 * G * (F_ * HW * (M + Nc) + M * Nc)))
    return gw


GROUPED_WGRAD_MAX_PIX = 100000  # below this many pixels a single weight gradient is a latency-class launch


def conv2d_wgrad_grouped(in1_list, in2_list, g_list, Cout, ks, arena=None, g_stacked=None):
    """the weight gradients of G convolutions of ONE shape (the K steps of a flow level) in one GEMM launch: list of G
    tensors [Cout, Cin, ks, ks].  Same operand choices as conv2d_wgrad_b3 (the small operand of a 3x3 gradient is
    expanded per step -- in one launch when the gradients are the slices of `g_stacked` -- and the layout fix-up is one
    copy for all groups)."""
    G = len(g_list)
    N, C1, H, W = in1_list[0].shape
    has2 = in2_list is not None and in2_list[0] is not None
    C2 = int(in2_list[0].shape[1]) if has2 else 0
    Cin = C1 + C2
    dev_ = in1_list[0].device
    if ks == 1:
        xs = in1_list if not has2 else [torch.cat((a, b), 1) for a, b in zip(in1_list, in2_list)]
        gw = gemm_wgrad_grouped(g_list, xs, Cout, Cin, arena)
        return [gw[i].view(Cout, Cin, 1, 1) for i in range(G)]
    if Cin <= Cout and W % 8 == 0 and os.environ.get("RFN_WGRAD_IMPLICIT") != "0":
        gw = _zeros(arena, G, Cout, 9 * Cin, device=dev_)
        gns, i1ns = L.frames(g_list[0], "g")[1], L.frames(in1_list[0], "in1")[1]
        i2ns = L.frames(in2_list[0], "in2")[1] if has2 else 0
        pg, p1 = L.ptr_array(g_list, "g"), L.ptr_array(in1_list, "in1")
        p2 = L.ptr_array(in2_list, "in2") if has2 else None
        pw = L.ptr_array([gw[i] for i in range(G)], "gw")
        L.call("rfn_conv3x3_wgrad_implicit_grouped_bf16x3", pg, _l(gns), _i(Cout), p1, _l(i1ns), _i(C1), p2, _l(i2ns),
               _i(C2), pw, _i(G), _i(N), _i(H), _i(W),
               meta=("wgrad", "gemm_wgrad_dma_impl_kernel<grouped 4,2,2,3>" if (
                         Cout > 128 and G * N * H * W >= 100000 and N * H * W >= 2048 and (H * W) % 32 == 0 and gns % 4 == 0
                         and os.environ.get("RFN_WGRAD_DMA", "1") != "0") else "gemm_wgrad_b3_kernel<grouped implicit>",
                     2.0 * G * N * H * W * Cout * 9 * Cin,
                     "G%d F%d %dx%d HW%d implicit3x3" % (G, N, Cout, 9 * Cin, H * W),
                     4.0 * G * (N * H * W * (Cout + Cin) + Cout * 9 * Cin)))
        return [gw[i].view(Cout, Cin, 3, 3) for i in range(G)]
    if Cin <= Cout:
        x9s = []
        for i in range(G):
            i1p, i1ns = L.frames(in1_list[i], "in1")
            i2p, i2ns = (None, 0) if not has2 else L.frames(in2_list[i], "in2")
            x9 = torch.empty((N, 9 * Cin, H, W), device=dev_, dtype=torch.float32)
            L.call("rfn_im2col3x3_f32", i1p, _l(i1ns), _i(C1), i2p, _l(i2ns), _i(C2), L.dev(x9), _i(N), _i(H), _i(W),
                   meta=_shell("im2col3x3", x9, 10.0 / 9.0))
            x9s.append(x9)
        gw = gemm_wgrad_grouped(g_list, x9s, Cout, 9 * Cin, arena)  # [g][co][tap*Cin + ci]
        gwt = gw.view(G, Cout, 9, Cin).permute(0, 1, 3, 2).reshape(G, Cout, Cin, 3, 3)  # one copy for all groups
        return [gwt[i] for i in range(G)]
    xs = in1_list if not has2 else [torch.cat((a, b), 1) for a, b in zip(in1_list, in2_list)]
    gss = []
    if g_stacked is not None and g_stacked.is_contiguous() and tuple(g_stacked.shape) == (G, N, Cout, H, W):
        # the G gradients are slices of one [G, N, Cout, H, W] buffer (in list order): one scatter launch over G*N frames
        gs_all = torch.empty((G, N, 9 * Cout, H, W), device=dev_, dtype=torch.float32)
        L.call("rfn_tap_scatter_f32", L.dev(g_stacked), L.dev(gs_all), _i(G * N), _i(Cout), _i(H), _i(W),
               meta=_shell("tap_scatter", gs_all, 10.0 / 9.0))
        gss = [gs_all[i] for i in range(G)]
    else:
        for i in range(G):
            gs = torch.empty((N, 9 * Cout, H, W), device=dev_, dtype=torch.float32)
            gi = g_list[i].contiguous()
            L.call("rfn_tap_scatter_f32", L.dev(gi), L.dev(gs), _i(N), _i(Cout), _i(H), _i(W),
                   meta=_shell("tap_scatter", gs, 10.0 / 9.0))
            gss.append(gs)
    gw = gemm_wgrad_grouped(gss, xs, 9 * Cout, Cin, arena)  # [g][tap*Cout + co][ci]
    gwt = gw.view(G, 3, 3, Cout, Cin).permute(0, 3, 4, 1, 2).contiguous()
    return [gwt[i] for i in range(G)]


def conv2d_wgrad(in1, in2, g, Cout, ks, arena=None):
    """returns gw [Cout, Cin, ks, ks]"""
    N, C1, H, W = in1.shape
    C2 = 0 if in2 is None else int(in2.shape[1])
    Cin = C1 + C2
    if bwd_b3() and (H * W) % 4 == 0:
        return conv2d_wgrad_b3(in1, in2, g, Cout, ks, arena)
    i1p, i1ns = L.frames(in1, "in1")
    i2p, i2ns = (None, 0) if in2 is None else L.frames(in2, "in2")
    gp, gns = L.frames(g, "g")
    gwt = _zeros(arena, ks * ks, Cout, Cin, device=in1.device)
    L.call("rfn_conv2d_wgrad_f32", i1p, _l(i1ns), _i(C1), i2p, _l(i2ns), _i(C2), gp, _l(gns), _i(Cout), L.dev(gwt),
           _i(N), _i(H), _i(W), _i(ks),
           meta=("wgrad", wgrad_kernel_name(Cout, Cin, ks, H * W), 2.0 * N * H * W * Cin * Cout * ks * ks,
                 "N%d %d->%d %dx%d k%d" % (N, Cin, Cout, H, W, ks), 4.0 * (N * H * W * (Cin + Cout) + Cin * Cout * ks * ks)))
    if ks == 1:
        return gwt.view(Cout, Cin, 1, 1)  # tap-major == torch layout when there is a single tap
    gw = torch.empty((Cout, Cin, ks, ks), device=in1.device, dtype=torch.float32)
    L.call("rfn_wgrad_finish_f32", L.dev(gwt), L.dev(gw), _i(Cout), _i(Cin), _i(ks), _i(0), meta=_shell("wgrad_finish", gw, 2))
    return gw


TAP_MAX_COUT = 8  # 3x3 convs with at most this many outputs run tap-expanded (1x1 to 9*C channels + shift-add)


def zeros_conv_uses_taps(w):
    return int(w.shape[2]) == 3 and int(w.shape[0]) <= TAP_MAX_COUT


def zeros_conv_fwd(x, w, b, logs, wpk=None, prec=None):
    """Conv2dZeros forward (glow_modules.py:119-121): (conv3x3(x) + b) * exp(3 logs).  Tiny Cout -> tap-expanded.
    `wpk` (optional): pre-packed weight (mode 2 = tap-expanded when zeros_conv_uses_taps(w), else mode 0), packed for
    `prec` (default: this map's forward arithmetic)."""
    C, Cin, ks = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
    N, _, H, W = x.shape
    if prec is None:
        prec = fwd_prec(H, W)
    if not zeros_conv_uses_taps(w):
        return conv2d_raw(x, None, wpk if wpk is not None else pack_weight(w, prec=prec), C, ks, 2, b, logs, 0, prec=prec)
    if wpk is None:
        wpk = pack_weight(w.detach().permute(2, 3, 0, 1).reshape(9 * C, Cin, 1, 1).contiguous(), prec=prec)  # [tap*C + co][ci]
    P = conv2d_raw(x, None, wpk, 9 * C, 1, prec=prec)
    o = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
    L.call("rfn_tap_gather_f32", L.dev(P), L.dev(b), L.dev(logs), L.dev(o), _i(N), _i(C), _i(H), _i(W),
           meta=_shell("tap_gather", P, 10.0 / 9.0))
    return o


def zeros_conv_wgrad(x, g_pre, C, ks, arena=None):
    """weight gradient of the conv inside Conv2dZeros given g_pre = grad wrt (conv + b); same switch as the forward."""
    if ks != 3 or C > TAP_MAX_COUT or (bwd_b3() and _hw(x) % 4 == 0):
        return conv2d_wgrad(x, None, g_pre, C, ks, arena)
    N, Cin, H, W = x.shape
    Gs = torch.empty((N, 9 * C, H, W), device=x.device, dtype=torch.float32)
    L.call("rfn_tap_scatter_f32", L.dev(g_pre.contiguous()), L.dev(Gs), _i(N), _i(C), _i(H), _i(W),
           meta=_shell("tap_scatter", Gs, 10.0 / 9.0))
    gw = conv2d_wgrad(x, None, Gs, 9 * C, 1, arena)  # [9C, Cin, 1, 1]
    return gw.view(3, 3, C, Cin).permute(2, 3, 0, 1).contiguous()


def conv_epilogue_bwd(y, gy, logs, ep_mode, act, want_gl=True, arena=None):
    """in-place on gy: gy <- gu ; returns (gu, gb, gl)"""
    N, C = gy.shape[0], gy.shape[1]
    yp, yns = (None, 0) if y is None else L.frames(y, "y")
    gp, gns = L.frames(gy, "gy")
    gb = _zeros(arena, C, device=gy.device)
    gl = _zeros(arena, C, device=gy.device) if want_gl else None
    L.call("rfn_conv_epilogue_bwd_f32", yp, _l(yns), gp, _l(gns), gp, _l(gns), L.dev(logs), L.dev(gb), L.dev(gl),
           _i(N), _i(C), _i(_hw(gy)), _i(ep_mode), _i(act), meta=_shell("conv_epilogue_bwd", gy, 3))
    return gy, gb, gl


def affine_coupling_(z, o, scale, scale_shift, logdet, clamp_type, reverse):
    """in place on z's second channel half; logdet [N] updated in place (may be None)."""
    N, C = z.shape[0], z.shape[1]
    zp, zns = L.frames(z, "z")
    op, ons = L.frames(o, "o")
    L.call("rfn_affine_coupling_f32", zp, _l(zns), op, _l(ons), L.dev(scale), L.dev(scale_shift), L.dev(logdet),
           _i(clamp_type), _i(1 if reverse else 0), _i(N), _i(C), _i(_hw(z)), meta=_shell("affine_coupling", z, 2))


def gather_affine_(z, o, P, b3, l3, scale, scale_shift, clamp_type):
    """fused shell tail of the forward Glow step (rfn_gather_affine_f32): with P the tap-expanded Conv2dZeros output is
    gathered, biased and scaled here (written to a fresh o); z's second channel half is coupled in place.
    Returns (o, dlogdet[N]) -- dlogdet is written by the kernel, not accumulated."""
    N, C, H, W = z.shape
    zp, zns = L.frames(z, "z")
    dlogdet = torch.empty(N, device=z.device, dtype=torch.float32)
    if P is not None:
        o = torch.empty((N, C, H, W), device=z.device, dtype=torch.float32)
        L.call("rfn_gather_affine_f32", L.dev(P), None, _l(0), L.dev(b3), L.dev(l3), L.dev(o), zp, _l(zns), L.dev(scale),
               L.dev(scale_shift), L.dev(dlogdet), _i(clamp_type), _i(N), _i(C), _i(H), _i(W),
               meta=_shell("gather_affine", z, 9 + 1 + 1))  # P read (9x), o written, z2 read + written (2 x 1/2)
    else:
        op, ons = L.frames(o, "o")
        L.call("rfn_gather_affine_f32", None, op, _l(ons), None, None, None, zp, _l(zns), L.dev(scale),
               L.dev(scale_shift), L.dev(dlogdet), _i(clamp_type), _i(N), _i(C), _i(H), _i(W),
               meta=_shell("gather_affine", z, 2))
    return o, dlogdet


def gauss_logp(z, o, layout, std_mode):
    N, Cz = z.shape[0], z.shape[1]
    zp, zns = L.frames(z, "z")
    op, ons = L.frames(o, "o")
    logp = torch.zeros(N, device=z.device, dtype=torch.float32)
    L.call("rfn_gauss_logp_f32", zp, _l(zns), op, _l(ons), L.dev(logp), _i(layout), _i(std_mode), _i(N), _i(Cz),
           _i(_hw(z)), meta=_shell("gauss_logp", z, 3))
    return logp


def gauss_sample(o, eps, layout, std_mode, temperature):
    N, C2 = o.shape[0], o.shape[1]
    Cz = C2 // 2
    z = torch.empty((N, Cz) + tuple(o.shape[2:]), device=o.device, dtype=torch.float32)
    op, ons = L.frames(o, "o")
    zp, zns = L.frames(z, "z")
    L.call("rfn_gauss_sample_f32", op, _l(ons), L.dev(eps.contiguous(), "eps"), zp, _l(zns),
           ctypes.c_float(float(temperature)), _i(layout), _i(std_mode), _i(N), _i(Cz), _i(_hw(o)))
    return z


# ----------------------------------------------------------------------------------------------- autograd Functions
class Squeeze2dFn(torch.autograd.Function):
    """Flow/glow_modules.py:298-310; backward = the opposite permutation."""

    @staticmethod
    def forward(ctx, x, undo):
        ctx.undo = undo
        return squeeze2d_raw(x, undo)

    @staticmethod
    def backward(ctx, g):
        return squeeze2d_raw(g.contiguous(), not ctx.undo), None


def fewcin_ok(in1, in2, w, ep_mode):
    """the direct fp32 kernel for 3x3 convolutions of 1 .. 4-channel images (csrc/conv.hip, rfn_conv3x3_fewcin_fwd_f32)"""
    return (in2 is None and ep_mode == 0 and tuple(w.shape[2:]) == (3, 3) and os.environ.get("RFN_FEWCIN") != "0"
            and bool(L.load().rfn_conv3x3_fewcin_supported(int(w.shape[1]), int(w.shape[0]))))


def conv3x3_fewcin(x, w):
    N, Cin, H, W = x.shape
    Cout = int(w.shape[0])
    xp, xns = L.frames(x, "x")
    out = torch.empty((N, Cout, H, W), device=x.device, dtype=torch.float32)
    wc = w.detach().contiguous()
    L.call("rfn_conv3x3_fewcin_fwd_f32", xp, _l(xns), _i(Cin), L.dev(wc), L.dev(out), _l(Cout * H * W), _i(Cout), _i(N),
           _i(H), _i(W), meta=_shell("conv3x3_fewcin_fwd", out, 1.0 + Cin / Cout))
    return out


def conv3x3_c1_wgrad16(x, g):
    N, _, H, W = x.shape
    xp, xns = L.frames(x, "x")
    gp, gns = L.frames(g, "g")
    gw = torch.zeros((16, 1, 3, 3), device=x.device, dtype=torch.float32)
    L.call("rfn_conv3x3_c1_wgrad16_f32", xp, _l(xns), gp, _l(gns), L.dev(gw), _i(N), _i(H), _i(W),
           meta=_shell("conv3x3_c1_wgrad", g, 1.0 + 1.0 / 16))
    return gw


class ConvFn(torch.autograd.Function):
    """epilogue(conv(cat(in1, in2), w)) with the epilogue of rfn_conv2d_fwd_f32.
    p0/p1: ep_mode 1 -> (actnorm bias, actnorm logs); 2 -> (conv bias, logs); 3 -> (conv bias, None)."""

    @staticmethod
    def forward(ctx, in1, in2, w, p0, p1, ep_mode, act, prec=None, packs=None):
        # packs: (forward pack, data-gradient pack or None) queued by the caller (run_time_batched), else packed here
        Cout, ks = int(w.shape[0]), int(w.shape[2])
        ctx.pk_b = None if packs is None else packs[1]
        p0f = None if p0 is None else p0.detach().reshape(-1).contiguous()
        p1f = None if p1 is None else p1.detach().reshape(-1).contiguous()
        fp = prec if prec is not None else fwd_prec(int(in1.shape[2]), int(in1.shape[3]))
        if fewcin_ok(in1, in2, w, ep_mode):
            y = conv3x3_fewcin(in1, w)   # the extractor's first convolution (1 .. 4 image channels): exact fp32 FMAs
        else:
            y = conv2d_raw(in1, in2, packs[0] if packs is not None and packs[0] is not None else pack_weight(w, prec=fp),
                           Cout, ks, ep_mode, p0f, p1f, act, prec=fp)
        ctx.save_for_backward(in1, in2, w, p1f, y)
        ctx.cfg = (ep_mode, act, None if p0 is None else p0.shape, None if p1 is None else p1.shape)
        return y

    @staticmethod
    def backward(ctx, gy):
        in1, in2, w, p1f, y = ctx.saved_tensors
        ep_mode, act, p0shape, p1shape = ctx.cfg
        Cout, Cin, ks = int(w.shape[0]), int(w.shape[1]), int(w.shape[2])
        gy = gy.contiguous().clone() if ep_mode != 0 else gy.contiguous()
        gp0 = gp1 = None
        if ep_mode != 0:
            gy, gb, gl = conv_epilogue_bwd(y if ep_mode != 3 else None, gy, p1f, ep_mode, act, ep_mode != 3)
            gp0 = gb.view(p0shape)
            gp1 = gl.view(p1shape) if gl is not None else None
        g1 = g2 = gw = None
        C1 = int(in1.shape[1])
        need1, need2 = ctx.needs_input_grad[0], in2 is not None and ctx.needs_input_grad[1]
        if need1 or need2:
            wt = ctx.pk_b if ctx.pk_b is not None else pack_weight(w, flip=True)
            N, _, H, W = in1.shape
            g1 = torch.empty(in1.shape, device=gy.device, dtype=torch.float32)
            if in2 is not None:
                g2 = torch.empty(in2.shape, device=gy.device, dtype=torch.float32)
            conv2d_raw(gy, None, wt, Cin, ks, 0, None, None, 0, out1=g1, out2=g2, cout_split=C1)
        if ctx.needs_input_grad[2]:
            if fewcin_ok(in1, in2, w, ep_mode) and Cin == 1 and Cout == 16:
                gw = conv3x3_c1_wgrad16(in1, gy)
            else:
                gw = conv2d_wgrad(in1, in2, gy, Cout, ks)
        return g1, g2, gw, gp0, gp1, None, None, None, None


def conv_ep(in1, in2, w, p0, p1, ep_mode, act, prec=None, packs=None):
    """`prec` (optional): forward arithmetic ('bf16x3' | 'bf16x6' | 'f32') instead of this map size's default;
    `packs` (optional): (forward pack in that arithmetic, data-gradient pack or None) already queued by the caller"""
    return ConvFn.apply(in1, in2, w, p0, p1, ep_mode, act, prec, packs)


def _f(t):
    return None if t is None else t.detach().reshape(-1).contiguous()


# per-step pack tuple handed to the coupling nets: [0..5] split-precision conv packs (w1 f, w1 d, w2 f, w2 d, w3 f, w3 d),
# [6] / [7] fused forward / backward streams, [8] / [9] / [10] dense small-map packs (w1 forward, w3 forward, w1 data gradient)
PACK_SLOTS = 11


def dgrad_small_ok(N, Cin, Cout, H, W, ks):
    """the dedicated few-output-channel 3x3 kernel (csrc/dgrad_small.hip) serves this data-gradient convolution"""
    return (ks == 3 and bwd_b3() and os.environ.get("RFN_DGRAD_SMALL") != "0"
            and bool(L.load().rfn_dgrad_small_supported(int(N), int(Cin), int(Cout), int(H), int(W))))


def conv3x3_smallcout(x, wpk, Cout, out1, out2=None, cout_split=None, acc1=False, acc2=False):
    """rfn_conv3x3_smallcout_bf16x3: 3x3 / pad 1 convolution of x [N, Cin, H, W] with the packed weight `wpk`
    (pack_weight(..., flip=True) for a data gradient) into out1 (channels [0, cout_split)) and out2 (the rest)."""
    N, Cin, H, W = x.shape
    if cout_split is None:
        cout_split = Cout
    xp, xns = L.frames(x, "x")
    o1p, o1ns = L.frames(out1, "out1")
    o2p, o2ns = (None, 0) if out2 is None else L.frames(out2, "out2")
    L.call("rfn_conv3x3_smallcout_bf16x3", xp, _l(xns), _i(Cin), L.dev(wpk), o1p, _l(o1ns), o2p, _l(o2ns), _i(Cout),
           _i(cout_split), _i(1 if acc1 else 0), _i(1 if acc2 else 0), _i(N), _i(H), _i(W),
           meta=("conv", "dgrad_small_kernel", 2.0 * N * H * W * Cin * Cout * 9,
                 "N%d %d->%d %dx%d k3 dgrad" % (N, Cin, Cout, H, W), 4.0 * (N * H * W * (Cin + Cout) + Cin * Cout * 9)))
    return out1


def _net_fwd(z, cond, w1, n1b, n1l, w2, n2b, n2l, w3, b3, l3, act, pk, want_masks=False):
    """coupling network of one Glow step (glow_modules.py:232-238) on z's first channel half and `cond`.
    Returns (h1, h2, o, P, masks): either o (finished Conv2dZeros output) or P (its tap-expanded pre-gather form, fused
    forward kernel) is None; masks: the fused kernel's activation masks for the fused backward kernel (or None).
    `pk`: the step's 8 pack-plan entries (see GlowStepFn.forward)."""
    N, C, H, W = z.shape
    Ch = C // 2
    Hd = int(w1.shape[0])
    z1 = z[:, :Ch]
    cin2 = cond if cond.shape[1] > 0 else None
    Cc_ = 0 if cin2 is None else int(cin2.shape[1])
    k33 = int(w1.shape[2]) == 3 and int(w3.shape[2]) == 3
    fp = fwd_prec(H, W)
    if coupling_po_ok(N, C, Cc_, Hd, H, W, w1, w3) and int(w2.shape[2]) == 1:
        # shallow levels: the whole coupling net in one kernel (csrc/coupling_po.hip), h1 / h2 written once
        po = pk[6]
        if po is None:
            plan = POPackPlan([(w1.detach(), w2.detach(), w3.detach())])
            plan.run()
            po = plan.bufs[0]
        want = want_masks and pk[7] is not None and coupling_po_bwd_ok(N, C, H, W)
        h1, h2, P, masks = coupling_po_fwd(z, cin2, po, _f(n1b), _f(n1l), _f(n2b), _f(n2l), C, act, want_masks=want)
        return h1, h2, None, P, masks
    # the two deepest levels (H*W <= 16): a launch is a few thousand pixels against megabytes of weights, the
    # 3x3 convolutions go through the dense small-map kernels (bf16x3 arithmetic: only where that is allowed)
    dense = k33 and smallmap_conv_ok(H, W, Ch, Cc_, Hd, N)
    dense3 = k33 and smallmap_conv_ok(H, W, Hd, 0, C, N) and not zeros_conv_uses_taps(w3)
    b3fwd = fp in ("bf16x3", "bf16x6")  # the caller's pack plan holds forward packs in this map's split arithmetic
    if dense:
        h1 = smallmap_conv(z1, cin2, pk[8] if pk[8] is not None else smallmap_pack(w1, H, W, False), Hd, 1, _f(n1b),
                           _f(n1l), act)
    else:
        h1 = conv2d_raw(z1, cin2, pk[0] if (pk[0] is not None and b3fwd) else pack_weight(w1, prec=fp), Hd,
                        int(w1.shape[2]), 1, _f(n1b), _f(n1l), act, prec=fp)
    h2 = conv2d_raw(h1, None, pk[2] if (pk[2] is not None and b3fwd) else pack_weight(w2, prec=fp), Hd,
                    int(w2.shape[2]), 1, _f(n2b), _f(n2l), act, prec=fp)
    if dense3:
        o = smallmap_conv(h2, None, pk[9] if pk[9] is not None else smallmap_pack(w3, H, W, False), C, 2, _f(b3), _f(l3), 0)
    else:
        o = zeros_conv_fwd(h2, w3, _f(b3), _f(l3), pk[4] if b3fwd else None, prec=fp)
    return h1, h2, o, None, None


def _net_bwd(go, out, cond, h1, h2, w1, n1l, w2, n2l, w3, act, pk, arena, gz, gcond, acc_cond, defer=None,
             n1b=None, n2b=None, masks=None, fin=None):
    """backward of the coupling network from `go` = gradient at conv3's output: returns the parameter gradients
    (gw1, gn1b, gn1l, gw2, gn2b, gn2l, gw3); the data gradient of conv1 is ADDED to gz[:, :C/2] and written (acc_cond:
    added) to gcond.  With `defer` (a dict of three lists) the weight gradients are NOT computed: their operands are
    appended to defer["w1" | "w2" | "w3"] and None is returned in their place (grouped launch by the caller).
    With `masks` (the fused forward kernel's activation masks; needs n1b, n2b and the step's backward stream pk[7]) the
    data-gradient chain conv3^T -> act' -> conv2^T -> act' is ONE kernel (rfn_coupling_po_bwd) that reads no
    activation; the four ActNorm gradients then come from the weight gradients (rfn_coupling_po_bwd_finish): a ticket
    is appended to `fin` (the caller runs coupling_po_bwd_finish once its weight gradients exist; ticket[2] / [5] = gw1 /
    gw2 are filled in by the caller when deferred) or, without `fin`, finished here."""
    N, C, H, W = out.shape
    Ch = C // 2
    Hd = int(w1.shape[0])
    Cc = int(cond.shape[1])
    k1, k2, k3 = int(w1.shape[2]), int(w2.shape[2]), int(w3.shape[2])
    dfr = defer is not None and bwd_b3() and Hd % 64 == 0 and _hw(h2) % 4 == 0
    gw3 = None if dfr else zeros_conv_wgrad(h2, go, C, k3, arena)
    fused = (pk[7] is not None and (masks is not None or act == 0) and n1b is not None and n2b is not None
             and k2 == 1 and k1 == 3 and k3 == 3 and Hd == 256 and coupling_po_bwd_ok(N, C, H, W))
    w3f = w2f = None
    if not fused:
        w3f = pk[5] if pk[5] is not None else pack_weight(w3, True)
        w2f = pk[3] if pk[3] is not None else pack_weight(w2, True)
    if fused:
        gh2, gh1, part = coupling_po_bwd(go.contiguous(), pk[7], _f(n1l), _f(n2l), masks, act)
        gw2 = None if dfr else conv2d_wgrad(h1, None, gh2, Hd, k2, arena)
        o4 = torch.empty((4, 256), device=go.device, dtype=torch.float32)
        gn1b, gn1l, gn2b, gn2l = o4[0], o4[1], o4[2], o4[3]
    elif bwd_b3() and Hd % 64 == 0:
        # data-gradient convs with the backward of the producer's ActNorm+activation fused into their epilogue
        gh2, gn2b, gn2l = conv2d_dgrad_act(go, w3f, h2, _f(n2l), act, Hd, k3, arena)
        gw2 = None if dfr else conv2d_wgrad(h1, None, gh2, Hd, k2, arena)
        gh1, gn1b, gn1l = conv2d_dgrad_act(gh2, w2f, h1, _f(n1l), act, Hd, k2, arena)
    else:
        gh2 = conv2d_raw(go, None, w3f, Hd, k3)
        # ---- actnorm2 + act bwd, conv2 (1x1) bwd
        gh2, gn2b, gn2l = conv_epilogue_bwd(h2, gh2, _f(n2l), 1, act, arena=arena)
        gw2 = conv2d_wgrad(h1, None, gh2, Hd, k2, arena)
        gh1 = conv2d_raw(gh2, None, w2f, Hd, k2)
        # ---- actnorm1 + act bwd, conv1 bwd (grad flows to z1 (accumulated into gz's first half) and to cond)
        gh1, gn1b, gn1l = conv_epilogue_bwd(h1, gh1, _f(n1l), 1, act, arena=arena)
    z1 = out[:, :Ch]
    has_cond = Cc > 0
    gw1 = None if dfr else conv2d_wgrad(z1, cond if has_cond else None, gh1, Hd, k1, arena)
    if fused:
        ticket = [part, w1, gw1, _f(n1b), w2, gw2, _f(n2b), o4]
        if fin is not None:
            fin.append(ticket)
        else:
            assert not dfr
            coupling_po_bwd_finish([ticket])
    if dfr:
        defer["w3"].append((h2, go))
        defer["w2"].append((h1, gh2))
        defer["w1"].append((z1, cond if has_cond else None, gh1))
    if k1 == 3 and k3 == 3 and smallmap_conv_ok(H, W, Hd, 0, Ch + Cc, N, bwd=True):
        smallmap_conv(gh1, None, pk[10] if pk[10] is not None else smallmap_pack(w1, H, W, True), Ch + Cc, 0, out1=gz[:, :Ch],
                      out2=gcond if has_cond else None, cout_split=Ch, acc1=True, acc2=acc_cond)
    elif dgrad_small_ok(N, Hd, Ch + Cc, H, W, k1):
        conv3x3_smallcout(gh1, pk[1] if pk[1] is not None else pack_weight(w1, True), Ch + Cc, gz[:, :Ch],
                          gcond if has_cond else None, Ch, True, acc_cond)
    else:
        conv2d_raw(gh1, None, pk[1] if pk[1] is not None else pack_weight(w1, True), Ch + Cc, k1, 0,
                   None, None, 0, out1=gz[:, :Ch], out2=gcond if has_cond else None, cout_split=Ch, acc1=True,
                   acc2=acc_cond)
    return gw1, gn1b, gn1l, gw2, gn2b, gn2l, gw3


def _net_arena_numel(C, Cc, Hd, k1, k2, k3):
    Ch = C // 2
    return (2 * Ch + 8 + 3 * 2 * (Hd + 4) + 2 * (C + 4) + k1 * k1 * Hd * (Ch + Cc) + k2 * k2 * Hd * Hd
            + max(k3 * k3 * C * Hd, 9 * C * Hd) + C * C + 2 * C + 64)


def _affine_zeros_bwd(out, o, gout, gdl, scale, scale_shift, l3, clamp_type, arena, go=None):
    """rfn_affine_zeros_bwd_f32: returns (gz, go, gscale, gshift, gb3, gl3)"""
    N, C, H, W = out.shape
    Ch, HW = C // 2, H * W
    gz = torch.empty_like(gout)
    if go is None:
        go = torch.empty_like(o)
    gscale = gshift = None
    if clamp_type == 0:
        gscale = arena.take(Ch)
        gshift = arena.take(Ch)
    gb3 = arena.take(C)
    gl3 = arena.take(C)
    op, ons = L.frames(o, "o")
    outp, outns = L.frames(out, "out")
    gop, gons = L.frames(gout, "gout")
    gzp, gzns = L.frames(gz, "gz")
    gonp, gonns = L.frames(go, "go")
    scf, shf, l3f = _f(scale), _f(scale_shift), _f(l3)
    L.call("rfn_affine_zeros_bwd_f32", outp, _l(outns), op, _l(ons), gop, _l(gons), L.dev(gdl), L.dev(scf),
           L.dev(shf), L.dev(l3f), gzp, _l(gzns), gonp, _l(gonns), L.dev(gscale), L.dev(gshift), L.dev(gb3),
           L.dev(gl3), _i(clamp_type), _i(N), _i(C), _i(HW), meta=_shell("affine_zeros_bwd", gout, 4.5))
    return gz, go, gscale, gshift, gb3, gl3


class GlowStepFn(torch.autograd.Function):
    """One forward Glow step (Flow/glow.py:31-36) as a single autograd node:
         y  = (x + an_bias) * exp(an_logs)                       glow_modules.py:38-45
         z  = Wm y                                               glow_modules.py:209-216
         h1 = act(actnorm(conv3x3(cat(z1, cond))))               glow_modules.py:232-238, 139-142
         h2 = act(actnorm(conv1x1(h1)))
         o  = (conv3x3(h2) + b3) * exp(3 logs3)                  glow_modules.py:119-121
         z2 <- (z2 + o[0::2]) * exp(clamp(o[1::2]))              glow_modules.py:276-285
       Returns (out, dlogdet[N]) where dlogdet holds only the data dependent Σ clamp(s) part; the parameter-only
       terms (Σlogs + Σlog_s)·H·W are added by the caller.
       Saved for backward: x, cond, out, h1, h2, o (activations stay resident in HBM, 288 GB is plenty).
       (The K steps of a level normally run as ONE node, GlowLevelFn; this one serves the first, data-initialising
       call and stand-alone coupling layers.)"""

    @staticmethod
    def forward(ctx, x, cond, Wm, an_bias, an_logs, w1, n1b, n1l, w2, n2b, n2l, w3, b3, l3, scale, scale_shift,
                act, clamp_type, packs=None):
        """`packs` (optional): (w1 fwd, w1 dgrad, w2 fwd, w2 dgrad, w3 fwd, w3 dgrad, fused-forward stream, fused-backward stream) packed
        buffers kept fresh by the caller's pack plans (entries may be None: packed on the fly); the forward entries are
        bf16x3 packs and only used where that is the forward arithmetic."""
        out = actnorm_invconv_fwd(x, _f(an_bias), _f(an_logs), Wm.detach())
        pk = tuple(packs) + (None,) * (PACK_SLOTS - len(packs)) if packs is not None else (None,) * PACK_SLOTS
        ctx.packs = pk
        h1, h2, o, P, masks = _net_fwd(out, cond, w1, n1b, n1l, w2, n2b, n2l, w3, b3, l3, act, pk,
                                       want_masks=any(ctx.needs_input_grad))
        o, dlogdet = gather_affine_(out, o, P, _f(b3), _f(l3), _f(scale), _f(scale_shift), clamp_type)
        ctx.save_for_backward(x, cond, Wm, an_bias, an_logs, w1, n1l, w2, n2l, w3, l3, scale, scale_shift, out, h1, h2, o,
                              n1b, n2b, *(masks if masks is not None else ()))
        ctx.cfg = (act, clamp_type)
        return out, dlogdet

    @staticmethod
    def backward(ctx, gout, gdl):
        (x, cond, Wm, an_bias, an_logs, w1, n1l, w2, n2l, w3, l3, scale, scale_shift, out, h1, h2, o, n1b, n2b) = ctx.saved_tensors[:19]
        masks = tuple(ctx.saved_tensors[19:21]) if len(ctx.saved_tensors) > 19 else None
        act, clamp_type = ctx.cfg
        N, C, H, W = x.shape
        gout = gout.contiguous()
        gdl = None if gdl is None else gdl.contiguous()
        Cc = int(cond.shape[1])
        k1, k2, k3 = int(w1.shape[2]), int(w2.shape[2]), int(w3.shape[2])
        # every accumulate-into output of this node lives in one zero-filled arena (1 fill launch instead of 14)
        arena = ZeroArena(_net_arena_numel(C, Cc, int(w1.shape[0]), k1, k2, k3), x.device)
        # ---- affine coupling bwd + Conv2dZeros epilogue bwd in one launch: gz (whole tensor), go = grad at conv3's output
        gz, go, gscale, gshift, gb3, gl3 = _affine_zeros_bwd(out, o, gout, gdl, scale, scale_shift, l3, clamp_type, arena)
        gcond = torch.empty_like(cond) if Cc > 0 else torch.zeros_like(cond)
        gw1, gn1b, gn1l, gw2, gn2b, gn2l, gw3 = _net_bwd(go, out, cond, h1, h2, w1, n1l, w2, n2l, w3, act, ctx.packs,
                                                         arena, gz, gcond, False, n1b=n1b, n2b=n2b, masks=masks)
        # ---- invconv + actnorm bwd
        gx, gW, gab, gal = actnorm_invconv_bwd(x, _f(an_bias), _f(an_logs), Wm.detach(), gz, arena)
        return (gx, gcond, gW, gab.view(an_bias.shape), gal.view(an_logs.shape), gw1, gn1b.view(1, -1, 1, 1),
                gn1l.view(n1l.shape), gw2, gn2b.view(1, -1, 1, 1), gn2l.view(n2l.shape), gw3, gb3, gl3.view(l3.shape),
                None if gscale is None else gscale.view(scale.shape),
                None if gshift is None else gshift.view(scale_shift.shape), None, None, None)


STEP_NPARAM = 13  # an_bias, an_logs, w1, n1b, n1l, w2, n2b, n2l, w3, b3, l3, scale, scale_shift


class GlowLevelFn(torch.autograd.Function):
    """The K Glow steps of one flow level (Flow/glow.py:105-117, inner loop) as ONE autograd node.  Between two steps a
    single launch each way does the shell work (rfn_glow_shell_fwd_f32: coupling tail of step k + ActNorm/InvConv head
    of step k+1; rfn_glow_shell_bwd_f32: the mirror image); the gradient wrt the shared condition map accumulates
    inside the data-gradient kernels and the per-frame log-det inside the shell kernel, so autograd adds nothing.
    apply(x, cond, Wst[K,C,C], act, clamp_type, packs (list of K 7-tuples or None), *13K step parameters)
    -> (out, dlogdet[N] = sum over the K steps of the data dependent log-det AND of the ActNorm parameter term
    H*W * sum_c logs[c]; the InvConv term sum log|s| * H*W stays with the caller, who builds the matrices)."""

    @staticmethod
    def forward(ctx, x, cond, Wst, act, clamp_type, packs, *flat):
        Kn = int(Wst.shape[0])
        assert len(flat) == STEP_NPARAM * Kn
        N, C, H, W = x.shape
        prm = [flat[STEP_NPARAM * k:STEP_NPARAM * (k + 1)] for k in range(Kn)]
        pks = [(tuple(packs[k]) + (None,) * PACK_SLOTS)[:PACK_SLOTS] if packs is not None and packs[k] is not None
               else (None,) * PACK_SLOTS
               for k in range(Kn)]
        want_masks = any(ctx.needs_input_grad)
        Wd = Wst.detach().contiguous()
        # log-det: every shell launch WRITES its per-block partial sums into its own slice; one reduce launch adds them
        # per frame in a fixed order (no float atomics anywhere in the forward pass: bit-reproducible)
        ldf = int(L.load().rfn_glow_shell_fwd_ld_floats(N, C, H, W))
        ldp = torch.empty((Kn + 1, ldf), device=x.device, dtype=torch.float32)
        dl = torch.empty(N, device=x.device, dtype=torch.float32)
        xp, xns = L.frames(x, "x")
        z = torch.empty_like(x)
        zp, zns = L.frames(z, "z")
        ab0, al0 = _f(prm[0][0]), _f(prm[0][1])
        L.call("rfn_glow_shell_fwd_f32", xp, _l(xns), None, None, _l(0), None, None, None, None, None, L.dev(ldp[0]), _i(0),
               L.dev(ab0), L.dev(al0), L.dev(Wd[0]), zp, _l(zns), _i(1), _i(N), _i(C), _i(H), _i(W),
               meta=_shell("glow_shell_fwd", x, 2))
        outs, h1s, h2s, os_, mks = [], [], [], [], []
        for k in range(Kn):
            (_, _, w1, n1b, n1l, w2, n2b, n2l, w3, b3, l3, scale, scale_shift) = prm[k]
            h1, h2, o, P, masks = _net_fwd(z, cond, w1, n1b, n1l, w2, n2b, n2l, w3, b3, l3, act, pks[k], want_masks)
            mks.append(masks)
            last = k == Kn - 1
            zn = None if last else torch.empty_like(z)
            znp, znns = (None, 0) if last else L.frames(zn, "znext")
            hold = [_f(b3), _f(l3), _f(scale), _f(scale_shift)] + ([None, None] if last else [_f(prm[k + 1][0]), _f(prm[k + 1][1])])
            zp, zns = L.frames(z, "z")
            if P is not None:
                o = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
                args = (L.dev(P), None, _l(0), L.dev(hold[0]), L.dev(hold[1]), L.dev(o))
                nt = 12 + (0 if last else 1)
            else:
                op, ons = L.frames(o, "o")
                args = (None, op, _l(ons), None, None, None)
                nt = 2 + (0 if last else 1.5)
            L.call("rfn_glow_shell_fwd_f32", zp, _l(zns), *args, L.dev(hold[2]), L.dev(hold[3]), L.dev(ldp[k + 1]),
                   _i(clamp_type), L.dev(hold[4]), L.dev(hold[5]), None if last else L.dev(Wd[k + 1]), znp, _l(znns),
                   _i(0 if last else 1), _i(N), _i(C), _i(H), _i(W), meta=_shell("glow_shell_fwd", z, nt))
            outs.append(z)
            h1s.append(h1)
            h2s.append(h2)
            os_.append(o)
            z = zn
        L.call("rfn_logdet_reduce_f32", L.dev(ldp), _i(Kn + 1), L.dev(dl), _i(0), _i(N), _i(C), _i(H), _i(W),
               meta=_shell("logdet_reduce", ldp, 1))
        has_masks = all(m is not None for m in mks)
        ctx.save_for_backward(x, cond, Wst, *flat, *outs, *h1s, *h2s, *os_,
                              *([m[0] for m in mks] + [m[1] for m in mks] if has_masks else []))
        ctx.cfg = (act, clamp_type, Kn, pks, has_masks)
        return outs[-1], dl

    @staticmethod
    def backward(ctx, gout, gdl):
        act, clamp_type, Kn, pks, has_masks = ctx.cfg
        sv = ctx.saved_tensors
        x, cond, Wst = sv[0], sv[1], sv[2]
        nf = STEP_NPARAM * Kn
        flat = sv[3:3 + nf]
        rest = sv[3 + nf:]
        outs, h1s, h2s, os_ = (rest[i * Kn:(i + 1) * Kn] for i in range(4))
        mks = [(rest[4 * Kn + k], rest[5 * Kn + k]) for k in range(Kn)] if has_masks else [None] * Kn
        fin = []   # ActNorm-gradient tickets of the fused backward kernel, finished in one launch at the end
        prm = [flat[STEP_NPARAM * k:STEP_NPARAM * (k + 1)] for k in range(Kn)]
        N, C, H, W = x.shape
        Ch, HW = C // 2, H * W
        Cc = int(cond.shape[1])
        gout = gout.contiguous()
        gdl = None if gdl is None else gdl.contiguous()
        Wd = Wst.detach().contiguous()
        w1 = prm[0][2]
        k1, k2, k3 = int(prm[0][2].shape[2]), int(prm[0][5].shape[2]), int(prm[0][8].shape[2])
        # one zero-filled arena for the accumulate-into outputs of all K steps
        arena = ZeroArena(Kn * _net_arena_numel(C, Cc, int(w1.shape[0]), k1, k2, k3), x.device)
        gWst = arena.take(Kn, C, C)
        gcond = torch.empty_like(cond) if Cc > 0 else torch.zeros_like(cond)
        grads = [None] * nf
        # last step: stand-alone coupling backward; earlier steps get theirs from the fused shell launch below
        (_, _, _, _, _, _, _, _, _, _, l3, scale, scale_shift) = prm[Kn - 1]
        # (with deferred weight gradients the K conv3-output gradients live in one buffer: one tap-scatter for all)
        go_all = (torch.empty((Kn,) + tuple(os_[0].shape), device=x.device, dtype=torch.float32)
                  if (N * HW <= GROUPED_WGRAD_MAX_PIX and 1 < Kn <= 16 and os.environ.get("RFN_WGRAD_GROUPED") != "0")
                  else None)
        gz, go, gscale, gshift, gb3, gl3 = _affine_zeros_bwd(outs[Kn - 1], os_[Kn - 1], gout, gdl, scale, scale_shift, l3,
                                                            clamp_type, arena, None if go_all is None else go_all[Kn - 1])
        gx = None
        # latency-class levels (deep levels, small batches): the 3 K weight gradients are computed at the end, K of one
        # shape per launch (their operands stay alive until then: a few hundred MB at most)
        defer = ({"w1": [], "w2": [], "w3": []}
                 if (N * HW <= GROUPED_WGRAD_MAX_PIX and 1 < Kn <= 16 and os.environ.get("RFN_WGRAD_GROUPED") != "0")
                 else None)
        for k in range(Kn - 1, -1, -1):
            (an_bias, an_logs, w1, n1b, n1l, w2, n2b, n2l, w3, b3, l3, scale, scale_shift) = prm[k]
            nfin = len(fin)
            gw1, gn1b, gn1l, gw2, gn2b, gn2l, gw3 = _net_bwd(go, outs[k], cond, h1s[k], h2s[k], w1, n1l, w2, n2l, w3, act,
                                                             pks[k], arena, gz, gcond, k != Kn - 1, defer,
                                                             n1b=n1b, n2b=n2b, masks=mks[k], fin=fin)
            if len(fin) > nfin:
                fin[-1].append(k)   # the step whose (possibly deferred) weight gradients the ticket needs
            base = STEP_NPARAM * k
            grads[base + 2:base + 13] = [gw1, gn1b.view(1, -1, 1, 1), gn1l.view(n1l.shape), gw2, gn2b.view(1, -1, 1, 1),
                                         gn2l.view(n2l.shape), gw3, gb3, gl3.view(l3.shape),
                                         None if gscale is None else gscale.view(scale.shape),
                                         None if gshift is None else gshift.view(scale_shift.shape)]
            gab, gal = arena.take(C), arena.take(C)
            grads[base], grads[base + 1] = gab.view(an_bias.shape), gal.view(an_logs.shape)
            abf, alf = _f(an_bias), _f(an_logs)
            gzp, gzns = L.frames(gz, "gz")
            if k == 0:
                gx = torch.empty_like(x)
                xp, xns = L.frames(x, "x")
                gxp, gxns = L.frames(gx, "gx")
                L.call("rfn_actnorm_invconv_bwd_ld_f32", xp, _l(xns), L.dev(abf), L.dev(alf), L.dev(Wd[0]), gzp, _l(gzns),
                       gxp, _l(gxns), L.dev(gWst[0]), L.dev(gab), L.dev(gal), L.dev(gdl), _i(N), _i(C), _i(HW),
                       meta=_shell("actnorm_invconv_bwd", x, 3))
                break
            # fused: ActNorm/InvConv backward of step k, coupling + Conv2dZeros-epilogue backward of step k-1
            (_, _, _, _, _, _, _, _, _, _, l3p, scalep, shiftp) = prm[k - 1]
            xin, op_ = outs[k - 1], os_[k - 1]
            gzn = torch.empty_like(gz)
            gon = torch.empty_like(op_) if go_all is None else go_all[k - 1]
            gscale = gshift = None
            if clamp_type == 0:
                gscale, gshift = arena.take(Ch), arena.take(Ch)
            gb3, gl3 = arena.take(C), arena.take(C)
            xp, xns = L.frames(xin, "x")
            opp, ons = L.frames(op_, "o")
            gznp, gznns = L.frames(gzn, "gz_prev")
            gonp, gonns = L.frames(gon, "gpre")
            hold = [_f(scalep), _f(shiftp), _f(l3p)]
            L.call("rfn_glow_shell_bwd_f32", xp, _l(xns), L.dev(abf), L.dev(alf), L.dev(Wd[k]), gzp, _l(gzns),
                   L.dev(gWst[k]), L.dev(gab), L.dev(gal), opp, _l(ons), L.dev(gdl), L.dev(hold[0]), L.dev(hold[1]),
                   L.dev(hold[2]), gznp, _l(gznns), gonp, _l(gonns), L.dev(gscale), L.dev(gshift), L.dev(gb3),
                   L.dev(gl3), _i(clamp_type), _i(1), _i(N), _i(C), _i(HW), meta=_shell("glow_shell_bwd", xin, 5.5))
            gz, go = gzn, gon
        if defer is not None and defer["w2"]:
            # entries were appended for k = Kn-1 .. 0: back to step order
            for key in ("w1", "w2", "w3"):
                defer[key].reverse()
            Hd_ = int(prm[0][2].shape[0])
            g1 = conv2d_wgrad_grouped([t[0] for t in defer["w1"]],
                                      None if defer["w1"][0][1] is None else [t[1] for t in defer["w1"]],
                                      [t[2] for t in defer["w1"]], Hd_, k1, arena)
            g2 = conv2d_wgrad_grouped([t[0] for t in defer["w2"]], None, [t[1] for t in defer["w2"]], Hd_, k2, arena)
            g3 = conv2d_wgrad_grouped([t[0] for t in defer["w3"]], None, [t[1] for t in defer["w3"]], C, k3, arena,
                                      g_stacked=go_all)
            for k in range(Kn):
                base = STEP_NPARAM * k
                grads[base + 2], grads[base + 5], grads[base + 8] = g1[k], g2[k], g3[k]
        if fin:
            for t in fin:
                base = STEP_NPARAM * t.pop()
                t[2], t[5] = grads[base + 2], grads[base + 5]
            coupling_po_bwd_finish(fin)
        return (gx, gcond, gWst, None, None, None) + tuple(grads)


class GlowStepRevFn(torch.autograd.Function):
    """Reverse Glow step (Flow/glow.py:37-41) for generation; no gradient (the reference samples under no_grad)."""

    @staticmethod
    def forward(ctx, x, cond, Winv, an_bias, an_logs, w1, n1b, n1l, w2, n2b, n2l, w3, b3, l3, scale, scale_shift,
                act, clamp_type, pkcache=None):
        """`pkcache` (optional dict, owned by the caller): packed weights of this step, filled on first use and reused
        while the caller keeps it -- autoregressive generation runs the same step once per frame on unchanged weights."""
        N, C, H, W = x.shape
        Ch = C // 2
        Hd = int(w1.shape[0])
        f = lambda t: None if t is None else t.detach().reshape(-1).contiguous()
        z = x.detach().clone()
        cin2 = cond if cond.shape[1] > 0 else None
        fp = fwd_prec(H, W)
        Cc_ = 0 if cin2 is None else int(cin2.shape[1])
        pkc = pkcache if pkcache is not None else {}

        def cached(key, make):
            if key not in pkc:
                pkc[key] = make()
            return pkc[key]
        if coupling_po_ok(N, C, Cc_, Hd, H, W, w1, w3) and int(w2.shape[2]) == 1:
            def make_po():
                plan = POPackPlan([(w1.detach(), w2.detach(), w3.detach())])
                plan.run()
                return plan.bufs[0]
            _, _, P, _ = coupling_po_fwd(z, cin2, cached("po", make_po), f(n1b), f(n1l), f(n2b), f(n2l), C, act)
            o = torch.empty((N, C, H, W), device=x.device, dtype=torch.float32)
            b3f, l3f = f(b3), f(l3)
            L.call("rfn_tap_gather_f32", L.dev(P), L.dev(b3f), L.dev(l3f), L.dev(o), _i(N), _i(C), _i(H), _i(W))
        else:
            h1 = conv2d_raw(z[:, :Ch], cin2, cached(("w1", fp), lambda: pack_weight(w1, prec=fp)), Hd, int(w1.shape[2]), 1,
                            f(n1b), f(n1l), act, prec=fp)
            h2 = conv2d_raw(h1, None, cached(("w2", fp), lambda: pack_weight(w2, prec=fp)), Hd, int(w2.shape[2]), 1,
                            f(n2b), f(n2l), act, prec=fp)
            if zeros_conv_uses_taps(w3):
                C3, Cin3 = int(w3.shape[0]), int(w3.shape[1])
                pk3 = cached(("w3t", fp), lambda: pack_weight(
                    w3.detach().permute(2, 3, 0, 1).reshape(9 * C3, Cin3, 1, 1).contiguous(), prec=fp))
            else:
                pk3 = cached(("w3", fp), lambda: pack_weight(w3, prec=fp))
            o = zeros_conv_fwd(h2, w3, f(b3), f(l3), pk3, prec=fp)
        dlogdet = torch.zeros(N, device=x.device, dtype=torch.float32)
        affine_coupling_(z, o, f(scale), f(scale_shift), dlogdet, clamp_type, True)
        out = invconv_actnorm_rev(z, f(an_bias), f(an_logs), Winv.detach())
        ctx.mark_non_differentiable(out, dlogdet)
        return out, dlogdet


class GaussLogpFn(torch.autograd.Function):
    """Σ log N(z; mean, std) per frame (glow_modules.py:358-365 layout 0/softplus; glow.py:135-140 layout 1/exp)."""

    @staticmethod
    def forward(ctx, z, o, layout, std_mode):
        ctx.save_for_backward(z, o)
        ctx.cfg = (layout, std_mode)
        return gauss_logp(z, o, layout, std_mode)

    @staticmethod
    def backward(ctx, g):
        z, o = ctx.saved_tensors
        layout, std_mode = ctx.cfg
        N, Cz = z.shape[0], z.shape[1]
        gz = torch.empty(z.shape, device=z.device, dtype=torch.float32)
        go = torch.empty(o.shape, device=z.device, dtype=torch.float32)
        zp, zns = L.frames(z, "z")
        op, ons = L.frames(o, "o")
        gzp, gzns = L.frames(gz, "gz")
        gop, gons = L.frames(go, "go")
        L.call("rfn_gauss_logp_bwd_f32", zp, _l(zns), op, _l(ons), L.dev(g.contiguous()), gzp, _l(gzns), gop, _l(gons),
               _i(layout), _i(std_mode), _i(N), _i(Cz), _i(_hw(z)), meta=_shell("gauss_logp_bwd", z, 6))
        return gz, go, None, None


# ------------------------------------------------------------------------------------------------ small-map dense convs
def smallmap_arith_ok(H, W):
    """the dense small-map kernels compute forward AND backward in bf16x3: allowed when that is this map's forward
    arithmetic ('bf16x3' mode; 'mixed' mode on maps of at most 2x2)"""
    return CONV_PRECISION == "bf16x3" or (CONV_PRECISION == "mixed" and H * W <= 4)


def smallmap_supported(conv, H, W):
    """3x3 / stride 1 / pad 1 convolution on an H x W <= 16 map whose sample rows are 16-byte friendly"""
    return (smallmap_arith_ok(H, W) and tuple(conv.kernel_size) == (3, 3) and tuple(conv.stride) == (1, 1)
            and tuple(conv.padding) == (1, 1) and tuple(conv.dilation) == (1, 1) and conv.groups == 1
            and conv.bias is not None and H * W <= 16 and (conv.in_channels * H * W) % 8 == 0
            and (conv.out_channels * H * W) % 8 == 0)


class _SmallmapPackDesc(ctypes.Structure):   # rfn_smallmap_pack_desc (include/rfn_hip.h)
    _fields_ = [("w", ctypes.c_void_p), ("packed", ctypes.c_void_p), ("Cout", ctypes.c_int), ("Cin", ctypes.c_int),
                ("H", ctypes.c_int), ("W", ctypes.c_int), ("transpose", ctypes.c_int), ("pad_", ctypes.c_int)]


# packs asked for but not launched yet: (descriptor fields, weight tensor kept alive).  The weights change every
# optimizer step, so a training step re-packs ~60 matrices (latent nets, ConvLSTM, the 2x2 flow level); they are queued
# here and go out in ONE launch (rfn_smallmap_pack_batched_bf16x3) right before the first kernel that reads any of them
# -- every consumer (smallmap_dense / _pair / _conv) calls smallmap_pack_flush() first.
_PACK_QUEUE = []


def smallmap_pack(w, H, W, transpose):
    """w [Cout, Cin, 3, 3] -> MFMA-fragment-ordered bf16 (hi, lo) dense matrix of the H x W map.  The buffer is returned
    at once; its contents exist after smallmap_pack_flush() (called by every consumer)."""
    Cout, Cin = int(w.shape[0]), int(w.shape[1])
    nbytes = int(L.load().rfn_smallmap_packed_size(Cout, Cin, H, W, 1 if transpose else 0))
    buf = torch.empty(nbytes // 4, device=w.device, dtype=torch.float32)
    wc = w.detach().contiguous()
    L.dev(wc, "w")
    _PACK_QUEUE.append(((wc.data_ptr(), buf.data_ptr(), Cout, Cin, int(H), int(W), 1 if transpose else 0, 0), wc, buf))
    L.PENDING_FLUSH = flush_packs
    return buf


def smallmap_pack_flush():
    """launch the queued packs (one launch per 64 matrices)"""
    if not _PACK_QUEUE:
        return
    q = list(_PACK_QUEUE)
    del _PACK_QUEUE[:]
    arr = (_SmallmapPackDesc * len(q))(*[_SmallmapPackDesc(*f) for f, _, _ in q])
    L.call("rfn_smallmap_pack_batched_bf16x3", ctypes.cast(arr, ctypes.c_void_p), _i(len(q)),
           meta=_shell("rfn_smallmap_pack_bf16x3", q[0][2], sum(b.numel() for _, _, b in q) / max(q[0][2].numel(), 1)))


def smallmap_dense(a, packed, n_channels, bias=None, slope_out=None, y=None, slope_in=0.0, want_a_out=False, add=None):
    """out[B, n_channels, H, W] = a'[B, C, H, W] (as rows) x packed (+ bias, leaky_relu) -- rfn_smallmap_dense_bf16x3.
    With y: a' = a * (y > 0 ? 1 : slope_in); returns (out, a') when want_a_out."""
    smallmap_pack_flush()
    a = a.contiguous()
    B, H, W = int(a.shape[0]), int(a.shape[2]), int(a.shape[3])
    K, HW = int(a.shape[1]) * H * W, H * W
    out = torch.empty((B, n_channels, H, W), device=a.device, dtype=torch.float32)
    a_out = torch.empty_like(a) if want_a_out else None
    yc = None if y is None else y.contiguous()
    addc = None if add is None else add.contiguous()  # held until the launch is enqueued
    L.call("rfn_smallmap_dense_bf16x3", L.dev(a), L.dev(yc), ctypes.c_float(slope_in), L.dev(packed), L.dev(bias),
           L.dev(addc), _i(0 if slope_out is None else 1), ctypes.c_float(0.0 if slope_out is None else slope_out), L.dev(out),
           L.dev(a_out), _i(B), _i(K), _i(n_channels * HW), _i(HW))
    return (out, a_out) if want_a_out else out


def smallmap_dense_pair(a0, packed0, n_ch0, a1, packed1, n_ch1, bias0=None, bias1=None, slope_out0=None, slope_out1=None,
                        y0=None, y1=None, slope_in0=0.0, slope_in1=0.0, want_a_out=False, add0=None, add1=None):
    """two independent smallmap_dense products in ONE launch (rfn_smallmap_dense_pair_bf16x3): same batch and map size.
    Returns (out0, out1) or (out0, a0', out1, a1') with want_a_out.  add0 / add1: optional [B, n_ch, H, W] addends."""
    smallmap_pack_flush()
    a0, a1 = a0.contiguous(), a1.contiguous()
    B, H, W = int(a0.shape[0]), int(a0.shape[2]), int(a0.shape[3])
    assert int(a1.shape[0]) == B and tuple(a1.shape[2:]) == (H, W)
    HW = H * W
    K0, K1 = int(a0.shape[1]) * HW, int(a1.shape[1]) * HW
    out0 = torch.empty((B, n_ch0, H, W), device=a0.device, dtype=torch.float32)
    out1 = torch.empty((B, n_ch1, H, W), device=a0.device, dtype=torch.float32)
    ao0 = torch.empty_like(a0) if want_a_out and y0 is not None else None
    ao1 = torch.empty_like(a1) if want_a_out and y1 is not None else None
    y0c = None if y0 is None else y0.contiguous()  # held until the launch is enqueued
    y1c = None if y1 is None else y1.contiguous()
    ad0 = None if add0 is None else add0.contiguous()
    ad1 = None if add1 is None else add1.contiguous()
    fl = ctypes.c_float
    L.call("rfn_smallmap_dense_pair_bf16x3",
           L.dev(a0), L.dev(y0c), fl(slope_in0), L.dev(packed0), L.dev(bias0), L.dev(ad0), _i(0 if slope_out0 is None else 1),
           fl(0.0 if slope_out0 is None else slope_out0), L.dev(out0), L.dev(ao0), _i(K0), _i(n_ch0 * HW),
           L.dev(a1), L.dev(y1c), fl(slope_in1), L.dev(packed1), L.dev(bias1), L.dev(ad1), _i(0 if slope_out1 is None else 1),
           fl(0.0 if slope_out1 is None else slope_out1), L.dev(out1), L.dev(ao1), _i(K1), _i(n_ch1 * HW), _i(B), _i(HW))
    if want_a_out:
        return out0, (ao0 if ao0 is not None else a0), out1, (ao1 if ao1 is not None else a1)
    return out0, out1


def smallmap_conv_ok(H, W, C1, C2, Cout, N, bwd=False):
    """3x3 conv on a map small enough for the dense kernels (rfn_smallmap_conv_bf16x3), and few enough frames: every
    32-frame row tile streams the whole dense matrix ((C1+C2)*HW x Cout*HW, on a 4x4 map more than half structural
    zeros), so the product is only used while that stream stays L2 / Infinity-Cache sized."""
    HW = H * W
    stream = -(-N // 32) * (C1 + C2) * HW * Cout * HW
    return ((bwd_b3() if bwd else smallmap_arith_ok(H, W)) and HW <= 16 and (C1 * HW) % 8 == 0 and ((C1 + C2) * HW) % 8 == 0
            and stream <= 32 * 1024 * 1024 and os.environ.get("RFN_SMALLMAP_GLOW") != "0")


def smallmap_conv(in1, in2, packed, Cout, ep_mode=0, p0=None, p1=None, act=0, out1=None, out2=None, cout_split=None,
                  acc1=False, acc2=False):
    """conv2d_raw's contract (3x3, pad 1) on an H*W <= 16 map through the dense split-precision product."""
    smallmap_pack_flush()
    N, C1, H, W = in1.shape
    C2 = 0 if in2 is None else int(in2.shape[1])
    if cout_split is None:
        cout_split = Cout
    if out1 is None:
        out1 = torch.empty((N, cout_split, H, W), device=in1.device, dtype=torch.float32)
    i1p, i1ns = L.frames(in1, "in1")
    i2p, i2ns = (None, 0) if in2 is None else L.frames(in2, "in2")
    o1p, o1ns = L.frames(out1, "out1")
    o2p, o2ns = (None, 0) if out2 is None else L.frames(out2, "out2")
    L.call("rfn_smallmap_conv_bf16x3", i1p, _l(i1ns), _i(C1), i2p, _l(i2ns), _i(C2), L.dev(packed), o1p, _l(o1ns), o2p,
           _l(o2ns), _i(Cout), _i(cout_split), _i((1 if acc1 else 0) | (2 if acc2 else 0)), _i(N), _i(H), _i(W), _i(ep_mode), L.dev(p0),
           L.dev(p1), _i(act),
           meta=("conv", "smallmap_dense_kernel", 2.0 * N * H * W * (C1 + C2) * Cout * 9,
                 "N%d %d+%d->%d %dx%d k3 ep%d dense" % (N, C1, C2, Cout, H, W, ep_mode),
                 4.0 * (N * H * W * (C1 + C2 + Cout) + (C1 + C2) * Cout * H * W * H * W)))
    return out1


INVCONV_MAX_STEPS, INVCONV_MAX_CHANNELS = 32, 96  # RFN_INVCONV_MAX_* of include/rfn_hip.h


def invconv_weights_ok(ics):
    """the K InvConv layers of a flow level the one-launch kernels take: LU parameterised, on the GPU, C <= 96"""
    return (0 < len(ics) <= INVCONV_MAX_STEPS and all(ic.LU_decomposed for ic in ics) and ics[0].lower.is_cuda
            and int(ics[0].lower.shape[0]) <= INVCONV_MAX_CHANNELS and os.environ.get("RFN_INVCONV_KERNEL") != "0")


class InvConvWeightsFn(torch.autograd.Function):
    """InvConv.get_weight (glow_modules.py:178-207) of the K steps of a flow level: (W [K, C, C], HW * sum log_s) in one
    launch, and one launch back to (lower, upper, log_s) gradients (rfn_invconv_weights_{fwd,bwd}_f32).
    apply(hw, K, p_1..p_K, sign_1..sign_K, lower_1..lower_K, upper_1..upper_K, log_s_1..log_s_K)."""

    @staticmethod
    def _pointers(P, S, Lw, U, LS):
        """host arrays of the K device pointers per parameter kind, in the entry points' argument order (the tensors are
        the modules' own parameters / buffers: dense, alive and at fixed addresses)"""
        out = []
        for grp, nm in ((P, "p"), (Lw, "lower"), (U, "upper"), (LS, "log_s"), (S, "sign_s")):
            for x in grp:
                L.dev(x, nm)  # fp32, device, contiguous -- raises otherwise
            out.append(L.ptr_array(list(grp), nm))
        return out

    @staticmethod
    def forward(ctx, hw, Kn, *t):
        ctx.set_materialize_grads(False)
        P, S, Lw, U, LS = (t[i * Kn:(i + 1) * Kn] for i in range(5))
        C = int(Lw[0].shape[0])
        arrs = InvConvWeightsFn._pointers(P, S, Lw, U, LS)
        W = torch.empty((Kn, C, C), device=Lw[0].device, dtype=torch.float32)
        c = torch.empty((), device=Lw[0].device, dtype=torch.float32)   # written by the kernel
        L.call("rfn_invconv_weights_fwd_f32", *arrs, L.dev(W), L.dev(c), _i(Kn), _i(C), _i(int(hw)),
               meta=_shell("invconv_weights_fwd", W, 4))
        ctx.cfg = (int(hw), Kn, C)
        ctx.save_for_backward(*t)
        return W, c

    @staticmethod
    def backward(ctx, gW, gc):
        hw, Kn, C = ctx.cfg
        t = ctx.saved_tensors
        P, S, Lw, U, LS = (t[i * Kn:(i + 1) * Kn] for i in range(5))
        dev_ = Lw[0].device
        if gW is None:
            gW = torch.zeros((Kn, C, C), device=dev_, dtype=torch.float32)
        arrs = InvConvWeightsFn._pointers(P, S, Lw, U, LS)
        gl = torch.empty((Kn, C, C), device=dev_, dtype=torch.float32)
        gu = torch.empty((Kn, C, C), device=dev_, dtype=torch.float32)
        gs = torch.empty((Kn, C), device=dev_, dtype=torch.float32)
        gcc = None if gc is None else gc.contiguous()
        L.call("rfn_invconv_weights_bwd_f32", *arrs, L.dev(gW.contiguous()), L.dev(gcc), L.dev(gl), L.dev(gu), L.dev(gs),
               _i(Kn), _i(C), _i(hw), meta=_shell("invconv_weights_bwd", gW, 4))
        return (None, None) + (None,) * (2 * Kn) + tuple(gl.unbind(0)) + tuple(gu.unbind(0)) + tuple(gs.unbind(0))


class LatentStepFn(torch.autograd.Function):
    """one SRNN latent step (RFN_new.py:167-184,206-207): from the raw outputs of the encoder / prior parameter convs
    to (z_t, z^x_t, KL, enc_mean, enc_std) in one kernel each way (rfn_latent_step_{fwd,bwd}_f32)."""

    @staticmethod
    def forward(ctx, enc, pri, eps_p, eps_q, res_q):
        ctx.set_materialize_grads(False)  # unused outputs (enc_mean / enc_std without overshooting) arrive as None
        B = int(enc.shape[0])
        shp = (B, enc.shape[1] // 2) + tuple(enc.shape[2:])
        ZHW = 1
        for d in shp[1:]:
            ZHW *= int(d)
        enc, pri, eps_p, eps_q = enc.contiguous(), pri.contiguous(), eps_p.contiguous(), eps_q.contiguous()
        outs = [torch.empty(shp, device=enc.device, dtype=torch.float32) for _ in range(5)]
        L.call("rfn_latent_step_fwd_f32", L.dev(enc), L.dev(pri), L.dev(eps_p), L.dev(eps_q), *[L.dev(o) for o in outs],
               _i(B), _i(ZHW), _i(1 if res_q else 0), meta=_shell("latent_step_fwd", enc, 5.5))
        ctx.save_for_backward(enc, pri, eps_p, eps_q)
        ctx.cfg = (B, ZHW, bool(res_q))
        return tuple(outs)

    @staticmethod
    def backward(ctx, g_zt, g_zxt, g_kl, g_em, g_es):
        enc, pri, eps_p, eps_q = ctx.saved_tensors
        B, ZHW, res_q = ctx.cfg
        # g_zt / g_zxt usually arrive as channel slices of the next step's input gradient: read in place (row stride)
        strided, keep = [], []
        for g in (g_zt, g_zxt):
            if g is None:
                strided += [None, _l(0)]
                continue
            if B > 0 and (not g[0].is_contiguous() or (B > 1 and g.stride(0) < ZHW)):
                g = g.contiguous()
            keep.append(g)
            gp, gns = L.frames(g, "g_z")
            strided += [gp, _l(gns)]
        # keep the contiguous copies alive until the launch is enqueued (a freed temporary's block would be reused)
        gs = [None if g is None else g.contiguous() for g in (g_kl, g_em, g_es)]
        g_enc, g_pri = torch.empty_like(enc), torch.empty_like(pri)
        L.call("rfn_latent_step_bwd_f32", L.dev(enc), L.dev(pri), L.dev(eps_p), L.dev(eps_q), *strided,
               *[L.dev(g) for g in gs], L.dev(g_enc), L.dev(g_pri), _i(B), _i(ZHW), _i(1 if res_q else 0))
        return g_enc, g_pri, None, None, None


class ConvLSTMCellFn(torch.autograd.Function):
    """ConvLSTMLayer.forward (Utils/modules.py:355-377): conv3x3(cat(x,h))+b on the MFMA conv kernel, then the fused
    gate update.  Peephole tensors may be None (== 0)."""

    @staticmethod
    def forward(ctx, x, h, c, w, b, wci, wcf, wco):
        N, Cx, H, W = x.shape
        Hc = int(w.shape[0]) // 4
        ks = int(w.shape[2])
        HW = H * W
        fp = fwd_prec(H, W)
        cc = conv2d_raw(x, h, pack_weight(w, prec=fp), 4 * Hc, ks, 3 if b is not None else 0,
                        None if b is None else b.detach().contiguous(), None, 0, prec=fp)
        h_out = torch.empty((N, Hc, H, W), device=x.device, dtype=torch.float32)
        c_out = torch.empty((N, Hc, H, W), device=x.device, dtype=torch.float32)
        gates = torch.empty((N, 4 * Hc, H, W), device=x.device, dtype=torch.float32)
        cp, cns = L.frames(c, "c")
        hp, hns = L.frames(h_out, "h_out")
        cop, cons = L.frames(c_out, "c_out")
        pe = [None if t is None else t.detach().reshape(-1).contiguous() for t in (wci, wcf, wco)]
        L.call("rfn_convlstm_gates_fwd_f32", L.dev(cc), cp, _l(cns), L.dev(pe[0]), L.dev(pe[1]), L.dev(pe[2]), hp,
               _l(hns), cop, _l(cons), L.dev(gates), _i(N), _i(Hc), _i(HW), meta=_shell("convlstm_gates_fwd", gates, 2.75))
        ctx.save_for_backward(x, h, c, w, gates, c_out, *[t for t in pe if t is not None])
        ctx.has_bias = b is not None
        ctx.has_pe = pe[0] is not None
        return h_out, c_out

    @staticmethod
    def backward(ctx, gh, gc):
        saved = ctx.saved_tensors
        x, h, c, w, gates, c_out = saved[:6]
        pe = list(saved[6:9]) if ctx.has_pe else [None, None, None]
        N, Cx, H, W = x.shape
        Hc = int(w.shape[0]) // 4
        ks = int(w.shape[2])
        HW = H * W
        gcc = torch.empty_like(gates)
        gc_prev = torch.empty_like(c)
        cp, cns = L.frames(c, "c")
        cop, cons = L.frames(c_out, "c_out")
        ghp, ghns = (None, 0) if gh is None else L.frames(gh.contiguous(), "gh")
        gcp, gcns = (None, 0) if gc is None else L.frames(gc.contiguous(), "gc")
        gpp, gpns = L.frames(gc_prev, "gc_prev")
        L.call("rfn_convlstm_gates_bwd_f32", L.dev(gates), cp, _l(cns), cop, _l(cons), ghp, _l(ghns), gcp, _l(gcns),
               L.dev(pe[0]), L.dev(pe[1]), L.dev(pe[2]), L.dev(gcc), gpp, _l(gpns), _i(N), _i(Hc), _i(HW),
               meta=_shell("convlstm_gates_bwd", gates, 3.25))
        gb = conv_epilogue_bwd(None, gcc, None, 3, 0, want_gl=False)[1] if ctx.has_bias else None
        gw = conv2d_wgrad(x, h, gcc, 4 * Hc, ks)
        gx = torch.empty_like(x)
        ghp_ = torch.empty_like(h)
        conv2d_raw(gcc, None, pack_weight(w, True), Cx + Hc, ks, 0, None, None, 0, out1=gx, out2=ghp_, cout_split=Cx)
        return gx, ghp_, gc_prev, gw, gb, None, None, None


def convlstm_seq_supported(w, Cx, H, W):
    """time-batched ConvLSTM path: 3x3 kernel on a small map, both weight halves dense-packable"""
    Hc = int(w.shape[0]) // 4
    return (smallmap_arith_ok(H, W) and tuple(w.shape[2:]) == (3, 3) and H * W <= 16 and (Cx * H * W) % 8 == 0
            and (Hc * H * W) % 8 == 0 and int(w.shape[1]) == Cx + Hc)


class ConvLSTMSeqFn(torch.autograd.Function):
    """A whole ConvLSTM sequence (Utils/modules.py:396-414 looping :355-377) on a small map as ONE autograd node.
    conv(cat(x_t, h_{t-1})) = Wx*x_t + Wh*h_{t-1}: the input projection of all S steps is one time-batched dense
    product, only Wh*h_{t-1} (Hc of the Cx+Hc input channels) stays in the recurrence; the backward pass keeps only
    Wh^T*gcc_t in its loop and computes the input gradient, the weight gradient and the bias gradient once over the S
    steps.  x_all [S,B,Cx,H,W] -> h_all [S,B,Hc,H,W], c_S."""

    @staticmethod
    def forward(ctx, x_all, h0, c0, w, b):
        S, B, Cx, H, W = (int(v) for v in x_all.shape)
        Hc, HW = int(w.shape[0]) // 4, H * W
        x_all, h0, c0 = x_all.contiguous(), h0.contiguous(), c0.contiguous()
        wd = w.detach()
        wx, wh = wd[:, :Cx].contiguous(), wd[:, Cx:].contiguous()
        pk_xf, pk_hf = smallmap_pack(wx, H, W, False), smallmap_pack(wh, H, W, False)
        pre = smallmap_dense(x_all.view(S * B, Cx, H, W), pk_xf, 4 * Hc, bias=None if b is None else b.detach())
        pre = pre.view(S, B, 4 * Hc, H, W)
        h_all = torch.empty((S, B, Hc, H, W), device=x_all.device, dtype=torch.float32)
        c_all = torch.empty((S, B, Hc, H, W), device=x_all.device, dtype=torch.float32)
        gates = torch.empty((S, B, 4 * Hc, H, W), device=x_all.device, dtype=torch.float32)
        h_prev, c_prev = h0, c0
        ns = Hc * HW
        for t in range(S):
            cc = smallmap_dense(h_prev, pk_hf, 4 * Hc, add=pre[t])
            L.call("rfn_convlstm_gates_fwd_f32", L.dev(cc), L.dev(c_prev), _l(ns), None, None, None, L.dev(h_all[t]),
                   _l(ns), L.dev(c_all[t]), _l(ns), L.dev(gates[t]), _i(B), _i(Hc), _i(HW),
                   meta=_shell("convlstm_gates_fwd", gates[t], 2.75))
            h_prev, c_prev = h_all[t], c_all[t]
        ctx.save_for_backward(x_all, h0, c0, w, h_all, c_all, gates)
        ctx.has_bias = b is not None
        return h_all, c_all[S - 1]

    @staticmethod
    def backward(ctx, g_hall, g_cS):
        x_all, h0, c0, w, h_all, c_all, gates = ctx.saved_tensors
        S, B, Cx, H, W = (int(v) for v in x_all.shape)
        Hc, HW = int(w.shape[0]) // 4, H * W
        ns = Hc * HW
        wd = w.detach()
        wx, wh = wd[:, :Cx].contiguous(), wd[:, Cx:].contiguous()
        pk_xb, pk_hb = smallmap_pack(wx, H, W, True), smallmap_pack(wh, H, W, True)
        g_hall = None if g_hall is None else g_hall.contiguous()
        gcc = torch.empty_like(gates)
        gh_rec, gc = None, (None if g_cS is None else g_cS.contiguous())
        for t in range(S - 1, -1, -1):
            if g_hall is None:
                gh = gh_rec
            elif gh_rec is None:
                gh = g_hall[t]
            else:
                gh = g_hall[t] + gh_rec
            c_prev = c_all[t - 1] if t > 0 else c0
            gc_prev = torch.empty_like(c0)
            L.call("rfn_convlstm_gates_bwd_f32", L.dev(gates[t]), L.dev(c_prev), _l(ns), L.dev(c_all[t]), _l(ns),
                   L.dev(gh), _l(ns if gh is not None else 0), L.dev(gc), _l(ns if gc is not None else 0), None, None, None,
                   L.dev(gcc[t]), L.dev(gc_prev), _l(ns), _i(B), _i(Hc), _i(HW),
                   meta=_shell("convlstm_gates_bwd", gates[t], 3.25))
            gh_rec = smallmap_dense(gcc[t], pk_hb, Hc)
            gc = gc_prev
        G = gcc.view(S * B, 4 * Hc, H, W)
        gx_all = smallmap_dense(G, pk_xb, Cx).view(S, B, Cx, H, W) if ctx.needs_input_grad[0] else None
        h_prev_all = torch.cat([h0.unsqueeze(0), h_all[:S - 1]], 0).view(S * B, Hc, H, W)
        gw = conv2d_wgrad(x_all.view(S * B, Cx, H, W), h_prev_all, G, 4 * Hc, 3)
        gb = G.sum(dim=(0, 2, 3)) if ctx.has_bias else None
        return gx_all, gh_rec, gc, gw, gb


class StepBatchNormActFn(torch.autograd.Function):
    """Training-mode BatchNorm2d with per-timestep statistics on a step-major time-batched tensor [S*B, C, H, W],
    fused with the activation that follows it (rfn_stepbn_{fwd,bwd}_f32: two launches each way).  Returns (y, mean[S, C],
    biased var[S, C]).  act: 0 none, 1 relu, 2 leaky_relu(slope), 3 tanh.  running (optional) = (running_mean, running_var,
    coef[S], coef_u[S], decay, num_batches_tracked or None): the S exponential-average updates of the step-wise calls,
    applied by the same launch (r <- decay r + sum_s coef[s] stat[s]); without it the caller does them."""

    @staticmethod
    def forward(ctx, x, gamma, beta, S, eps, act, slope, running=None):
        ctx.set_materialize_grads(False)  # mean / var are not differentiable: no zero gradients built for them
        x = x.contiguous()
        from . import dist as rdist
        if rdist.sync_batchnorm_on():
            return StepBatchNormActFn._forward_sync(ctx, x, gamma, beta, S, eps, act, slope, running)
        ctx.world = 1
        SB, C, H, W = (int(v) for v in x.shape)
        B, HW = SB // S, H * W
        mean = torch.empty((S, C), device=x.device, dtype=torch.float32)
        var = torch.empty((S, C), device=x.device, dtype=torch.float32)
        y = torch.empty_like(x)
        gm = None if gamma is None else gamma.detach().contiguous()
        bt = None if beta is None else beta.detach().contiguous()
        nscr = int(L.load().rfn_stepbn_scratch_floats(S, B, C))  # partial sums of the split reductions (either way)
        acc = torch.empty((nscr,), device=x.device, dtype=torch.float32)
        rm = rv = cf = cfu = nbt = None
        decay = 1.0
        if running is not None:
            rm, rv, cf, cfu, decay, nbt = running
            if nbt is not None and (nbt.dtype != torch.int64 or not nbt.is_cuda):
                raise RuntimeError("num_batches_tracked must be an int64 device tensor")
        L.call("rfn_stepbn_fwd_f32", L.dev(x), L.dev(gm), L.dev(bt), L.dev(y), L.dev(mean), L.dev(var), L.dev(acc),
               L.dev(rm), L.dev(rv), L.dev(cf), L.dev(cfu), ctypes.c_float(decay),
               None if nbt is None else ctypes.c_void_p(nbt.data_ptr()), _i(S), _i(B), _i(C), _i(HW), ctypes.c_float(eps),
               _i(act), ctypes.c_float(slope), meta=_shell("stepbn_fwd", x, 3))
        ctx.save_for_backward(x, mean, var, gm, bt)
        ctx.cfg = (S, B, C, HW, eps, act, slope, gamma is not None, nscr)
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def _forward_sync(ctx, x, gamma, beta, S, eps, act, slope, running):
        """synchronised BatchNorm (rfn_hip.dist.sync_batchnorm_on): the statistics of every step are those of the GLOBAL
        batch -- local moments, one all-gather of [2, S, C] floats, equal-count combination, apply with the given
        statistics; the running statistics follow the global moments (torch ops on [C] vectors)."""
        import torch.distributed as dist
        from . import dist as rdist
        SB, C, H, W = (int(v) for v in x.shape)
        B, HW = SB // S, H * W
        world = dist.get_world_size()
        mean = torch.empty((S, C), device=x.device, dtype=torch.float32)
        var = torch.empty((S, C), device=x.device, dtype=torch.float32)
        y = torch.empty_like(x)
        gm = None if gamma is None else gamma.detach().contiguous()
        bt = None if beta is None else beta.detach().contiguous()
        nscr = int(L.load().rfn_stepbn_scratch_floats(S, B, C))
        acc = torch.empty((nscr,), device=x.device, dtype=torch.float32)
        L.call("rfn_stepbn_fwd_f32", L.dev(x), L.dev(gm), L.dev(bt), L.dev(y), L.dev(mean), L.dev(var), L.dev(acc),
               None, None, None, None, ctypes.c_float(1.0), None, _i(S), _i(B), _i(C), _i(HW), ctypes.c_float(eps),
               _i(act), ctypes.c_float(slope))
        st = rdist.all_gather_cat(torch.stack((mean, var)).unsqueeze(0))          # [world, 2, S, C]
        mean = st[:, 0].mean(0).contiguous()
        var = (st[:, 1].mean(0) + (st[:, 0] - mean).pow(2).mean(0)).contiguous()  # equal counts per rank
        L.call("rfn_stepbn_apply_f32", L.dev(x), L.dev(gm), L.dev(bt), L.dev(y), L.dev(mean), L.dev(var), _i(S), _i(B),
               _i(C), _i(HW), ctypes.c_float(eps), _i(act), ctypes.c_float(slope))
        if running is not None:
            rm, rv, cf, cfu, decay, nbt = running
            n = B * HW * world
            cfu_g = cf * (n / max(n - 1, 1))          # unbiased variance of the global batch
            rm.mul_(decay).add_((cf.view(-1, 1) * mean).sum(0))
            rv.mul_(decay).add_((cfu_g.view(-1, 1) * var).sum(0))
            if nbt is not None:
                nbt.add_(S)
        ctx.save_for_backward(x, mean, var, gm, bt)
        ctx.cfg = (S, B, C, HW, eps, act, slope, gamma is not None, nscr)
        ctx.world = world
        ctx.mark_non_differentiable(mean, var)
        return y, mean, var

    @staticmethod
    def backward(ctx, g, _gm, _gv):
        x, mean, var, gm, bt = ctx.saved_tensors
        S, B, C, HW, eps, act, slope, affine, nscr = ctx.cfg
        if g is None:
            return (None,) * PACK_SLOTS
        g = g.contiguous()
        sums = torch.empty((nscr,), device=x.device, dtype=torch.float32)
        gx = torch.empty_like(x)
        ggamma = torch.empty((C,), device=x.device, dtype=torch.float32) if affine else None
        gbeta = torch.empty((C,), device=x.device, dtype=torch.float32) if affine else None
        args = (L.dev(x), L.dev(gm), L.dev(bt), L.dev(g), L.dev(mean), L.dev(var), L.dev(sums), L.dev(gx), L.dev(ggamma),
                L.dev(gbeta), _i(S), _i(B), _i(C), _i(HW), ctypes.c_float(eps), _i(act), ctypes.c_float(slope))
        if ctx.world > 1:   # synchronised: partial sums, added over the ranks, apply
            from . import dist as rdist
            L.call("rfn_stepbn_bwd_f32", *args, _i(1), _i(ctx.world))
            rdist.all_reduce_sum_(sums)
            L.call("rfn_stepbn_bwd_f32", *args, _i(2), _i(ctx.world))
        else:
            L.call("rfn_stepbn_bwd_f32", *args, _i(0), _i(1))
        return gx, ggamma, gbeta, None, None, None, None, None
