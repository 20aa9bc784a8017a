"""GPU parity of the individual gfx950 kernels (through the C ABI via ctypes) against the CPU oracle / plain
PyTorch fp32 references on seeded inputs.  Integer/index work is bit exact; float work within 1e-4 relative
(north_star), with the tolerance written at each check."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["mixed", "bf16x3", "f32"], autouse=True)
def conv_precision(request):
    """every test runs with the three arithmetics of the convolutions (RFN_CONV_PRECISION): the shipped 'mixed' path
    (forward fp32-grade: fused f16x3s kernel / fp32 MFMA, gradients bf16x3), all-bf16x3 and all-fp32-MFMA."""
    from rfn_hip import ops
    old = ops.CONV_PRECISION
    ops.CONV_PRECISION = request.param
    yield request.param
    ops.CONV_PRECISION = old

from oracle import rfn_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def K():
    from rfn_hip import ops
    assert torch.cuda.is_available(), "GPU tests need a device"
    return ops


def cu(t):
    return t.cuda()


def ctol():
    """conv tolerance factor: the split-precision (bf16x3) MFMA path carries ~1e-5 relative error per contraction
    (2^-16 per product), the fp32 MFMA path ~1e-6; thresholds below are written for fp32 and scaled by this."""
    from rfn_hip import ops
    return 1.0 if ops.CONV_PRECISION == "f32" else 5.0


def relerr(a, b):
    """max-norm relative error: max |a - b| / max |b| (accumulated fp32 sums of different order agree norm-wise, not
    element-wise next to zero crossings); per-sample log-likelihoods are checked element-wise in test_hip_modules"""
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize("shape", [(2, 3, 4, 6), (3, 1, 64, 64), (5, 16, 2, 2)])
def test_squeeze2d_bit_exact(K, shape):
    g = torch.Generator().manual_seed(0)
    x = torch.randn(shape, generator=g)
    y = K.squeeze2d_raw(cu(x), False)
    assert torch.equal(y.cpu(), O.squeeze2d(x))
    xb = K.squeeze2d_raw(y, True)
    assert torch.equal(xb.cpu(), x)
    # strided (channel-slice) input, as produced by Split2d
    big = torch.randn(shape[0], 2 * shape[1], shape[2], shape[3], generator=g)
    v = cu(big)[:, : shape[1]]
    assert torch.equal(K.squeeze2d_raw(v, False).cpu(), O.squeeze2d(big[:, : shape[1]]))


def test_squeeze2d_golden(K, golden):
    f = golden("modules.pt")["squeeze"]
    assert torch.equal(K.squeeze2d_raw(cu(f["x"]), False).cpu(), f["y"])
    assert torch.equal(K.squeeze2d_raw(cu(f["y"]), True).cpu(), f["x_back"])


def test_channel_stats(K):
    g = torch.Generator().manual_seed(1)
    x = torch.randn(7, 5, 6, 3, generator=g) * 3 + 1.5
    mean, var = K.channel_stats(cu(x))
    xt = x.transpose(0, 1).reshape(5, -1)
    torch.testing.assert_close(mean.cpu(), xt.mean(1), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(var.cpu(), xt.var(1, unbiased=True), rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("N,C,H,W", [(3, 4, 5, 3), (2, 8, 4, 4), (5, 64, 2, 2), (2, 12, 8, 8), (70, 2, 2, 2)])
def test_actnorm_invconv_fwd_bwd_rev(K, N, C, H, W):
    g = torch.Generator().manual_seed(2)
    x = torch.randn(N, C, H, W, generator=g, requires_grad=True)
    b = torch.randn(C, generator=g) * 0.3
    l = torch.randn(C, generator=g) * 0.2
    Wm = torch.linalg.qr(torch.randn(C, C, generator=g))[0] + 0.05 * torch.randn(C, C, generator=g)
    b.requires_grad_(True); l.requires_grad_(True); Wm.requires_grad_(True)
    y = (x + b.view(1, C, 1, 1)) * l.view(1, C, 1, 1).exp()
    z = torch.einsum("oc,bchw->bohw", Wm, y)
    gz = torch.randn(z.shape, generator=g)
    z.backward(gz)
    zk = K.actnorm_invconv_fwd(cu(x.detach()), cu(b.detach()), cu(l.detach()), cu(Wm.detach()))
    assert relerr(zk, z) < 1e-5
    gx, gW, gb, gl = K.actnorm_invconv_bwd(cu(x.detach()), cu(b.detach()), cu(l.detach()), cu(Wm.detach()), cu(gz))
    assert relerr(gx, x.grad) < 1e-5
    assert relerr(gW, Wm.grad) < 1e-4
    assert relerr(gb, b.grad) < 1e-4
    assert relerr(gl, l.grad) < 1e-4
    xr = K.invconv_actnorm_rev(zk, cu(b.detach()), cu(l.detach()), cu(torch.inverse(Wm.detach())))
    assert relerr(xr, x) < 1e-4


CONV_CASES = [
    # N, C1, C2, Cout, H, W, ks
    (2, 5, 0, 7, 6, 6, 3),       # odd sizes, non power-of-two map
    (3, 2, 16, 256, 32, 32, 3),  # level-0 conv1 shape (z1 | cond)
    (2, 256, 0, 256, 16, 16, 1),  # conv2
    (2, 256, 0, 4, 32, 32, 3),   # level-0 conv3 (tiny Cout)
    (9, 32, 256, 256, 2, 2, 3),  # level-4 conv1
    (5, 256, 0, 64, 2, 2, 3),    # level-4 conv3
    (3, 12, 0, 40, 4, 4, 1),     # 1x1 with Cin not multiple of 8, Cout 33..64 config
    (1, 3, 0, 130, 8, 8, 3),     # Cout > 128 (two cout blocks), Cin < 8
    (2, 9, 0, 20, 64, 64, 3),    # W = 64 (two column tiles)
    (2, 40, 0, 33, 3, 5, 1),     # 3x5 map
    (32, 512, 200, 800, 2, 2, 3),  # ConvLSTM conv: few pixels, huge K -> 32-pixel tiles + split-K (atomic combine)
    (6, 100, 0, 200, 4, 4, 1),   # few-pixel 1x1, split-K
    (70, 256, 0, 256, 16, 16, 1),   # weight-stationary 1x1 kernel (>= 16384 pixels, Cout 256, 128 < Cin <= 256)
    (131, 200, 0, 256, 12, 12, 1),  # same kernel: Cin not a multiple of 16, non power-of-two map, ragged last tile
    (20, 2, 16, 256, 32, 32, 3),    # weight-stationary 3x3 kernel (Cin <= 24, Cout 256, >= 16384 pixels): level-0 conv1
    (70, 4, 14, 256, 16, 16, 3),    # same kernel, 2 x 16 pixel tiles
    (300, 10, 0, 256, 8, 8, 3),     # same kernel, 4 x 8 pixel tiles, single source, Cin not a multiple of 8
    (70, 4, 32, 256, 16, 16, 3),    # same kernel, 45-unit variant (24 < Cin <= 40): level-1 conv1
    (70, 6, 12, 512, 16, 16, 3),    # weight-stationary 3x3 with two 256-channel output blocks (Hd = 512)
    (70, 200, 0, 512, 16, 16, 1),   # weight-stationary 1x1 with two output blocks
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_dgrad_wgrad(K, case):
    N, C1, C2, Cout, H, W, ks = case
    g = torch.Generator().manual_seed(3)
    Cin = C1 + C2
    x1 = torch.randn(N, C1, H, W, generator=g)
    x2 = torch.randn(N, C2, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, ks, ks, generator=g) / (Cin * ks * ks) ** 0.5
    xin = (torch.cat([x1, x2], 1) if C2 else x1).clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = F.conv2d(xin, wr, None, padding=ks // 2)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    out = K.conv2d_raw(cu(x1), cu(x2) if C2 else None, K.pack_weight(cu(w)), Cout, ks)
    assert relerr(out, ref) < 2e-5 * ctol(), "forward"
    # data gradient through the same kernel with flipped/transposed packing, split into the two sources
    g1 = torch.empty(N, C1, H, W, device="cuda")
    g2 = torch.empty(N, C2, H, W, device="cuda") if C2 else None
    K.conv2d_raw(cu(gy), None, K.pack_weight(cu(w), True), Cin, ks, out1=g1, out2=g2, cout_split=C1)
    assert relerr(g1, xin.grad[:, :C1]) < 2e-5 * ctol(), "dgrad in1"
    if C2:
        assert relerr(g2, xin.grad[:, C1:]) < 2e-5 * ctol(), "dgrad in2"
    gw = K.conv2d_wgrad(cu(x1), cu(x2) if C2 else None, cu(gy), Cout, ks)
    assert relerr(gw, wr.grad) < 1e-4, "wgrad"  # both arithmetics


@pytest.mark.parametrize("N,Cmid,Cnext,S,ks,act", [(70, 256, 8, 16, 3, 2), (70, 256, 256, 16, 1, 2), (66, 512, 4, 16, 3, 1),
                                                   (3, 64, 6, 8, 3, 2), (5, 128, 64, 4, 1, 1)])
def test_dgrad_with_fused_activation_backward(K, N, Cmid, Cnext, S, ks, act):
    """rfn_conv2d_dgrad_act_bf16x3: data gradient of a conv fused with the backward of the producer's ActNorm +
    activation (gu, grad bias, grad logs) against autograd of  y = act((u + b) * exp(l)),  o = conv(y, w).  The first
    three cases have >= 16384 pixels and take the weight-stationary kernels (per-workgroup partial sums)."""
    if K.CONV_PRECISION != "bf16x3":
        pytest.skip("fused epilogue exists in the split-precision kernels only")
    g = torch.Generator().manual_seed(33)
    u = torch.randn(N, Cmid, S, S, generator=g).requires_grad_(True)
    b = (torch.randn(Cmid, generator=g) * 0.3).requires_grad_(True)
    l = (torch.randn(Cmid, generator=g) * 0.2).requires_grad_(True)
    w = torch.randn(Cnext, Cmid, ks, ks, generator=g) / (Cmid * ks * ks) ** 0.5
    pre = (u + b.view(1, -1, 1, 1)) * l.view(1, -1, 1, 1).exp()
    y = F.relu(pre) if act == 1 else F.leaky_relu(pre, 0.2)
    o = F.conv2d(y, w, None, padding=ks // 2)
    go = torch.randn(o.shape, generator=g)
    o.backward(go)
    gu, gb, gl = K.conv2d_dgrad_act(cu(go), K.pack_weight(cu(w), True), cu(y.detach()), cu(l.detach()), act, Cmid, ks)
    assert relerr(gu, u.grad) < 3e-5
    assert relerr(gb, b.grad) < 1e-4
    assert relerr(gl, l.grad) < 1e-4


@pytest.mark.parametrize("ep_mode,act", [(1, 1), (1, 2), (1, 0), (2, 0), (3, 0)])
def test_conv_epilogues_fwd_bwd(K, ep_mode, act):
    g = torch.Generator().manual_seed(4)
    N, Cin, Cout, H, W = 3, 6, 10, 4, 4
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.2).requires_grad_(True)
    p0 = (torch.randn(Cout, generator=g) * 0.3).requires_grad_(True)
    p1 = (torch.randn(Cout, generator=g) * 0.2).requires_grad_(True)
    xr = x.clone().requires_grad_(True)
    u = F.conv2d(xr, w, None, padding=1)
    if ep_mode == 1:
        ref = O.act_fun((u + p0.view(1, -1, 1, 1)) * p1.view(1, -1, 1, 1).exp(), ["none", "relu", "leakyrelu"][act]) \
            if act else (u + p0.view(1, -1, 1, 1)) * p1.view(1, -1, 1, 1).exp()
    elif ep_mode == 2:
        ref = (u + p0.view(1, -1, 1, 1)) * (3 * p1.view(1, -1, 1, 1)).exp()
    else:
        ref = u + p0.view(1, -1, 1, 1)
    gy = torch.randn(ref.shape, generator=g)
    ref.backward(gy)
    xk = cu(x).requires_grad_(True)
    wk = cu(w.detach()).requires_grad_(True)
    p0k = cu(p0.detach()).requires_grad_(True)
    p1k = cu(p1.detach()).requires_grad_(True) if ep_mode != 3 else None
    out = K.conv_ep(xk, None, wk, p0k, p1k, ep_mode, act)
    assert relerr(out, ref) < 2e-5 * ctol()
    out.backward(cu(gy))
    assert relerr(xk.grad, xr.grad) < 5e-5 * ctol()
    assert relerr(wk.grad, w.grad) < 1e-4
    assert relerr(p0k.grad, p0.grad) < 1e-4
    if ep_mode != 3:
        assert relerr(p1k.grad, p1.grad) < 1e-4


@pytest.mark.parametrize("clamp", ["realnvp", "glow", "softclamp", "none"])
def test_affine_kernel_fwd_rev(K, clamp):
    g = torch.Generator().manual_seed(5)
    N, C, H, W = 4, 6, 3, 5
    z = torch.randn(N, C, H, W, generator=g)
    o = torch.randn(N, C, H, W, generator=g)
    sd = {"scale": torch.randn(C // 2, 1, 1, generator=g) * 0.5, "scale_shift": torch.randn(C // 2, 1, 1, generator=g) * 0.1}
    shift, s = O.split_feature(o, "cross")
    ls = O.clamp_log_scale(sd, "", s, clamp)
    ref = torch.cat((z[:, : C // 2], (z[:, C // 2:] + shift) * ls.exp()), 1)
    ref_ld = ls.sum(dim=[1, 2, 3])
    zk = cu(z.clone())
    ld = torch.zeros(N, device="cuda")
    K.affine_coupling_(zk, cu(o), cu(sd["scale"].reshape(-1)), cu(sd["scale_shift"].reshape(-1)), ld, K.CLAMP[clamp], False)
    assert relerr(zk, ref) < 1e-5
    assert relerr(ld, ref_ld) < 1e-5
    K.affine_coupling_(zk, cu(o), cu(sd["scale"].reshape(-1)), cu(sd["scale_shift"].reshape(-1)), ld, K.CLAMP[clamp], True)
    assert relerr(zk, z) < 1e-5
    assert float(ld.abs().max()) < 1e-4


@pytest.mark.parametrize("N,Cin,Cout,S,split,acc1,acc2", [
    (70, 256, 18, 32, 2, True, False),    # level 0 of the canonical flow: dgrad of conv1 (256 -> 2 + 16)
    (70, 256, 36, 16, 4, True, True),     # level 1 (256 -> 4 + 32), condition gradient accumulating
    (3, 32, 18, 32, 18, False, False),    # single output tensor
    (2, 64, 50, 32, 6, False, True),      # two output tiles at W = 32 (BAIR-like widths)
    (5, 32, 7, 16, 3, True, False),       # two chunks, one output tile at W = 16
    (609, 256, 18, 32, 2, True, True),    # more frames than CUs x 2, odd count
])
def test_conv3x3_smallcout_dgrad_kernel(K, N, Cin, Cout, S, split, acc1, acc2):
    """rfn_conv3x3_smallcout_bf16x3 (csrc/dgrad_small.hip) as the data gradient of a forward conv [Cin -> Cout'] with
    weight w [Cin, Cout, 3, 3]: against torch's conv_transpose2d / conv2d on the CPU in fp32 (N >= 70 shapes at the
    BASELINE channel counts included).  bf16x3 arithmetic: 5e-5 of the output range."""
    if not K.bwd_b3():
        pytest.skip("bf16x3 backward arithmetic only")
    g = torch.Generator().manual_seed(70 + Cout)
    Nc = min(N, 8)  # the CPU reference covers the first frames and, for large N, the last ones
    x = torch.randn(N, Cin, S, S, generator=g)
    w = torch.randn(Cin, Cout, 3, 3, generator=g) * 0.05   # FORWARD weight of a conv Cout -> Cin
    base1 = torch.randn(N, split, S, S, generator=g)
    base2 = torch.randn(N, Cout - split, S, S, generator=g) if Cout > split else None
    out1 = cu(base1.clone())
    out2 = cu(base2.clone()) if base2 is not None else None
    wd = cu(w)
    K.conv3x3_smallcout(cu(x), K.pack_weight(wd, flip=True), Cout, out1, out2, split, acc1, acc2)

    def ref(sl):
        r = F.conv_transpose2d(x[sl], w, padding=1)  # = data gradient of conv2d(., w, padding=1)
        r1 = r[:, :split] + (base1[sl] if acc1 else 0)
        r2 = (r[:, split:] + (base2[sl] if acc2 else 0)) if base2 is not None else None
        return r1, r2

    for sl in (slice(0, Nc), slice(N - Nc, N)):
        r1, r2 = ref(sl)
        scale = float(r1.abs().max())
        assert float((out1[sl].cpu() - r1).abs().max()) < 5e-5 * scale + 1e-6
        if r2 is not None:
            assert float((out2[sl].cpu() - r2).abs().max()) < 5e-5 * float(r2.abs().max()) + 1e-6
    # and the generic kernel agrees (same arithmetic, different summation order)
    g1 = cu(base1.clone())
    g2 = cu(base2.clone()) if base2 is not None else None
    K.conv2d_raw(cu(x), None, K.pack_weight(wd, flip=True), Cout, 3, 0, None, None, 0, out1=g1, out2=g2,
                 cout_split=split, acc1=acc1, acc2=acc2)
    assert relerr(out1, g1) < 2e-5
    if g2 is not None:
        assert relerr(out2, g2) < 2e-5


@pytest.mark.parametrize("clamp", ["realnvp", "glow", "softclamp", "none"])
@pytest.mark.parametrize("N,C,H,W", [(3, 4, 6, 6), (2, 8, 4, 4), (5, 64, 2, 2)])
def test_fused_shell_tail_fwd_bwd(K, clamp, N, C, H, W):
    """rfn_gather_affine_f32 / rfn_affine_zeros_bwd_f32 (the shell around conv3 of a Glow step) against autograd of the
    oracle's Conv2dZeros epilogue + affine coupling (glow_modules.py:119-121, 276-285)."""
    from rfn_hip import lib as L
    import ctypes
    g = torch.Generator().manual_seed(15)
    Ch = C // 2
    z = torch.randn(N, C, H, W, generator=g)
    P = torch.randn(N, 9 * C, H, W, generator=g) * 0.3
    b3 = torch.randn(C, generator=g) * 0.2
    l3 = (torch.randn(C, generator=g) * 0.1).requires_grad_(True)
    sd = {"scale": (torch.randn(Ch, 1, 1, generator=g) * 0.5).requires_grad_(True),
          "scale_shift": (torch.randn(Ch, 1, 1, generator=g) * 0.1).requires_grad_(True)}
    # reference: 3x3 shift-and-add of the tap-expanded output = what conv3x3 would give
    conv = torch.zeros(N, C, H, W)
    Pp = F.pad(P.view(N, 9, C, H, W), (1, 1, 1, 1))
    for t in range(9):
        conv = conv + Pp[:, t, :, t // 3:t // 3 + H, t % 3:t % 3 + W]
    conv = conv.requires_grad_(True)
    b3r = b3.clone().requires_grad_(True)
    zr = z.clone().requires_grad_(True)
    o_ref = (conv + b3r.view(1, -1, 1, 1)) * (l3.view(1, -1, 1, 1) * 3).exp()
    shift, s_ = O.split_feature(o_ref, "cross")
    ls = O.clamp_log_scale(sd, "", s_, clamp)
    out_ref = torch.cat((zr[:, :Ch], (zr[:, Ch:] + shift) * ls.exp()), 1)
    ld_ref = ls.sum(dim=[1, 2, 3])
    sc, sh = cu(sd["scale"].detach().reshape(-1)), cu(sd["scale_shift"].detach().reshape(-1))
    b3c, l3c = cu(b3), cu(l3.detach())
    # forward, gathering from P
    zk = cu(z.clone())
    o_k, ld = K.gather_affine_(zk, None, cu(P), b3c, l3c, sc, sh, K.CLAMP[clamp])
    assert relerr(o_k, o_ref) < 1e-5
    assert relerr(zk, out_ref) < 1e-5
    assert relerr(ld, ld_ref) < 1e-5
    # forward, o given
    zk2 = cu(z.clone())
    o2, ld2 = K.gather_affine_(zk2, o_k, None, None, None, sc, sh, K.CLAMP[clamp])
    assert o2 is o_k and torch.equal(zk2, zk) and torch.equal(ld2, ld)
    # backward
    gout = torch.randn(N, C, H, W, generator=g)
    gld = torch.randn(N, generator=g)
    ((out_ref * gout).sum() + (ld_ref * gld).sum()).backward()
    goutc, gldc = cu(gout), cu(gld)
    gz = torch.empty_like(goutc)
    gpre = torch.empty_like(goutc)
    acc = torch.zeros(2 * Ch + 2 * C, device="cuda")
    gsc, gsh, gb3, gl3 = acc[:Ch], acc[Ch:2 * Ch], acc[2 * Ch:2 * Ch + C], acc[2 * Ch + C:]
    st = C * H * W
    rn = clamp == "realnvp"
    L.call("rfn_affine_zeros_bwd_f32", L.dev(zk), ctypes.c_long(st), L.dev(o_k), ctypes.c_long(st), L.dev(goutc),
           ctypes.c_long(st), L.dev(gldc), L.dev(sc), L.dev(sh), L.dev(l3c), L.dev(gz), ctypes.c_long(st), L.dev(gpre),
           ctypes.c_long(st), L.dev(gsc, check_contiguous=False) if rn else None,
           L.dev(gsh, check_contiguous=False) if rn else None, L.dev(gb3, check_contiguous=False),
           L.dev(gl3, check_contiguous=False), ctypes.c_int(K.CLAMP[clamp]), ctypes.c_int(N), ctypes.c_int(C),
           ctypes.c_int(H * W))
    assert relerr(gz, zr.grad) < 1e-5
    assert relerr(gpre, conv.grad) < 1e-5
    assert relerr(gb3, b3r.grad) < 1e-4
    assert relerr(gl3, l3.grad) < 1e-4
    if rn:
        assert relerr(gsc, sd["scale"].grad.reshape(-1)) < 1e-4
        assert relerr(gsh, sd["scale_shift"].grad.reshape(-1)) < 1e-4


@pytest.mark.parametrize("clamp", ["realnvp", "glow", "softclamp", "none"])
def test_standalone_affine_backward_matches_fused_kernel(K, clamp):
    """rfn_affine_coupling_bwd_f32 (the stand-alone coupling backward of the C ABI; the host path uses the fused
    rfn_affine_zeros_bwd_f32) gives the same gz2, go and clamp-parameter gradients as the fused kernel with Conv2dZeros
    logs = 0 (exp(0) = 1: gpre == go), which test_fused_shell_tail_fwd_bwd checks against autograd."""
    from rfn_hip import lib as L
    import ctypes
    g = torch.Generator().manual_seed(23)
    N, C, H, W = 3, 6, 4, 5
    Ch, HW, st = C // 2, H * W, C * H * W
    zout, o, gout = (cu(torch.randn(N, C, H, W, generator=g)) for _ in range(3))
    gld = cu(torch.randn(N, generator=g))
    sc, sh = cu(torch.randn(Ch, generator=g) * 0.5), cu(torch.randn(Ch, generator=g) * 0.1)
    rn = clamp == "realnvp"
    ct, _l, _i = K.CLAMP[clamp], ctypes.c_long, ctypes.c_int
    gz_a = gout.clone()
    go_a = torch.empty_like(o)
    acc_a = torch.zeros(2, Ch, device="cuda")
    L.call("rfn_affine_coupling_bwd_f32", L.dev(zout), _l(st), L.dev(o), _l(st), L.dev(gout), _l(st), L.dev(gld),
           L.dev(sc), L.dev(sh), L.dev(gz_a), _l(st), L.dev(go_a), _l(st), L.dev(acc_a[0]) if rn else None,
           L.dev(acc_a[1]) if rn else None, _i(ct), _i(N), _i(C), _i(HW))
    gz_b, go_b = torch.empty_like(gout), torch.empty_like(o)
    acc_b = torch.zeros(2, Ch, device="cuda")
    junk = torch.zeros(2, C, device="cuda")
    l3 = torch.zeros(C, device="cuda")
    L.call("rfn_affine_zeros_bwd_f32", L.dev(zout), _l(st), L.dev(o), _l(st), L.dev(gout), _l(st), L.dev(gld), L.dev(sc),
           L.dev(sh), L.dev(l3), L.dev(gz_b), _l(st), L.dev(go_b), _l(st), L.dev(acc_b[0]) if rn else None,
           L.dev(acc_b[1]) if rn else None, L.dev(junk[0]), L.dev(junk[1]), _i(ct), _i(N), _i(C), _i(HW))
    assert torch.equal(gz_a[:, :Ch], gout[:, :Ch])      # first half untouched by the stand-alone kernel
    assert relerr(gz_a[:, Ch:], gz_b[:, Ch:]) < 1e-6 and torch.equal(gz_b[:, :Ch], gout[:, :Ch])
    assert relerr(go_a, go_b) < 1e-6
    if rn:
        assert relerr(acc_a, acc_b) < 1e-5


@pytest.mark.parametrize("layout,std_mode", [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gauss_logp_fwd_bwd_sample(K, layout, std_mode):
    g = torch.Generator().manual_seed(6)
    N, Cz, H, W = 3, 5, 4, 2
    z = torch.randn(N, Cz, H, W, generator=g, requires_grad=True)
    o = (torch.randn(N, 2 * Cz, H, W, generator=g)).requires_grad_(True)
    mean, raw = O.split_feature(o, "cross" if layout == 0 else "split")
    std = F.softplus(raw) + 1e-8 if std_mode == 0 else raw.exp()
    ref = O.normal_log_prob(z, mean, std).sum(dim=(1, 2, 3))
    gl = torch.randn(N, generator=g)
    ref.backward(gl)
    zk = cu(z.detach()).requires_grad_(True)
    ok = cu(o.detach()).requires_grad_(True)
    lp = K.GaussLogpFn.apply(zk, ok, layout, std_mode)
    assert relerr(lp, ref) < 1e-5
    lp.backward(cu(gl))
    assert relerr(zk.grad, z.grad) < 1e-5
    assert relerr(ok.grad, o.grad) < 1e-5
    eps = torch.randn(N, Cz, H, W, generator=g)
    zs = K.gauss_sample(cu(o.detach()), cu(eps), layout, std_mode, 0.7)
    assert relerr(zs, (mean + std * 0.7 * eps)) < 1e-5


@pytest.mark.parametrize("M,Nc,F_,H,view", [(256, 256, 100, 32, False), (256, 162, 100, 32, False), (200, 324, 400, 16, False),
                                            (256, 256, 3, 8, False), (256, 256, 131, 32, True), (36, 256, 100, 32, False),
                                            (72, 256, 401, 16, True), (144, 256, 1700, 8, False), (200, 256, 1700, 8, False)])
def test_gemm_wgrad_tilings(K, M, Nc, F_, H, view):
    """rfn_gemm_wgrad_bf16x3: gw[m][n] = sum over frames and pixels of a*b, against an fp64 einsum, on every kernel it
    selects: the LDS-DMA ring kernel (>= 100000 pixels, HW % 32 == 0, 256-column gradients: 256 x 256 and <= 64 x 256
    tiles; 131 frames of 32x32 = 4192 stages over 256 workgroups: uneven shares; `view`: operands are channel slices of
    wider tensors, i.e. frame strides that are not M * HW), the register-staged 8-wave 256 x 192 / 256 x 256 tilings
    (162 / 324 columns; 144 rows) and the 128 x 128 one."""
    if not K.bwd_b3():
        pytest.skip("split-precision GEMM only")
    g = torch.Generator().manual_seed(70)
    a = torch.randn(F_, M + (8 if view else 0), H, H, generator=g)
    b = torch.randn(F_, Nc + (4 if view else 0), H, H, generator=g)
    av, bv = a[:, :M], b[:, 4:] if view else b
    ref = torch.einsum("fmp,fnp->mn", av.flatten(2).double(), bv.flatten(2).double())
    ag, bg = cu(a), cu(b)
    gw = K.gemm_wgrad(ag[:, :M], bg[:, 4:] if view else bg, M, Nc)
    torch.cuda.synchronize()
    assert relerr(gw, ref.float()) < 2e-5


@pytest.mark.parametrize("N,Cin,Cout,H,W", [(5, 1, 16, 64, 64), (3, 3, 32, 64, 64), (2, 1, 16, 7, 5), (2, 4, 16, 8, 8)])
def test_conv3x3_few_input_channels(K, N, Cin, Cout, H, W):
    """rfn_conv3x3_fewcin_fwd_f32 / rfn_conv3x3_c1_wgrad16_f32 (the extractor's first convolution: 1 .. 4 image channels
    into 16 / 32 maps as plain fp32 FMAs) through rfn_hip.ops.ConvFn against F.conv2d and its gradients: fp32 arithmetic,
    so 1e-6 forward; the weight gradient sums 10^4 .. 10^5 products per entry in another order: 2e-5."""
    g = torch.Generator().manual_seed(90)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * 0.3)
    gy = torch.randn(N, Cout, H, W, generator=g)
    xr, wr = x.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = F.conv2d(xr, wr, padding=1)
    ref.backward(gy.double())
    xc, wc = cu(x).requires_grad_(True), cu(w).requires_grad_(True)
    assert K.fewcin_ok(xc, None, wc, 0)
    y = K.conv_ep(xc, None, wc, None, None, 0, 0)
    assert relerr(y, ref.detach()) < 1e-6
    y.backward(cu(gy))
    assert relerr(wc.grad, wr.grad) < 2e-5
    assert relerr(xc.grad, xr.grad) < 2e-5 * ctol()


@pytest.mark.parametrize("G,N,C1,C2,Cout,S,ks", [(3, 600, 8, 64, 256, 8, 3), (4, 40, 2, 16, 256, 32, 3), (3, 50, 4, 0, 64, 8, 3),
                                                 (3, 70, 256, 0, 256, 8, 1), (2, 60, 256, 0, 16, 8, 3), (3, 33, 16, 128, 256, 4, 3)])
def test_conv2d_wgrad_grouped(K, G, N, C1, C2, Cout, S, ks):
    """conv2d_wgrad_grouped: the weight gradients of G convolutions of one shape in one launch (the K steps of a flow
    level) against one fp64 F.conv2d weight gradient per group.  256 output channels on 8x8 / 32x32 maps with
    G * pixels >= 100000 take the grouped form of the LDS-DMA ring kernel (level 2 of the canonical flow at the bench's
    size; level 0 at a local batch of 2), the others the 128 x 128 tiles: implicit 3x3, plain 1x1, tap-scattered 3x3
    (few output channels), im2col 3x3 (4x4 maps)."""
    if not K.bwd_b3():
        pytest.skip("split-precision GEMM only")
    g = torch.Generator().manual_seed(72)
    in1 = [torch.randn(N, C1, S, S, generator=g) for _ in range(G)]
    in2 = [torch.randn(N, C2, S, S, generator=g) for _ in range(G)] if C2 else None
    gy = torch.randn(G, N, Cout, S, S, generator=g)
    refs = []
    for i in range(G):
        xin = in1[i] if in2 is None else torch.cat((in1[i], in2[i]), 1)
        w = torch.zeros(Cout, C1 + C2, ks, ks, dtype=torch.float64, requires_grad=True)
        F.conv2d(xin.double(), w, padding=ks // 2).backward(gy[i].double())
        refs.append(w.grad.float())
    gyc = cu(gy)
    gws = K.conv2d_wgrad_grouped([cu(t) for t in in1], None if in2 is None else [cu(t) for t in in2],
                                 [gyc[i] for i in range(G)], Cout, ks, g_stacked=gyc)
    torch.cuda.synchronize()
    for i in range(G):
        assert relerr(gws[i], refs[i]) < 2e-5, i


@pytest.mark.parametrize("N,C1,C2,Cout,S", [(100, 2, 16, 256, 32), (401, 4, 32, 256, 16), (131, 6, 0, 200, 32)])
def test_conv3x3_wgrad_implicit_big(K, N, C1, C2, Cout, S):
    """rfn_conv3x3_wgrad_implicit_bf16x3 at >= 100000 pixels and more than 128 output channels -- conv1's weight gradient
    at the two finest flow levels -- where the gradient image goes through the LDS-DMA ring and the shifted input planes
    through the register-staged double buffer (gemm_wgrad_dma_impl_kernel): two-source input whose first source is a
    channel slice of a wider tensor (z[:, :C/2]), 324 columns = two column tiles, uneven stage shares (401 frames of
    16x16 = 3208 stages over 128 workgroups), Cout not a multiple of the tile; against F.conv2d's weight gradient (fp64)."""
    if not K.bwd_b3():
        pytest.skip("split-precision GEMM only")
    g = torch.Generator().manual_seed(71)
    z = torch.randn(N, 2 * C1, S, S, generator=g)
    cond = torch.randn(N, C2, S, S, generator=g) if C2 else None
    gy = torch.randn(N, Cout, S, S, generator=g)
    xin = z[:, :C1] if cond is None else torch.cat((z[:, :C1], cond), 1)
    w = torch.zeros(Cout, C1 + C2, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xin.double(), w, padding=1).backward(gy.double())
    zg = cu(z)
    gw = K.conv2d_wgrad(zg[:, :C1], None if cond is None else cu(cond), cu(gy), Cout, 3)
    torch.cuda.synchronize()
    assert relerr(gw, w.grad.float()) < 2e-5


@pytest.mark.parametrize("B,Cin,Cout,H,W", [(32, 768, 256, 2, 2), (4, 256, 112, 2, 2), (3, 10, 6, 4, 4), (40, 16, 8, 1, 2),
                                             (5, 6, 10, 2, 4)])
@pytest.mark.parametrize("slope", [0.2, None])
def test_smallmap_dense_conv_fwd_bwd(K, B, Cin, Cout, H, W, slope):
    """rfn_smallmap_{pack,dense}_bf16x3: conv3x3(pad 1) + bias [+ leaky_relu] on H*W <= 16 maps as one dense product,
    and its backward (activation backward fused, data gradient through the transposed pack) -- the per-timestep
    prior / encoder layers (Utils/modules.py:216-244 called from RFN_new.py:167-179) -- against F.conv2d autograd."""
    if K.CONV_PRECISION != "bf16x3":
        pytest.skip("split-precision kernels only")
    g = torch.Generator().manual_seed(80)
    x = torch.randn(B, Cin, H, W, generator=g, requires_grad=True)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)).requires_grad_(True)
    b = torch.randn(Cout, generator=g)
    pre = F.conv2d(x, w, b, padding=1)
    ref = F.leaky_relu(pre, slope) if slope is not None else pre
    gout = torch.randn(B, Cout, H, W, generator=g)
    ref.backward(gout)
    gpre_ref = gout * ((pre > 0).float() + (pre <= 0).float() * slope) if slope is not None else gout
    pf, pb = K.smallmap_pack(cu(w.detach()), H, W, False), K.smallmap_pack(cu(w.detach()), H, W, True)
    y = K.smallmap_dense(cu(x.detach()), pf, Cout, bias=cu(b), slope_out=slope)
    assert relerr(y, ref) < 2e-5
    if slope is not None:
        gx, gpre = K.smallmap_dense(cu(gout), pb, Cin, y=y, slope_in=slope, want_a_out=True)
        assert relerr(gpre, gpre_ref) < 1e-6
    else:
        gx = K.smallmap_dense(cu(gout), pb, Cin)
    assert relerr(gx, x.grad) < 2e-5


def test_smallmap_dense_pair_equals_two_single_launches(K):
    """rfn_smallmap_dense_pair_bf16x3 (encoder and prior layer of a timestep in one launch) against two
    rfn_smallmap_dense_bf16x3 launches: forward with bias + leaky_relu, and the data gradient with activation backward."""
    if K.CONV_PRECISION != "bf16x3":
        pytest.skip("split-precision kernels only")
    g = torch.Generator().manual_seed(95)
    B, H, W = 32, 2, 2
    shapes = [(768, 256), (256, 112)]
    xs = [cu(torch.randn(B, ci, H, W, generator=g)) for ci, _ in shapes]
    ws = [cu(torch.randn(co, ci, 3, 3, generator=g) / (3 * ci ** 0.5)) for ci, co in shapes]
    bs = [cu(torch.randn(co, generator=g)) for _, co in shapes]
    pf = [K.smallmap_pack(w, H, W, False) for w in ws]
    pb = [K.smallmap_pack(w, H, W, True) for w in ws]
    y0, y1 = K.smallmap_dense_pair(xs[0], pf[0], 256, xs[1], pf[1], 112, bias0=bs[0], bias1=bs[1], slope_out0=0.2,
                                   slope_out1=None)
    r0 = K.smallmap_dense(xs[0], pf[0], 256, bias=bs[0], slope_out=0.2)
    r1 = K.smallmap_dense(xs[1], pf[1], 112, bias=bs[1])
    assert torch.equal(y0, r0) and torch.equal(y1, r1)
    g0, g1 = cu(torch.randn(B, 256, H, W, generator=g)), cu(torch.randn(B, 112, H, W, generator=g))
    gx0, gp0, gx1, gp1 = K.smallmap_dense_pair(g0, pb[0], 768, g1, pb[1], 256, y0=y0, y1=None, slope_in0=0.2,
                                               want_a_out=True)
    s0, sp0 = K.smallmap_dense(g0, pb[0], 768, y=y0, slope_in=0.2, want_a_out=True)
    s1 = K.smallmap_dense(g1, pb[1], 256)
    assert torch.equal(gx0, s0) and torch.equal(gp0, sp0) and torch.equal(gx1, s1) and gp1 is g1


@pytest.mark.parametrize("N,C1,C2,Cout,H,W,ep_mode,act,split", [(70, 32, 256, 256, 2, 2, 1, 2, None),
                                                                 (40, 16, 24, 64, 4, 4, 1, 1, None),
                                                                 (33, 256, 0, 64, 2, 2, 2, 0, None),
                                                                 (50, 64, 0, 40, 4, 4, 0, 0, 16),
                                                                 (5, 6, 2, 10, 2, 4, 3, 0, None)])
def test_smallmap_conv_interface(K, N, C1, C2, Cout, H, W, ep_mode, act, split):
    """rfn_smallmap_conv_bf16x3 (dense kernels behind conv2d_raw's contract: two-source input with frame strides, conv
    epilogues 0-3, split + accumulated outputs) against the reference formulation of Conv2dNorm / Conv2dZeros
    (glow_modules.py:119-121,139-142) and of a data gradient written into a channel-slice view."""
    if K.CONV_PRECISION != "bf16x3":
        pytest.skip("split-precision kernels only")
    g = torch.Generator().manual_seed(90)
    big = torch.randn(N, 2 * C1, H, W, generator=g)          # in1 is a channel-slice view, like z1 = x[:, :C/2]
    x1 = big[:, :C1]
    x2 = torch.randn(N, C2, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / (3 * (C1 + C2) ** 0.5)
    p0, p1 = torch.randn(Cout, generator=g) * 0.3, torch.randn(Cout, generator=g) * 0.2
    u = F.conv2d(torch.cat([x1, x2], 1) if C2 else x1, w, None, padding=1)
    if ep_mode == 1:
        ref = (u + p0.view(1, -1, 1, 1)) * p1.view(1, -1, 1, 1).exp()
        ref = O.act_fun(ref, ["none", "relu", "leakyrelu"][act]) if act else ref
    elif ep_mode == 2:
        ref = (u + p0.view(1, -1, 1, 1)) * (3 * p1.view(1, -1, 1, 1)).exp()
    elif ep_mode == 3:
        ref = u + p0.view(1, -1, 1, 1)
    else:
        ref = u
    pk = K.smallmap_pack(cu(w), H, W, False)
    bigk = cu(big)
    kw = dict(ep_mode=ep_mode, p0=cu(p0) if ep_mode else None, p1=cu(p1) if ep_mode in (1, 2) else None, act=act)
    if split is None:
        out = K.smallmap_conv(bigk[:, :C1], cu(x2) if C2 else None, pk, Cout, **kw)
        assert relerr(out, ref) < 3e-5
    else:
        base = torch.randn(N, 2 * split, H, W, generator=g)   # out1 = first half of a bigger tensor, accumulated into
        basek = cu(base)
        o2 = torch.empty(N, Cout - split, H, W, device="cuda")
        K.smallmap_conv(bigk[:, :C1], cu(x2) if C2 else None, pk, Cout, out1=basek[:, :split], out2=o2, cout_split=split,
                        acc1=True, **kw)
        assert relerr(basek[:, :split], base[:, :split] + ref[:, :split]) < 3e-5
        assert relerr(basek[:, split:], base[:, split:]) == 0
        assert relerr(o2, ref[:, split:]) < 3e-5


@pytest.mark.parametrize("res_q", [False, True])
@pytest.mark.parametrize("use", ["all", "kl_only", "no_kl"])
def test_latent_step_fwd_bwd(K, res_q, use):
    """rfn_latent_step_*: chunk + softplus + res_q shift + both rsamples + KL of RFN.loss's per-step glue
    (RFN_new.py:167-184,206-207) against the oracle's torch.distributions formulation, values and gradients."""
    import torch.distributions as td
    g = torch.Generator().manual_seed(60)
    B, Z, H, W = 3, 7, 2, 2
    enc = (torch.randn(B, 2 * Z, H, W, generator=g) * 1.5).requires_grad_(True)
    pri = (torch.randn(B, 2 * Z, H, W, generator=g) * 1.5).requires_grad_(True)
    with torch.no_grad():
        enc[0, Z, 0, 0] = 25.0  # softplus threshold branch
        pri[1, Z + 1, 1, 0] = 30.0
    ep, eq = torch.randn(B, Z, H, W, generator=g), torch.randn(B, Z, H, W, generator=g)
    em, eraw = enc.chunk(2, 1)
    pm, praw = pri.chunk(2, 1)
    es, ps = F.softplus(eraw), F.softplus(praw)
    if res_q:
        em = pm + em
    refs = [pm + ps * ep, em + es * eq, td.kl_divergence(td.Normal(em, es), td.Normal(pm, ps)), em, es]
    gouts = [torch.randn(B, Z, H, W, generator=g) for _ in range(5)]
    sel = {"all": [0, 1, 2, 3, 4], "kl_only": [2], "no_kl": [0, 1, 3, 4]}[use]
    sum((refs[i] * gouts[i]).sum() for i in sel).backward()
    ek, pk = cu(enc.detach()).requires_grad_(True), cu(pri.detach()).requires_grad_(True)
    outs = K.LatentStepFn.apply(ek, pk, cu(ep), cu(eq), res_q)
    for o, r in zip(outs, refs):
        assert relerr(o, r) < 1e-5
    # non-contiguous incoming gradients (as the slices of a cat's gradient are in RFN.loss)
    sum((outs[i] * cu(gouts[i].repeat_interleave(2, dim=-1))[..., ::2]).sum() for i in sel).backward()
    assert relerr(ek.grad, enc.grad) < 1e-5
    assert relerr(pk.grad, pri.grad) < 1e-5


@pytest.mark.parametrize("N,Cin,C,H,W", [(3, 40, 4, 8, 8), (2, 256, 8, 16, 16), (5, 24, 2, 3, 5)])
def test_tap_expanded_zeros_conv_fwd_wgrad(K, N, Cin, C, H, W):
    """Conv2dZeros with tiny Cout: 1x1 conv to 9C channels + tap gather == the 3x3 conv; scatter + 1x1 wgrad == wgrad"""
    g = torch.Generator().manual_seed(8)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = (torch.randn(C, Cin, 3, 3, generator=g) * 0.1).requires_grad_(True)
    b = torch.randn(C, generator=g) * 0.2
    l = torch.randn(C, generator=g) * 0.1
    ref = (F.conv2d(x, w, b, padding=1)) * torch.exp(3 * l).view(1, C, 1, 1)
    gpre = torch.randn(N, C, H, W, generator=g)
    F.conv2d(x, w, None, padding=1).backward(gpre)
    o = K.zeros_conv_fwd(cu(x), cu(w.detach()), cu(b), cu(l))
    assert relerr(o, ref) < 2e-5 * ctol()
    gw = K.zeros_conv_wgrad(cu(x), cu(gpre), C, 3)
    assert relerr(gw, w.grad) < 1e-4


@pytest.mark.parametrize("N,C,Cc,S,act", [(2, 4, 16, 32, 1), (4, 8, 32, 16, 2), (41, 4, 16, 32, 2), (3, 4, 20, 32, 1),
                                          (2, 16, 64, 8, 1), (600, 16, 64, 8, 2), (6, 16, 60, 8, 1),
                                          (2, 12, 64, 32, 1), (41, 12, 64, 32, 2)])
def test_coupling_po_fused_forward(K, N, C, Cc, S, act, conv_precision):
    """the fused coupling-net forward (csrc/coupling_po.hip: conv3x3 -> ActNorm -> act -> conv1x1 -> ActNorm -> act ->
    tap-expanded conv3x3, hidden activations handed over in registers, scaled two-piece fp16 arithmetic "f16x3s") against plain fp32
    torch ops on the CPU: h1, h2 and the Conv2dZeros output o = gather(P).  N=41 frames of 32x32 is 328 rounds, more
    than one per persistent workgroup; Cc=20 leaves padded input channels in the last 8-channel group.
    8x8 maps (level 2 of the canonical flow, Cin 72): two frames per 128-pixel round, 600 frames = 300 rounds, Cc=60 a
    ragged last channel group; C=12 / Cc=64 on 32x32 maps is level 0 of the BAIR-shaped flow (with_skip conditions).
    Tolerance: 2e-5 of the tensor's largest magnitude (fp32-equivalent arithmetic, different summation order)."""
    if conv_precision != "mixed":
        pytest.skip("the fused kernel is the forward path of the 'mixed' arithmetic")
    g = torch.Generator().manual_seed(21)
    Ch, Cin = C // 2, C // 2 + Cc
    z = torch.randn(N, C, S, S, generator=g)
    cond = torch.randn(N, Cc, S, S, generator=g)
    w1 = torch.randn(256, Cin, 3, 3, generator=g) * 0.05
    w2 = torch.randn(256, 256, 1, 1, generator=g) * 0.05
    w3 = torch.randn(C, 256, 3, 3, generator=g) * 0.05
    n1b, n1l = torch.randn(256, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
    n2b, n2l = torch.randn(256, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
    b3, l3 = torch.randn(C, generator=g) * 0.1, torch.randn(C, generator=g) * 0.1
    actf = (lambda t: F.relu(t)) if act == 1 else (lambda t: F.leaky_relu(t, 0.2))
    xin = torch.cat((z[:, :Ch], cond), 1)
    h1r = actf((F.conv2d(xin.double(), w1.double(), padding=1) + n1b.double().view(1, -1, 1, 1)) * n1l.double().exp().view(1, -1, 1, 1))
    h2r = actf((F.conv2d(h1r, w2.double()) + n2b.double().view(1, -1, 1, 1)) * n2l.double().exp().view(1, -1, 1, 1))
    orf = (F.conv2d(h2r, w3.double(), b3.double(), padding=1)) * torch.exp(3 * l3.double()).view(1, -1, 1, 1)
    assert K.coupling_po_ok(N, C, Cc, 256, S, S, w1, w3, any_size=True)
    w1c, w2c, w3c = cu(w1), cu(w2), cu(w3)
    plan = K.POPackPlan([(w1c, w2c, w3c)])
    plan.run()
    h1, h2, P, _ = K.coupling_po_fwd(cu(z), cu(cond), plan.bufs[0], cu(n1b), cu(n1l), cu(n2b), cu(n2l), C, act)
    o = torch.empty((N, C, S, S), device="cuda")
    from rfn_hip import lib as L
    import ctypes
    b3c, l3c = cu(b3), cu(l3)  # (kept alive across the launch)
    L.call("rfn_tap_gather_f32", L.dev(P), L.dev(b3c), L.dev(l3c), L.dev(o), ctypes.c_int(N), ctypes.c_int(C),
           ctypes.c_int(S), ctypes.c_int(S))
    torch.cuda.synchronize()
    assert relerr(h1, h1r) < 2e-5
    assert relerr(h2, h2r) < 2e-5
    assert relerr(o, orf) < 2e-5


@pytest.mark.parametrize("N,C,Cc,S,act", [(2, 4, 16, 32, 1), (4, 8, 32, 16, 2), (41, 4, 16, 32, 1), (3, 8, 32, 16, 1),
                                          (2, 16, 64, 8, 1), (600, 16, 64, 8, 2), (2, 12, 64, 32, 2), (41, 12, 64, 32, 1)])
def test_coupling_po_fused_backward(K, N, C, Cc, S, act, conv_precision):
    """the fused data-gradient chain of the coupling net (csrc/coupling_po.hip, BWD instantiation: conv3^T -> act' ->
    conv2^T -> act', the intermediate gradient handed over in registers, act' read from the forward kernel's 1-bit masks)
    against torch autograd in fp64 on the CPU: the gradients at the outputs of conv2 and conv1 (ga2, ga1), and the four
    ActNorm gradients that rfn_coupling_po_bwd_finish derives from the per-workgroup sums and the weight gradients
    (gnl[c] = sum_k w[c][k] gw[c][k] + nb[c] gnb[c]).  N=41 frames of 32x32 = 328 rounds: more than one per workgroup.
    8x8 maps / C=16 is level 2 of the canonical flow (two frames per round, 600 frames = 300 rounds), C=12 on 32x32 maps
    level 0 of the BAIR-shaped flow (two channel groups of the gradient image).
    Tolerance 2e-5 of the tensor's largest magnitude for the data gradients (f16x3s arithmetic), 1e-4 for the ActNorm
    gradients (they inherit the weight gradients' bf16x3 arithmetic)."""
    if conv_precision != "mixed":
        pytest.skip("the fused kernels are the path of the 'mixed' arithmetic")
    g = torch.Generator().manual_seed(23)
    Ch, Cin = C // 2, C // 2 + Cc
    z = torch.randn(N, C, S, S, generator=g)
    cond = torch.randn(N, Cc, S, S, generator=g)
    w1 = (torch.randn(256, Cin, 3, 3, generator=g) * 0.05)
    w2 = (torch.randn(256, 256, 1, 1, generator=g) * 0.05)
    w3 = (torch.randn(C, 256, 3, 3, generator=g) * 0.05)
    n1b, n1l = torch.randn(256, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
    n2b, n2l = torch.randn(256, generator=g) * 0.1, torch.randn(256, generator=g) * 0.1
    go = torch.randn(N, C, S, S, generator=g)
    assert K.coupling_po_ok(N, C, Cc, 256, S, S, w1, w3, any_size=True) and K.coupling_po_bwd_ok(N, C, S, S)
    w1c, w2c, w3c = cu(w1), cu(w2), cu(w3)
    plan = K.POPackPlan([(w1c, w2c, w3c)])
    plan.run()
    assert plan.bwd_bufs[0] is not None
    n1bc, n1lc, n2bc, n2lc = cu(n1b), cu(n1l), cu(n2b), cu(n2l)
    h1, h2, P, masks = K.coupling_po_fwd(cu(z), cu(cond), plan.bufs[0], n1bc, n1lc, n2bc, n2lc, C, act, want_masks=True)
    assert masks is not None
    ga2, ga1, part = K.coupling_po_bwd(cu(go), plan.bwd_bufs[0], n1lc, n2lc, masks, act)
    torch.cuda.synchronize()

    # Reference: fp64 autograd through the same network with the activation's branch (act' = 1 or the negative slope)
    # taken where the GPU's forward pass took it.  act' is discontinuous at 0, and among 10^7 hidden activations one or two
    # sit within the forward arithmetic's error of the kink (the forward test's tolerance: 2e-5 of the largest
    # magnitude); a branch taken the other way changes that element's gradient by a factor and, through conv2^T, all 256
    # channels of the next gradient at its pixel.  The branches themselves are checked first:
    # they may differ from the fp64 forward pass only inside that band, and only in a vanishing fraction of elements.
    slope = 0.0 if act == 1 else 0.2
    d = lambda t: t.double().clone().requires_grad_(True)
    w1d, w2d, w3d, n1bd, n1ld, n2bd, n2ld = d(w1), d(w2), d(w3), d(n1b), d(n1l), d(n2b), d(n2l)
    xin = torch.cat((z[:, :Ch], cond), 1).double()

    def act_like_gpu(y, h_gpu):
        on = h_gpu.detach().cpu() > 0
        flips = on != (y.detach() > 0)
        band = 2e-5 * float(y.detach().abs().max())
        assert int(flips.sum()) <= max(2, y.numel() // 1000000), int(flips.sum())
        assert not bool((flips & (y.detach().abs() > band)).any())
        return torch.where(on, y, slope * y)

    a1 = F.conv2d(xin, w1d, padding=1)
    a1.retain_grad()
    h1r = act_like_gpu((a1 + n1bd.view(1, -1, 1, 1)) * n1ld.exp().view(1, -1, 1, 1), h1)
    a2 = F.conv2d(h1r, w2d)
    a2.retain_grad()
    h2r = act_like_gpu((a2 + n2bd.view(1, -1, 1, 1)) * n2ld.exp().view(1, -1, 1, 1), h2)
    F.conv2d(h2r, w3d, padding=1).backward(go.double())
    assert relerr(ga2, a2.grad) < 2e-5
    assert relerr(ga1, a1.grad) < 2e-5
    # the weight gradients as the product computes them, then the finishing launch
    gw2 = K.conv2d_wgrad(h1, None, ga2, 256, 1)
    gw1 = K.conv2d_wgrad(cu(z)[:, :Ch], cu(cond), ga1, 256, 3)
    assert relerr(gw2, w2d.grad) < 1e-4 and relerr(gw1, w1d.grad) < 1e-4
    out = torch.empty((4, 256), device="cuda")
    K.coupling_po_bwd_finish([[part, w1c, gw1.contiguous(), n1bc, w2c, gw2.contiguous(), n2bc, out]])
    torch.cuda.synchronize()
    for got, ref in zip(out, (n1bd.grad, n1ld.grad, n2bd.grad, n2ld.grad)):
        assert relerr(got, ref) < 1e-4
