"""BASELINE config 1 — "TwoMoon 2D RealNVP density (data_generators/halfmoon.py) on CPU PyTorch — plumbing, no GPU".

The reference's two-moons experiment lives in a notebook (Notebooks/TwoMoonFlows.ipynb, code cells 8-9: `AffineCoupling`
with an MLP 2 -> 64 -> 64 -> 2 on the masked input, six couplings alternating the masked coordinate, standard-normal
prior, Adam lr 5e-3) on `RotatingTwoMoonsConditionalSampler.conditioned_sample` (data_generators/halfmoon.py:14-24:
sklearn.make_moons, centred by (0.5, 0.25), rotated by theta).  SURVEY.md §2 row 19 scopes it as a restatement in the
test directory: it is not product code and not on the hot path; it checks that the flow plumbing this repository shares
with that experiment (affine coupling forward / reverse / log-det bookkeeping, NLL training) behaves.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn as nn


def two_moons(n, theta, noise, seed):
    """data_generators/halfmoon.py:14-24 with a pinned generator"""
    from sklearn.datasets import make_moons
    X, y = make_moons(n_samples=n, shuffle=True, noise=noise, random_state=seed)
    X = X - np.array([0.5, 0.25])
    c, s = math.cos(theta), math.sin(theta)
    R = np.array([[c, -s], [s, c]])
    return torch.from_numpy((X @ R.T).astype(np.float32)), torch.from_numpy(y)


class Coupling(nn.Module):
    """notebook cell 8: x' = x * exp(log_s(x*b)) + t(x*b) on the unmasked coordinate; returns (x', log_s [B, 2])"""

    def __init__(self, keep, hidden=64):
        super().__init__()
        self.register_buffer("mask", torch.tensor([1.0, 0.0] if keep == "x_dim" else [0.0, 1.0]))
        self.mlp = nn.Sequential(nn.Linear(2, hidden), nn.ReLU(), nn.Linear(hidden, hidden), nn.ReLU(), nn.Linear(hidden, 2))

    def forward(self, x, reverse=False):
        b = self.mask.expand(x.shape[0], 2)
        log_s, t = self.mlp(x * b).split(1, dim=1)
        t, log_s = t * (1.0 - b), log_s * (1.0 - b)
        x = (x - t) * torch.exp(-log_s) if reverse else x * torch.exp(log_s) + t
        return x, log_s


class RealNVP(nn.Module):
    def __init__(self, n=6):
        super().__init__()
        self.transforms = nn.ModuleList([Coupling("y_dim" if i % 2 == 0 else "x_dim") for i in range(n)])

    def flow(self, x):
        z, log_det = x, torch.zeros_like(x)
        for op in self.transforms:
            z, d = op(z)
            log_det = log_det + d
        return z, log_det

    def invert_flow(self, z):
        for op in reversed(self.transforms):
            z, _ = op(z, reverse=True)
        return z

    def log_prob(self, x):
        z, log_det = self.flow(x)
        return log_det.sum(1) + torch.distributions.Normal(0.0, 1.0).log_prob(z).sum(1)

    def nll(self, x):
        return -self.log_prob(x).mean()


def test_two_moons_realnvp_plumbing():
    torch.manual_seed(0)
    torch.set_num_threads(2)
    x, y = two_moons(2000, theta=0.3 * math.pi, noise=0.05, seed=1)
    xt, _ = two_moons(500, theta=0.3 * math.pi, noise=0.05, seed=2)
    assert x.shape == (2000, 2) and x.dtype == torch.float32 and set(y.tolist()) == {0, 1}
    m = RealNVP()
    # bijection and change of variables before any training: g(f(x)) = x, log-det = log |det J|
    z, ld = m.flow(x[:64])
    assert float((m.invert_flow(z) - x[:64]).abs().max()) < 1e-4
    J = torch.autograd.functional.jacobian(lambda v: m.flow(v.unsqueeze(0))[0].squeeze(0), x[0])
    assert abs(float(torch.linalg.slogdet(J)[1]) - float(ld[0].sum())) < 1e-4
    nll0 = float(m.nll(xt))
    opt = torch.optim.Adam(m.parameters(), lr=5e-3)
    for step in range(800):
        idx = torch.randint(0, 2000, (500,))
        loss = m.nll(x[idx])
        opt.zero_grad()
        loss.backward()
        opt.step()
    nll1 = float(m.nll(xt))
    assert nll1 == nll1 and nll1 < nll0 - 0.8, (nll0, nll1)                      # the density has been learnt
    z, _ = m.flow(xt)
    assert float((m.invert_flow(z) - xt).abs().max()) < 1e-3                   # still a bijection after training
    # samples land on the moons: most of them within 0.25 of a training point
    with torch.no_grad():
        s = m.invert_flow(torch.randn(200, 2, generator=torch.Generator().manual_seed(3)))
        d = torch.cdist(s, x).min(1).values
    assert float((d < 0.25).float().mean()) > 0.7
