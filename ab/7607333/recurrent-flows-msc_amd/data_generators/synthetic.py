"""SM-MNIST-shaped synthetic video (no dataset or network here): `num_digits` 28x28 soft blobs bouncing in a
64x64 frame with a random velocity re-drawn on wall contact, like data_generators/stochasticMovingMnist.py:62-111 of
the reference does with MNIST digits.  Output contract identical to the reference datasets: `__getitem__` ->
[T, C, H, W] float32 in [0, 1], deterministic in the sample index."""
import numpy as np
import torch
from torch.utils.data import Dataset


class SyntheticMovingMNIST(Dataset):
    def __init__(self, seq_len=20, image_size=64, digit_size=28, num_digits=2, step_length=4, channels=1,
                 length=10000, seed=0):
        self.seq_len, self.image_size, self.digit_size = seq_len, image_size, digit_size
        self.num_digits, self.step_length, self.channels, self.length, self.seed = (num_digits, step_length, channels,
                                                                                    length, seed)
        ax = np.arange(digit_size, dtype=np.float32) - (digit_size - 1) / 2.0
        self._r2 = ax[:, None] ** 2 + ax[None, :] ** 2

    def __len__(self):
        return self.length

    def _glyph(self, rng):
        """an anisotropic ring-ish blob: bright, sparse, values in [0,1] like a digit stroke"""
        s = rng.uniform(4.0, 7.0)
        ring = rng.uniform(3.0, 8.0)
        g = np.exp(-((np.sqrt(self._r2) - ring) ** 2) / (2 * (s / 3.0) ** 2))
        return (g / g.max()).astype(np.float32)

    def __getitem__(self, index):
        rng = np.random.RandomState(self.seed * 1000003 + index)
        S, D, T = self.image_size, self.digit_size, self.seq_len
        x = np.zeros((T, S, S), dtype=np.float32)
        for _ in range(self.num_digits):
            g = self._glyph(rng)
            sx, sy = rng.randint(S - D), rng.randint(S - D)
            dx, dy = rng.randint(-self.step_length, self.step_length + 1), rng.randint(-self.step_length,
                                                                                       self.step_length + 1)
            for t in range(T):
                if sy < 0:
                    sy, dy = 0, rng.randint(1, self.step_length + 1)
                    dx = rng.randint(-self.step_length, self.step_length + 1)
                elif sy >= S - D:
                    sy, dy = S - D - 1, rng.randint(-self.step_length, 0)
                    dx = rng.randint(-self.step_length, self.step_length + 1)
                if sx < 0:
                    sx, dx = 0, rng.randint(1, self.step_length + 1)
                    dy = rng.randint(-self.step_length, self.step_length + 1)
                elif sx >= S - D:
                    sx, dx = S - D - 1, rng.randint(-self.step_length, 0)
                    dy = rng.randint(-self.step_length, self.step_length + 1)
                x[t, sy:sy + D, sx:sx + D] += g
                sy += dy
                sx += dx
        x = np.clip(x, 0.0, 1.0)
        out = torch.from_numpy(x).unsqueeze(1)
        if self.channels == 3:
            out = out.repeat(1, 3, 1, 1)
        return out
