"""Synthetic image stacks of the benchmark configurations, generated on the
device so that multi-GiB stacks never cross PCIe.  Harness code, not part of the solver."""
from __future__ import annotations

import math

import numpy as np


def texture_parameters(n, seed, n_modes=64):
    """Fourier-mode parameters of the exactly translating "actin-like" texture; the draw order
    (f, g, a, phi from numpy's default_rng(seed)) is part of the recipe."""
    rng = np.random.default_rng(seed)
    fm = max(2, n // 16)
    f = rng.integers(-fm, fm + 1, n_modes).astype(np.float64)
    g = rng.integers(-fm, fm + 1, n_modes).astype(np.float64)
    a = rng.random(n_modes) + 0.5
    phi = 2 * np.pi * rng.random(n_modes)
    return f, g, a, phi


def texture_stack_torch(n, n_frames, seed, device, shift=(0.3, 0.6), first_frame=0, n_modes=64, out=None):
    """(n_frames, n, n) float64 stack on ``device``:
    ``I_t = clip(0.5 + 0.45 sqrt(K)/(3 sum a) * sum_k a_k cos(2 pi (f_k (i - sx t) + g_k (j - sy t))/n + phi_k), 0, 1)``.
    True flow = ``shift`` pixels/frame, true remodelling = 0."""
    import torch
    f, g, a, phi = (torch.from_numpy(v).to(device) for v in texture_parameters(n, seed, n_modes))
    i = torch.arange(n, dtype=torch.float64, device=device)
    scale = 0.45 * math.sqrt(n_modes) / (3.0 * float(a.sum()))
    if out is None:
        out = torch.empty((n_frames, n, n), dtype=torch.float64, device=device)
    for t in range(n_frames):
        tt = first_frame + t
        p = 2 * math.pi * f[:, None] * (i[None, :] - shift[0] * tt) / n + phi[:, None]
        q = 2 * math.pi * g[:, None] * (i[None, :] - shift[1] * tt) / n
        A = (a[:, None] * torch.cos(p)).T @ torch.cos(q) - (a[:, None] * torch.sin(p)).T @ torch.sin(q)
        out[t] = torch.clamp(0.5 + scale * A, 0.0, 1.0)
    return out


def texture_stack_numpy(n, n_frames, seed=0, shift=(0.3, 0.6), first_frame=0, n_modes=64):
    """Host (numpy) version of ``texture_stack_torch``: same recipe, same parameters."""
    f, g, a, phi = texture_parameters(n, seed, n_modes)
    i = np.arange(n, dtype=np.float64)
    scale = 0.45 * math.sqrt(n_modes) / (3.0 * float(a.sum()))
    out = np.empty((n_frames, n, n))
    for t in range(n_frames):
        tt = first_frame + t
        p = 2 * math.pi * f[:, None] * (i[None, :] - shift[0] * tt) / n + phi[:, None]
        q = 2 * math.pi * g[:, None] * (i[None, :] - shift[1] * tt) / n
        A = (a[:, None] * np.cos(p)).T @ np.cos(q) - (a[:, None] * np.sin(p)).T @ np.sin(q)
        out[t] = np.clip(0.5 + scale * A, 0.0, 1.0)
    return out
