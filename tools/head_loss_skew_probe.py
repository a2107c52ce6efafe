#!/usr/bin/env python3
"""The one-pass head + loss kernel on labelled edges whose endpoints are SKEWED (a few hub nodes incident to a large share of
the edges, as in real interaction graphs) against uniformly random endpoints of the same count — the S2 shape (T = 65,
N = 3 800, E = 3.25 M labelled edges, F = 6, C = 2), loss + all gradients in one launch.   python tools/head_loss_skew_probe.py"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tmgcn_amd import ops  # noqa: E402

dev = "cuda"
T, N, E, F, C = 65, 3800, 3_250_000, 6, 2
rng = np.random.default_rng(0)
Z0 = torch.randn(T, N, F, device=dev)
U0 = torch.randn(2 * F, C, device=dev)
w = torch.tensor([0.9, 0.1], device=dev)
for kind in ("uniform", "zipf 1.0", "zipf 1.5"):
    t = rng.integers(0, T, E)
    if kind == "uniform":
        i, j = rng.integers(0, N, E), rng.integers(0, N, E)
    else:
        a = float(kind.split()[1])
        p = np.arange(1, N + 1, dtype=np.float64) ** (-a)
        p /= p.sum()
        i, j = rng.choice(N, E, p=p), rng.integers(0, N, E)
    edges = ops.EdgeIndex(torch.from_numpy(np.stack([t, i, j])), N, dev, T=T)
    target = torch.from_numpy(rng.integers(0, C, E)).to(dev)
    deg = np.bincount(t * N + i, minlength=T * N) + np.bincount(t * N + j, minlength=T * N)
    Z, U = Z0.clone().requires_grad_(True), U0.clone().requires_grad_(True)

    def step():
        Z.grad = U.grad = None
        ops.head_loss(Z, edges, U, target, w).backward()

    for _ in range(3):
        step()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        step()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e))
    ts.sort()
    print(f"{kind:9s}: most labelled-edge entries on one row {deg.max():>8,d} (mean {deg.mean():.1f}): loss + gradients {ts[3] * 1e3:8.1f} us", flush=True)
