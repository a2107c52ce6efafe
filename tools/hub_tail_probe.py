#!/usr/bin/env python3
"""What ONE very long row costs a launch: a one-slice S4 launch (N = 2 M, 33 entries per row, F = 128: 66 M entries) with one
row replaced by a hub of H entries, H = 0 / 1e4 / 1e5 / 1e6.  A long row runs on the four waves of ONE block (csrc/spmm_row.h);
heavy tiles are taken first, so inside a big launch the hub hides under everything else — but a launch cannot be shorter than
its longest row, unless the giant-row plan (csr.BatchedCSR.giant_plan) has it summed chunk by chunk by many blocks in front of
the launch.  Prints the fused kernel's time per H, with and without the plan.    python tools/hub_tail_probe.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tmgcn_amd import ops, synth  # noqa: E402
from tmgcn_amd.csr import BatchedCSR  # noqa: E402

dev = "cuda"
N, F = 2_000_000, 128
base = synth.device_er_csr(1, N, 32, dev)
X = torch.rand(1, N, F, device=dev)
W = torch.randn(F, F, device=dev) * 0.1
g = torch.Generator(device=dev).manual_seed(1)
for H in (0, 10_000, 100_000, 1_000_000, 4_000_000):
    if H:
        r = 777_777
        a, b = int(base.rowptr[r]), int(base.rowptr[r + 1])
        hub = torch.sort(torch.randint(0, N, (H,), device=dev, dtype=torch.int32, generator=g)).values
        col = torch.cat([base.col[:a], hub, base.col[b:]])
        val = torch.cat([base.val[:a], torch.full((H,), 1.0 / H, device=dev), base.val[b:]])
        rowptr = base.rowptr.clone()
        rowptr[r + 1:] += H - (b - a)
        A = BatchedCSR(rowptr, col, val, 1, N)
    else:
        A = base
    line = f"hub of {H:>9,d} entries in a {A.nnz / 1e6:.0f} M-entry launch:"
    for plan in (True, False):        # with the giant-row plan (rows > 32 768 entries summed chunk by chunk in front of the launch) / without
        if not plan:
            A._blocks["giant"] = (None, None)
        for _ in range(2):
            ops.kernels.spmm_gemm(A, X, W)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            ops.kernels.spmm_gemm(A, X, W)
            e.record()
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e))
        ts.sort()
        line += f"  {'with' if plan else 'without'} plan {ts[2]:7.3f} ms"
    print(line, flush=True)
