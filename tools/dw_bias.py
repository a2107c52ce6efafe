#!/usr/bin/env python3
"""Signed error of the dW kernels against an fp64 product as the reduction length R grows.
A rounding error that is random grows like sqrt(R); a systematic one (a truncating accumulator)
grows like R and shows up as a non-zero MEAN signed error over the K x Nf outputs.

    python tools/dw_bias.py            -> JSON lines: operand distribution, R, algo, mean signed / max error
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmgcn_amd import ops  # noqa: E402

K = Nf = 128
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
for dist_name in ("A~U(0,1) dY~N(0,1)", "A~N(0,1) dY~N(0,1)", "A~U(0,1) dY~U(0,1)"):
    for R in (1 << 20, 1 << 22, 1 << 24, 1 << 25):
        A = torch.rand(1, R, K, device=dev, generator=g) if "A~U" in dist_name else torch.randn(1, R, K, device=dev, generator=g)
        dY = torch.rand(1, R, Nf, device=dev, generator=g) if "dY~U" in dist_name else torch.randn(1, R, Nf, device=dev, generator=g)
        ref = torch.zeros(K, Nf, dtype=torch.float64, device=dev)
        step = 1 << 20
        for r in range(0, R, step):
            ref += A[0, r:r + step].double().t() @ dY[0, r:r + step].double()
        for algo in ("auto", "f32mfma"):
            got = ops.kernels.gemm_dw(A, dY, False, algo=algo).double()
            err = got - ref
            print(json.dumps({"operands": dist_name, "R": R, "algo": algo,
                              "mean_signed_err": float(err.mean()), "max_abs_err": float(err.abs().max()),
                              "max_abs_ref": float(ref.abs().max()),
                              "max_rel": float(err.abs().max() / ref.abs().max()),
                              "mean_signed_rel": float(err.mean() / ref.abs().max())}), flush=True)
        del A, dY
