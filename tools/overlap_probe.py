#!/usr/bin/env python3
"""Does the power-bound dW kernel hide under the HBM-bound backward gather when the two run on separate
HIP streams?  S4 shard (T = 16, N = 2 M, F = 128), backward of the fused layer:
   dX = (Âᵀ·dY)·Wᵀ  (spmm_gemm on the transposed CSR)   and   dW = AXᵀ·dY  (gemm_dw)  — independent.
Times, interleaved in one process: sequential; dW launched first on a side stream; the gather launched
first with `grid_reserve` block slots left free and dW on the side stream.
   python tools/overlap_probe.py [--slices 16] [--nodes 2000000]"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tmgcn_amd import ops, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--slices", type=int, default=16)
ap.add_argument("--nodes", type=int, default=2_000_000)
ap.add_argument("--feat", type=int, default=128)
ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
dev = "cuda"
T, N, F = a.slices, a.nodes, a.feat
A = synth.device_er_csr(T, N, 32, dev)
At = A.transpose()
g = torch.Generator(device=dev).manual_seed(0)
dY = torch.randn(T, N, F, device=dev, generator=g)
AX = torch.rand(T, N, F, device=dev, generator=g)
W = torch.randn(F, F, device=dev, generator=g) * 0.1
dX = torch.empty(T, N, F, device=dev)
K = ops.kernels
main = torch.cuda.current_stream()
side = torch.cuda.Stream()


def gather(reserve=0):
    K.spmm_gemm(At, dY, W, trans_w=True, out=(dX, None, None), grid_reserve=reserve)


def sequential():
    gather()
    return K.gemm_dw(AX, dY, per_slice=False)


def dw_first():
    side.wait_stream(main)
    with torch.cuda.stream(side):
        dW = K.gemm_dw(AX, dY, per_slice=False)
    gather()
    main.wait_stream(side)
    return dW


def gather_first(reserve):
    def run():
        ready = torch.cuda.Event()
        ready.record(main)                 # dW needs AX and dY (ready here), not the gather's result
        gather(reserve)
        side.wait_event(ready)
        with torch.cuda.stream(side):
            dW = K.gemm_dw(AX, dY, per_slice=False)
        main.wait_stream(side)
        return dW
    return run


modes = {"sequential": sequential, "dW first on a side stream": dw_first}
for r in (0, 128, 256, 512):
    modes[f"gather first, grid_reserve={r}, dW on a side stream"] = gather_first(r)
ref = sequential()
torch.cuda.synchronize()
res = {k: [] for k in modes}
for rep in range(a.reps + 1):
    for name, fn in modes.items():
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        dW = fn()
        e.record()
        torch.cuda.synchronize()
        if rep:
            res[name].append(s.elapsed_time(e))
        assert torch.equal(dW, ref), name       # the overlap changes when, never what
out = {"workload": f"T={T} N={N} F={F} deg=32+1", "ms": {k: {"median": round(statistics.median(v), 3), "min": round(min(v), 3)} for k, v in res.items()}}
print(json.dumps(out, indent=1))
