#!/usr/bin/env python3
"""Timing of the fused layer-1+2 kernels (csrc/layer12.hip) at a reference-shaped config, over the lanes-per-row choice
(the `avg_nnz_per_row` hint picks G) and against the unfused pair.   python tools/ab_layer12.py [S1|S3]"""
import os
import statistics
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from tmgcn_amd import _lib, ops, synth  # noqa: E402
from tmgcn_amd.csr import BatchedCSR  # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else "S1"
g = synth.dynamic_graph(**synth.CONFIGS[cfg], seed=0)
A = BatchedCSR.from_coo_list(g.At_list(), N=g.N, device="cuda")
At = A.transpose()
T, N = g.T, g.N
gen = torch.Generator(device="cuda").manual_seed(0)
H = torch.randn(T, N, 2, device="cuda", generator=gen)
W1 = torch.randn(2, 6, device="cuda", generator=gen)
W2 = torch.randn(6, 6, device="cuda", generator=gen)
dZ = torch.randn(T, N, 6, device="cuda", generator=gen)
print(f"{cfg}: T={T} N={N} rows={T * N} nnz={A.nnz} avg={A.avg_nnz_per_row:.2f}")
o = ops.kernels.ops
SELU = _lib.ACT_IDS["selu"]


def timeit(fn, reps=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        torch.cuda.synchronize()
        ts.append(s.elapsed_time(e) * 1e3)
    return statistics.median(ts)


for avg in (1.0, 3.0, 6.0, 12.0, 24.0):
    w1, w2 = W1.clone().requires_grad_(True), W2.clone().requires_grad_(True)

    def fwd():
        return o.layer12(H, w1, w2, A.rowptr, A.col, A.val, At.rowptr, At.col, At.val, N, avg, SELU, 0)

    Z = fwd()

    def bwd():
        w1.grad = None
        w2.grad = None
        Z.backward(dZ, retain_graph=True)

    print(f"hint {avg:5.1f}: fwd {timeit(fwd):7.1f} us   bwd (dW1 kernel + dW2 kernel) {timeit(bwd):7.1f} us")
with torch.no_grad():
    print(f"unfused forward pair: {timeit(lambda: ops.spmm_feature_gemm(A, ops.feature_gemm(H, W1, act='selu'), W2)):7.1f} us")
    print(f"no-grad fused forward: {timeit(lambda: ops.layer12(H, W1, 'selu', A, W2)):7.1f} us")
