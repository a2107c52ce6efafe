#!/usr/bin/env python3
"""Kernel-level timing on the GPU box (no oracle): each C-ABI kernel alone at bench-like sizes.
    python tools/perf_kernels.py [mtransform] [gemm] [spmm] ...
Prints ms and achieved GB/s (algorithmic bytes) / TFLOP/s per kernel; used to iterate on one kernel
without paying for the whole bench."""
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from tmgcn_amd import ops, synth  # noqa: E402

dev = "cuda"
K = ops.kernels


def timeit(fn, reps=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True)
    e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def mtransform(T=16, N=2_000_000, F=128, band=20, tl=0):
    X = torch.rand(T, N, F, device=dev)
    op = ops.MOperator(synth.band_M(T, band, "matlab"), dev)
    for tr in (False, True):
        ms = timeit(lambda: K.mtransform(op, X, transpose=tr, y_group_rows=0 if tr else tl, x_group_rows=tl if tr else 0))
        print(f"mtransform T={T} N={N} F={F} band={band} transpose={tr} tl={tl}: {ms:.2f} ms  "
              f"{2 * X.numel() * 4 / ms / 1e6:.0f} GB/s")


def gemm(T=16, N=2_000_000, Kf=128, Nf=128):
    A = torch.rand(T, N, Kf, device=dev)
    W = torch.randn(Kf, Nf, device=dev)
    dY = torch.rand(T, N, Nf, device=dev)
    fl = 2.0 * T * N * Kf * Nf
    by = (A.numel() + T * N * Nf) * 4
    for name, fn in (("gemm", lambda: K.gemm(A, W)), ("gemm_dA", lambda: K.gemm(dY, W, trans_w=True)),
                     ("gemm_dW", lambda: K.gemm_dw(A, dY, False))):
        ms = timeit(fn)
        print(f"{name} R={T * N} {Kf}x{Nf}: {ms:.2f} ms  {fl / ms / 1e9:.1f} TFLOP/s  {by / ms / 1e6:.0f} GB/s")


def spmm(T=4, N=2_000_000, F=128, deg=32):
    A = synth.device_er_csr(T, N, deg, dev)
    X = torch.rand(T, N, F, device=dev)
    W = torch.randn(F, F, device=dev) * 0.1
    by = A.nnz * (8 + F * 4 + (4 + F * 4) / (A.nnz / A.n_rows))
    ms = timeit(lambda: K.spmm(A, X))
    print(f"spmm T={T} N={N} F={F}: {ms:.2f} ms  {by / ms / 1e6:.0f} GB/s (algorithmic)")
    ms = timeit(lambda: K.spmm_gemm(A, X, W))
    print(f"spmm_gemm (no AX): {ms:.2f} ms  {by / ms / 1e6:.0f} GB/s (P2 bytes only)")
    ms = timeit(lambda: K.spmm_gemm(A, X, W, want_ax=True))
    print(f"spmm_gemm (+AX store): {ms:.2f} ms  {by / ms / 1e6:.0f} GB/s (P2 bytes only)")


def wide(T=4, N=1_000_000, deg=32):
    """Feature widths beyond the fused kernel's K <= 128: the plain SpMM (vec4 up to F = 256, the generic kernel above) and
    the standalone GEMM (bf16-split, k-chunks above 128) — what a layer of that width runs as two launches."""
    A = synth.device_er_csr(T, N, deg, dev)
    for F in (64, 128, 256, 512):
        X = torch.rand(T, N, F, device=dev)
        by = A.nnz * (8 + F * 4 + (4 + F * 4) / (A.nnz / A.n_rows))
        ms = timeit(lambda: K.spmm(A, X))
        line = f"wide F={F}: spmm {ms:.2f} ms {by / ms / 1e6:.0f} GB/s"
        W = torch.randn(F, F, device=dev) * 0.1
        msg = timeit(lambda: K.gemm(X, W))
        line += f" | gemm {F}x{F} {msg:.2f} ms {2 * X.numel() * 4 / msg / 1e6:.0f} GB/s {2.0 * T * N * F * F / msg / 1e9:.1f} TFLOP/s"
        if K.spmm_gemm_supported(F, F):
            msf = timeit(lambda: K.spmm_gemm(A, X, W))
            line += f" | fused {msf:.2f} ms"
        print(line, flush=True)
        del X


def mtransform_dense(T=128, N=250_000, F=128):
    import numpy as np
    X = torch.rand(T, N, F, device=dev)
    op = ops.MOperator(synth.band_M(T, 20, "matlab"), dev).inverse()  # dense lower-triangular
    for tr in (False, True):
        ms = timeit(lambda: K.mtransform(op, X, transpose=tr))
        fl = 2.0 * (T * (T + 1) / 2) * N * F
        print(f"mtransform Minv (lower-triangular dense) T={T} N={N} F={F} transpose={tr}: {ms:.2f} ms  "
              f"{2 * X.numel() * 4 / ms / 1e6:.0f} GB/s  {fl / ms / 1e9:.1f} TFLOP/s")


if __name__ == "__main__":
    which = sys.argv[1:] or ["mtransform", "gemm", "spmm"]
    for w in which:
        if w == "mtransform":
            mtransform()
            mtransform(T=128, N=250_000, tl=16)
            mtransform(T=128, N=250_000)
        elif w == "gemm":
            gemm()
        elif w == "spmm":
            spmm()
        elif w == "wide":
            wide()
        elif w == "dense":
            mtransform_dense()
            mtransform_dense(T=64, N=20000, F=6)
