#!/usr/bin/env python3
"""Interleaved A/B timing (one process, one device) of the SpMM / fused SpMM+GEMM kernels across
library variants built by tools/ab_variants.sh.   python tools/ab_fused.py [variant ...]"""
import ctypes as C
import glob
import os
import statistics
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
from tmgcn_amd import synth  # noqa: E402

p, i32, i64 = C.c_void_p, C.c_int32, C.c_int64


def load(path):
    lib = C.CDLL(path)
    lib.tmgcn_spmm_gemm_f32_hint.argtypes = [p, p, p, p, i64, i32, i32, p, i32, i32, i64, i64, i32, p, p, p, i32, C.c_float, p]
    lib.tmgcn_spmm_csr_batched_f32.argtypes = [p, p, p, p, p, i64, i32, i32, p]
    return lib


names = sys.argv[1:] or sorted(os.path.basename(os.path.dirname(f)) for f in glob.glob(root + "/build/variants/*/libtmgcn_hip.so"))
libs = {"default": load(root + "/tm-gcn_amd/libtmgcn_hip.so")}
for n in names:
    libs[n] = load(f"{root}/build/variants/{n}/libtmgcn_hip.so")

T, N, F, deg = (int(os.environ.get("AB_T", 4)), int(os.environ.get("AB_N", 2_000_000)), int(os.environ.get("AB_F", 128)),
                int(os.environ.get("AB_DEG", 32)))
A = synth.device_csr(os.environ.get("AB_GRAPH", "er"), T, N, deg, "cuda")       # er | powerlaw | powerlaw_sym | chess_tiled
N = A.N                                                                          # chess_tiled rounds N to a multiple of 7 301
X = torch.rand(T, N, F, device="cuda")
W = torch.randn(F, F, device="cuda") * 0.1
Y = torch.empty_like(X)
AX = torch.empty_like(X)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())


def run(lib, which):
    if which == "spmm":
        return lib.tmgcn_spmm_csr_batched_f32(ptr(A.rowptr), ptr(A.col), ptr(A.val), ptr(X), ptr(Y), A.n_rows, N, F, st)
    ax = ptr(AX) if which == "fused+ax" else None
    return lib.tmgcn_spmm_gemm_f32_hint(ptr(A.rowptr), ptr(A.col), ptr(A.val), ptr(X), A.n_rows, N, F, ptr(W), F, 0, 0, 0, 0,
                                        ptr(Y), ax, None, 0, float(A.avg_nnz_per_row), st)   # the hint the torch ops pass


res = {}
KINDS = ("spmm", "fused", "fused+ax") if F <= 128 else ("spmm",)          # the fused kernel takes K <= 128
for which in KINDS:
    for name, lib in libs.items():
        assert run(lib, which) == 0
    torch.cuda.synchronize()
    for rnd in range(7):
        for name, lib in libs.items():
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            run(lib, which)
            e.record()
            torch.cuda.synchronize()
            res.setdefault((which, name), []).append(s.elapsed_time(e))
# every variant computes the same thing: Y (and AX) bit for bit against the default library
ref = None
for name, lib in libs.items():
    Y.zero_(); AX.zero_()
    assert run(lib, KINDS[-1]) == 0
    torch.cuda.synchronize()
    if ref is None:
        ref = (Y.clone(), AX.clone())
    else:
        print(f"check     {name:16s} Y bit-equal: {bool(torch.equal(Y, ref[0]))}  AX bit-equal: {bool(torch.equal(AX, ref[1]))}  "
              f"max|dY| = {float((Y - ref[0]).abs().max()):.2e}")
del ref
by = A.nnz * (8 + F * 4 + (4 + F * 4) / (A.nnz / A.n_rows))
for (which, name), ms in res.items():
    med = statistics.median(ms)
    print(f"{which:9s} {name:16s} median {med:8.3f} ms  min {min(ms):8.3f}  {by / med / 1e6:6.0f} GB/s")
