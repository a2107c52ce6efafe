#!/usr/bin/env python3
"""Interleaved A/B timing of the standalone GEMM / dW kernels across library variants
(tools/ab_variants.sh).   python tools/ab_gemm.py [variant ...]"""
import ctypes as C
import glob
import os
import statistics
import sys

import torch

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
p, i32, i64 = C.c_void_p, C.c_int32, C.c_int64


def load(path):
    lib = C.CDLL(path)
    lib.tmgcn_gemm_f32.argtypes = [p, p, p, p, i64, i32, i32, i32, i64, i64, i32, i32, p]
    if hasattr(lib, "tmgcn_gemm_bf16w_f32"):
        lib.tmgcn_gemm_bf16w_f32.argtypes = [p, p, p, p, i64, i32, i32, i32, i64, i64, i32, i32, p]
    lib.tmgcn_gemm_dw_f32.argtypes = [p, p, p, i64, i32, i32, i64, i32, p, i64, p]
    lib.tmgcn_gemm_dw_workspace_bytes.restype = i64
    lib.tmgcn_gemm_dw_workspace_bytes.argtypes = [i64, i32, i32, i64]
    return lib


ALGO = int(os.environ.get("TMGCN_AB_GEMM_ALGO", "0"))   # 0 = auto (bf16x3), 1 = exact-f32 MFMA
names = sys.argv[1:] or sorted(os.path.basename(os.path.dirname(f)) for f in glob.glob(root + "/build/variants/*/libtmgcn_hip.so"))
libs = {"default": load(root + "/tm-gcn_amd/libtmgcn_hip.so")}
for n in names:
    libs[n] = load(f"{root}/build/variants/{n}/libtmgcn_hip.so")
R, K, Nf = int(os.environ.get("AB_R", 8_000_000)), int(os.environ.get("AB_K", 128)), int(os.environ.get("AB_NF", 128))
A = torch.rand(R, K, device="cuda")
W = torch.randn(K, Nf, device="cuda")
Wh = W.to(torch.bfloat16)
dY = torch.rand(R, Nf, device="cuda")
Y = torch.empty(R, max(K, Nf), device="cuda")     # gemm writes [R,Nf], gemm_dA [R,K]
dW = torch.empty(K, Nf, device="cuda")
ws = torch.empty(int(libs["default"].tmgcn_gemm_dw_workspace_bytes(R, K, Nf, 0)), dtype=torch.uint8, device="cuda")
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
ptr = lambda t: C.c_void_p(t.data_ptr())


def run(lib, which):
    if which == "gemm":
        return lib.tmgcn_gemm_f32(ptr(A), ptr(W), ptr(Y), None, R, K, Nf, 0, 0, 0, 0, ALGO, st)
    if which == "gemm_dA":
        return lib.tmgcn_gemm_f32(ptr(dY), ptr(W), ptr(Y), None, R, Nf, K, 1, 0, 0, 0, ALGO, st)
    if which == "gemm_bf16w":      # weight stored in bf16: three plane products per term
        if not hasattr(lib, "tmgcn_gemm_bf16w_f32"):
            return 0
        return lib.tmgcn_gemm_bf16w_f32(ptr(A), ptr(Wh), ptr(Y), None, R, K, Nf, 0, 0, 0, 0, ALGO, st)
    return lib.tmgcn_gemm_dw_f32(ptr(A), ptr(dY), ptr(dW), R, K, Nf, 0, 0, ptr(ws), ws.numel(), st)


res = {}
for which in ("gemm", "gemm_bf16w", "gemm_dA", "gemm_dW"):
    for lib in libs.values():
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
fl = 2.0 * R * K * Nf
for (which, name), ms in res.items():
    med = statistics.median(ms)
    print(f"{which:10s} {name:14s} median {med:6.2f} ms  {fl / med / 1e9:6.1f} TFLOP/s  {(R * (K + Nf) * 4) / med / 1e6:6.0f} GB/s")
